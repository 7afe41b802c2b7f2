"""Experiment 2 (round 4, VERDICT r3 #1c): on the SAME dictionaries, what does each inference-side switch do to the attack
success rate, and how large is the run-to-run spread of the product's own ASR?  Structured workload, ResNet-50 with a
fitted head, NDICT dictionaries learned by the bf16 product (different seeded initialisations, T iterations), N_EVAL
held-out images each, ASR judged by the plain fp32 network (the classifier the reference attacks) and by the attacked one.

Inference variants per dictionary (paired: same dictionary, same images):
  bf16_streams        ADIL.forward, bf16 image streams + bf16 FusedResNet                 (the product as benchmarked)
  fp32_streams        fp32 image streams into the same bf16 FusedResNet (its stem kernels read fp32 pixels): the adversarial
                      image x + D v is not rounded to bf16 (ulp 2e-3 at 0.5 against a budget of 3.1e-2) before the network sees it
  bf16_streams_head32 bf16 streams, classifier head (pooling + last linear layer) in fp32
and for the first dictionary the ORACLE's fp32 inference against the plain fp32 network (N_ORACLE images).
Prints one JSON object."""
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import torch

from attacks import ADIL
from dl_attack_on_imagenet_amd import engine, zoo
from oracle import adil_oracle as O
from structured import fit_centroid_head, structured_images

n, k, eps, dev = 512, 50, 8 / 255, "cuda"
T, S = int(os.environ.get("T", 300)), int(os.environ.get("S", 100))
n_eval, bs = int(os.environ.get("N_EVAL", 4096)), int(os.environ.get("BS", 512))
ndict, n_oracle = int(os.environ.get("NDICT", 4)), int(os.environ.get("N_ORACLE", 2048))
images, labels = structured_images(n, 10, seed=3)
held, held_labels = structured_images(n_eval, 10, seed=3, draw=1)
tmp = tempfile.mkdtemp()

ref = zoo.build_classifier("resnet50", seed=0, device=dev)
fit_centroid_head(ref, images, labels, 10, dev, target_margin=10.0)
path = os.path.join(tmp, "fitted.pt")
torch.save(ref[-1].state_dict(), path)
ref = zoo.build_classifier("resnet50", seed=0, weights=path, device=dev)
kw = dict(seed=0, weights=path, device=dev, dtype=torch.bfloat16, channels_last=True, fuse_bn_act=True, fuse_stem=True)
fast = zoo.build_classifier("resnet50", **kw)
fast_h32 = zoo.build_classifier("resnet50", head_fp32=True, **kw)


@torch.no_grad()
def fooled(net, x, adv):
    return int((net(adv).argmax(-1) != net(x).argmax(-1)).sum())


def evaluate(attack_fn, attacked, in_dtype, count):
    f_self = f_32 = 0
    t0 = time.time()
    for lo in range(0, count, bs):
        x = held[lo:lo + bs].to(dev).to(in_dtype)
        adv = attack_fn(x)
        f_self += fooled(attacked, x, adv)
        f_32 += fooled(ref, x.float(), adv.float())
    torch.cuda.synchronize()
    return dict(asr_judged_by_attacked_net=f_self / count, asr_judged_by_fp32_net=f_32 / count,
                images_per_sec=count / (time.time() - t0))


out = {"T": T, "S": S, "n_eval": n_eval, "dictionaries": []}
x16, index = images.to(dev).to(torch.bfloat16), torch.arange(n, device=dev)
lab = engine.predict(fast, x16)
lab0 = torch.zeros(bs, dtype=torch.long, device=dev)
for di in range(ndict):
    g = torch.Generator().manual_seed(33 + 1000 * di)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    learner = engine.DictionaryLearner(d0.to(dev), v0.to(dev), eps, 0.01, "logits", False, 50.0)
    fl = [learner.step(fast, x16, index, lab)[1] for _ in range(T)]
    name = f"d{di}"
    torch.save([learner.d.cpu(), learner.v.cpu(), [], [], torch.tensor(0.)], os.path.join(tmp, f"ImageNet_{name}.bin"))
    rec = {"seed": 33 + 1000 * di, "fooled_while_learning_last": [int(f) for f in fl[-3:]]}
    for tag, net, sdt in (("bf16_streams", fast, torch.bfloat16), ("fp32_streams", fast, None),
                          ("bf16_streams_head32", fast_h32, torch.bfloat16)):
        atk = ADIL(net, eps=eps, n_atoms=k, attack="supervised", model_name=name, loss="logits", steps_inference=S, dict_dir=tmp,
                   stream_dtype=sdt)
        rec[tag] = evaluate(lambda xx: atk(xx, lab0[:xx.shape[0]]), net, sdt or torch.float32, n_eval)
        del atk
    if di == 0 and n_oracle > 0:
        rec["oracle_fp32_inference_fp32_net"] = evaluate(
            lambda xx: O.forward_supervised_ddrague(ref, xx, learner.d, eps, S, "logits"), ref, torch.float32, n_oracle)
        rec["oracle_images"] = n_oracle
    out["dictionaries"].append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
    del learner
for tag in ("bf16_streams", "fp32_streams", "bf16_streams_head32"):
    vals = torch.tensor([r[tag]["asr_judged_by_fp32_net"] for r in out["dictionaries"]], dtype=torch.float64)
    out[f"{tag}_mean_std_pp"] = [100 * float(vals.mean()), 100 * float(vals.std(unbiased=True)) if len(vals) > 1 else 0.0]
print(json.dumps(out))
