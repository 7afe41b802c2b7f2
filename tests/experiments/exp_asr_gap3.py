"""Experiment 3 (round 4): the distribution of the attack success rate over INDEPENDENT runs of the whole
pipeline (learn a dictionary, then attack held-out images with it), per configuration — because a single run of the
chaotic learner says little: experiment 2 measured a seed-to-seed spread of the bf16 product of ~1-2 pp.

Configurations (each learns AND attacks with its own classifier; ASR judged by the plain fp32 network and by itself):
  C   bf16 FusedResNet, bf16 streams                                  (rounds 1-3 benchmark configuration)
  H   the same with the classifier head (pooling + last linear layer) in fp32 (zoo head_fp32)
  A   fp32 oracle learner + oracle inference + plain fp32 network     (the reference configuration; SEEDS_A seeds only:
      a run costs minutes)
Seeds index the initial (D, V).  Prints one JSON object with per-run figures and mean / standard deviation per leg."""
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
n_eval_a = int(os.environ.get("N_EVAL_A", 2048))
seeds = [int(s) for s in os.environ.get("SEEDS", "33,1033,2033,3033,4033,5033").split(",")]
seeds_a = [int(s) for s in os.environ.get("SEEDS_A", "1033,2033,3033").split(",") if s]
images, labels = structured_images(n, 10, seed=3)
held, held_labels = structured_images(n_eval, 10, seed=3, draw=1)
tmp = tempfile.mkdtemp()

ref = zoo.build_classifier("resnet50", seed=0, device=dev)
fit_centroid_head(ref, images, labels, 10, dev, target_margin=10.0)
path = os.path.join(tmp, "fitted.pt")
torch.save(ref[-1].state_dict(), path)
ref = zoo.build_classifier("resnet50", seed=0, weights=path, device=dev)
kw = dict(seed=0, weights=path, device=dev, dtype=torch.bfloat16, channels_last=True, fuse_bn_act=True, fuse_stem=True)
nets = {"C": zoo.build_classifier("resnet50", **kw), "H": zoo.build_classifier("resnet50", head_fp32=True, **kw)}


@torch.no_grad()
def fooled(net, x, adv):
    return int((net(adv).argmax(-1) != net(x).argmax(-1)).sum())


def evaluate(attack_fn, attacked, in_dtype, count):
    f_self = f_32 = 0
    for lo in range(0, count, bs):
        x = held[lo:lo + bs].to(dev).to(in_dtype)
        adv = attack_fn(x)
        f_self += fooled(attacked, x, adv)
        f_32 += fooled(ref, x.float(), adv.float())
    return f_self / count, f_32 / count


def init(seed):
    g = torch.Generator().manual_seed(seed)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    return d0, O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)


out = {"T": T, "S": S, "n_eval": n_eval, "n_eval_A": n_eval_a, "runs": []}
x16, index = images.to(dev).to(torch.bfloat16), torch.arange(n, device=dev)
lab0 = torch.zeros(bs, dtype=torch.long, device=dev)
for seed in seeds:
    d0, v0 = init(seed)
    for tag, net in nets.items():
        t0 = time.time()
        learner = engine.DictionaryLearner(d0.to(dev), v0.to(dev), eps, 0.01, "logits", False, 50.0)
        lab = engine.predict(net, x16)
        fl = [learner.step(net, x16, index, lab)[1] for _ in range(T)]
        name = f"{tag}{seed}"
        torch.save([learner.d.cpu(), learner.v.cpu(), [], [], torch.tensor(0.)], os.path.join(tmp, f"ImageNet_{name}.bin"))
        atk = ADIL(net, eps=eps, n_atoms=k, attack="supervised", model_name=name, loss="logits", steps_inference=S, dict_dir=tmp,
                   stream_dtype=torch.bfloat16)
        a_self, a_32 = evaluate(lambda xx: atk(xx, lab0[:xx.shape[0]]), net, torch.bfloat16, n_eval)
        rec = dict(leg=tag, seed=seed, asr_judged_by_attacked_net=a_self, asr_judged_by_fp32_net=a_32,
                   fooled_while_learning_last=int(fl[-1]), seconds=time.time() - t0)
        out["runs"].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
        del learner, atk
for seed in seeds_a:
    t0 = time.time()
    d0, v0 = init(seed)
    d, v = d0.to(dev), v0.to(dev)
    sd, sv = O.AdamWState(d, 0.01), O.AdamWState(v, 0.01)
    x = images.to(dev)
    with torch.no_grad():
        lab = ref(x).argmax(-1)
    for _ in range(T):
        _, fl = O.learn_step_a(ref, x, index, d, v, sd, sv, eps, "logits", -1.0, 50.0, labels=lab)
    a_self, _ = evaluate(lambda xx: O.forward_supervised_ddrague(ref, xx, d, eps, S, "logits"), ref, torch.float32, n_eval_a)
    rec = dict(leg="A", seed=seed, asr_judged_by_fp32_net=a_self, fooled_while_learning_last=int(fl), seconds=time.time() - t0)
    out["runs"].append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
    del d, v, sd, sv
for tag in ("C", "H", "A"):
    vals = torch.tensor([r["asr_judged_by_fp32_net"] for r in out["runs"] if r["leg"] == tag], dtype=torch.float64)
    if len(vals):
        out[f"{tag}_mean_std_pp_judged_by_fp32_net"] = [100 * float(vals.mean()), 100 * float(vals.std(unbiased=True)) if len(vals) > 1 else 0.0]
print(json.dumps(out))
