"""Experiment 5 (round 4): the fp32 classifier head ONLY inside the DDrague inference loop (zoo head_fp32="inference",
engine.precise_head) against the bf16 head, paired on dictionaries learned by the product with the bf16 head.
Structured workload, 4096 held-out images, ASR judged by the plain fp32 network.  Prints one JSON object."""
import json
import os
import sys
import tempfile

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
seeds = [int(s) for s in os.environ.get("SEEDS", "6033,7033,8033,9033,10033,11033").split(",")]
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
fast_i = zoo.build_classifier("resnet50", head_fp32="inference", **kw)
lab0 = torch.zeros(bs, dtype=torch.long, device=dev)


@torch.no_grad()
def fooled(net, x, adv):
    return int((net(adv).argmax(-1) != net(x).argmax(-1)).sum())


def evaluate(atk, attacked):
    f_self = f_32 = 0
    for lo in range(0, n_eval, bs):
        x = held[lo:lo + bs].to(dev).to(torch.bfloat16)
        adv = atk(x, lab0[:x.shape[0]])
        f_self += fooled(attacked, x, adv)
        f_32 += fooled(ref, x.float(), adv.float())
    return f_self / n_eval, f_32 / n_eval


out = {"runs": []}
x16, index = images.to(dev).to(torch.bfloat16), torch.arange(n, device=dev)
lab = engine.predict(fast_i, x16)
for seed in seeds:
    g = torch.Generator().manual_seed(seed)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    learner = engine.DictionaryLearner(d0.to(dev), v0.to(dev), eps, 0.01, "logits", False, 50.0)
    for _ in range(T):
        learner.step(fast_i, x16, index, lab)                  # the learner does not switch the precise head on: bf16 head
    name = f"s{seed}"
    torch.save([learner.d.cpu(), learner.v.cpu(), [], [], torch.tensor(0.)], os.path.join(tmp, f"ImageNet_{name}.bin"))
    rec = {"seed": seed}
    for tag, net in (("bf16_head", fast), ("fp32_head_at_inference", fast_i)):
        atk = ADIL(net, eps=eps, n_atoms=k, attack="supervised", model_name=name, loss="logits", steps_inference=S, dict_dir=tmp,
                   stream_dtype=torch.bfloat16)
        rec[tag] = dict(zip(("asr_judged_by_the_bf16_head_net", "asr_judged_by_fp32_net"), evaluate(atk, fast)))
        del atk
    out["runs"].append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
    del learner
for tag in ("bf16_head", "fp32_head_at_inference"):
    vals = torch.tensor([r[tag]["asr_judged_by_fp32_net"] for r in out["runs"]], dtype=torch.float64)
    out[f"{tag}_mean_std_pp"] = [100 * float(vals.mean()), 100 * float(vals.std(unbiased=True))]
print(json.dumps(out))
