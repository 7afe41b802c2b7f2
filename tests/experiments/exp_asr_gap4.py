"""Experiment 4 (round 4): anatomy of a BAD run of the bf16 product.  Experiment 3 found the product's ASR bimodal over
independent runs (four of six at 99.5-99.9 %, two at 94.5 / 97.6 %) while the fp32 reference sits at 99.4 +- 0.15 %.  This
script repeats the product's pipeline until a run lands below LOW (default 98.5 %), then asks of THAT dictionary:
  * which held-out images are not fooled: per class, per evaluation batch;
  * how far from the decision boundary they end: top-2 margin of the adversarial image under the attacked bf16 network and
    under the fp32 network;
  * what other inference settings do with the same dictionary: 300 iterations instead of 100, the fp32-head classifier,
    fp32 streams + the plain fp32 network (the product's fp32 path).
Prints one JSON object."""
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
low, max_runs = float(os.environ.get("LOW", 0.985)), int(os.environ.get("MAX_RUNS", 6))
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
lab0 = torch.zeros(bs, dtype=torch.long, device=dev)


@torch.no_grad()
def margins(net, x, adv):
    """(fooled flags, top-2 margin of the adversarial logits signed by 'clean label still wins')."""
    clean = net(x).argmax(-1)
    out = net(adv).float()
    own = out.gather(1, clean[:, None]).squeeze(1)
    other = out.scatter(1, clean[:, None], float("-inf")).max(1).values
    return out.argmax(-1) != clean, own - other


def attack_all(atk, in_dtype):
    advs = []
    for lo in range(0, n_eval, bs):
        x = held[lo:lo + bs].to(dev).to(in_dtype)
        advs.append(atk(x, lab0[:x.shape[0]]).cpu())
    return torch.cat(advs)


def judged(net, adv, in_dtype):
    f, m = [], []
    for lo in range(0, n_eval, bs):
        ff, mm = margins(net, held[lo:lo + bs].to(dev).to(in_dtype), adv[lo:lo + bs].to(dev).to(in_dtype))
        f.append(ff.cpu()); m.append(mm.cpu())
    return torch.cat(f), torch.cat(m)


out = {"runs": []}
x16, index = images.to(dev).to(torch.bfloat16), torch.arange(n, device=dev)
lab = engine.predict(fast, x16)
bad = None
# (a first version repeated ONE seed: six runs inside one process gave the identical ASR, 0.990479 — the bf16 pipeline is
# deterministic within a process; the spread of experiments 1-3 is between processes / boxes (library algorithm choices)
# and between initialisations.  So: one run per seed.)
seed0 = int(os.environ.get("SEED0", 4033))
for run in range(max_runs):
    g = torch.Generator().manual_seed(seed0 + 1000 * run)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    learner = engine.DictionaryLearner(d0.to(dev), v0.to(dev), eps, 0.01, "logits", False, 50.0)
    for _ in range(T):
        learner.step(fast, x16, index, lab)
    name = f"run{run}"
    torch.save([learner.d.cpu(), learner.v.cpu(), [], [], torch.tensor(0.)], os.path.join(tmp, f"ImageNet_{name}.bin"))
    atk = ADIL(fast, eps=eps, n_atoms=k, attack="supervised", model_name=name, loss="logits", steps_inference=S, dict_dir=tmp,
               stream_dtype=torch.bfloat16)
    adv = attack_all(atk, torch.bfloat16)
    f32, m32 = judged(ref, adv, torch.float32)
    asr = float(f32.float().mean())
    out["runs"].append([seed0 + 1000 * run, asr])
    print(f"run {run}: ASR judged by the fp32 network {asr:.4f}", file=sys.stderr, flush=True)
    if asr < low:
        bad = (name, learner.d.clone(), adv, f32, m32)
        break
    del learner, atk
if bad is not None:
    name, d_bad, adv, f32, m32 = bad
    f16, m16 = judged(fast, adv, torch.bfloat16)
    unf = ~f32
    rep = {"asr_fp32_judge": float(f32.float().mean()), "asr_bf16_judge": float(f16.float().mean()),
           "unfooled_per_class": [int((unf & (held_labels == c)).sum()) for c in range(10)],
           "unfooled_per_batch": [int(unf[lo:lo + bs].sum()) for lo in range(0, n_eval, bs)],
           "margin_fp32_net_of_unfooled_quantiles": [float(q) for q in m32[unf].quantile(torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0]))],
           "margin_bf16_net_of_unfooled_quantiles": [float(q) for q in m16[unf].quantile(torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0]))],
           "margin_fp32_net_of_fooled_quantiles": [float(q) for q in m32[f32].quantile(torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0]))]}
    for tag, net, sdt, steps in (("bf16 net, 300 iterations", fast, torch.bfloat16, 300), ("fp32-head net, 100 iterations", fast_h32, torch.bfloat16, S),
                                 ("fp32 streams + plain fp32 net, 100 iterations", ref, None, S)):
        a2 = ADIL(net, eps=eps, n_atoms=k, attack="supervised", model_name=name, loss="logits", steps_inference=steps, dict_dir=tmp,
                  stream_dtype=sdt)
        adv2 = attack_all(a2, sdt or torch.float32)
        ff, _ = judged(ref, adv2, torch.float32)
        rep[f"same dictionary, {tag}: ASR (fp32 judge)"] = float(ff.float().mean())
        rep[f"same dictionary, {tag}: unfooled per class"] = [int(((~ff) & (held_labels == c)).sum()) for c in range(10)]
    out["bad_run"] = rep
print(json.dumps(out))
