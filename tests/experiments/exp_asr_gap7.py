"""Experiment 7 (round 4): is the bf16 ADiL path itself inside north_star's ASR tolerance when the classifier is the
reference's fp32 one?  Whole pipelines (learn a dictionary, attack held-out images with it), seeds on which leg A — the
fp32 oracle end to end — is already recorded (profiles/r04_asr_gap.md §2: 33 -> 99.19, 1033 -> 99.46, 2033 -> 99.41,
3033 -> 99.66 %), so the comparison is paired:
  M    the PRODUCT with bf16 image streams (x + D v and dLoss/dx in bf16: every ADiL kernel in its benchmarked dtype)
       against the plain fp32 ResNet-50 — the network sees the bf16 pixels through a cast, its input gradient is rounded to
       bf16 on the way back;
  P32  the PRODUCT in fp32 streams against the plain fp32 ResNet-50 (the parity configuration).
Also reports the cost: learner steps/s and attacked images/s of each leg.  Prints one JSON object."""
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
n_eval, bs = int(os.environ.get("N_EVAL", 2048)), int(os.environ.get("BS", 512))
seeds = [int(s) for s in os.environ.get("SEEDS", "33,1033,2033").split(",")]
legs = os.environ.get("LEGS", "M,P32").split(",")
images, labels = structured_images(n, 10, seed=3)
held, held_labels = structured_images(n_eval, 10, seed=3, draw=1)
tmp = tempfile.mkdtemp()
ref = zoo.build_classifier("resnet50", seed=0, device=dev)
fit_centroid_head(ref, images, labels, 10, dev, target_margin=10.0)
path = os.path.join(tmp, "fitted.pt")
torch.save(ref[-1].state_dict(), path)
ref = zoo.build_classifier("resnet50", seed=0, weights=path, device=dev)


class CastIn(torch.nn.Module):
    """The fp32 network behind bf16 image streams."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x):
        return self.net(x.float())


nets = {"M": (CastIn(ref).eval(), torch.bfloat16), "P32": (ref, None)}
lab0 = torch.zeros(bs, dtype=torch.long, device=dev)


@torch.no_grad()
def fooled(x, adv):
    return int((ref(adv.float()).argmax(-1) != ref(x.float()).argmax(-1)).sum())


out = {"T": T, "S": S, "n_eval": n_eval, "runs": []}
index = torch.arange(n, device=dev)
for seed in seeds:
    g = torch.Generator().manual_seed(seed)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    for tag in legs:
        net, sdt = nets[tag]
        x = images.to(dev).to(sdt or torch.float32)
        lab = engine.predict(net, x)
        learner = engine.DictionaryLearner(d0.to(dev), v0.to(dev), eps, 0.01, "logits", False, 50.0)
        torch.cuda.synchronize(); t0 = time.time()
        for it in range(T):
            fl = learner.step(net, x, index, lab)[1]
            if it % 100 == 99:
                print(f"  {tag} seed {seed}: learning iteration {it + 1}", file=sys.stderr, flush=True)
        torch.cuda.synchronize(); t_learn = time.time() - t0
        name = f"{tag}{seed}"
        torch.save([learner.d.cpu(), learner.v.cpu(), [], [], torch.tensor(0.)], os.path.join(tmp, f"ImageNet_{name}.bin"))
        atk = ADIL(net, eps=eps, n_atoms=k, attack="supervised", model_name=name, loss="logits", steps_inference=S, dict_dir=tmp,
                   stream_dtype=sdt)
        f = 0
        torch.cuda.synchronize(); t0 = time.time()
        for lo in range(0, n_eval, bs):
            xe = held[lo:lo + bs].to(dev).to(sdt or torch.float32)
            adv = atk(xe, lab0[:xe.shape[0]])
            f += fooled(xe, adv)
            print(f"  {tag} seed {seed}: attacked {lo + bs} images", file=sys.stderr, flush=True)
        torch.cuda.synchronize(); t_atk = time.time() - t0
        rec = dict(leg=tag, seed=seed, asr_judged_by_fp32_net=f / n_eval, fooled_while_learning_last=int(fl),
                   learner_images_per_sec=n * T / t_learn, attacked_images_per_sec=n_eval / t_atk)
        out["runs"].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
        del learner, atk
for tag in legs:
    vals = torch.tensor([r["asr_judged_by_fp32_net"] for r in out["runs"] if r["leg"] == tag], dtype=torch.float64)
    out[f"{tag}_mean_std_pp"] = [100 * float(vals.mean()), 100 * float(vals.std(unbiased=True)) if len(vals) > 1 else 0.0]
print(json.dumps(out))
