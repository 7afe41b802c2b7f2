"""Experiment: ASR of the bf16 product vs the fp32 reference configuration on the structured workload (tests/structured.py).
Env knobs: N K T EPS NOISE CLASSES MODEL LOSS.  Prints one JSON line."""
import json, os, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
from dl_attack_on_imagenet_amd import engine, ops
from oracle import adil_oracle as O
from structured import fitted_classifiers, structured_images

E = os.environ.get
n, k, T = int(E("N", 512)), int(E("K", 50)), int(E("T", 100))
eps, noise, classes = float(E("EPS", 8 / 255)), float(E("NOISE", 0.10)), int(E("CLASSES", 10))
name, loss = E("MODEL", "resnet50"), E("LOSS", "logits")
dev = "cuda"
images, labels = structured_images(n, classes, seed=3, noise=noise)
t0 = time.time()
ref, fast, margins, pred = fitted_classifiers(name, images, labels, classes, dev, tempfile.mkdtemp(), target_margin=float(E("MARGIN", 10.0)))
fit_s = time.time() - t0
with torch.no_grad():
    p32 = torch.cat([ref(c.to(dev)).argmax(-1).cpu() for c in images.split(64)])
    o16 = torch.cat([fast(c.to(dev).to(torch.bfloat16)).float().cpu() for c in images.split(64)])
p16 = o16.argmax(-1)
t2 = o16.topk(2, dim=1).values
g = torch.Generator().manual_seed(33)
d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
index = torch.arange(n, device=dev)
out = dict(n=n, k=k, T=T, eps=eps, noise=noise, model=name, loss=loss, fit_seconds=fit_s,
           clean_accuracy_fp32=float((p32 == labels).float().mean()), clean_label_agreement_bf16_fp32=float((p16 == p32).float().mean()),
           margin_fp32_median=float(margins.median()), margin_fp32_min=float(margins.min()),
           margin_bf16_median=float((t2[:, 0] - t2[:, 1]).median()), margin_bf16_min=float((t2[:, 0] - t2[:, 1]).min()))
print(json.dumps(out), flush=True)
# leg A: fp32 oracle maths + fp32 plain network
d, v = d0.clone().to(dev), v0.clone().to(dev)
sd, sv = O.AdamWState(d, 0.01), O.AdamWState(v, 0.01)
x32 = images.to(dev)
fa = []
t0 = time.time()
for _ in range(T):
    _, fl = O.learn_step_a(ref, x32, index, d, v, sd, sv, eps, loss, -1.0, 50.0); fa.append(int(fl))
out["leg_A_seconds"] = time.time() - t0
with torch.no_grad():
    adv = O.synth(x32, d, v)
    asr_a = float(torch.cat([(ref(a).argmax(-1) != ref(c).argmax(-1)).float() for a, c in zip(adv.split(64), x32.split(64))]).mean())
del adv
# leg C: the product (HIP kernels, bf16 streams, bf16 FusedResNet)
x16 = images.to(dev).to(torch.bfloat16)
learner = engine.DictionaryLearner(d0.clone().to(dev), v0.clone().to(dev), eps, 0.01, loss, False, 50.0)
fc = []
t0 = time.time()
for _ in range(T):
    _, fl = learner.step(fast, x16, index); fc.append(fl)
fc = [int(f) for f in fc]
out["leg_C_seconds"] = time.time() - t0
with torch.no_grad():
    advc = ops.synth(x16, learner.d, ops.pack_codes(learner.v, None, n), n)
    asr_c = float((engine.predict(fast, advc) != engine.predict(fast, x16)).float().mean())
    asr_c_fp32judge = float(torch.cat([(ref(a.float()).argmax(-1) != ref(c).argmax(-1)).float() for a, c in zip(advc.split(64), x32.split(64))]).mean())
# leg P: the product in fp32 (HIP fp32 streams + the fp32 network)
fp, asr_p = [], None
if E("SKIP_P", "0") != "1":
    learner32 = engine.DictionaryLearner(d0.clone().to(dev), v0.clone().to(dev), eps, 0.01, loss, False, 50.0)
    fp = [int(learner32.step(ref, x32, index)[1]) for _ in range(T)]
    with torch.no_grad():
        advp = ops.synth(x32, learner32.d, ops.pack_codes(learner32.v, None, n), n)
        asr_p = float(torch.cat([(ref(a).argmax(-1) != ref(c).argmax(-1)).float() for a, c in zip(advp.split(64), x32.split(64))]).mean())
out.update(fooled_A=fa, fooled_C=fc, fooled_P=fp, asr_A=asr_a, asr_C=asr_c, asr_C_judged_by_fp32_net=asr_c_fp32judge, asr_P=asr_p)
print(json.dumps(out), flush=True)
