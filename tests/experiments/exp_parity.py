"""Experiment: where does the config-1 D mismatch come from? CPU-oracle vs GPU-oracle (same code on cuda tensors) vs HIP."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dl_attack_on_imagenet_amd import engine, zoo
from oracle import adil_oracle as O
n, k, T, eps = 32, 10, int(os.environ.get("T", 20)), 8 / 255
g = torch.Generator().manual_seed(21)
images = torch.rand(n, 3, 224, 224, generator=g)
d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
cpu_model = zoo.build_classifier("resnet18", seed=5)
gpu_model = zoo.build_classifier("resnet18", seed=5, device="cuda")
idx = torch.arange(n)
# first-step gradient comparison
def gin(model, x, d, v):
    lab = model(x).argmax(-1)
    xt = O.synth(x, d, v)
    out, ls, g_ = O._input_grad(model, xt, lab, "logits", -1.0, 50.0, "sum")
    return g_, out
gc, oc = gin(cpu_model, images, d0, v0)
gg, og = gin(gpu_model, images.cuda(), d0.cuda(), v0.cuda())
rel = ((gg.cpu() - gc).abs() / (gc.abs() + 1e-12))
print("G: |g| median %.3e max %.3e ; rel err median %.3e 90%% %.3e 99%% %.3e ; frac sign flips %.3e" % (
    gc.abs().median(), gc.abs().max(), rel.median(), rel.flatten().kthvalue(int(rel.numel()*0.9)).values, rel.flatten().kthvalue(int(rel.numel()*0.99)).values,
    (torch.sign(gg.cpu()) != torch.sign(gc)).float().mean()))
gdc, _ = O.grad_dv(gc, d0, v0); gdg, _ = O.grad_dv(gg.cpu(), d0, v0)
print("grad_d sign flips cpu-vs-gpu G: %.3e ; |grad_d| median %.3e" % ((torch.sign(gdc) != torch.sign(gdg)).float().mean(), gdc.abs().median()))
runs = {}
for name, model, dev in (("cpu_oracle", cpu_model, "cpu"), ("gpu_oracle", gpu_model, "cuda")):
    d, v = d0.clone().to(dev), v0.clone().to(dev)
    sd, sv = O.AdamWState(d, 0.01), O.AdamWState(v, 0.01)
    fooled = []
    for _ in range(T):
        ls, fl = O.learn_step_a(model, images.to(dev), idx.to(dev), d, v, sd, sv, eps, "logits", -1.0, 50.0); fooled.append(fl)
    runs[name] = (d.cpu(), v.cpu(), fooled)
learner = engine.DictionaryLearner(d0.cuda(), v0.cuda(), eps, 0.01, "logits", False, 50.0)
fooled = []
for _ in range(T):
    ls, fl = learner.step(gpu_model, images.cuda(), idx.cuda()); fooled.append(int(fl))
runs["hip"] = (learner.d.cpu(), learner.v.cpu(), fooled)
for a, b in (("cpu_oracle", "gpu_oracle"), ("gpu_oracle", "hip"), ("cpu_oracle", "hip")):
    da, va, fa = runs[a]; db, vb, fb = runs[b]
    dd = (da - db).abs()
    print(f"{a:11s} vs {b:11s}: |dD| max {dd.max():.3e} mean {dd.mean():.3e} frac>1e-3 {(dd > 1e-3).float().mean():.3e} | |dV| {float((va - vb).abs().max()):.3e} | "
          f"|d(Dv)| {float((va @ da.reshape(-1, k).t() - vb @ db.reshape(-1, k).t()).abs().max()):.3e} | fooled equal {fa == fb}")
for name, (_, _, f) in runs.items():
    print(f"fooled {name:11s} {f}")
