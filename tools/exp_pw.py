"""Experiment: fused pointwise-conv kernel at the ResNet-50 / B=512 shapes vs its HBM floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import ops, _lib
dev = torch.device("cuda")
lib = _lib.load()
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 512
shapes = [(56, 64, 64, 0), (56, 64, 256, 1), (56, 256, 64, 0), (56, 256, 128, 0), (28, 128, 512, 1), (28, 512, 128, 0), (28, 512, 256, 0),
          (14, 256, 1024, 1), (14, 1024, 256, 0), (14, 1024, 512, 0), (7, 512, 2048, 1), (7, 2048, 512, 0)]
mult = [1, 4, 2, 1, 4, 3, 1, 6, 5, 1, 3, 2]
tot = tot_floor = 0
for (hw, k, n, has_res), mu in zip(shapes, mult):
    m = B * hw * hw
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.05
    sc = torch.rand(n, device=dev) + 0.5; sh = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev, dtype=torch.bfloat16) if has_res else None
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: lib.adil_pw_conv_fwd(ops._ptr(x), ops._ptr(w), ops._ptr(sc), ops._ptr(sh), ops._ptr(r), ops._ptr(y), m, k, n, 1, None, None, 0, 0, ops._stream()))
    byt = (m * k + m * n * (2 if has_res else 1) + n * k) * 2
    fl = byt / 5.3e6
    tot += t * mu; tot_floor += fl * mu
    print(f"hw {hw:3d} K {k:5d} N {n:5d} res {has_res}  {t:7.1f} us   floor@5.3TB/s {fl:7.1f} us  ({fl/t*100:4.0f}%)  x{mu}", flush=True)
print(f"weighted total {tot/1e3:.2f} ms, floor {tot_floor/1e3:.2f} ms")

print("---- input gradient (adil_pw_conv_bwd): g (+g2), y -> gres, gx")
tot = tot_floor = 0
for (hw, k, n, has_res), mu in zip(shapes, mult):
    m = B * hw * hw
    g = torch.randn(m, n, device=dev, dtype=torch.bfloat16)
    g2 = torch.randn(m, n, device=dev, dtype=torch.bfloat16) if has_res else None
    yy = torch.relu(torch.randn(m, n, device=dev)).bfloat16()
    wt = torch.randn(k, n, device=dev, dtype=torch.bfloat16) * 0.05
    sc = torch.rand(n, device=dev) + 0.5
    gx = torch.empty(m, k, device=dev, dtype=torch.bfloat16)
    gres = torch.empty(m, n, device=dev, dtype=torch.bfloat16) if has_res else None
    t = timeit(lambda: lib.adil_pw_conv_bwd(ops._ptr(g), ops._ptr(g2), ops._ptr(yy), ops._ptr(sc), ops._ptr(wt), ops._ptr(gx), ops._ptr(gres), m, k, n, 1, None, None, None, None, 0, 0, ops._stream()))
    byt = (m * k + m * n * (4 if has_res else 2) + n * k) * 2
    fl = byt / 5.3e6
    tot += t * mu; tot_floor += fl * mu
    print(f"hw {hw:3d} K {k:5d} N {n:5d} res {has_res}  {t:7.1f} us   floor@5.3TB/s {fl:7.1f} us  ({fl/t*100:4.0f}%)  x{mu}", flush=True)
print(f"weighted total {tot/1e3:.2f} ms, floor {tot_floor/1e3:.2f} ms")
