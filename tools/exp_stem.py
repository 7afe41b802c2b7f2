"""Experiment: the four stem kernels at B=512, 224x224 (time per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import ops, _lib
dev = torch.device("cuda"); lib = _lib.load()
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H = 512, 224
x = torch.rand(B, 3, H, H, device=dev).bfloat16()
w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
wf, wb = ops.pack_stem_weights(w)
sc = torch.rand(64, device=dev) + 0.5; sh = torch.randn(64, device=dev) * 0.1
y1 = torch.empty(B, 112, 112, 64, device=dev, dtype=torch.bfloat16)
p = torch.empty(B, 56, 56, 64, device=dev, dtype=torch.bfloat16); idx = torch.empty(B, 56, 56, 64, device=dev, dtype=torch.uint8)
g = torch.randn(B, 56, 56, 64, device=dev).bfloat16(); gy = torch.empty_like(y1); gx = torch.empty_like(x)
m, s = [0.485, 0.456, 0.406], [1 / 0.229, 1 / 0.224, 1 / 0.225]
print("stem_conv_fwd  %7.1f us" % timeit(lambda: lib.adil_stem_conv_fwd(ops._ptr(x), 1, ops._ptr(wf), *m, *s, ops._ptr(sc), ops._ptr(sh), ops._ptr(y1), B, H, H, ops._stream())))
print("maxpool_fwd    %7.1f us" % timeit(lambda: lib.adil_maxpool_fwd(ops._ptr(y1), ops._ptr(p), ops._ptr(idx), B, 112, 112, 64, ops._stream())))
print("stem_pool_bwd  %7.1f us" % timeit(lambda: lib.adil_stem_pool_bwd(ops._ptr(g), ops._ptr(idx), ops._ptr(p), ops._ptr(sc), ops._ptr(gy), B, 112, 112, 64, ops._stream())))
print("stem_conv_bwd  %7.1f us" % timeit(lambda: lib.adil_stem_conv_bwd(ops._ptr(gy), ops._ptr(wb), *s, ops._ptr(gx), 1, B, H, H, ops._stream())))
