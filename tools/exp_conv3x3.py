"""Experiment: adil_conv3x3 vs MIOpen at the ResNet-50 / B=512 conv2 shapes (forward; the input gradient is the same kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
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
B = 512
for hw, c in ((56, 64), (28, 128), (14, 256), (7, 512)):
    x = torch.randn(B, hw, hw, c, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(c, c, 3, 3, device=dev) * 0.05)
    wp = w.permute(0, 2, 3, 1).reshape(c, 9, c).contiguous().bfloat16()
    y = torch.empty(B * hw * hw, c, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: lib.adil_conv3x3(ops._ptr(x), ops._ptr(wp), ops._ptr(y), B, hw, hw, c, c, ops._stream()))
    xt = x.permute(0, 3, 1, 2); wt = w.bfloat16().contiguous(memory_format=torch.channels_last)
    tm = timeit(lambda: F.conv2d(xt, wt, padding=1))
    fl = 2 * B * hw * hw * c * c * 9
    print(f"hw {hw:3d} C {c:4d}: adil_conv3x3 {t:7.1f} us = {fl/t/1e6:7.1f} TFLOP/s | MIOpen (incl. its zero-fill) {tm:7.1f} us = {fl/tm/1e6:7.1f} TFLOP/s", flush=True)
