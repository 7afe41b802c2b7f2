"""Experiment: synth time vs atom count K at the BASELINE stream shape (access-pattern ceiling vs MFMA/LDS cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import ops
dev = torch.device("cuda")
B, P = int(os.environ.get("B", 512)), 150528
def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for dt, s in ((torch.bfloat16, 2), (torch.float32, 4)):
    x = torch.rand(B, 1, 1, P, device=dev).to(dt); out = torch.empty_like(x)
    for K in (1, 16, 32, 48, 50, 64, 100, 128):
        d = torch.rand(1, 1, P, K, device=dev)
        v = (torch.randn(B, K) * 0.01).to(dev); vp = ops.pack_codes(v, None, B)
        t = timeit(lambda: ops.synth(x, d, vp, B, out=out))
        byt = 2 * B * P * s + P * K * 4
        print(f"{str(dt):15s} K {K:4d} {t*1e3:8.1f} us  {byt/t/1e6:8.1f} GB/s", flush=True)
