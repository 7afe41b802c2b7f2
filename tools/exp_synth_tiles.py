"""Experiment: synth time vs number of 128-pixel tiles (workgroups), to expose launch-quantisation effects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import ops
dev = torch.device("cuda")
B, K = int(os.environ.get("B", 512)), int(os.environ.get("K", 50))
def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
v = (torch.randn(B, K) * 0.01).to(dev); vp = ops.pack_codes(v, None, B)
for nt in (768, 1176, 2048, 2304, 4096):
    P = 128 * nt
    d = torch.rand(1, 1, P, K, device=dev)
    for dt, s in ((torch.bfloat16, 2), (torch.float32, 4)):
        x = torch.rand(B, 1, 1, P, device=dev).to(dt); out = torch.empty_like(x)
        t = timeit(lambda: ops.synth(x, d, vp, B, out=out))
        byt = 2 * B * P * s + P * K * 4
        tc = timeit(lambda: out.copy_(x))
        print(f"tiles {nt:5d} {str(dt):15s} {t*1e3:8.1f} us  {byt/t/1e6:8.1f} GB/s  {t*1e6/nt:7.1f} ns/tile | torch copy {tc*1e3:8.1f} us {2*B*P*s/tc/1e6:8.1f} GB/s", flush=True)
