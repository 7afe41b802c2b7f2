"""Does the synth kernel lose time in a partially filled last round of workgroups?  GB/s vs number of 128-pixel tiles
(B=512, K=50, bf16; 4 workgroups of 4 waves are resident per CU -> 1024 slots)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import ops
dev = torch.device("cuda"); B, K = 512, 50
def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for tiles in (1024, 1176, 1280, 1536, 2048, 2200, 2560, 3072):
    P = 128 * tiles
    d = (-1 + 2 * torch.rand(P, K)).to(dev).reshape(1, 1, P, K)
    v = (torch.randn(B, K) * 0.01).to(dev); vp = ops.pack_codes(v, None, B)
    x = torch.rand(B, 1, 1, P).to(dev).to(torch.bfloat16); out = torch.empty_like(x)
    t = timeit(lambda: ops.synth(x, d, vp, B, out=out))
    byt = 2 * B * P * 2 + P * K * 4
    print(f"tiles {tiles:5d} ({tiles / 1024:.2f} rounds)  {t * 1e3:7.1f} us  {byt / t / 1e6:7.1f} GB/s   {t * 1e6 / tiles:6.1f} ns/tile", flush=True)
    del d, x, out
