"""adamw_l1ball_ over ALL N code rows (quirk Q3) with the batch gradient as slabs vs dense, N = B and N >> B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dl_attack_on_imagenet_amd import ops
dev = torch.device("cuda")
B, K, P = 512, 50, 150528
g0 = torch.Generator().manual_seed(0)
d = (-1 + 2 * torch.rand(3, 224, 224, K, generator=g0)).to(dev)
g = (torch.randn(B, 3, 224, 224, generator=g0) * 1e-3).to(dev).to(torch.bfloat16)
h = ops.AdamWSchedule(0.01).next()
for N in (512, 4096, 50000):
    v = ops.l1ball_project_(torch.rand(N, K, generator=g0).to(dev), 8 / 255)
    m, s = torch.zeros_like(v), torch.zeros_like(v)
    pos = torch.full((N,), -1, dtype=torch.int32, device=dev)
    idx = torch.randperm(N, generator=g0)[:B].to(dev)
    for defer in (False, True):
        def step():
            vp, vpt = ops.pack_codes(v, idx, B, pos=pos, transposed=torch.bfloat16)
            _, gv = ops.grad(g, d, vp, B, vpt=vpt, defer_v=defer)
            e0.record(); ops.adamw_l1ball_(v, gv, pos, m, s, h, 8 / 255, reset_pos=True); e1.record()
        ts = []
        for i in range(25):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            step(); torch.cuda.synchronize()
            if i >= 5: ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"N={N:6d} slabs={defer!s:5s} adamw_l1ball bracket {sum(ts)/len(ts):7.1f} us (min {min(ts):.1f})", flush=True)
