"""Time adil_spd_inverse alone (HIP events over 50 back-to-back launches) at several K, and check it against torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dl_attack_on_imagenet_amd import ops
dev = torch.device("cuda")
for k in (10, 50, 64, 100, 128):
    g = torch.Generator().manual_seed(k)
    x = torch.randn(4 * k + 50, k, generator=g, dtype=torch.float64)
    a = (x.t() @ x).float().to(dev)
    inv = ops.spd_inverse(a)
    err = float((inv.double().cpu() - torch.linalg.inv(a.double().cpu())).abs().max() / torch.linalg.inv(a.double().cpu()).abs().max())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.spd_inverse(a)
    e1.record(); torch.cuda.synchronize()
    print(f"K={k:4d}  {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per call   rel err vs fp64 inverse {err:.2e}", flush=True)
