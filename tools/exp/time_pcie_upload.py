"""Host -> HBM rate of one 512-image bf16 batch (154 MB) from pinned memory, and of loader.ResidentImages on an in-memory set."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dl_attack_on_imagenet_amd import loader
dev = torch.device("cuda")
x = torch.rand(512, 3, 224, 224).to(torch.bfloat16).pin_memory()
y = torch.empty_like(x, device=dev)
for _ in range(3):
    y.copy_(x, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    y.copy_(x, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"pinned H2D copy of one 512-image bf16 batch: {x.numel() * 2 / 1e6:.0f} MB in {dt * 1e3:.2f} ms = {x.numel() * 2 / dt / 1e9:.1f} GB/s")
ds = torch.utils.data.TensorDataset(torch.rand(2048, 3, 224, 224), torch.zeros(2048, dtype=torch.long))
t0 = time.perf_counter()
res = loader.ResidentImages(ds, dev, torch.bfloat16)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ResidentImages upload of 2048 fp32 host images -> bf16 resident: {2048 * 150528 * 4 / 1e9:.2f} GB of host data in {dt:.2f} s = "
      f"{2048 * 150528 * 4 / dt / 1e9:.1f} GB/s ({2048 / dt:.0f} images/s; per-item fetch + pinned staging + copy + convert)")
