"""Experiment: ResNet-50 1x1 convolutions (channels_last bf16, B=512) as MIOpen convs vs hipBLASLt GEMMs, fwd and bwd-data."""
import os, sys, torch, torch.nn.functional as F
dev = torch.device("cuda")
B = int(os.environ.get("B", 512))
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(56, 64, 64), (56, 64, 256), (56, 256, 64), (28, 256, 128), (28, 128, 512), (28, 512, 128), (14, 512, 256), (14, 256, 1024),
          (14, 1024, 256), (7, 1024, 512), (7, 512, 2048), (7, 2048, 512)]
tot = {"conv_fwd": 0, "mm_fwd": 0, "addmm_relu": 0, "conv_bwd": 0, "mm_bwd": 0}
for (hw, cin, cout) in shapes:
    x = torch.randn(B, cin, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 1, 1, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last) * 0.05
    bias = torch.randn(cout, device=dev, dtype=torch.bfloat16)
    x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)          # view: (B*H*W, cin)
    assert x2.is_contiguous()
    w2 = w.reshape(cout, cin)
    wt = w2.t().contiguous()
    y = F.conv2d(x, w)
    y2 = x2 @ w2.t()
    err = (y.permute(0, 2, 3, 1).reshape(-1, cout).float() - y2.float()).abs().max().item()
    gy = torch.randn_like(y); gy2 = gy.permute(0, 2, 3, 1).reshape(-1, cout)
    t_cf = timeit(lambda: F.conv2d(x, w))
    t_mf = timeit(lambda: torch.mm(x2, wt))
    t_ar = timeit(lambda: torch._addmm_activation(bias, x2, wt, use_gelu=False))
    t_cb = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
    t_mb = timeit(lambda: torch.mm(gy2, w2))
    mbytes = (x.numel() + y.numel()) * 2 / 1e6
    print(f"hw {hw:3d} {cin:5d}->{cout:5d}  conv fwd {t_cf:7.1f}  mm {t_mf:7.1f}  addmm+relu {t_ar:7.1f} | conv bwd {t_cb:7.1f}  mm {t_mb:7.1f} us | min traffic {mbytes:7.1f} MB = {mbytes/5.3e3*1e3:6.1f} us   err {err:.3f}", flush=True)
    for k, v in zip(tot, (t_cf, t_mf, t_ar, t_cb, t_mb)): tot[k] += v
print(tot)
