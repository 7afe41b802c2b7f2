"""What plain streaming kernels reach on this MI355X (context for the roofline fractions, which are priced against 8 TB/s):
torch's copy (read + write), an in-place add (read + write of one buffer), a read-only reduction and a fill (write only), at the
sizes the ADiL kernels move.  Stand-alone HIP events; buffers far beyond the 256 MB Infinity Cache are the relevant rows."""
import torch

dev = torch.device("cuda")


def timeit(fn, n=20, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print("| buffer | copy (read + write) | in-place add (read + write) | sum (read only) | fill (write only) |")
print("|---|---|---|---|---|")
for mb in (154, 308, 925, 1880, 4096):
    n = mb * 1000 * 1000 // 4
    x = torch.rand(n, device=dev)
    y = torch.empty_like(x)
    t_copy = timeit(lambda: y.copy_(x))
    t_add = timeit(lambda: x.add_(1.0))
    t_sum = timeit(lambda: x.sum())
    t_fill = timeit(lambda: y.fill_(0.5))
    b = n * 4
    print(f"| {mb} MB fp32 | {2 * b / t_copy / 1e12:.2f} TB/s ({t_copy * 1e6:.0f} us) | {2 * b / t_add / 1e12:.2f} TB/s | "
          f"{b / t_sum / 1e12:.2f} TB/s | {b / t_fill / 1e12:.2f} TB/s |", flush=True)
    del x, y
