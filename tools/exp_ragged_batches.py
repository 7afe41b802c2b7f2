"""What does a NEW batch size cost the frozen classifier?  performance.py's correctly-classified filter
(performance.py:163-165) hands the attack a different number of images per batch, and every convolution configuration
MIOpen has not seen yet goes through its find / compile step.  Times the first and the second forward+backward at each
batch size (not part of the product)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import zoo, engine

dev = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "mobilenet"
dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
model = zoo.build_classifier(name, seed=0, device=dev, dtype=dtype)
sizes = [20, 19, 18, 17, 13, 20, 19, 18, 17, 13, 11, 7]
for n in sizes:
    x = torch.rand(n, 3, 224, 224, device=dev).to(dtype)
    lab = torch.zeros(n, dtype=torch.long, device=dev)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        engine.input_gradient(model, x, lab, "ce", -1.0, 50.0, "mean")
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name} {dtype} N={n:3d}  first {ts[0]:9.1f} ms   second {ts[1]:7.2f} ms   third {ts[2]:7.2f} ms", flush=True)
