"""Is the single-image attack (main.py's use) launch-bound?  ms per DDrague iteration and per learning step vs batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import engine, ops, zoo
dev = "cuda"
for name in ("mobilenet", "resnet18", "resnet50"):
    model = zoo.build_classifier(name, seed=0, device=dev)
    d = (-1 + 2 * torch.rand(3, 224, 224, 100)).to(dev)
    pinv = engine.PseudoInverse(d)
    for B in (1, 8, 32):
        x = torch.rand(B, 3, 224, 224, device=dev)
        s = engine.DDragueSolver(model, x, d, 8 / 255, "logits", pinv=pinv)
        for _ in range(3): s.iterate()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): s.iterate()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{name:10s} B={B:3d}  DDrague iteration {dt*1e3:7.2f} ms", flush=True)
