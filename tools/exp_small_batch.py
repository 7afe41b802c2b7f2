"""Is the single-image attack (main.py's use) launch-bound?  ms per DDrague iteration vs batch size, eager loop and hipGraph
replay (engine.DDragueSolver.run(use_graph=True), three iterations per launch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import engine, ops, zoo
dev = "cuda"
N = 60
for name in ("mobilenet", "resnet18", "resnet50"):
    model = zoo.build_classifier(name, seed=0, device=dev)
    d = (-1 + 2 * torch.rand(3, 224, 224, 100)).to(dev)
    pinv = engine.PseudoInverse(d)
    for B in (1, 8, 32):
        x = torch.rand(B, 3, 224, 224, device=dev)
        row = []
        for use_graph in (False, True):
            s = engine.DDragueSolver(model, x, d, 8 / 255, "logits", pinv=pinv)
            s.run(9, use_graph)                               # warm-up (and capture)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s.run(N, use_graph)
            torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / N)
        print(f"{name:10s} B={B:3d}  DDrague iteration: eager {row[0]*1e3:6.2f} ms   hipGraph {row[1]*1e3:6.2f} ms   x{row[0]/row[1]:.1f}", flush=True)
