"""Where does ViT-B/16 at 512 images spend its time / stop?  Prints a timestamp after every phase (flushed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import engine, ops, zoo
B = int(os.environ.get("B", 512)); K = int(os.environ.get("K", 100))
t0 = time.time()
def mark(s):
    torch.cuda.synchronize(); print(f"{time.time() - t0:7.2f}s  {s}", flush=True)
dev = "cuda"
model = zoo.build_classifier("vit_b_16", seed=0, device=dev, dtype=torch.bfloat16)
mark("model built")
x = torch.rand(B, 3, 224, 224, device=dev).to(torch.bfloat16)
with torch.no_grad():
    y = model(x[:8])
mark("forward B=8")
with torch.no_grad():
    y = model(x)
mark(f"forward B={B} no-grad")
xt = x.clone().requires_grad_(True)
out = model(xt)
mark(f"forward B={B} with graph; mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
(g,) = torch.autograd.grad(out.float().sum(), xt)
mark("backward")
d = (-1 + 2 * torch.rand(3, 224, 224, K, device=dev))
v = ops.l1ball_project_(torch.rand(B, K, device=dev), 8 / 255)
learner = engine.DictionaryLearner(d, v, 8 / 255, 0.01, "logits")
idx = torch.arange(B, device=dev)
for i in range(3):
    learner.step(model, x, idx)
    mark(f"learner step {i}")
