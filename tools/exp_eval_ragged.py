"""End-to-end effect of engine.classifier_batch_bucket on performance.performance (the correctly-classified filter makes
every batch a different size): wall time of one evaluation pass with and without it.  Run each variant in its own
process with its own MIOpen user database (`bucket` / `nobucket` as argv[1]) so that neither inherits the other's find
results.  Not part of the product."""
import os, sys, tempfile, time
variant = sys.argv[1] if len(sys.argv) > 1 else "bucket"
name = sys.argv[2] if len(sys.argv) > 2 else "mobilenet"
tmp = tempfile.mkdtemp(prefix=f"miopen_{variant}_")
os.environ["MIOPEN_USER_DB_PATH"] = tmp
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = tmp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import performance as perf
from attacks import ADIL
from dl_attack_on_imagenet_amd import zoo, engine

dev = torch.device("cuda")
model = zoo.build_classifier(name, seed=0, device=dev)
g = torch.Generator().manual_seed(0)
n, bs, k = 240, 20, 20
x = torch.rand(n, 3, 224, 224, generator=g)
with torch.no_grad():
    y = torch.cat([engine.predict(model, x[i:i + bs].to(dev)).cpu() for i in range(0, n, bs)])
flip = torch.rand(n, generator=g) < 0.3                     # 30 % "misclassified": every batch keeps a different count
y = torch.where(flip, (y + 1) % 1000, y)
loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=bs, shuffle=False)
d = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
os.makedirs(os.path.join(tmp, "dicts"), exist_ok=True)
torch.save([d, torch.zeros(1), [], [], torch.tensor(0.)], os.path.join(tmp, "dicts", f"ImageNet_{name}.bin"))
atk = ADIL(model, eps=8 / 255, n_atoms=k, attack="supervised", model_name=name, loss="logits", steps_inference=10,
           dict_dir=os.path.join(tmp, "dicts"))
if variant == "nobucket":
    perf._loader_batch_size = lambda data: 0
sizes = set()
hook = model.register_forward_hook(lambda m, inp, out: sizes.add(inp[0].shape[0]))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = perf.performance(atk, model, loader, dev)
    torch.cuda.synchronize()
    print(f"{name} {variant} pass {rep}: {time.perf_counter() - t0:7.2f} s  classifier batch sizes seen {sorted(sizes)}  "
          f"fooling_rate {float(res['fooling_rate']):.4f} rmse {float(res['rmse']):.5f}", flush=True)
