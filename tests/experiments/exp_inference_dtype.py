"""Experiment: with ONE dictionary (learned by the bf16 product, 300 iterations), which part of the bf16 configuration
costs attack success at inference — the bf16 image streams or the bf16 classifier?  1024 held-out structured images."""
import json, os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
import performance as perf
from attacks import ADIL
from dl_attack_on_imagenet_amd import engine, loader
from oracle import adil_oracle as O
from structured import fitted_classifiers, structured_images

n, k, T, S, eps, dev = 512, 50, int(os.environ.get("T", 300)), 100, 8 / 255, "cuda"
images, labels = structured_images(n, 10, seed=3)
held, held_labels = structured_images(1024, 10, seed=3, draw=1)
tmp = tempfile.mkdtemp()
ref, fast, margins, pred = fitted_classifiers("resnet50", images, labels, 10, dev, tmp)
g = torch.Generator().manual_seed(33)
d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
learner = engine.DictionaryLearner(d0.to(dev), v0.to(dev), eps, 0.01, "logits", False, 50.0)
x16, index = images.to(dev).to(torch.bfloat16), torch.arange(n, device=dev)
for _ in range(T):
    learner.step(fast, x16, index)
torch.save([learner.d.cpu(), learner.v.cpu(), [], [], torch.tensor(0.)], os.path.join(tmp, "ImageNet_x.bin"))


class AsFp32(torch.nn.Module):
    def __init__(self, net):
        super().__init__(); self.net = net
    def forward(self, x):
        return self.net(x.to(torch.bfloat16)).float()


ds = torch.utils.data.TensorDataset(held, held_labels)
out = {}
batches = [(held[lo:lo + 128].to(dev), held_labels[lo:lo + 128].to(dev)) for lo in range(0, 1024, 128)]
out["oracle inference, fp32 streams, fp32 net"] = O.performance(lambda xx, yy: O.forward_supervised_ddrague(ref, xx, learner.d, eps, S, "logits"), ref, batches)["fooling_rate"]
from dl_attack_on_imagenet_amd import zoo
plain16 = zoo.build_classifier("resnet50", seed=0, weights=os.path.join(tmp, "resnet50_fitted.pt"), device=dev, dtype=torch.bfloat16)
for tag, model, sdt in (("product inference, fp32 streams, fp32 net", ref, None),
                        ("product inference, bf16 streams, PLAIN PyTorch bf16 net (no kernels of this repo in the classifier)", plain16, torch.bfloat16),
                        ("product inference, fp32 streams, bf16 net behind an fp32 interface", AsFp32(fast), None),
                        ("product inference, bf16 streams, bf16 net (configs[1])", fast, torch.bfloat16)):
    atk = ADIL(model, eps=eps, n_atoms=k, attack="supervised", model_name="x", loss="logits", steps_inference=S, dict_dir=tmp, stream_dtype=sdt)
    res = loader.ResidentBatches(ds, held_labels, 128, dev, sdt or torch.float32)
    out[tag] = float(perf.performance(atk, model, res)["fooling_rate"])
print(json.dumps(out))
