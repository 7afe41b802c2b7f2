"""Experiment: configs[3] (transfer evaluation) at real image size, product vs oracle: one dictionary (learned by the bf16
product against ResNet-50), the attack against the fp32 ResNet-50, the adversary scored on the six classifiers of the
reference CLI — every network with a head fitted to the structured workload.  256 held-out images."""
import json, os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
import performance as perf
from attacks import ADIL
from dl_attack_on_imagenet_amd import engine, loader, zoo
from oracle import adil_oracle as O
from structured import fitted_classifiers, structured_images

n, k, T, S, eps, dev = 256, 50, int(os.environ.get("T", 200)), 100, 8 / 255, "cuda"
images, labels = structured_images(n, 10, seed=3)
held, held_labels = structured_images(n, 10, seed=3, draw=1)
tmp = tempfile.mkdtemp()
ref, fast, _, _ = fitted_classifiers("resnet50", images, labels, 10, dev, tmp)
targets = {}
for name in ("resnet18", "densenet121", "googlenet", "inception_v3", "mobilenet_v2", "vgg11"):
    m = zoo.build_classifier(name, seed=1, device=dev)
    margins, pred = zoo.fit_centroid_head(m, images, labels, 10, dev)
    targets[name] = m
    print(name, "fitted: accuracy", float((pred == labels).float().mean()), flush=True)
g = torch.Generator().manual_seed(33)
d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
learner = engine.DictionaryLearner(d0.to(dev), v0.to(dev), eps, 0.01, "logits", False, 50.0)
x16, index = images.to(dev).to(torch.bfloat16), torch.arange(n, device=dev)
for _ in range(T):
    learner.step(fast, x16, index)
torch.save([learner.d.cpu(), learner.v.cpu(), [], [], torch.tensor(0.)], os.path.join(tmp, "ImageNet_t.bin"))
batches = [(held[lo:lo + 64].to(dev), held_labels[lo:lo + 64].to(dev)) for lo in range(0, n, 64)]
po = O.transfer_performance(lambda xx, yy: O.forward_supervised_ddrague(ref, xx, learner.d, eps, S, "logits"), targets, batches, n)
atk = ADIL(ref, eps=eps, n_atoms=k, attack="supervised", model_name="t", loss="logits", steps_inference=S, dict_dir=tmp)
res = loader.ResidentBatches(torch.utils.data.TensorDataset(held, held_labels), held_labels, 64, dev)
pp = perf.get_transfer_performance({"adil": [atk]}, targets, res, device=torch.device(dev))["adil"]
out = {name: {"oracle": po[name], "product": {kk: float(vv) for kk, vv in pp[name].items()}} for name in targets}
print(json.dumps(out))
