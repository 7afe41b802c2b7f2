"""Experiment: at which eps / iteration count do DenseNet-121 and ViT-B/16 (fitted head, 16 structured images) get fooled?"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
from dl_attack_on_imagenet_amd import engine, zoo
from oracle import adil_oracle as O
from structured import fit_centroid_head, structured_images
dev = "cuda"
for name, k in (("densenet121", 50), ("vit_b_16", 100)):
    for classes, noise in ((4, 0.15), (8, 0.25)):
        images, labels = structured_images(16, classes=classes, seed=7, noise=noise)
        model = zoo.build_classifier(name, seed=1, device=dev)
        margins, pred = fit_centroid_head(model, images, labels, classes, dev, target_margin=2.0)
        for eps in (8 / 255, 32 / 255):
            g = torch.Generator().manual_seed(5)
            d0 = (-1 + 2 * torch.rand(3, 224, 224, k, generator=g)).to(dev)
            v0 = O.project_onto_l1_ball(torch.rand(16, k, generator=g), eps).to(dev)
            learner = engine.DictionaryLearner(d0, v0, eps, 0.01, "logits", False, 50.0)
            x, index = images.to(dev), torch.arange(16, device=dev)
            fooled = [int(learner.step(model, x, index)[1]) for _ in range(60)]
            print(json.dumps(dict(model=name, classes=classes, noise=noise, eps=eps, clean_ok=bool((pred == labels).all()),
                                  margin_min=float(margins.min()), fooled=fooled)), flush=True)
