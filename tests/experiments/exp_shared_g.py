"""Experiment: teacher-forced steps with ONE shared classifier evaluation (tests/parity_tools.py): what bounds hold?"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
from dl_attack_on_imagenet_amd import engine, zoo
from oracle import adil_oracle as O
from parity_tools import shared_gradient_step
dev = "cuda"
for name, n, k, T, dt, seed in (("resnet18", 32, 10, 20, torch.float32, 21), ("resnet50", int(os.environ.get("N2", 512)), 50, int(os.environ.get("T2", 100)), torch.bfloat16, 33),
                                ("resnet50", 128, 50, 20, torch.float32, 33)):
    eps = 8 / 255
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(n, 3, 224, 224, generator=g)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    if dt == torch.bfloat16:
        model = zoo.build_classifier(name, seed=0, device=dev, dtype=dt, channels_last=True, fuse_bn_act=True, fuse_stem=True)
    else:
        model = zoo.build_classifier(name, seed=5 if name == "resnet18" else 0, device=dev)
    x = images.to(dev).to(dt).contiguous()
    index = torch.arange(n, device=dev)
    labels = engine.predict(model, x)
    d, v = d0.clone().to(dev), v0.clone().to(dev)
    sd, sv = O.AdamWState(d, 0.01), O.AdamWState(v, 0.01)
    learner = engine.DictionaryLearner(d.clone(), v.clone(), eps, 0.01, "logits", False, 50.0)
    twin = engine.DictionaryLearner(d.clone(), v.clone(), eps, 0.01, "logits", False, 50.0)
    worst = {}
    fooled = []
    for it in range(T):
        r = shared_gradient_step(O, engine, model, learner, twin, x, index, labels, d, v, sd, sv, eps, "logits")
        fooled.append((r["fooled"], r["fooled_on_oracle_synth"]))
        for key, val in r.items():
            if key not in ("fooled", "fooled_on_oracle_synth", "loss", "synth_worst"):
                worst[key] = max(worst.get(key, 0), val)
    print(json.dumps(dict(model=name, n=n, k=k, T=T, dtype=str(dt), worst=worst, fooled=fooled)), flush=True)
