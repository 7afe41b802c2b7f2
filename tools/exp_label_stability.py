"""Does an image's clean pseudo-label (adil.py:172: model(x).argmax) depend on the batch it is in?  VERDICT r2 next #6.

N seeded U[0,1) images (the bench's workload: a random-init network separates them by margins of the size of its own
rounding noise — the worst case for label stability), labelled once in a fixed order and then in 3 shuffled orders with
the learner's batch size (ragged last batch included: N is not a multiple of the batch size), for the bf16 FusedResNet
of the bench and for the plain fp32 network.  Reports how many labels differ from the fixed-order pass."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import engine, zoo

N, B = int(os.environ.get("N", 10000)), int(os.environ.get("B", 512))
dev = torch.device("cuda")
out = {"images": N, "batch": B, "ragged_last_batch": N % B, "shuffles": 3, "models": {}}
gen = torch.Generator().manual_seed(123)
images = torch.empty(N, 3, 224, 224, dtype=torch.bfloat16)
for lo in range(0, N, 500):
    images[lo:lo + 500] = torch.rand(min(500, N - lo), 3, 224, 224, generator=gen).to(torch.bfloat16)
for tag, kw, dt in (("bf16 FusedResNet-50 (the bench's classifier)", dict(dtype=torch.bfloat16, channels_last=True, fuse_bn_act=True, fuse_stem=True), torch.bfloat16),
                    ("fp32 plain ResNet-50", dict(), torch.float32)):
    model = zoo.build_classifier("resnet50", seed=0, device=dev, **kw)

    def labels_in_order(order):
        lab = torch.empty(N, dtype=torch.int64)
        margin = torch.empty(N)
        for lo in range(0, N, B):
            idx = order[lo:lo + B]
            with torch.no_grad():
                o = model(images[idx].to(dev).to(dt)).float()
            top = o.topk(2, dim=1).values
            lab[idx], margin[idx] = o.argmax(1).cpu(), (top[:, 0] - top[:, 1]).cpu()
        return lab, margin

    base, margin = labels_in_order(torch.arange(N))
    again, _ = labels_in_order(torch.arange(N))
    rec = {"same_order_repeat_flips": int((again != base).sum()), "median_margin": float(margin.median()),
           "images_with_margin_below_1e-3": int((margin < 1e-3).sum()), "flips_per_shuffle": [], "flipped_margins_max": 0.0}
    flipped_any = torch.zeros(N, dtype=torch.bool)
    for s in range(3):
        order = torch.randperm(N, generator=torch.Generator().manual_seed(1000 + s))
        lab, _ = labels_in_order(order)
        diff = lab != base
        flipped_any |= diff
        rec["flips_per_shuffle"].append(int(diff.sum()))
        if bool(diff.any()):
            rec["flipped_margins_max"] = max(rec["flipped_margins_max"], float(margin[diff].max()))
    rec["images_whose_label_ever_changed"] = int(flipped_any.sum())
    rec["flip_rate"] = float(flipped_any.float().mean())
    out["models"][tag] = rec
    del model
    torch.cuda.empty_cache()
print(json.dumps(out))
