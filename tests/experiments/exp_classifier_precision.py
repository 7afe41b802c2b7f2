"""Experiment: how far are the two bf16 classifiers (plain PyTorch modules; zoo.FusedResNet on this repo's stem / pointwise /
3x3 kernels) from the fp32 network — logits and the input gradient of the attack loss — on the structured workload?"""
import json, os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
from dl_attack_on_imagenet_amd import engine, zoo
from structured import fitted_classifiers, structured_images
dev = "cuda"
images, labels = structured_images(256, 10, seed=3)
tmp = tempfile.mkdtemp()
ref, fast, margins, pred = fitted_classifiers("resnet50", images, labels, 10, dev, tmp)
plain16 = zoo.build_classifier("resnet50", seed=0, weights=os.path.join(tmp, "resnet50_fitted.pt"), device=dev, dtype=torch.bfloat16)
fused_nostem = zoo.build_classifier("resnet50", seed=0, weights=os.path.join(tmp, "resnet50_fitted.pt"), device=dev, dtype=torch.bfloat16,
                                    channels_last=True, fuse_bn_act=True, fuse_stem=False)
g0 = torch.Generator().manual_seed(1)
x = (images + 0.02 * torch.randn(images.shape, generator=g0)).clamp(0, 1).to(dev)     # slightly perturbed images
lab = labels.to(dev)
def run(model, xin):
    out, ls, g = engine.input_gradient(model, xin, lab, "logits", -1.0, 50.0, "sum")
    return out.float(), g.float()
o32, g32 = run(ref, x)
res = {}
for tag, m in (("plain PyTorch bf16", plain16), ("FusedResNet bf16 (stem kernels)", fast), ("FusedResNet bf16 (no stem kernels)", fused_nostem)):
    o, g = run(m, x.to(torch.bfloat16))
    cos = torch.nn.functional.cosine_similarity(g.flatten(1), g32.flatten(1), dim=1)
    res[tag] = dict(logit_rel_err=float((o[:, :10] - o32[:, :10]).abs().max() / o32[:, :10].abs().max()),
                    grad_rel_l2=float((g - g32).flatten(1).norm(dim=1).div(g32.flatten(1).norm(dim=1)).median()),
                    grad_cosine_median=float(cos.median()), grad_cosine_min=float(cos.min()),
                    grad_norm_ratio_median=float(g.flatten(1).norm(dim=1).div(g32.flatten(1).norm(dim=1)).median()))
print(json.dumps(res, indent=1))
