"""A synthetic workload with CLASS STRUCTURE for the ASR-parity tests (VERDICT r2 next #1c).

On seeded U[0,1) images a random-init classifier's logit margins are of the size of bf16 rounding, so a bf16 and an fp32
copy of the same network disagree about labels and fooling counts for reasons that have nothing to do with the attack
kernels.  Here the images are noisy copies of a few coarse colour patterns and the classifier's last layer is FITTED
to them (nearest class centroid in the frozen random backbone's standardised feature space — a closed form, seeded, a
few seconds), so clean margins are large against bf16 rounding while the deep random backbone stays as sensitive to
input perturbations as before: the attack has real work to do and its success rate is a stable aggregate."""
import os

import torch
import torch.nn.functional as F


def structured_images(n, classes=10, seed=0, size=224, noise=0.10, cells=7, draw=0):
    """(images (n,3,size,size) in [0,1], labels (n,)): class c = a coarse cells x cells colour pattern (fixed by `seed`),
    bilinearly upsampled, plus per-pixel Gaussian noise (`draw` selects an independent set of the same classes, e.g. a
    held-out evaluation split)."""
    g = torch.Generator().manual_seed(seed)
    protos = torch.rand(classes, 3, cells, cells, generator=g)
    protos = F.interpolate(protos, size=(size, size), mode="bilinear", align_corners=False) * 0.6 + 0.2
    labels = torch.arange(n) % classes
    if draw:
        g = torch.Generator().manual_seed(seed * 7919 + 104729 * draw)
    images = (protos[labels] + noise * torch.randn(n, 3, size, size, generator=g)).clamp_(0.0, 1.0)
    return images, labels


def fit_centroid_head(model, images, labels, classes, device, target_margin=10.0, chunk=64):
    """The product's closed-form head fit (zoo.fit_centroid_head; it also serves `demo_dL_attack.py --synthetic`)."""
    from dl_attack_on_imagenet_amd import zoo
    return zoo.fit_centroid_head(model, images, labels, classes, device, target_margin=target_margin, chunk=chunk)


def fitted_classifiers(name, images, labels, classes, device, tmp_dir, seed=0, target_margin=10.0):
    """(fp32 plain network, the product's bf16 network) sharing ONE set of weights incl. the fitted head: the fp32
    network is fitted, its state_dict saved, and both are rebuilt from that file through zoo.build_classifier(weights=)
    — the same route a torchvision checkpoint takes."""
    from dl_attack_on_imagenet_amd import zoo
    ref = zoo.build_classifier(name, seed=seed, device=device)
    margins, pred = fit_centroid_head(ref, images, labels, classes, device, target_margin=target_margin)
    path = os.path.join(str(tmp_dir), f"{name}_fitted.pt")
    torch.save(ref[-1].state_dict(), path)
    fused = zoo.canonical_name(name).startswith("resnet")
    # the product's configuration: bf16 head while learning, fp32 logits inside the DDrague inference loop (round 4)
    fast = zoo.build_classifier(name, seed=seed, weights=path, device=device, dtype=torch.bfloat16, channels_last=fused,
                                head_fp32="inference" if fused else False,
                                fuse_bn_act=fused, fuse_stem=fused)
    ref = zoo.build_classifier(name, seed=seed, weights=path, device=device)
    return ref, fast, margins, pred
