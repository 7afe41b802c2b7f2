"""Tiny frozen classifiers used by the parity tests and the golden fixtures.

Smooth activations (SiLU) on purpose: the parity tests compare gradients
computed on different devices, and a kink (ReLU) would make dLoss/dx jump for
pre-activations that round to opposite sides of zero.
"""
import numpy as np
import torch
import torch.nn as nn


class TinyNet(nn.Module):
    """(B,3,H,W) in [0,1] -> (B,n_classes) logits."""

    def __init__(self, n_classes: int = 10, width: int = 8):
        super().__init__()
        self.c1 = nn.Conv2d(3, width, 3, stride=2, padding=1)
        self.c2 = nn.Conv2d(width, 2 * width, 3, stride=2, padding=1)
        self.fc = nn.Linear(2 * width, n_classes)
        self.act = nn.SiLU()

    def forward(self, x):
        h = self.act(self.c1(x - 0.5))
        h = self.act(self.c2(h))
        return self.fc(h.mean(dim=(2, 3))) * 8.0


def make_tinynet(seed: int, n_classes: int = 10, width: int = 8) -> TinyNet:
    g = torch.Generator().manual_seed(seed)
    net = TinyNet(n_classes, width)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.6 if p.dim() > 1 else 0.1))
    net.eval()
    for p in net.parameters():
        p.requires_grad_(False)
    return net


def state_to_npz_dict(net: nn.Module, prefix: str = "net.") -> dict:
    return {prefix + k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}


def tinynet_from_npz(z, prefix: str = "net.", n_classes: int = 10, width: int = 8) -> TinyNet:
    net = TinyNet(n_classes, width)
    sd = {k[len(prefix):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(prefix)}
    net.load_state_dict(sd)
    net.eval()
    for p in net.parameters():
        p.requires_grad_(False)
    return net
