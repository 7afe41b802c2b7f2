"""Host-side mirror of the reference's attacks/utils.py on the HIP kernels.

Same names, argument meaning and return conventions as the reference (cited per function);
tensors must live on the GPU — there is no CPU path (ops raises otherwise).
`fit_laplace*` (utils.py:60-150, scipy statistics for the broken ADILR class) is out of scope."""
import torch
import torch.nn as nn

from .. import ops
from .base import Attack  # noqa: F401  (re-exported like the reference's `from torchattacks.attack import Attack`)


def clamp_image(image, max_val=1, min_val=0):
    """utils.py:17-18."""
    return torch.clamp(image, max=max_val, min=min_val)


def project_onto_l1_ball(x, eps):
    """Row-wise Euclidean projection onto the l1 ball of radius eps (utils.py:21-41), one wavefront per row,
    no host sync (the reference's `rho.cpu()` is quirk Q9).  Returns a new tensor of x's shape."""
    out = x.detach().to(torch.float32).contiguous().clone()
    ops.l1ball_project_(out.view(out.shape[0], -1), float(eps))
    return out.view(x.shape)


def constraint_dict(d, constr_set='l2ball'):
    """Per-atom constraint on D (C,H,W,K), IN PLACE like the reference, and returned (utils.py:44-57): 'l2sphere',
    'l2ball', anything else = the reference's else-branch, every (channel, atom) row onto the l1 ball of radius 1."""
    tmp = d if d.is_contiguous() else d.contiguous()
    if constr_set in ('l2ball', 'l2sphere'):
        ops.atom_l2_project_(tmp, sphere=(constr_set == 'l2sphere'))
    else:
        ops.atom_l1_project_(tmp, 1.0)                                        # utils.py:55-56
    if tmp is not d:
        d.copy_(tmp)
    return d


class _SoftThreshold(nn.Module):
    def __init__(self, lambd):
        super().__init__()
        self.lambd = float(lambd)

    def forward(self, x):
        out = x.detach().to(torch.float32).contiguous().clone()
        return ops.ista_step_(out, None, 0.0, self.lambd)


def get_prox_l1(param):
    """Soft-thresholding operator, the reference's torch.nn.Softshrink(lambd=param) (utils.py:159-161)."""
    return _SoftThreshold(param)


def get_slices(n, step):
    """utils.py:153-156: consecutive index lists of length `step` covering range(n)."""
    return [list(range(lo, min(lo + step, n))) for lo in range(0, n, step)]


def get_target(img, label, targeted, classifier):
    """utils.py:164-174: the second most probable class when targeted, else the label."""
    with torch.no_grad():
        if targeted:
            return classifier(img).sort().indices[:, -2]
        return label


class QuickAttackDataset(torch.utils.data.Dataset):
    """utils.py:177-186."""

    def __init__(self, images, labels):
        self.images, self.labels = images, labels

    def __len__(self):
        return len(self.images)

    def __getitem__(self, item):
        return self.images[item], self.labels[item]


def compute_fooling_rate(dataset, attack, model, device):
    """Fooling rate of a FIXED additive perturbation `attack` over a dataset (utils.py:189-200)."""
    with torch.no_grad():
        loader = torch.utils.data.DataLoader(dataset, batch_size=128, shuffle=False)
        fooled = 0
        for x, _ in loader:
            x = x.to(device=device)
            fooled += torch.sum(model(x).argmax(dim=1) != model(x + attack).argmax(dim=1))
        return fooled / len(dataset)
