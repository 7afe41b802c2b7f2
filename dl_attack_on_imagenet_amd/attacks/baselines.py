"""Comparison baselines of the reference.

`UAPPGD` (uappgd.py:29-178) is the ADiL hot path with ONE atom and the constant code 1 — the reference itself writes
its perturbation as `tensordot(ones(B,1), attack)` (uappgd.py:92,96) — so it runs on the same kernels: `adil_synth`
(K = 1) for x + e, `adil_grad` (grad_d only) for de = sum_b dLoss/dx_b, and the fused Adam + clamp / gradient step /
l2-ball projection kernels for the update.  GPU only, like the rest of the product (no CPU fallback).
`FastUAP` (fast_uap.py) stays an importable name that raises (out of scope)."""
import os

import torch
import torch.nn.functional as F

from .. import ops
from .base import Attack
from .utils import QuickAttackDataset, compute_fooling_rate


class UAPPGD(Attack):
    """UAP by projected gradient descent [Shafahi et al., 2020]; constructor signature of uappgd.py:38-39.
    Extra keyword `model_dir` (default: the reference's relative directory) says where the learned attack is kept."""

    def __init__(self, model, data_train=None, data_val=None, steps=10, batch_size=100, beta=9, step_size=0.01, norm='l2',
                 eps=.1, optimizer='adam', distributed=None, model_name=None,
                 model_dir='dict_model_ImageNet_version_constrained/'):
        super().__init__("UAPPGD", model)
        self.beta, self.steps, self.step_size, self.batch_size = beta, steps, step_size, batch_size
        self.norm, self.eps, self.optimizer = norm, eps, optimizer
        self.model_name = os.path.join(model_dir, 'UAPPGD_model_test.bin')          # uappgd.py:49-50
        self.fooling_rate = None
        if distributed:
            raise NotImplementedError("UAPPGD.learn_attack_distributed (uappgd.py:109-164) is outside the built scope")
        if not os.path.exists(self.model_name) and data_train is not None:         # uappgd.py:52-58
            self.learn_attack(dataset=data_train, val=data_val)

    # ------------------------------------------------------------------ #
    def project_(self, e):
        """UAPPGD.project (uappgd.py:60-68), in place on the (C,H,W,1) fp32 perturbation."""
        if self.norm.lower() == 'l2':
            ops.atom_l2_project_(e, sphere=False, radius=float(self.eps))           # eps * e / max(||e||, eps)
        else:
            e.clamp_(-float(self.eps), float(self.eps))
        return e

    def learn_attack(self, dataset, val=None, batches=None):
        """uappgd.py:70-107.  `batches` (list of epochs, each a list of index lists) overrides the shuffled DataLoader —
        used by the parity tests to replay the reference's order."""
        if self.device.type != "cuda":
            raise RuntimeError("UAPPGD runs on the HIP kernels only (no CPU fallback)")
        dev = self.device
        x0, _ = dataset[0]
        c, h, w = x0.shape
        e = torch.zeros(c, h, w, 1, dtype=torch.float32, device=dev)               # the single atom (uappgd.py:78)
        m, s = torch.zeros_like(e), torch.zeros_like(e)
        sched = ops.AdamWSchedule(self.step_size, weight_decay=0.0)                # torch.optim.Adam == AdamW with wd = 0
        ge = torch.empty_like(e)
        loader = None
        if batches is None:
            loader = torch.utils.data.DataLoader(dataset, batch_size=self.batch_size, shuffle=True, pin_memory=True)
        fooling_rate, self.train_fooled = [], []
        linf = self.norm.lower() != 'l2'
        for epoch in range(int(self.steps)):
            fool_s = torch.zeros((), dtype=torch.int64, device=dev)
            if batches is not None:
                it = ((torch.stack([dataset[i][0] for i in idx]), torch.as_tensor([int(dataset[i][1]) for i in idx]))
                      for idx in batches[epoch])
            else:
                it = iter(loader)
            for x, y in it:
                x, y = x.to(dev, torch.float32).contiguous(), y.to(dev)
                b = x.shape[0]
                ones = torch.ones(b, 1, dtype=torch.float32, device=dev)
                vp = ops.pack_codes(ones, None, b)
                xt = ops.synth(x, e, vp, b).requires_grad_(True)                    # x + e            (uappgd.py:96)
                with torch.enable_grad():
                    out = self.model(xt)
                    loss = torch.clamp_min(-F.cross_entropy(out.float(), y, reduction='mean'), -float(self.beta))
                    (g,) = torch.autograd.grad(loss, xt)                            # frozen classifier: input gradient only
                fool_s += (out.argmax(-1) != y).sum()
                ops.grad(g.contiguous(), e, vp, b, want_d=True, want_v=False, grad_d=ge)   # de = sum_b g_b
                if self.optimizer.lower() == 'sgd':
                    ops.ista_step_(e.view(-1), ge.view(-1), float(self.step_size), 0.0)    # e -= lr * de
                    self.project_(e)
                elif linf:
                    ops.adamw_clamp_(e, ge, m, s, sched.next(), -float(self.eps), float(self.eps))   # Adam + clamp, one pass
                else:
                    ops.adamw_clamp_(e, ge, m, s, sched.next(), -float('inf'), float('inf'))
                    self.project_(e)
            attack = e.reshape(1, c, h, w)                                          # K = 1: (C,H,W,1) is (1,C,H,W) in memory
            if val is not None:
                fooling_rate.append(compute_fooling_rate(dataset=val, attack=attack, model=self.model, device=dev))
            self.train_fooled.append(int(fool_s))
        self.attack_tensor, self.fooling_rate = e.reshape(1, c, h, w).clone(), fooling_rate
        os.makedirs(os.path.dirname(self.model_name) or ".", exist_ok=True)
        torch.save([self.attack_tensor, fooling_rate], self.model_name)            # uappgd.py:107
        return self.attack_tensor

    def forward(self, images, labels):
        """clamp(images + attack, 0, 1) (uappgd.py:166-178); learns on the given images first if nothing was learned."""
        images = images.clone().detach().to(self.device)
        if not os.path.exists(self.model_name):
            print('The UAP attack has not been learned. It is now being learned on the given dataset.')
            self.learn_attack(dataset=QuickAttackDataset(images=images.cpu(), labels=labels.cpu()), val=None)
            attack = self.attack_tensor
        else:
            attack, _ = torch.load(self.model_name)
        attack = attack.to(self.device, torch.float32)
        b = images.shape[0]
        c, h, w = attack.shape[1:]
        vp = ops.pack_codes(torch.ones(b, 1, dtype=torch.float32, device=self.device), None, b)
        return ops.synth(images.float().contiguous(), attack.reshape(c, h, w, 1).contiguous(), vp, b, pixel_clamp=True)


class FastUAP(Attack):
    def __init__(self, model, *args, **kwargs):
        raise NotImplementedError("FastUAP (fast_uap.py) is a comparison baseline outside the ADiL hot path")
