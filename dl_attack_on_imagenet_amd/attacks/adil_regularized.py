"""ISTA / soft-threshold family of the reference's attacks/attacks_classes/adil_regularized.py on the HIP
kernels: objective  sum_b coeff*CE_sum(model(x + D v), target) + 0.5*lambda_2*||D v||^2 + lambda_1*||V||_1.

The synthesis and its adjoint run through ops.dict_synth (HIP synth / grad kernels), the prox through
ops.ista_step_, the dictionary constraint through ops.atom_l2_project_.  Line-search bookkeeping (a handful of
scalars per iteration) stays on the host exactly as upstream.  Citations: adil_regularized.py.
The class ADILR is dead code upstream (its constructor always raises TypeError); the name is
kept importable and raises a clear error instead.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .base import Attack
from .utils import constraint_dict, get_prox_l1, get_slices, get_target


class Attack_dict_model(nn.Module):
    """Regularised variant: prox on v, per-atom l2 ball on d (adil_regularized.py:10-28)."""

    def __init__(self, d, v, prox):
        super().__init__()
        self.d = nn.Parameter(d)
        self.v = nn.Parameter(v)
        self.eps = prox

    def forward(self, x, index, model):
        xt = ops.dict_synth(x, self.d, self.v, index)
        return model(xt), xt - x

    def update_v(self):
        ops.ista_step_(self.v.data, None, 0.0, float(self.eps))

    def update_d(self):
        constraint_dict(self.d.data)


def _smooth_loss(model, dataset_batches, d, v, indices, coeff, l2, targeted, criterion, device):
    """sum over batches of coeff*CE_sum(model(x+Dv), target) + .5*l2*||Dv||^2 (adil_regularized.py:109-114)."""
    total = 0
    for bi, (x, y) in enumerate(dataset_batches):
        x, y = x.to(device=device), y.to(device=device)
        ind = torch.as_tensor(indices[bi], dtype=torch.int64, device=device)
        xt = ops.dict_synth(x, d, v, ind)
        dv = xt - x
        total = total + coeff * criterion(model(xt), get_target(x, y, targeted, model)) + .5 * l2 * torch.sum(dv ** 2)
    return total


def learn_coding_vectors(dataset, model, targeted=True, niter=1e2, lambda_l1=1., lambda_l2=1., batch_size=None,
                         step_size=torch.tensor(.1), n_atom=10, dict_set='l2ball', device=None, dictionary=None,
                         verbose=False):
    """ISTA on V with D fixed + backtracking (delta=.9, beta=.5, <= 11 trials) — adil_regularized.py:508-628."""
    device = dictionary.device if device is None else torch.device(device)
    n_img = len(dataset)
    delta, gamma, beta = .9, 1, .5
    batch_size = n_img if batch_size is None else batch_size
    coeff = 1. if targeted else -1.
    indices = get_slices(n_img, batch_size)
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False)
    criterion = nn.CrossEntropyLoss(reduction='sum')
    d = dictionary.to(device=device, dtype=torch.float32).contiguous()
    v = torch.zeros(n_img, d.shape[-1], device=device)
    step_size = float(step_size)
    loss_all = [np.nan]
    for _ in range(int(niter)):
        v = v.detach().requires_grad_(True)
        loss_smooth = _smooth_loss(model, loader, d, v, indices, coeff, lambda_l2, targeted, criterion, device)
        loss_old = (loss_smooth + lambda_l1 * v.abs().sum()).item()
        (grad_v,) = torch.autograd.grad(loss_smooth, v)
        with torch.no_grad():
            v_old = v.detach().clone()
            v = ops.ista_step_(v_old.clone(), grad_v.contiguous(), step_size, step_size * lambda_l1)   # :570-573
            d_v = v - v_old
            h = (torch.sum(d_v * grad_v) + .5 * (gamma / step_size) * torch.norm(d_v, 'fro') ** 2
                 + lambda_l1 * v.abs().sum() - lambda_l1 * v_old.abs().sum()).item()                  # :579-580
            index_i = 0
            while True:                                                                               # :585-620
                new_v = (v_old + (delta ** index_i) * d_v).contiguous()
                loss_full = (_smooth_loss(model, loader, d, new_v, indices, coeff, lambda_l2, targeted, criterion,
                                          device) + lambda_l1 * new_v.abs().sum()).item()
                if index_i == 0:
                    loss_cur = loss_full
                if loss_full <= loss_old + beta * (delta ** index_i) * h:
                    if loss_cur > loss_full:
                        v = new_v
                        step_size = step_size * delta ** index_i
                        loss_all.append(loss_full)
                    else:
                        loss_all.append(loss_cur)
                    break
                index_i += 1
                if index_i > 10:
                    v = new_v
                    loss_all.append(loss_full)
                    break
        if loss_all[-2] - loss_all[-1] < 1e-6:                                                        # :625
            break
    return v.detach()


def adil(dataset, model, targeted=True, niter=1e3, lambdaCoding=1., l2_fool=1., batchsize=None, step_size=.1,
         n_atom=10, dict_set='l2ball', device="cuda", dictionary=None, init_dictionary=None):
    """Full-batch forward-backward on (D, V) with a secant Lipschitz estimate and a <= 51-trial line search
    (delta=.5, beta=.5) — adil_regularized.py:31-197.  Returns (d, v, loss_all).
    `init_dictionary` (not upstream) injects the already-constrained initial D instead of the RNG draw of :78-79."""
    device = torch.device(device)
    n_img = len(dataset)
    x0, _ = next(iter(dataset))
    nc, nx, ny = x0.shape
    delta, gamma, beta = .5, 1, .5
    lipschitz = .9 / step_size
    batchsize = n_img if batchsize is None else batchsize
    coeff = 1. if targeted else -1.
    indices = get_slices(n_img, batchsize)
    loader = torch.utils.data.DataLoader(dataset, batch_size=batchsize, shuffle=False)
    criterion = nn.CrossEntropyLoss(reduction='sum')
    learn_d = dictionary is None
    if learn_d and init_dictionary is not None:
        d = init_dictionary.to(device=device, dtype=torch.float32).contiguous().clone()
    elif learn_d:
        d = constraint_dict(torch.randn(3, nx, ny, n_atom, device=device), constr_set=dict_set)      # :78-79
    else:
        d = dictionary.to(device=device, dtype=torch.float32).contiguous()
    v = torch.zeros(n_img, d.shape[-1], device=device)
    d_old, v_old = torch.zeros_like(d), torch.zeros_like(v)
    grad_v_old, grad_d_old = torch.zeros_like(v), torch.zeros_like(d)
    loss_all = np.nan * np.ones(int(niter))
    loss_non_smooth_old = 0
    flag_stop = False
    for iteration in range(int(niter)):
        if flag_stop:
            continue
        v = v.detach().requires_grad_(True)
        d = d.detach().requires_grad_(learn_d)
        loss_non_smooth = lambdaCoding * torch.sum(torch.abs(v))
        loss_smooth = _smooth_loss(model, loader, d, v, indices, coeff, l2_fool, targeted, criterion, device)
        loss_full = loss_smooth + loss_non_smooth
        if learn_d:
            grad_v, grad_d = torch.autograd.grad(loss_smooth, [v, d])
        else:
            (grad_v,) = torch.autograd.grad(loss_smooth, v)
            grad_d = torch.zeros_like(d)
        with torch.no_grad():
            v, d = v.detach(), d.detach()
            if iteration > 1:                                                                        # :126-130
                lipschitz = torch.sqrt(torch.norm(grad_v - grad_v_old, 'fro') ** 2
                                       + torch.norm(grad_d - grad_d_old, 'fro') ** 2) \
                    / torch.sqrt(torch.norm(v - v_old, 'fro') ** 2 + torch.norm(d - d_old, 'fro') ** 2)
                lipschitz = lipschitz.item()
            d_old.copy_(d); v_old.copy_(v); grad_v_old.copy_(grad_v); grad_d_old.copy_(grad_d)
            loss_old = loss_full.detach()
            step = .9 / lipschitz
            v = ops.ista_step_(v.clone(), grad_v.contiguous(), step, step * lambdaCoding)            # :141-144
            if learn_d:
                d = constraint_dict((d - step * grad_d).contiguous(), constr_set=dict_set)          # :146-147
            d_v, d_d = v - v_old, d - d_old
            h = torch.sum(d_d * grad_d) + torch.sum(d_v * grad_v) + .5 * (gamma / step) * (
                torch.norm(d_d, 'fro') ** 2 + torch.norm(d_v, 'fro') ** 2) + loss_non_smooth.detach() \
                - loss_non_smooth_old                                                                # :154-156
            index_i = 0
            while True:                                                                              # :161-192
                new_v = (v_old + (delta ** index_i) * d_v).contiguous()
                new_d = (d_old + (delta ** index_i) * d_d).contiguous()
                loss_non_smooth = lambdaCoding * torch.sum(torch.abs(new_v))
                loss_full = _smooth_loss(model, loader, new_d, new_v, indices, coeff, l2_fool, targeted, criterion,
                                         device) + loss_non_smooth
                if loss_full <= loss_old + beta * (delta ** index_i) * h:
                    v, d = new_v, new_d
                    loss_non_smooth_old = loss_non_smooth.detach()
                    break
                index_i += 1
                if index_i > 50:
                    flag_stop = True
                    break
            loss_all[iteration] = loss_full.item()
    return d.detach(), v.detach(), loss_all


def sadil(dataset, model, targeted=True, nepochs=1e3, batchsize=1, lambdaCoding=1., l2_fool=1., stepsize=1.,
          n_atom=5, dict_set='l2ball', device=torch.device("cuda"), model_file=None, init_dictionary=None):
    """Stochastic D-step / V-step per batch — adil_regularized.py:200-312.

    Upstream quirk Q13 is reproduced: v stays one leaf whose .grad is never zeroed, so the V-step uses the
    gradient accumulated over every earlier backward pass (including the D-step passes once v tracks grad)."""
    device = torch.device(device)
    nimg = len(dataset)
    x0, _ = next(iter(dataset))
    nc, nx, ny = x0.shape
    loader = torch.utils.data.DataLoader(dataset, batch_size=batchsize, shuffle=False)
    coeff = 1. if targeted else -1.
    indices = get_slices(nimg, batchsize)
    criterion = nn.CrossEntropyLoss(reduction='sum')
    if init_dictionary is not None:
        D = init_dictionary.to(device=device, dtype=torch.float32).contiguous().clone()
    else:
        D = constraint_dict(torch.randn(3, nx, ny, n_atom, device=device), constr_set=dict_set)      # :240-241
    v = torch.zeros(nimg, n_atom, device=device)
    grad_v_acc = torch.zeros_like(v)
    v_tracks_grad = False

    def total_loss():
        with torch.no_grad():
            return (_smooth_loss(model, loader, D, v, indices, coeff, l2_fool, targeted, criterion, device)
                    + lambdaCoding * torch.sum(torch.abs(v))).item()

    loss = [total_loss()]
    for _ in range(int(nepochs)):
        for bi, (x, y) in enumerate(loader):
            x, y = x.to(device=device), y.to(device=device)
            ind = torch.as_tensor(indices[bi], dtype=torch.int64, device=device)
            tgt = get_target(x, y, targeted, model)
            # ---------- D-step (:265-284)
            Dg = D.detach().requires_grad_(True)
            vg = v.detach().requires_grad_(v_tracks_grad)
            xt = ops.dict_synth(x, Dg, vg, ind)
            ls = coeff * criterion(model(xt), tgt) + .5 * l2_fool * torch.sum((xt - x) ** 2)
            if v_tracks_grad:
                grad_D, gv = torch.autograd.grad(ls, [Dg, vg])
                grad_v_acc += gv
            else:
                (grad_D,) = torch.autograd.grad(ls, Dg)
            with torch.no_grad():
                D = constraint_dict((D - stepsize * grad_D).contiguous(), constr_set=dict_set)
            # ---------- V-step (:286-304)
            v_tracks_grad = True
            vg = v.detach().requires_grad_(True)
            xt = ops.dict_synth(x, D, vg, ind)
            ls = coeff * criterion(model(xt), tgt) + .5 * l2_fool * torch.sum((xt - x) ** 2)
            (gv,) = torch.autograd.grad(ls, vg)
            grad_v_acc += gv
            with torch.no_grad():
                rows = v[ind].contiguous()
                ops.ista_step_(rows, grad_v_acc[ind].contiguous(), stepsize, stepsize * lambdaCoding)
                v[ind] = rows
        loss.append(total_loss())
        if abs(loss[-1] - loss[-2]) < 1e-6:
            break
    if model_file is not None:
        torch.save([D, loss], model_file)                                                            # :310
    return D, v, None


def sadil_updated(dataset, model, targeted=True, nepochs=1e3, batchsize=1, lambdaCoding=1., l2_fool=1., stepsize=1.,
                  n_atom=5, dict_set='l2ball', device="cuda", model_file=None, init_dictionary=None):
    """Large-scale variant: per-batch ISTA step on the codes with a backtracking probe, one dictionary step per epoch
    with a line search (delta = beta = .5, <= 5 trials) — adil_regularized.py:315-501.  Returns (D, v).

    Upstream bookkeeping is reproduced exactly (quirk Q13 and neighbours): v.grad and D.grad are never zeroed, so the
    V-step uses the gradient accumulated over every earlier backward pass, grad_D at the end of an epoch is the sum
    of all D-pass gradients plus the V-pass gradients from the 2nd batch on (:450, :461); the V backtracking result is
    discarded and only shrinks stepsize_v (:442-446, :460); inside the probe the l1 term is not scaled by lambda (:439)."""
    device = torch.device(device)
    nimg = len(dataset)
    x0, _ = next(iter(dataset))
    nc, nx, ny = x0.shape
    delta, beta = .5, .5
    loader = torch.utils.data.DataLoader(dataset, batch_size=batchsize, shuffle=False)
    coeff = 1. if targeted else -1.
    indices = get_slices(nimg, batchsize)
    stepsize_D = stepsize_v = stepsize
    criterion = nn.CrossEntropyLoss(reduction='sum')
    if init_dictionary is not None:
        D = init_dictionary.to(device=device, dtype=torch.float32).contiguous().clone()
    else:
        D = constraint_dict(torch.randn(3, nx, ny, n_atom, device=device), constr_set=dict_set)      # :358-359
    v = torch.zeros(nimg, n_atom, device=device)
    grad_v_acc, grad_D_acc = torch.zeros_like(v), torch.zeros_like(D)
    d_tracks = False
    label, pred = [], []

    def loss_all(vec, Dict):
        with torch.no_grad():
            return (_smooth_loss(model, loader, Dict, vec, indices, coeff, l2_fool, targeted, criterion, device)
                    + lambdaCoding * torch.sum(torch.abs(vec))).item()

    def batch_smooth(x, tgt, ind, vec, Dict):
        xt = ops.dict_synth(x, Dict, vec, ind)
        return coeff * criterion(model(xt), tgt) + .5 * l2_fool * torch.sum((xt - x) ** 2)

    loss = [loss_all(v, D)]
    for i_bar in range(int(nepochs)):
        i_max = 0
        for bi, (x, y) in enumerate(loader):
            x, y = x.to(device=device), y.to(device=device)
            ind = torch.as_tensor(indices[bi], dtype=torch.int64, device=device)
            if i_bar == 0:
                label = label + y.tolist()
                pred = pred + model(x).sort().indices[:, -1].tolist()
            tgt = get_target(x, y, targeted, model)
            # ---------- V pass (:393-416)
            vg = v.detach().requires_grad_(True)
            Dg = D.detach().requires_grad_(d_tracks)
            ls = batch_smooth(x, tgt, ind, vg, Dg)
            if d_tracks:
                gv, gD = torch.autograd.grad(ls, [vg, Dg])
                grad_D_acc += gD
            else:
                (gv,) = torch.autograd.grad(ls, vg)
            grad_v_acc += gv
            with torch.no_grad():
                v_old = v[ind].clone()
                loss_batch_old = (ls.detach() + lambdaCoding * torch.sum(torch.abs(v_old))).item()
                rows = v_old.clone()
                ops.ista_step_(rows, grad_v_acc[ind].contiguous(), stepsize_v, stepsize_v * lambdaCoding)
                v[ind] = rows
                # ---------- backtracking probe (:419-446)
                v_cur = rows.clone()
                loss_batch_cur = (batch_smooth(x, tgt, ind, v, D) + lambdaCoding * torch.sum(torch.abs(v_cur))).item()
                loss_batch_cur_0 = loss_batch_cur
                delta_h = (torch.sum(grad_v_acc[ind] * (v_cur - v_old))
                           + 1 / 2 / stepsize_v * torch.norm(v_cur - v_old) ** 2).item()
                i = 0
                while loss_batch_cur > loss_batch_old + delta_h * beta and i < 5:
                    i += 1
                    v[ind] = (delta ** i) * v_cur + (1 - delta ** i) * v_old
                    loss_batch_cur = (batch_smooth(x, tgt, ind, v, D) + torch.sum(torch.abs(v[ind]))).item()
                    delta_h = delta_h * delta
                if not (loss_batch_cur_0 <= loss_batch_cur):
                    i_max = max(i, i_max)
                v[ind] = v_cur
            # ---------- D pass (:448-458)
            d_tracks = True
            vg = v.detach().requires_grad_(True)
            Dg = D.detach().requires_grad_(True)
            gv, gD = torch.autograd.grad(batch_smooth(x, tgt, ind, vg, Dg), [vg, Dg])
            grad_v_acc += gv
            grad_D_acc += gD
        stepsize_v = max(stepsize_v * (delta ** i_max), 1e-5)
        grad_D = grad_D_acc
        if torch.max(torch.abs(grad_D)).item() < 1e-4:
            continue
        D_old = D.detach().clone()
        loss_i_old = loss_all(v, D_old)
        with torch.no_grad():
            D_cur = constraint_dict((D - stepsize_D * grad_D).contiguous(), constr_set=dict_set)
            loss_i_cur = loss_all(v, D_cur)
            loss_i_cur_0 = loss_i_cur
            delta_h_D = (torch.sum(grad_D * (D_cur - D_old)) + 1 / 2 / stepsize_D * torch.norm(D_cur - D_old) ** 2).item()
            i = 0
            while loss_i_cur > loss_i_old + delta_h_D * beta and i < 5:
                i += 1
                loss_i_cur = loss_all(v, ((delta ** i) * D_cur + (1 - delta ** i) * D_old).contiguous())
                delta_h_D = delta_h_D * delta
            if loss_i_cur_0 <= loss_i_cur:
                loss.append(loss_i_cur_0)
            else:
                stepsize_D = max(stepsize_D * delta ** i, 1e-6)
                loss.append(loss_i_cur)
            D = D_cur                                            # fresh tensor upstream: its gradient sum restarts
            grad_D_acc = torch.zeros_like(D)
            d_tracks = False
        if abs(loss[-1] - loss[-2]) < 1e-6:
            break
    if model_file is not None:
        torch.save([D, label, pred, v, loss], model_file)        # :499
    return D, v


class ADILR(Attack):
    """Kept importable for `from attacks import ADILR` (attacks/__init__.py:1).  Upstream the constructor always
    raises TypeError (adil_regularized.py:689 vs :722); use ADIL, or the functions of this module."""

    def __init__(self, model, *args, **kwargs):
        raise NotImplementedError(
            "ADILR is dead code in the reference (its constructor raises TypeError); the regularised ADiL "
            "algorithms are available as adil(), sadil() and learn_coding_vectors() in this module")
