"""Evaluation harness around the attack path — drop-in for the reference's performance.py
(get_args / get_atks / get_performance / performance / transfer evaluation / metrics).
The per-image squared-error sums run in one fused HIP kernel (ops.image_metrics) instead of three
elementwise passes; everything else is bookkeeping.  `select_hyperparameter` (performance.py:51-110,
offline grid bookkeeping) is out of scope."""
import itertools
import time

import numpy as np
import torch

from dl_attack_on_imagenet_amd import engine, ops
from dl_attack_on_imagenet_amd import dist as adist


def _rank_world():
    """(rank, world) of the evaluation: data-parallel over the loader's batches when a process group is initialised
    (one process per GPU, torchrun env -> dist.init_from_env), else (0, 1).  Evaluation shards trivially: the dictionary
    is replicated (every rank loads the same file), batch i is attacked and scored by rank i % world, and only the final
    sums cross ranks (BASELINE.json configs[3])."""
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        return tdist.get_rank(), tdist.get_world_size()
    return 0, 1


def _sum_over_ranks(values, device):
    """Element-wise sum of a flat list of python floats over all ranks (one small all-reduce)."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    return adist.all_reduce_(t).tolist()


def _loader_batch_size(data):
    """Batch size of a DataLoader-like object (0 = unknown: no bucketing)."""
    return int(getattr(data, "batch_size", 0) or 0)


def get_args(args):
    """All combinations of ('name', values, 'name', values, ...) as keyword dicts (performance.py:6-37;
    the reference hard-codes up to three nested loops, this handles any number)."""
    names, grids = list(args[0::2]), list(args[1::2])
    if not names:
        return [dict()]
    return [dict(zip(names, combo)) for combo in itertools.product(*grids)]


def get_atks(model, atk, *args, **kwargs):
    """One attack object per hyper-parameter combination (performance.py:40-48)."""
    atks = []
    for arg_val in get_args(args):
        kwargs.update(arg_val)
        atks.append(atk(model, **kwargs))
    return atks


def get_performance(atks, model, data, verbose=False, device=torch.device('cpu')):
    """performance.py:116-151. The result key is `{name}_atoms_{K}_loss_{loss}_` for ADiL attacks (as upstream)
    and the plain name otherwise (upstream leaves `sub_name` undefined there)."""
    out = {'fooling_rate': {}, 'rmse': {}, 'mse': {}, 'time': {}}
    for name, attack_list in atks.items():
        if verbose:
            print(name, '...')
        cols = {k: [] for k in out}
        sub_name = name
        for atk in attack_list:
            print('Attack Image learning with {}'.format(atk))
            if name == 'adil':
                sub_name = f'{name}_atoms_{atk.n_atoms}_loss_{atk.loss}_'
            start = time.time()
            perf_tmp = performance(attack=atk, model=model.to(device=device), data=data, device=device)
            elapsed = time.time() - start
            print('time costing {}s'.format(elapsed))
            print('performance: {}'.format(perf_tmp))
            for k in ('fooling_rate', 'rmse', 'mse'):
                cols[k].append(perf_tmp[k])
            cols['time'].append(elapsed)
        for k in out:
            out[k][sub_name] = cols[k]
    return out


def performance(attack, model, data, device=torch.device('cpu')):
    """Attack the correctly-classified samples of every batch and average fooling / rmse / mse over them
    (performance.py:154-177)."""
    num_samples, fooling, rmse, mse = 0, 0, 0, 0
    device = attack.device
    rank, world = _rank_world()
    # the filter below gives every batch its own size; the classifier still only ever runs at the loader's batch size
    # (engine.classifier_batch_bucket: a new size costs MIOpen seconds of find / compile on this stack)
    with engine.classifier_batch_bucket(_loader_batch_size(data)):
        for i, (x, y) in enumerate(data):
            if i % world != rank:
                continue
            x, y = x.to(device=device), y.to(device=device)
            keep = engine.predict(model, x) == y                             # performance.py:162-164
            x, y = x[keep].contiguous(), y[keep]
            num_samples += torch.sum(keep)
            adversary = attack(x, y)
            if isinstance(adversary, tuple):                                 # unsupervised attack returns a tuple
                adversary = adversary[0]
            adversary = adversary.detach()
            fooling += compute_fooling_rate(model=model.eval(), adversary=adversary, clean=x)
            r, m = _rmse_mse(adversary, x)
            rmse += r
            mse += m
    if world > 1:
        num_samples, fooling, rmse, mse = _sum_over_ranks([float(num_samples), fooling, rmse, mse], device)
    print(num_samples)
    return {"fooling_rate": fooling / num_samples, "rmse": rmse / num_samples, "mse": mse / num_samples}


def get_transfer_performance(atks, models, data, device=torch.device('cpu')):
    """performance.py:183-195."""
    perf_transfer = dict()
    for name in atks.keys():
        if len(atks[name]) > 0:
            perf_transfer[name] = get_transfer_performance_aux(atks[name][0], models, data=data, device=device)
        else:
            perf_transfer[name] = empty_transfer_performance(models)
    return perf_transfer


def empty_transfer_performance(model_transfer):
    return {name: {'fooling_rate': np.nan, 'rmse': np.nan, 'mse': np.nan} for name in model_transfer.keys()}


def get_transfer_performance_aux(attack, model_transfer, data, device=torch.device('cpu')):
    """The adversary is computed ONCE per batch against the source model, then every target model is evaluated on
    it; sums are divided by the dataset size (performance.py:205-232)."""
    num_samples = len(data.dataset)
    perf = {name: {'fooling_rate': 0., 'rmse': 0., 'mse': 0.} for name in model_transfer.keys()}
    rank, world = _rank_world()
    with engine.classifier_batch_bucket(_loader_batch_size(data)):       # the last batch is usually smaller
        for i, (x, y) in enumerate(data):
            if i % world != rank:
                continue
            x, y = x.to(device=device), y.to(device=device)
            adversary = attack(x, y)
            if isinstance(adversary, tuple):
                adversary = adversary[0]
            adversary = adversary.detach()
            r, m = _rmse_mse(adversary, x)
            for model_name, target in model_transfer.items():
                target = target.to(device=device)
                perf[model_name]['fooling_rate'] += compute_fooling_rate(model=target, adversary=adversary,
                                                                         clean=x) / num_samples
                perf[model_name]['rmse'] += r / num_samples
                perf[model_name]['mse'] += m / num_samples
    if world > 1:
        keys = [(name, k) for name in perf for k in ('fooling_rate', 'rmse', 'mse')]
        for (name, k), v in zip(keys, _sum_over_ranks([perf[n][k] for n, k in keys], device)):
            perf[name][k] = v
    return perf


# ---------- metrics (performance.py:238-266) ----------- #
def _rmse_mse(adversary, clean):
    se, sn = ops.image_metrics(adversary.contiguous(), clean.contiguous().to(adversary.dtype))
    return torch.sum(se / sn).item(), torch.sum(se).item()


def compute_fooling_rate(model, adversary, clean, reduction='sum'):
    different = engine.predict(model.eval(), clean) != engine.predict(model.eval(), adversary)
    return different.float().sum().item() if reduction == 'sum' else different.float().mean().item()


def compute_rmse(adversary, clean, reduction='sum'):
    se, sn = ops.image_metrics(adversary.contiguous(), clean.contiguous().to(adversary.dtype))
    ratio = se / sn
    return torch.sum(ratio).item() if reduction == 'sum' else torch.mean(ratio).item()


def compute_mse(adversary, clean, reduction='sum'):
    se, _ = ops.image_metrics(adversary.contiguous(), clean.contiguous().to(adversary.dtype))
    return torch.sum(se).item() if reduction == 'sum' else torch.mean(se).item()
