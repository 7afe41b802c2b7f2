"""Data-parallel plumbing of the ADiL learner: one process per GPU, images and their code
rows sharded, the dictionary replicated, ONE all-reduce(SUM) of grad_d per step.

Replaces env_setting.py (SLURM-derived NCCL bootstrap, env_setting.py:10-28) and the DDP
wrapper of learn_dictionary_distributed (adil.py:362-419), which cannot run as written
(SURVEY.md §2.1).  The parity target is the single-process learner at the GLOBAL batch:
SUM (not mean) reproduces the reference's `reduction='sum'` loss (adil.py:136)."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's env; initialises the default process group when
    WORLD_SIZE > 1.  backend defaults to 'nccl' (= RCCL on ROCm) when a GPU is present, else 'gloo'."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("ADIL_FORCE_REDUCER") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous ownership [lo, hi) of n items; the first n % world ranks own one extra item."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(index: List[int], rank: int, world: int) -> List[int]:
    """This rank's slice of one GLOBAL batch (contiguous rows, as §8e of SURVEY.md)."""
    lo, hi = shard_bounds(len(index), rank, world)
    return list(index[lo:hi])


class DictGradReducer:
    """The single collective of a learning step: grad_d <- sum over ranks (in place)."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("DictGradReducer needs an initialised process group (see init_from_env)")
        self.group = group
        self.world = dist.get_world_size(group)

    def all_reduce_(self, grad_d: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(grad_d, op=dist.ReduceOp.SUM, group=self.group)
        return grad_d

    def sum_scalars(self, *values) -> List[float]:
        """Per-epoch bookkeeping (loss, fooled counts): mirrors dist.reduce at adil.py:418-419."""
        dev = values[0].device if isinstance(values[0], torch.Tensor) else torch.device("cpu")
        t = torch.stack([torch.as_tensor(v, dtype=torch.float64, device=dev).reshape(()) for v in values])
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.tolist()

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        dist.broadcast(t, src=src, group=self.group)
        return t
