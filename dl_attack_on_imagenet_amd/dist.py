"""Data-parallel plumbing of the ADiL learner: one process per GPU, images and their code
rows sharded, the dictionary replicated, ONE all-reduce(SUM) of grad_d per step.

Replaces env_setting.py (SLURM-derived NCCL bootstrap, env_setting.py:10-28) and the DDP
wrapper of learn_dictionary_distributed (adil.py:362-419), which cannot run as written
(SURVEY.md §2.1).  The parity target is the single-process learner at the GLOBAL batch:
SUM (not mean) reproduces the reference's `reduction='sum'` loss (adil.py:136)."""
from __future__ import annotations

import os
import warnings
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def local_device_index(local_rank: int) -> int:
    """The GPU of this rank: its local rank, one process per GPU.  ADIL_SHARE_GPU=1 (rehearsals on a box with fewer
    GPUs than ranks, together with ADIL_DIST_BACKEND=gloo — RCCL refuses two ranks on one device) wraps around."""
    if os.environ.get("ADIL_SHARE_GPU") == "1":
        return local_rank % max(1, torch.cuda.device_count())
    return local_rank


IPC_ENV = "HSA_ENABLE_IPC_MODE_LEGACY"


def init_from_env(backend: Optional[str] = None, slurm: bool = False) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's env; initialises the default process group when
    WORLD_SIZE > 1.  backend: argument, else $ADIL_DIST_BACKEND, else 'nccl' (= RCCL on ROCm) when a GPU is
    present, else 'gloo'.

    One environment for every launch form (bench.py / demo_dL_attack.py starting their own ranks, or a driver calling
    `python -m torch.distributed.run` itself): HSA_ENABLE_IPC_MODE_LEGACY=0 is set HERE, before this process makes its
    first HIP call, unless the caller exported a value.  Source: the deployment notes of this MI355X pool — the host
    driver only supports dmabuf IPC, and with the legacy IPC mode RCCL's (and torch's) cross-process sharing of device
    memory fails with `hipIpcGetMemHandle: invalid argument`.  The ROCr runtime reads the variable when it is
    initialised, so it has to be in place before `torch.cuda.*` touches the device — callers invoke init_from_env first.

    slurm=True: under plain `srun` (no torchrun variables) take rank / world size / rendezvous from SLURM's task variables
    first (`adopt_slurm_env`) — the reference's launch form; never implied, so that a single-process run that merely sits inside
    a multi-task allocation is not turned into rank 0 of N."""
    if slurm:                            # opt-in (the demo's --distributed): a bench run inside a SLURM allocation stays one process
        adopt_slurm_env()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("ADIL_FORCE_REDUCER") == "1":
        if IPC_ENV not in os.environ and torch.cuda.is_initialized():
            # the ROCr runtime read its environment at this process's first HIP call, which has already
            # happened (e.g. the classifier was moved to the GPU before ADIL.learn_dictionary_distributed): setting the
            # variable now cannot take effect any more
            warnings.warn(f"{IPC_ENV} was not set when this process initialised the GPU; on hosts whose driver only supports "
                          f"dmabuf IPC, RCCL then fails with `hipIpcGetMemHandle: invalid argument`.  Export {IPC_ENV}=0 "
                          "in the launcher's environment, or call dist.init_from_env() before the first torch.cuda call.",
                          RuntimeWarning)
        os.environ.setdefault(IPC_ENV, "0")
    if (world > 1 or os.environ.get("ADIL_FORCE_REDUCER") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = os.environ.get("ADIL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_device_index(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def first_host(nodelist: str) -> str:
    """First host of a SLURM node list: 'gpu[017-020,031],login1' -> 'gpu017' (what hostlist.expand_hostlist(...)[0] gives
    the reference, env_setting.py:10-11; the `hostlist` package is not in this image)."""
    head = nodelist.strip()
    lb = head.find("[")
    comma = head.find(",")
    if lb < 0 or (0 <= comma < lb):
        return head.split(",", 1)[0]
    rb = head.index("]", lb)
    first = head[lb + 1:rb].split(",", 1)[0].split("-", 1)[0]
    tail = head[rb + 1:].split(",", 1)[0]
    return head[:lb] + first + tail


def adopt_slurm_env() -> bool:
    """Under `srun` without torchrun (how the reference launches, env_setting.py:7-16): translate SLURM's task variables into
    the RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT this package reads — rank SLURM_PROCID of SLURM_NTASKS,
    device SLURM_LOCALID, rendezvous at the first host of the job's node list, port 12345 + the lowest GPU id of the step
    (env_setting.py:25).  Nothing is touched when torchrun's variables are present."""
    env = os.environ
    if "WORLD_SIZE" in env or "RANK" in env or "SLURM_NTASKS" not in env or "SLURM_PROCID" not in env:
        return False
    env["WORLD_SIZE"], env["RANK"] = env["SLURM_NTASKS"], env["SLURM_PROCID"]
    env["LOCAL_RANK"] = env.get("SLURM_LOCALID", "0")
    nodes = env.get("SLURM_JOB_NODELIST") or env.get("SLURM_STEP_NODELIST")
    if nodes:
        env.setdefault("MASTER_ADDR", first_host(nodes))
    gpus = [int(g) for g in env.get("SLURM_STEP_GPUS", "").split(",") if g.strip().isdigit()]
    env.setdefault("MASTER_PORT", str(12345 + (min(gpus) if gpus else 0)))
    return True


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous ownership [lo, hi) of n items; the first n % world ranks own one extra item."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(index: List[int], rank: int, world: int) -> List[int]:
    """This rank's slice of one GLOBAL batch (contiguous rows, as §8e of SURVEY.md)."""
    lo, hi = shard_bounds(len(index), rank, world)
    return list(index[lo:hi])


def owned_rows(index, lo: int, hi: int) -> List[int]:
    """The members of one GLOBAL batch whose images (and code rows) this rank owns, in batch order."""
    return [int(i) for i in index if lo <= int(i) < hi]


def global_epoch_batches(n: int, batch_size: int, world: int, seed: int, epoch: int) -> List[List[int]]:
    """The GLOBAL batches of one epoch, computed identically on every rank (seeded, no communication).

    Every rank shuffles its own contiguous shard and global batch s is the union of the ranks' s-th chunks of
    batch_size // world rows, so a global batch is balanced across ranks by construction and all ranks take the SAME
    number of steps, ceil(largest shard / chunk): a rank whose shard is exhausted contributes an empty chunk and
    still joins that step's all-reduce (no collective can ever pair with a different one)."""
    if batch_size < world:
        warnings.warn(f"global batch {batch_size} < {world} ranks: every rank still takes one image per step, "
                      f"so the global batch is {world}", stacklevel=2)
    # rank r takes batch_size // world rows per step and the remainder batch_size % world is dealt out round-robin,
    # continuing from step to step (8 ranks, batch 100: ranks 0-3 take 13 in even steps, ranks 4-7 in odd ones), so the
    # global batch is `batch_size` exactly and equal shards run out together: 1000 images = 10 steps of 100, like the
    # single-process learner (batch_size // world alone gave 11 steps of 96)
    base, rem = divmod(batch_size, world)
    if base == 0:
        base, rem = 1, 0
    perms = []
    for r in range(world):
        lo, hi = shard_bounds(n, r, world)
        g = torch.Generator().manual_seed(1_000_003 * (seed + 1) + 7919 * epoch + r)
        perms.append((lo + torch.randperm(hi - lo, generator=g)).tolist())
    taken = [0] * world
    batches, s = [], 0
    while any(taken[r] < len(perms[r]) for r in range(world)):
        first = (s * rem) % world
        batch = []
        for r in range(world):
            c = base + (1 if (r - first) % world < rem else 0)
            batch += perms[r][taken[r]:taken[r] + c]
            taken[r] += c
        batches.append(batch)
        s += 1
    return batches


def _staged(t: torch.Tensor, group=None) -> bool:
    """gloo rehearsals (ADIL_DIST_BACKEND=gloo) move device tensors through the host; RCCL works on them in place."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_(t: torch.Tensor, op=dist.ReduceOp.SUM, group=None) -> torch.Tensor:
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


class DictGradReducer:
    """The single collective of a learning step: grad_d <- sum over ranks (in place)."""

    def __init__(self, group=None, timing: bool = False):
        if not dist.is_initialized():
            raise RuntimeError("DictGradReducer needs an initialised process group (see init_from_env)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        # bench.py: HIP-event bracket around every start -> wait.  ADIL_REDUCER_TIMING=1 switches it on for a learner that
        # builds its own reducer (ADIL.learn_dictionary_distributed), so that any multi-rank run can say what its
        # collective cost (tests/dist_learn_worker.py records it)
        self.timing = bool(timing) or os.environ.get("ADIL_REDUCER_TIMING") == "1"
        self._brackets, self.bytes_per_call = [], 0
        self.saw_async_work = False              # the asynchronous (RCCL) branch of all_reduce_start has been taken

    # -- measurement (bench.py's config.collective) ------------------------------------------------------------------ #
    def describe(self) -> dict:
        """Backend, world size and — with timing on — the mean time between the start of the step's all-reduce and the
        point where the compute stream has passed its wait (HIP events on the compute stream: it contains the AdamW +
        projection of the code rows that runs underneath, i.e. it is an upper bound of what the collective can cost a
        step), over the brackets recorded so far.  Call after a device synchronisation."""
        out = {"backend": self.backend, "world_size": self.world, "bytes_per_allreduce": self.bytes_per_call,
               IPC_ENV: os.environ.get(IPC_ENV)}
        if self._brackets:
            ms = [a.elapsed_time(b) for a, b in self._brackets]
            out["allreduce_ms_start_to_wait_mean"] = sum(ms) / len(ms)
            out["allreduce_ms_start_to_wait_max"] = max(ms)
            out["allreduce_brackets"] = len(ms)
        return out

    def reset_timing(self) -> None:
        self._brackets = []

    def all_reduce_(self, grad_d: torch.Tensor) -> torch.Tensor:
        return all_reduce_(grad_d, dist.ReduceOp.SUM, self.group)

    def all_reduce_start(self, grad_d: torch.Tensor):
        """Start the step's collective and return a handle whose .wait() orders the CURRENT stream behind it (RCCL runs on
        its own stream): the caller puts the work that does not need the reduced gradient — AdamW + projection of the code
        rows — between start and wait.  The gloo rehearsal path reduces synchronously (its handle's wait() only closes the
        timing bracket)."""
        self.bytes_per_call = grad_d.numel() * grad_d.element_size()
        e0 = None
        if self.timing and grad_d.is_cuda:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if _staged(grad_d, self.group):
            all_reduce_(grad_d, dist.ReduceOp.SUM, self.group)
            work = None
        else:
            work = dist.all_reduce(grad_d, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.saw_async_work = True
        return _Pending(self, work, e0)

    def max_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place MAX over ranks of a small tensor (the stop slot of a sharded solver)."""
        return all_reduce_(t, dist.ReduceOp.MAX, self.group)

    def _closed(self, e0) -> None:
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._brackets.append((e0, e1))

    def sum_scalars(self, *values) -> List[float]:
        """Per-epoch bookkeeping (loss, fooled counts): mirrors dist.reduce at adil.py:418-419."""
        dev = values[0].device if isinstance(values[0], torch.Tensor) else torch.device("cpu")
        t = torch.stack([torch.as_tensor(v, dtype=torch.float64, device=dev).reshape(()) for v in values])
        return all_reduce_(t, dist.ReduceOp.SUM, self.group).tolist()

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if _staged(t, self.group):
            h = t.cpu()
            dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src, group=self.group)
        return t

    def gather_rows(self, rows: torch.Tensor, counts: List[int]) -> torch.Tensor:
        """Concatenate the ranks' row blocks (counts[r] rows on rank r, any sizes): blocks are padded to the largest
        count for the all-gather (RCCL needs equal shapes) and trimmed afterwards.  Used once, when saving V."""
        width = max(counts)
        pad = torch.zeros((width,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        pad[:rows.shape[0]] = rows
        if _staged(pad, self.group):
            pad = pad.cpu()
        parts = [torch.empty_like(pad) for _ in counts]
        dist.all_gather(parts, pad, group=self.group)
        return torch.cat([p[:c] for p, c in zip(parts, counts)]).to(rows.device)


class _Pending:
    """Handle of a started all-reduce: .wait() orders the current stream behind it (the host does not block)."""
    __slots__ = ("reducer", "work", "e0")

    def __init__(self, reducer: DictGradReducer, work, e0):
        self.reducer, self.work, self.e0 = reducer, work, e0

    def wait(self) -> None:
        if self.work is not None:
            self.work.wait()
        self.reducer._closed(self.e0)
