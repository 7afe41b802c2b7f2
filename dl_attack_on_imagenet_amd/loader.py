"""The data step in front of the ADiL path: the whole training / validation set resident in HBM, batches formed by
ONE gather kernel from a device index vector.

Replaces, for the learner, the reference's `DataLoader(dataset, batch_size, shuffle=True, pin_memory=True,
num_workers=0)` (adil.py:130-133) and its per-batch `x.to(device)` (adil.py:170): there every batch is B Python-level
`dataset[i]` calls, a `torch.stack` on the host and a PCIe copy — every epoch again.  Here every image crosses PCIe
exactly once (pinned, double-buffered staging, copies on their own stream), lives in HBM in the stream dtype (50 000
images of 3x224x224: 15 GB in bf16, 30 GB in fp32 — of 288 GB), and a batch is `adil_gather_images`: B*P*2*s bytes.

The batch ORDER is the reference's: `shuffled_batches` draws from the global torch RNG exactly what a
`DataLoader(shuffle=True)` iterator draws (its base seed, then the RandomSampler's seed), so a seeded run visits the
images in the same order as upstream (checked against a real DataLoader in tests/test_cabi_host.py).
The `indexed` protocol of imagenet_loading.Subset_I (imagenet_loading.py:8-18) is what the learner consumes: a batch
is (index, x) with `index` the positions in the dataset = the rows of V."""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import torch
from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler

from . import ops

Tensor = torch.Tensor


def shuffled_batches(n: int, batch_size: int, shuffle: bool = True) -> List[List[int]]:
    """Index batches of one pass of DataLoader(range(n), batch_size, shuffle) — same global-RNG consumption, same
    order, no item is fetched."""
    if not shuffle:
        return [list(b) for b in BatchSampler(SequentialSampler(range(n)), batch_size, False)]
    torch.empty((), dtype=torch.int64).random_()          # the iterator's base seed (_BaseDataLoaderIter.__init__)
    return [list(b) for b in BatchSampler(RandomSampler(range(n)), batch_size, False)]


def _item_image(dataset, i: int) -> Tensor:
    item = dataset[i]
    return item[1] if getattr(dataset, "indexed", False) else item[0]


class _ImagesOnly(torch.utils.data.Dataset):
    """Rows `rows` of a dataset, image part only (what a worker process of the upload fetches)."""

    def __init__(self, dataset, rows):
        self.dataset, self.rows = dataset, rows

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        return _item_image(self.dataset, self.rows[i])


class ResidentImages:
    """Images `rows` of `dataset` (default: all) as one (R,C,H,W) device tensor in `dtype`.

    Upload: chunks of `chunk` images are stacked into one of two pinned staging buffers on the host, copied
    asynchronously on a side stream and converted into the resident tensor by the gather kernel (index = identity);
    the host fills the other buffer meanwhile.  The one-time upload is bound by the per-item fetch of the dataset
    (2 300 images/s for in-memory tensors, far less when every item is a JPEG to decode): `num_workers` > 0 fetches
    the items through a torch DataLoader with that many worker processes (same rows, same order, same result)."""

    def __init__(self, dataset, device, dtype: torch.dtype = torch.float32, rows: Optional[Sequence[int]] = None,
                 chunk: int = 256, num_workers: int = 0):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("ResidentImages keeps the dataset in HBM: it needs a CUDA/ROCm device (no CPU path)")
        was_indexed = getattr(dataset, "indexed", False)
        if hasattr(dataset, "indexed"):
            dataset.indexed = False
        try:
            self.rows = list(range(len(dataset))) if rows is None else [int(r) for r in rows]
            n = len(self.rows)
            first = _item_image(dataset, self.rows[0]) if n else torch.zeros(3, 8, 8)
            self.shape = tuple(first.shape)
            self.images = torch.empty((n,) + self.shape, dtype=dtype, device=device)
            if n == 0:
                return
            chunk = max(1, min(chunk, n))
            src_dtype = first.dtype if first.dtype in (torch.float32, torch.bfloat16) else torch.float32
            pinned = [torch.empty((chunk,) + self.shape, dtype=src_dtype).pin_memory() for _ in range(2)]
            staged = [torch.empty((chunk,) + self.shape, dtype=src_dtype, device=device) for _ in range(2)]
            free = [torch.cuda.Event(), torch.cuda.Event()]              # staging pair `b` may be refilled
            copy_stream = torch.cuda.Stream(device=device)
            main = torch.cuda.current_stream(device)
            fetched = None
            if num_workers > 0:                                          # items decoded / fetched by worker processes
                fetched = iter(torch.utils.data.DataLoader(_ImagesOnly(dataset, self.rows), batch_size=chunk, shuffle=False,
                                                           num_workers=int(num_workers)))
            for c, lo in enumerate(range(0, n, chunk)):
                b, hi = c & 1, min(lo + chunk, n)
                if c >= 2:
                    free[b].synchronize()                                # its previous conversion has consumed it
                if fetched is not None:
                    pinned[b][:hi - lo].copy_(next(fetched))
                else:
                    for j, r in enumerate(self.rows[lo:hi]):
                        pinned[b][j].copy_(first if (c == 0 and j == 0) else _item_image(dataset, r))
                with torch.cuda.stream(copy_stream):
                    staged[b][:hi - lo].copy_(pinned[b][:hi - lo], non_blocking=True)
                main.wait_stream(copy_stream)
                ops.gather_images(staged[b], None, out=self.images[lo:hi])
                free[b].record(main)
                copy_stream.wait_event(free[b])                          # the next copy into staged[b] waits for it
            main.synchronize()
        finally:
            if hasattr(dataset, "indexed"):
                dataset.indexed = was_indexed

    def __len__(self) -> int:
        return self.images.shape[0]

    @property
    def device(self):
        return self.images.device

    def gather(self, index, dtype: Optional[torch.dtype] = None) -> Tensor:
        """Batch (B,C,H,W) of resident rows `index` (positions in `rows`), one kernel."""
        if not isinstance(index, torch.Tensor):
            index = torch.as_tensor(list(index), dtype=torch.int64)
        index = index.to(device=self.images.device, dtype=torch.int64)
        return ops.gather_images(self.images, index, dtype=dtype or self.images.dtype)

    def batches(self, order: Iterable[Sequence[int]]) -> Iterator[Tuple[Tensor, Tensor]]:
        """(index, x) per batch of `order` (lists of resident row numbers), index as an int64 device tensor."""
        for idx in order:
            index = torch.as_tensor([int(i) for i in idx], dtype=torch.int64).to(self.images.device, non_blocking=True)
            yield index, ops.gather_images(self.images, index)


class ResidentBatches:
    """A DataLoader-like view of an evaluation set that is resident in HBM: iterating yields (x, y) DEVICE batches in
    order, each x formed by one gather kernel — what performance.performance / get_transfer_performance_aux consume
    (they use `for x, y in data`, `data.batch_size` and `len(data.dataset)`, performance.py:156-160, :207-213).
    Replaces the host DataLoader of the reference's evaluation (demo_dL_attack.py:81-86: per-item fetch, stack and a
    PCIe copy per batch, every pass again): every image crosses PCIe once, when the set is built.

    Data-parallel evaluation (one process per GPU): performance.py gives batch i to rank i % world, so with
    `shard=(rank, world)` only THIS rank's batches are uploaded and kept (50 000 images over 8 GPUs: 6 250 each); the
    batches of the other ranks are yielded as (None, None) placeholders, which the consumer skips without looking."""

    def __init__(self, dataset, labels: Tensor, batch_size: int, device, dtype: torch.dtype = torch.float32,
                 shard: Tuple[int, int] = (0, 1)):
        n, bs = len(dataset), int(batch_size)
        if labels.shape[0] != n:
            raise ValueError("ResidentBatches: one label per image")
        self.batch_size, self.shard = bs, (int(shard[0]), int(shard[1]))
        self.dataset = range(n)                                  # only its length is used (performance.py:207)
        rank, world = self.shard
        self._bounds = [(lo, min(lo + bs, n)) for lo in range(0, n, bs)]
        owned = [r for i, (lo, hi) in enumerate(self._bounds) if i % world == rank for r in range(lo, hi)]
        self.images = ResidentImages(dataset, device, dtype, rows=owned)
        self.labels = labels[torch.as_tensor(owned, dtype=torch.int64)].to(device=self.images.device, dtype=torch.int64) \
            if owned else torch.zeros(0, dtype=torch.int64, device=self.images.device)

    def __len__(self) -> int:
        return len(self._bounds)

    def __iter__(self) -> Iterator[Tuple[Optional[Tensor], Optional[Tensor]]]:
        rank, world = self.shard
        at = 0
        for i, (lo, hi) in enumerate(self._bounds):
            if i % world != rank:
                yield None, None
                continue
            index = torch.arange(at, at + hi - lo, device=self.images.device)
            at += hi - lo
            yield self.images.gather(index), self.labels[index]
