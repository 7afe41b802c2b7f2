"""Clean top-1 accuracy of a classifier over a dataset — drop-in for the reference's model_accuracy.py, which
demo_dL_attack.py:65-66 calls before anything is attacked.

`model_accuracy(dataset, model, device)` is torchmetrics' micro-averaged `Accuracy()` of the reference (:50-63) restated
as correct / seen over batches of 128 (torchmetrics is not in this image); the value returned is a 0-d float tensor in
[0, 1], as `metric.compute()` is.  `model_accuracy_distributed` replaces the SLURM / DDP variant (:13-47): every rank
scores the images rank, rank + world, ... and the two counts are summed over the process group this package already
runs on (no DDP wrapper — there is no gradient — and no padded sampler: every image is counted exactly once, where the
reference's DistributedSampler repeats images to even out the shards and still divides by len(dataset))."""
import torch
from torch.utils.data import DataLoader, Subset


def _count_correct(dataset, model, device, batch_size):
    param = next(model.parameters(), None)
    dtype = param.dtype if param is not None else torch.float32
    model.eval()
    correct, seen = torch.zeros((), dtype=torch.int64, device=device), 0
    with torch.no_grad():
        for x, y in DataLoader(dataset, batch_size=batch_size):
            pred = model(x.to(device=device, dtype=dtype)).argmax(dim=-1)
            correct += (pred == y.to(device)).sum()
            seen += len(y)
    return correct, seen


def model_accuracy(dataset, model, device='cpu', batch_size=128):
    model = model.to(device)
    correct, seen = _count_correct(dataset, model, torch.device(device), batch_size)
    return correct.float() / max(seen, 1)


def model_accuracy_distributed(dataset, model, device, batch_size=128):
    """Inside an initialised process group (dl_attack_on_imagenet_amd.dist.init_from_env): the accuracy over the WHOLE
    dataset, identical on every rank."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    correct, seen = _count_correct(Subset(dataset, range(rank, len(dataset), world)), model, torch.device(device), batch_size)
    counts = torch.stack([correct.double(), torch.tensor(float(seen), dtype=torch.float64, device=correct.device)])
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return (counts[0] / counts[1].clamp_min(1.0)).float()
