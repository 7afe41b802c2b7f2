"""CPU, world_size 2, gloo: the data-parallel path of the learner (dl_attack_on_imagenet_amd.dist) — rendezvous from
torchrun-style env, contiguous sharding, ONE all-reduce(SUM) of grad_d per step, scalar bookkeeping.

The HIP kernels cannot run here, so each rank evaluates its shard's contributions with the ORACLE (test-side stand-in
for the kernels) and the product's reducer combines them; the result must equal the single-process oracle learner
at the GLOBAL batch (the parity target of SURVEY.md §8e): D bit-for-bit identical on both ranks, V rows owned per rank."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from dl_attack_on_imagenet_amd import dist as adist
    from oracle import adil_oracle as O
    from tinynet import make_tinynet

    r, w, lr = adist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    reducer = adist.DictGradReducer()

    # identical problem on every rank (seeded), sharded by ownership
    g = torch.Generator().manual_seed(0)
    n, k, eps, steps, gb = 16, 4, 0.5, 3, 8                     # gb = GLOBAL batch
    images = torch.rand(n, 3, 16, 16, generator=g)
    d = -1 + 2 * torch.rand(3, 16, 16, k, generator=g)
    v_all = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    net = make_tinynet(11)
    lo, hi = adist.shard_bounds(n, rank, world)
    v = v_all[lo:hi].clone()                                     # this rank's code rows
    opt_d, opt_v = O.AdamWState(d, 0.01), O.AdamWState(v, 0.01)
    # global batches = union of per-rank batches of gb/world owned images (deterministic order)
    local_batches = [[(s * (gb // world) + j) % (hi - lo) for j in range(gb // world)] for s in range(steps)]
    tot_loss, tot_fooled = 0.0, 0
    for idx in local_batches:
        index = torch.tensor(idx)
        x = images[lo:hi][index]
        with torch.no_grad():
            label = net(x).argmax(-1)
        xt = O.synth(x, d, v[index])
        out, ls, gin = O._input_grad(net, xt, label, "logits", -1.0, 50.0, "sum")
        gd, gvr = O.grad_dv(gin, d, v[index])
        reducer.all_reduce_(gd)                                  # <- the product's single collective per step
        gv = torch.zeros_like(v)
        gv[index] = gvr
        opt_d.step(d, gd)
        opt_v.step(v, gv)
        v.copy_(O.project_onto_l1_ball(v, eps))
        d.clamp_(-1, 1)
        tot_loss += float(ls)
        tot_fooled += int((out.argmax(-1) != label).sum())
    sums = reducer.sum_scalars(torch.tensor(tot_loss), torch.tensor(tot_fooled))
    d0 = d.clone()
    reducer.broadcast_(d0, 0)
    assert torch.equal(d0, d), "replicated dictionary diverged across ranks"
    torch.save(dict(d=d, v=v, lo=lo, hi=hi, sums=sums, local_batches=local_batches), os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_learning_matches_single_process_global_batch(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    sys.path.insert(0, HERE)
    from oracle import adil_oracle as O
    from tinynet import make_tinynet
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert torch.equal(res[0]["d"], res[1]["d"])

    # single-process reference at the global batch = union of the shards' batches
    g = torch.Generator().manual_seed(0)
    n, k, eps = 16, 4, 0.5
    images = torch.rand(n, 3, 16, 16, generator=g)
    d0 = -1 + 2 * torch.rand(3, 16, 16, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    net = make_tinynet(11)
    steps = len(res[0]["local_batches"])
    epochs = [[[res[r]["lo"] + i for r in range(world) for i in res[r]["local_batches"][s]]] for s in range(steps)]
    ref = O.learn_dictionary_a(net, images, d0, v0, epochs, eps, 0.01, "logits", False, 50.0)
    assert float((ref["d"] - res[0]["d"]).abs().max()) < 2e-6                     # fp32 summation order only
    v = torch.cat([res[r]["v"] for r in range(world)])
    assert float((ref["v"] - v).abs().max()) < 2e-6
    total_loss = sum(l * n for l in ref["loss_all"])
    assert abs(res[0]["sums"][0] - total_loss) < 1e-3 * max(1.0, abs(total_loss))
    assert res[0]["sums"][1] == sum(f * n for f in ref["fooling_rate_all"])


def test_init_from_env_single_process(monkeypatch):
    sys.path.insert(0, ROOT)
    from dl_attack_on_imagenet_amd import dist as adist
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert adist.init_from_env() == (0, 1, 0)
    with pytest.raises(RuntimeError):
        adist.DictGradReducer()
