"""CPU, world_size 2, gloo: the data-parallel learner of the PRODUCT — `ADIL.learn_dictionary_distributed` itself
(dl_attack_on_imagenet_amd/attacks/adil.py): rendezvous from torchrun-style env, image / code-row ownership, global
batches filtered by ownership (ragged shards, a rank that owns nothing of a batch), ONE all-reduce(SUM) of grad_d per
step through the product's DictGradReducer, the equal step count on every rank, scalar bookkeeping, sharded
validation, the padded gather of V and the saved file.

The HIP kernels cannot run here, so the three places the learner touches them are swapped for test-side stand-ins
built on the ORACLE (the (D, V) update, the validation solver, the resident image store); everything else is the
shipped code.  The result must equal the single-process oracle learner at the GLOBAL batches (the parity target of
SURVEY.md §8e): D bit-for-bit identical on both ranks, V rows owned per rank."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

N_IMG, N_VAL, K, EPS, STEPS, BATCH = 17, 7, 4, 0.5, 3, 6          # 17 and 7 do not divide by 2: ragged shards
# explicit global batches (second scenario): batch 0 is owned by rank 0 only, batch 1 by rank 1 only
EXPLICIT = [[[0, 1, 2, 3], [9, 10, 11, 12, 13], [4, 16, 8]], [[5, 6, 14, 15, 7], [3, 2, 1], [16, 9, 0]]]
EXPLICIT_VAL = [[[0, 1, 2], [3, 4, 5, 6]], [[6, 0, 3], [1, 2, 4, 5]]]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(n_img=N_IMG, n_val=N_VAL):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from oracle import adil_oracle as O
    from tinynet import make_tinynet
    g = torch.Generator().manual_seed(0)
    images = torch.rand(n_img, 3, 16, 16, generator=g)
    val = torch.rand(n_val, 3, 16, 16, generator=g)
    d0 = -1 + 2 * torch.rand(3, 16, 16, K, generator=g)
    v0raw = torch.rand(n_img, K, generator=g)
    return O, make_tinynet(11), images, val, d0, v0raw


# world 8 = the target node: 205 images over 8 ranks (shards of 26 and 25), global batch 100 = 13 / 12 images
# per rank and step, a last step of 5 images of which most ranks own nothing; 5 validation images: three ranks own no
# validation image at all and still take part in the sharded validation's collectives
N_IMG8, N_VAL8, BATCH8, STEPS8 = 205, 5, 100, 2


class _Indexed(torch.utils.data.Dataset):
    def __init__(self, images):
        self.images, self.indexed = images, False

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return (i, self.images[i], 0) if self.indexed else (self.images[i], 0)


def _install_oracle_backend(O):
    """Swap the HIP-touching pieces of the learner for oracle-backed stand-ins (test side only)."""
    from dl_attack_on_imagenet_amd import engine, ops
    from dl_attack_on_imagenet_amd.attacks import adil as A

    class OracleLearner:
        """Interface of engine.DictionaryLearner; maths by the oracle, collective by the product's reducer."""

        def __init__(self, d, v, eps, step_size=0.01, loss="ce", targeted=False, kappa=50.0, lr_d=None, lr_v=None,
                     reducer=None):
            self.d, self.v, self.eps, self.loss, self.kappa = d, v, eps, loss, kappa
            self.coeff = 1.0 if targeted else -1.0
            self.opt_d, self.opt_v = O.AdamWState(d, step_size), O.AdamWState(v, step_size)
            self.reducer = reducer
            self.collectives = 0

        def step(self, model, x, index, labels=None):
            if x.shape[0]:
                with torch.no_grad():
                    label = model(x).argmax(-1) if labels is None else labels     # ADIL(cache_labels=True) hands them in
                out, ls, gin = O._input_grad(model, O.synth(x, self.d, self.v[index]), label, self.loss, self.coeff,
                                             self.kappa, "sum")
                gd, gvr = O.grad_dv(gin, self.d, self.v[index])
                fooled = (out.argmax(-1) != label).sum()
            else:
                gd, gvr, ls, fooled = torch.zeros_like(self.d), None, torch.zeros(()), torch.zeros((), dtype=torch.int64)
            self.reducer.all_reduce_(gd)                          # <- the product's single collective per step
            self.collectives += 1
            gv = torch.zeros_like(self.v)
            if gvr is not None:
                gv[index] = gvr
            self.opt_d.step(self.d, gd)
            if self.v.shape[0]:
                self.opt_v.step(self.v, gv)
                self.v.copy_(O.project_onto_l1_ball(self.v, self.eps))
            self.d.clamp_(-1, 1)
            return ls, fooled

    class HostImages:
        """Interface of loader.ResidentImages on host tensors."""

        def __init__(self, dataset, device, dtype=torch.float32, rows=None, chunk=256, num_workers=0):
            dataset.indexed = False
            self.rows = list(range(len(dataset))) if rows is None else list(rows)
            self.images = torch.stack([dataset[r][0] for r in self.rows]) if self.rows else torch.zeros(0, 3, 16, 16)

        def __len__(self):
            return len(self.rows)

        def gather(self, index, dtype=None):
            return self.images[torch.as_tensor(list(index), dtype=torch.int64)]

        def batches(self, order):
            for idx in order:
                index = torch.as_tensor([int(i) for i in idx], dtype=torch.int64)
                yield index, self.images[index]

    def solve_codes(model, images, d, eps, loss="ce", targeted=False, kappa=50.0, norm="linf", mode="train",
                    max_iter=100, labels=None, return_codes=False, mean_over=None, reducer=None):
        assert loss == "logits", "the stand-in ignores mean_over, which only matters for the mean-reduced CE"
        if images.shape[0] == 0:
            return torch.zeros((), dtype=torch.int64)
        return torch.as_tensor(O.forward_supervised_adamw(model, images, d, eps, loss=loss, targeted=targeted, kappa=kappa,
                                                          norm=norm, mode=mode))

    A.ADIL._learner_cls = OracleLearner
    A.ResidentImages = HostImages
    engine.solve_codes_adamw = solve_codes
    ops.l1ball_project_ = lambda x, r: x.copy_(O.project_onto_l1_ball(x, r))


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    O, net, images, val, d0, v0raw = _problem()
    from dl_attack_on_imagenet_amd import dist as adist
    from dl_attack_on_imagenet_amd.attacks.adil import ADIL

    r, w, _ = adist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    _install_oracle_backend(O)
    for tag, explicit, vexplicit in (("seeded", None, None), ("explicit", EXPLICIT, EXPLICIT_VAL)):
        steps = STEPS if explicit is None else len(explicit)
        atk = ADIL(net, eps=EPS, steps=steps, n_atoms=K, batch_size=BATCH, data_train=_Indexed(images),
                   data_val=_Indexed(val), model_name=f"dist_{tag}", step_size=0.01, is_distributed=True, loss="logits",
                   kappa=50.0, init_d=d0, init_v=v0raw, epoch_batches=explicit, val_batches=vexplicit,
                   dict_dir=os.path.join(out_dir, "dicts"), shuffle_seed=5)
        assert os.path.exists(atk.model_file)                      # rank 0 saved before the final barrier
    # the same explicit scenario with the two classifier-work savers: cached pseudo-labels per OWNED image (local row
    # numbers) and validation after the last epoch only — the file must come out identical
    calls = {"n": 0}
    hook = net.register_forward_hook(lambda *a: calls.__setitem__("n", calls["n"] + 1))
    atk = ADIL(net, eps=EPS, steps=len(EXPLICIT), n_atoms=K, batch_size=BATCH, data_train=_Indexed(images),
               data_val=_Indexed(val), model_name="dist_savers", step_size=0.01, is_distributed=True, loss="logits",
               kappa=50.0, init_d=d0, init_v=v0raw, epoch_batches=EXPLICIT, val_batches=EXPLICIT_VAL,
               dict_dir=os.path.join(out_dir, "dicts"), shuffle_seed=5, cache_labels=True, val_every=0)
    hook.remove()
    torch.save(calls["n"], os.path.join(out_dir, f"savers_forwards_rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _worker8(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    O, net, images, val, d0, v0raw = _problem(N_IMG8, N_VAL8)
    from dl_attack_on_imagenet_amd import dist as adist
    from dl_attack_on_imagenet_amd.attacks.adil import ADIL

    r, w, _ = adist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    _install_oracle_backend(O)
    from dl_attack_on_imagenet_amd.attacks import adil as A
    created = []
    base = A.ADIL._learner_cls

    class Counting(base):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            created.append(self)

    A.ADIL._learner_cls = Counting
    atk = ADIL(net, eps=EPS, steps=STEPS8, n_atoms=K, batch_size=BATCH8, data_train=_Indexed(images),
               data_val=_Indexed(val), model_name="dist_w8", step_size=0.01, is_distributed=True, loss="logits",
               kappa=50.0, init_d=d0, init_v=v0raw, dict_dir=os.path.join(out_dir, "dicts"), shuffle_seed=5)
    assert os.path.exists(atk.model_file)
    lo, hi = adist.shard_bounds(N_IMG8, rank, world)
    torch.save(dict(collectives=created[0].collectives, rows=created[0].v.shape[0], bounds=(lo, hi), d=created[0].d.clone()),
               os.path.join(out_dir, f"w8_rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_product_distributed_learner_world_8(tmp_path):
    """The node size the path is built for: 8 ranks (gloo, CPU).  205 images = shards of 26 and 25, the requested global
    batch of 100 dealt 13 / 12 per rank with the remainder going round, a last step of 5 images (several ranks own
    nothing and still join the step's all-reduce), validation batches some ranks own nothing of — against the
    single-process oracle at the same global batches; every rank ends on the same dictionary, bit for bit, after the same
    number of collectives (one per step)."""
    world = 8
    port = _free_port()
    mp.start_processes(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    O, net, images, val, d0, v0raw = _problem(N_IMG8, N_VAL8)
    from dl_attack_on_imagenet_amd import dist as adist
    v0 = O.project_onto_l1_ball(v0raw, EPS)
    batches = [adist.global_epoch_batches(N_IMG8, BATCH8, world, 5, e) for e in range(STEPS8)]
    vbatches = [adist.global_epoch_batches(N_VAL8, BATCH8, world, 6, e) for e in range(STEPS8)]
    assert [len(b) for b in batches[0]] == [100, 100, 5] and sorted(i for b in batches[0] for i in b) == list(range(N_IMG8))
    own = [[len(adist.owned_rows(b, *adist.shard_bounds(N_IMG8, r, world))) for r in range(world)] for b in batches[0]]
    assert all(sorted(row) == [12] * 4 + [13] * 4 for row in own[:2])          # 100 over 8 ranks: 13 / 12, never 96
    assert sum(own[2]) == 5 and own[2].count(0) >= 3                            # the ragged last step: ranks that own nothing
    vown = [[len(adist.owned_rows(b, *adist.shard_bounds(N_VAL8, r, world))) for r in range(world)] for b in vbatches[0]]
    assert any(0 in row for row in vown)                                        # a validation batch some ranks own nothing of
    ref = O.learn_dictionary_a(net, images, d0, v0, batches, EPS, 0.01, "logits", False, 50.0, val_images=val,
                               val_batches=vbatches)
    d, v, loss_all, fooling_rate_all, val_fool = torch.load(tmp_path / "dicts" / "ImageNet_dist_w8.bin")
    assert v.shape == (N_IMG8, K)
    assert float((ref["d"] - d).abs().max()) < 5e-6 and float((ref["v"] - v).abs().max()) < 5e-6   # summation order over 8 partial grad_d
    assert list(fooling_rate_all) == list(ref["fooling_rate_all"]) and abs(float(val_fool) - ref["val_fool"]) < 1e-6
    assert max(abs(a - b) for a, b in zip(ref["loss_all"], loss_all)) < 1e-4 * max(1.0, max(map(abs, ref["loss_all"])))
    recs = [torch.load(tmp_path / f"w8_rank{r}.pt") for r in range(world)]
    assert sorted(rec["rows"] for rec in recs) == [25] * 3 + [26] * 5
    assert all(rec["collectives"] == STEPS8 * 3 for rec in recs)                # one all-reduce per step on EVERY rank
    assert all(torch.equal(rec["d"], recs[0]["d"]) for rec in recs)             # replicated D stays bit-identical


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_product_distributed_learner_matches_single_process_global_batch(world, tmp_path):
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    O, net, images, val, d0, v0raw = _problem()
    from dl_attack_on_imagenet_amd import dist as adist
    v0 = O.project_onto_l1_ball(v0raw, EPS)
    for tag, explicit, vexplicit in (("seeded", None, None), ("explicit", EXPLICIT, EXPLICIT_VAL)):
        d, v, loss_all, fooling_rate_all, val_fool = torch.load(tmp_path / "dicts" / f"ImageNet_dist_{tag}.bin")
        if explicit is None:
            explicit = [adist.global_epoch_batches(N_IMG, BATCH, world, 5, e) for e in range(STEPS)]
            vexplicit = [adist.global_epoch_batches(N_VAL, BATCH, world, 6, e) for e in range(STEPS)]
            sizes = [len(b) for b in explicit[0]]
            assert sorted(i for b in explicit[0] for i in b) == list(range(N_IMG))      # one epoch = every image once
            # 2 ranks: 9 + 8 images in chunks of 3; 4 ranks: 5 + 4 + 4 + 4 images in chunks of 1 or 2 (the remainder 6 % 4
            # goes round the ranks): the global batch is the requested 6, equal step count, ragged tail
            assert sizes == [6, 6, 5]
        # single-process reference at the same GLOBAL batches
        ref = O.learn_dictionary_a(net, images, d0, v0, explicit, EPS, 0.01, "logits", False, 50.0, val_images=val,
                                   val_batches=vexplicit)
        assert d.shape == d0.shape and v.shape == (N_IMG, K)
        assert float((ref["d"] - d).abs().max()) < 2e-6, tag                       # fp32 summation order only
        assert float((ref["v"] - v).abs().max()) < 2e-6, tag
        assert max(abs(a - b) for a, b in zip(ref["loss_all"], loss_all)) < 1e-4 * max(1.0, max(map(abs, ref["loss_all"])))
        assert list(fooling_rate_all) == list(ref["fooling_rate_all"])
        assert abs(float(val_fool) - ref["val_fool"]) < 1e-6
    a = torch.load(tmp_path / "dicts" / "ImageNet_dist_explicit.bin")
    b = torch.load(tmp_path / "dicts" / "ImageNet_dist_savers.bin")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and float(a[4]) == float(b[4])
    assert all(torch.load(tmp_path / f"savers_forwards_rank{r}.pt") > 0 for r in range(world))


def _stop_worker(rank, world, port, out_dir, deltas):
    """Rank 0 plays a validation shard that converges early (the slot protocol of adamw_l1ball_kernel restated on the host:
    skip + stay-stopped when the previous slot is below the threshold, else clear the successor's slot and max the own
    one), rank 1 owns nothing of the batch (engine._idle_rank_stop_loop); both max-reduce the stop slot every iteration and
    poll it every STOP_POLL iterations, as engine.solve_codes_adamw does."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from dl_attack_on_imagenet_amd import dist as adist, engine, ops
    adist.init_from_env(backend="gloo")
    red = adist.DictGradReducer()
    calls = {"n": 0}
    real_max = red.max_
    red.max_ = lambda t: (calls.__setitem__("n", calls["n"] + 1), real_max(t))[1]
    dev = torch.device("cpu")
    max_iter = 100
    if rank == 1:
        engine._idle_rank_stop_loop(dev, red, max_iter)
        iters = None
    else:
        stop = ops.StopTest(dev, 1e-6)
        iters = 0
        for it in range(max_iter):
            iters += 1
            t = stop.t
            stop.t += 1
            cur, prev, nxt = t % 3, (t + 2) % 3, (t + 1) % 3
            if float(stop.slots[prev]) < stop.threshold:
                stop.slots[cur] = 0.0                              # a skipped launch: stay stopped
            else:
                stop.slots[nxt] = 0.0
                stop.slots[cur] = max(float(stop.slots[cur]), deltas[min(it, len(deltas) - 1)])
            red.max_(stop.last_slot())
            if (it + 1) % engine.STOP_POLL == 0 and stop.converged():
                break
    torch.save(dict(collectives=calls["n"], iters=iters), os.path.join(out_dir, f"stop_rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("deltas,expect", [([1e-2, 1e-3, 1e-5, 5e-7], 4), ([1e-2] * 5 + [1e-7], 8), ([1e-2], 100)])
def test_idle_rank_pairs_up_with_an_early_converging_shard(deltas, expect, tmp_path):
    """sharded validation reduces the stop slot once per iteration; a rank that owns nothing of the batch must
    issue exactly as many of those collectives as the rank that solves it and leave at the same poll point — for a shard
    that converges at iteration 4 (left at the first poll), at iteration 6 (left at the second), and never (max_iter)."""
    port = _free_port()
    mp.start_processes(_stop_worker, args=(2, port, str(tmp_path), deltas), nprocs=2, join=True, start_method="spawn")
    a, b = torch.load(tmp_path / "stop_rank0.pt"), torch.load(tmp_path / "stop_rank1.pt")
    assert a["iters"] == expect and a["collectives"] == b["collectives"] == expect


def test_global_epoch_batches_equal_steps_and_ownership():
    sys.path.insert(0, ROOT)
    from dl_attack_on_imagenet_amd import dist as adist
    # the advisor's example: 1001 images, 2 ranks, batch 100 -> both ranks must take the same number of steps
    batches = adist.global_epoch_batches(1001, 100, 2, 0, 0)
    bounds = [adist.shard_bounds(1001, r, 2) for r in range(2)]
    per_rank = [[adist.owned_rows(b, *bounds[r]) for b in batches] for r in range(2)]
    assert len(per_rank[0]) == len(per_rank[1]) == len(batches) == 11
    assert sorted(i for b in batches for i in b) == list(range(1001))
    assert [len(x) for x in per_rank[1]][-1] == 0                  # rank 1 (500 images) is empty on the last step ...
    assert [len(x) for x in per_rank[0]][-1] == 1                  # ... where rank 0 (501) still has one image
    assert adist.global_epoch_batches(1001, 100, 2, 0, 0) == batches                    # deterministic
    # a batch size that is no multiple of the world size keeps its size (100 over 8 ranks used to give 96) and
    # equal shards run out together: 1000 images = 10 steps of exactly 100, every image once
    b8 = adist.global_epoch_batches(1000, 100, 8, 0, 0)
    assert [len(b) for b in b8] == [100] * 10 and sorted(i for b in b8 for i in b) == list(range(1000))
    own = [[len(adist.owned_rows(b, *adist.shard_bounds(1000, r, 8))) for r in range(8)] for b in b8]
    assert all(sorted(row) == [12] * 4 + [13] * 4 for row in own)
    with pytest.warns(UserWarning):
        tiny = adist.global_epoch_batches(20, 3, 8, 0, 0)                               # fewer images per step than ranks
    assert sorted(i for b in tiny for i in b) == list(range(20)) and len(tiny[0]) == 8
    assert adist.global_epoch_batches(1001, 100, 2, 0, 1) != batches                    # reshuffled every epoch


def test_init_from_env_single_process(monkeypatch):
    sys.path.insert(0, ROOT)
    from dl_attack_on_imagenet_amd import dist as adist
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert adist.init_from_env() == (0, 1, 0)
    with pytest.raises(RuntimeError):
        adist.DictGradReducer()


def test_slurm_environment_is_adopted(monkeypatch):
    """srun without torchrun (the reference's launch, env_setting.py:7-16, 25): SLURM's task variables become the
    rendezvous this package reads; torchrun's own variables win when present."""
    sys.path.insert(0, ROOT)
    from dl_attack_on_imagenet_amd import dist as adist
    assert adist.first_host("gpu[017-020,031],login1") == "gpu017"
    assert adist.first_host("node7") == "node7" and adist.first_host("a1,b[2-3]") == "a1"
    assert adist.first_host("rack[3,5-6]-ib") == "rack3-ib"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.setenv(k, "placeholder")                      # registered with monkeypatch, so whatever the code under
        monkeypatch.delenv(k)                                      # test writes is rolled back to the original state
    for k, v in dict(SLURM_NTASKS="1", SLURM_PROCID="0", SLURM_LOCALID="0", SLURM_JOB_NODELIST="gpu[017-020]",
                     SLURM_STEP_GPUS="5,4").items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SLURM_NTASKS", "8")
    assert adist.init_from_env() == (0, 1, 0) and "WORLD_SIZE" not in os.environ   # never implied: a bench run inside an allocation
    monkeypatch.setenv("SLURM_NTASKS", "1")
    assert adist.init_from_env(slurm=True) == (0, 1, 0)            # one task: no process group, but the env is translated
    assert os.environ["MASTER_ADDR"] == "gpu017" and os.environ["MASTER_PORT"] == str(12345 + 4)
    monkeypatch.setenv("SLURM_PROCID", "3")
    assert adist.adopt_slurm_env() is False and os.environ["RANK"] == "0"      # already translated / torchrun present: untouched
