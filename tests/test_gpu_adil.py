"""GPU parity of the drop-in surface (attacks.ADIL, Attack_dict_model, ISTA functions, performance) against the
golden vectors generated from the reference: learned D, per-image V, adversarial images within a stated fp32
tolerance, argmax label decisions / fooling counts bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, t
from tinynet import tinynet_from_npz

pytestmark = pytest.mark.gpu
DEV = "cuda"

# fp32 tolerances: the classifier's conv/gemm backward runs on different libraries (MIOpen/rocBLAS vs MKL), and
# AdamW's m/sqrt(s) amplifies relative gradient error for a few steps; 2e-4 absolute on O(1) dictionary entries.
TOL_D = 2e-4
TOL_V = 2e-4
TOL_ADV = 2e-4


class IndexedTensorDataset(torch.utils.data.Dataset):
    """The reference's `indexed` dataset protocol (imagenet_loading.py:8-18) over in-memory tensors."""

    def __init__(self, images):
        self.images, self.indexed = images, False

    def __len__(self):
        return len(self.images)

    def __getitem__(self, item):
        return (item, self.images[item], 0) if self.indexed else (self.images[item], 0)


def close(a, b, tol, what=""):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float((a - b).abs().max()) if a.numel() else 0.0
    assert err <= tol, (what, err, tol)
    return err


def net_on_gpu(z):
    return tinynet_from_npz(z).to(DEV)


@pytest.mark.parametrize("tag", ["ce", "logits"])
def test_learn_dictionary_a_golden(tag, tmp_path):
    from attacks import ADIL
    z = load_golden("g7_learn_a")
    net = net_on_gpu(z)
    atk = ADIL(net, eps=float(z[f"{tag}_eps"]), steps=int(z["steps"]), norm="linf", n_atoms=int(z["k"]),
               batch_size=int(z["batch_size"]), data_train=IndexedTensorDataset(t(z["images"])),
               data_val=IndexedTensorDataset(t(z["val"])), model_name=f"g7{tag}", step_size=float(z["step_size"]),
               loss=tag, method="gd", kappa=float(z["kappa"]), init_d=t(z[f"{tag}_d0"]), init_v=t(z[f"{tag}_v0raw"]),
               epoch_batches=z[f"{tag}_batches"].tolist(), val_batches=z[f"{tag}_val_batches"].tolist(),
               dict_dir=str(tmp_path))
    d, v, loss_all, fooling_rate_all, val_fool = torch.load(atk.model_file, map_location="cpu")
    assert d.shape == z[f"{tag}_d"].shape and d.dtype == torch.float32          # on-disk layout (adil.py:210)
    close(d, z[f"{tag}_d"], TOL_D, "D")
    close(v, z[f"{tag}_v"], TOL_V, "V")
    close(loss_all, z[f"{tag}_loss_all"], 1e-3 * max(1.0, float(np.abs(z[f"{tag}_loss_all"]).max())), "loss")
    assert list(fooling_rate_all) == list(z[f"{tag}_fooling_rate_all"])       # bit-exact label decisions
    assert float(val_fool) == float(z[f"{tag}_val_fool"])


def test_learn_dictionary_b_golden(tmp_path):
    from attacks import ADIL
    z = load_golden("g8_learn_b")
    net = net_on_gpu(z)
    atk = ADIL(net, eps=float(z["eps"]), steps=int(z["steps"]), norm="linf", n_atoms=int(z["k"]),
               batch_size=int(z["batch_size"]), data_train=IndexedTensorDataset(t(z["images"])), data_val=None,
               model_name="g8", step_size=float(z["step_size"]), loss="logits", method="alter",
               steps_in=int(z["steps_in"]), kappa=float(z["kappa"]), init_d=t(z["d0"]),
               init_v=torch.zeros(z["v0"].shape), epoch_batches=z["batches"].tolist(), dict_dir=str(tmp_path))
    d, v, loss_all, fooling_rate_all, _ = torch.load(atk.model_file, map_location="cpu")
    close(d, z["d"], TOL_D, "D"); close(v, z["v"], TOL_V, "V")
    close(loss_all, z["loss_all"], 1e-3 * max(1.0, float(np.abs(z["loss_all"]).max())))
    assert list(fooling_rate_all) == list(z["fooling_rate_all"])


def test_attack_dict_model_dropin():
    """Attack_dict_model + torch.optim.AdamW + update_v/update_d used exactly as the reference's loop uses them
    (adil.py:153-188) reproduces the G6 trajectory — i.e. the class works as a drop-in under autograd."""
    from attacks import Attack_dict_model
    z = load_golden("g6_adamw_steps")
    m = Attack_dict_model(t(z["d0"], DEV).clone(), t(z["v0"], DEV).clone(), float(z["eps"]))
    opt = torch.optim.AdamW(m.parameters(), lr=float(z["lr"]))
    x = t(z["x"], DEV)
    for step in range(z["g"].shape[0]):
        opt.zero_grad()
        m(x, t(z["index"][step]), lambda q: q).backward(t(z["g"][step], DEV))
        opt.step(); m.update_v(); m.update_d()
        close(m.d.data, z["d_hist"][step], 5e-6); close(m.v.data, z["v_hist"][step], 5e-6)


@pytest.mark.parametrize("tag", ["ce", "logits"])
def test_attack_ddrague_golden(tag, tmp_path):
    from attacks import ADIL
    z = load_golden("g9_ddrague")
    net = net_on_gpu(z)
    os.makedirs(tmp_path, exist_ok=True)
    torch.save([t(z["d"]), torch.zeros(1), [], [], torch.tensor(0.)], os.path.join(tmp_path, "ImageNet_g9.bin"))
    atk = ADIL(net, eps=float(z["eps"]), n_atoms=z["d"].shape[-1], attack="supervised", model_name="g9", loss=tag,
               steps_inference=int(z[f"{tag}_steps"]), kappa=float(z["kappa"]), dict_dir=str(tmp_path))
    images = t(z["images"], DEV)
    adv = atk(images, t(z["labels"], DEV))
    assert adv.shape == images.shape and adv.device.type == "cuda"
    close(adv, z[f"{tag}_adv"], TOL_ADV, "adv")
    assert net(adv).argmax(-1).cpu().tolist() == z[f"{tag}_adv_labels"].tolist()      # bit-exact decisions
    assert float(adv.min()) >= 0.0 and float(adv.max()) <= 1.0
    assert float((adv - images).abs().max()) > float(z["eps"])                        # quirk Q6 reproduced
    adv2 = atk(images, t(z["labels"], DEV))                                           # cached dictionary / pinv path
    close(adv2, adv, 0)             # bitwise reproducible (no float atomics in any kernel)


@pytest.mark.parametrize("tag", ["ce", "logits"])
def test_forward_supervised_adamw_golden(tag, tmp_path):
    from attacks import ADIL
    z = load_golden("g10_adamw_inference")
    net = net_on_gpu(z)
    atk = ADIL(net, eps=float(z["eps"]), n_atoms=z["d"].shape[-1], model_name="g10", loss=tag,
               kappa=float(z["kappa"]), dict_dir=str(tmp_path))
    d, images = t(z["d"], DEV), t(z["images"], DEV)
    cnt = atk.forward_supervised_AdamW(images, None, d, "train")
    assert int(cnt) == int(z[f"{tag}_count"])
    adv = atk.forward_supervised_AdamW(images, None, d, "attack")
    close(adv, z[f"{tag}_adv"], TOL_ADV)


@pytest.mark.parametrize("norm", ["linf", "l2"])
def test_forward_unsupervised_golden(norm, tmp_path):
    from attacks import ADIL
    z = load_golden("g11_unsupervised")
    net = net_on_gpu(z)
    torch.save([t(z["d"]), torch.zeros(1), [], [], torch.tensor(0.)], os.path.join(tmp_path, "ImageNet_g11.bin"))
    trials = [t(s, DEV) for s in z[f"{norm}_v_trials"]]
    atk = ADIL(net, eps=float(z["eps"]), n_atoms=z["d"].shape[-1], attack="unsupervised", model_name="g11",
               trials=len(trials), norm=norm, dict_dir=str(tmp_path))
    it = iter(trials)
    atk.sample_sphere = lambda n: next(it)                 # inject the reference's random draws
    adv, dv_norm = atk(t(z["images"], DEV), t(z["labels"], DEV))
    close(adv, z[f"{norm}_adv"], 2e-6)
    close(dv_norm, z[f"{norm}_dv_norm_inf"], 2e-6)
    # the sampler itself, from the recorded uniforms
    atk2 = ADIL(net, eps=float(z["eps"]), n_atoms=z["d"].shape[-1], model_name="g11", norm=norm, dict_dir=str(tmp_path))
    u = t(z[f"{norm}_u"][0])
    if norm == "linf":
        got = atk2.projection_v(atk2.eps + atk2.eps * u)
    else:
        var = 2 * u - 1
        got = atk2.eps * var / var.norm(p="fro", dim=1, keepdim=True)
    close(got, z[f"{norm}_v_trials"][0], 1e-6)


def test_ista_family_golden():
    from attacks.attacks_classes.adil_regularized import adil, learn_coding_vectors, sadil
    from attacks.utils import QuickAttackDataset
    z = load_golden("g12_ista_metrics")
    net = net_on_gpu(z)
    ds = QuickAttackDataset(t(z["images"]), t(z["labels"]))
    lam, step = float(z["lam"]), float(z["step"])
    v = learn_coding_vectors(ds, net, targeted=True, niter=6, lambda_l1=float(z["lcv_lambda_l1"]), lambda_l2=lam,
                             batch_size=3, step_size=torch.tensor(step), n_atom=4, dictionary=t(z["d"], DEV))
    close(v, z["lcv_v"], 1e-4, "lcv")
    d, v, loss = adil(ds, net, targeted=True, niter=4, lambdaCoding=lam, l2_fool=lam, batchsize=3, step_size=step,
                      n_atom=4, device=DEV, init_dictionary=t(z["adil_d0"]))
    close(d, z["adil_d"], 2e-4, "adil D"); close(v, z["adil_v"], 2e-4, "adil V")
    close(loss, z["adil_loss"], 1e-3 * float(np.abs(z["adil_loss"]).max()))
    d, v, _ = sadil(ds, net, targeted=True, nepochs=2, batchsize=3, lambdaCoding=lam, l2_fool=lam, stepsize=step,
                    n_atom=4, device=DEV, init_dictionary=t(z["sadil_d0"]))
    close(d, z["sadil_d"], 2e-4, "sadil D"); close(v, z["sadil_v"], 2e-4, "sadil V")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_sadil_updated_golden(tag, tmp_path):
    from attacks.attacks_classes.adil_regularized import sadil_updated
    from attacks.utils import QuickAttackDataset
    z = load_golden("g13_sadil_updated")
    net = net_on_gpu(z)
    ds = QuickAttackDataset(t(z["images"]), t(z["labels"]))
    out = str(tmp_path / "su.bin")
    d, v = sadil_updated(ds, net, targeted=True, nepochs=3, batchsize=2, lambdaCoding=float(z[f"{tag}_lam"]),
                         l2_fool=float(z["l2"]), stepsize=float(z[f"{tag}_step"]), n_atom=4, device=DEV, model_file=out,
                         init_dictionary=t(z[f"{tag}_d0"]))
    close(d, z[f"{tag}_d"], 5e-4, "D"); close(v, z[f"{tag}_v"], 5e-4, "V")
    saved = torch.load(out, map_location="cpu")
    assert len(saved) == 5 and len(saved[1]) == len(ds)                   # [D, label, pred, v, loss] (:499)
    close(saved[4], z[f"{tag}_loss"], 2e-3 * float(np.abs(z[f"{tag}_loss"]).max()))


def test_performance_metrics_golden():
    import performance as perf
    z = load_golden("g12_ista_metrics")
    net = net_on_gpu(z)
    images, adv = t(z["images"], DEV), t(z["metric_adv"], DEV)
    assert perf.compute_fooling_rate(net, adv, images) == float(z["fooling"])
    assert abs(perf.compute_rmse(adv, images) - float(z["rmse"])) <= 1e-5
    assert abs(perf.compute_mse(adv, images) - float(z["mse"])) <= 1e-3

    class FixedAttack:
        device = torch.device(DEV)

        def __call__(self, x, y):
            return (x + 0.08 * torch.sign(x - 0.5)).clamp(0, 1)
    ylab = t(z["perf_labels"])
    loader = [(t(z["images"])[:3], ylab[:3]), (t(z["images"])[3:], ylab[3:])]
    out = perf.performance(FixedAttack(), net, loader)
    assert abs(float(out["fooling_rate"]) - float(z["perf_fooling_rate"])) <= 1e-6
    assert abs(float(out["rmse"]) - float(z["perf_rmse"])) <= 1e-6
    assert abs(float(out["mse"]) - float(z["perf_mse"])) <= 1e-4


def test_end_to_end_round_trip_full_size(tmp_path):
    """BASELINE-size images (3x224x224): learn a small dictionary on a tiny classifier, attack, and check the
    size-independent invariants: ||v||_1 <= eps, |D| <= 1, adv in [0,1], dictionary file round-trips."""
    from attacks import ADIL
    from tinynet import make_tinynet
    net = make_tinynet(5).to(DEV)
    g = torch.Generator().manual_seed(1)
    images = torch.rand(24, 3, 224, 224, generator=g)
    eps = 8 / 255
    atk = ADIL(net, eps=eps, steps=3, n_atoms=10, batch_size=12, data_train=IndexedTensorDataset(images),
               data_val=None, model_name="e2e", loss="logits", dict_dir=str(tmp_path))
    d, v, loss_all, fr, _ = torch.load(atk.model_file, map_location="cpu")
    assert d.shape == (3, 224, 224, 10) and v.shape == (24, 10) and len(loss_all) == 3
    assert float(d.abs().max()) <= 1.0
    assert float(v.abs().sum(1).max()) <= eps * (1 + 1e-5)
    adv = atk(images[:8].to(DEV), torch.zeros(8, dtype=torch.long, device=DEV))
    assert adv.shape == (8, 3, 224, 224) and float(adv.min()) >= 0 and float(adv.max()) <= 1
    assert torch.isfinite(adv).all()


def test_transfer_performance_golden(tmp_path):
    """configs[3] path: performance.get_transfer_performance on the product ADIL (DDrague inference against the source
    net) against the reference's own numbers (golden G15): fooling rates exact on all three targets."""
    import performance as perf
    from attacks import ADIL
    z = load_golden("g15_transfer")
    targets = {name: tinynet_from_npz(z, prefix=f"{name}.").to(DEV) for name in ("src", "t1", "t2")}
    torch.save([t(z["d"]), torch.zeros(1), [], [], torch.tensor(0.)], os.path.join(tmp_path, "ImageNet_g15.bin"))
    atk = ADIL(targets["src"], eps=float(z["eps"]), n_atoms=z["d"].shape[-1], attack="supervised", model_name="g15",
               loss="logits", steps_inference=int(z["steps"]), kappa=float(z["kappa"]), dict_dir=str(tmp_path))
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(t(z["images"]), t(z["labels"])),
                                         batch_size=int(z["batch_size"]), shuffle=False)
    out = perf.get_transfer_performance({"adil": [atk], "none": []}, targets, loader, device=torch.device(DEV))
    assert set(out) == {"adil", "none"} and all(np.isnan(v["mse"]) for v in out["none"].values())
    for name in targets:
        assert abs(out["adil"][name]["fooling_rate"] - float(z[f"{name}_fooling_rate"])) <= 1e-9, name
        assert abs(out["adil"][name]["rmse"] - float(z[f"{name}_rmse"])) <= 1e-6
        assert abs(out["adil"][name]["mse"] - float(z[f"{name}_mse"])) <= 1e-3


def test_warm_start_reads_the_reference_4tuple(tmp_path, monkeypatch):
    """adil.py:139-143: warm_start loads D from dict_model_ImageNet_version_constrained/ImageNet_{model}_num_atom_{K}
    _nepoch_{steps}_AdamW_200.bin, a 4-TUPLE whose first element is the dictionary.  A run warm-started from that file
    equals a run handed the same dictionary through init_d, bit for bit."""
    from attacks import ADIL
    from tinynet import make_tinynet
    monkeypatch.chdir(tmp_path)
    net = make_tinynet(7).to(DEV)
    g = torch.Generator().manual_seed(2)
    images = torch.rand(10, 3, 16, 16, generator=g)
    d_ws = -1 + 2 * torch.rand(3, 16, 16, 4, generator=g)
    v0 = torch.rand(10, 4, generator=g)
    os.makedirs("dict_model_ImageNet_version_constrained")
    torch.save((d_ws, torch.zeros(10, 4), [0.5], [0.1]),
               "dict_model_ImageNet_version_constrained/ImageNet_ws_num_atom_4_nepoch_2_AdamW_200.bin")
    batches = [[[0, 1, 2, 3, 4], [5, 6, 7, 8, 9]]] * 2
    kw = dict(eps=0.3, steps=2, n_atoms=4, batch_size=5, data_val=None, loss="logits", init_v=v0, epoch_batches=batches)
    ADIL(net, data_train=IndexedTensorDataset(images), model_name="ws", warm_start=True, dict_dir="warm", **kw)
    ADIL(net, data_train=IndexedTensorDataset(images), model_name="ws", init_d=d_ws, dict_dir="cold", **kw)
    a, b = torch.load("warm/ImageNet_ws.bin", map_location="cpu"), torch.load("cold/ImageNet_ws.bin", map_location="cpu")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
    assert not torch.equal(a[0], d_ws)                                  # it did learn from there


def test_resident_loader_and_reference_batch_order(tmp_path):
    """The data step (loader.ResidentImages): every image uploaded once through the pinned double buffer (chunk < N),
    batches by one gather kernel; and the learner's default batch order is the reference's shuffled-DataLoader order:
    a seeded run without injected batches equals the run handed the batches a real DataLoader serves for that seed."""
    from attacks import ADIL
    from dl_attack_on_imagenet_amd.loader import ResidentImages
    from tinynet import make_tinynet
    g = torch.Generator().manual_seed(4)
    images, val = torch.rand(23, 3, 16, 16, generator=g), torch.rand(9, 3, 16, 16, generator=g)
    for dt in (torch.float32, torch.bfloat16):
        res = ResidentImages(IndexedTensorDataset(images), DEV, dt, chunk=5)        # 5 staging rounds, 2 buffers
        assert torch.equal(res.images.cpu(), images.to(dt))
        sub = ResidentImages(IndexedTensorDataset(images), DEV, dt, rows=range(7, 19), chunk=4)
        assert torch.equal(sub.gather([0, 11, 3]).cpu(), images[[7, 18, 10]].to(dt))
    net = make_tinynet(8).to(DEV)
    d0, v0 = -1 + 2 * torch.rand(3, 16, 16, 4, generator=g), torch.rand(23, 4, generator=g)
    kw = dict(eps=0.3, steps=2, n_atoms=4, batch_size=6, loss="logits", init_d=d0, init_v=v0)
    torch.manual_seed(77)
    ADIL(net, data_train=IndexedTensorDataset(images), data_val=IndexedTensorDataset(val), model_name="auto",
         dict_dir=str(tmp_path), **kw)
    torch.manual_seed(77)                                                # what the reference's two DataLoaders would serve
    tr = torch.utils.data.DataLoader(torch.arange(23), batch_size=6, shuffle=True)
    va = torch.utils.data.DataLoader(torch.arange(9), batch_size=6, shuffle=True)
    eb, vb = [], []
    for _ in range(2):
        eb.append([b.tolist() for b in tr])
        vb.append([b.tolist() for b in va])
    ADIL(net, data_train=IndexedTensorDataset(images), data_val=IndexedTensorDataset(val), model_name="inj",
         dict_dir=str(tmp_path), epoch_batches=eb, val_batches=vb, **kw)
    a, b = torch.load(tmp_path / "ImageNet_auto.bin", map_location="cpu"), torch.load(tmp_path / "ImageNet_inj.bin", map_location="cpu")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and float(a[4]) == float(b[4])


def test_single_rank_rccl_reducer_is_bitwise_neutral(tmp_path):
    """The RCCL path on one rank (process group 'nccl', world 1): a learner whose grad_d goes through
    DictGradReducer.all_reduce_ equals the learner without reducer bit for bit, and the product's data-parallel
    learner (learn_dictionary_distributed) equals learn_dictionary_a on the same global batches."""
    import socket
    import torch.distributed as tdist
    from attacks import ADIL
    from dl_attack_on_imagenet_amd import dist as adist, engine
    from tinynet import make_tinynet
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ADIL_FORCE_REDUCER="1")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        adist.init_from_env(backend="nccl")
        net = make_tinynet(9).to(DEV)
        g = torch.Generator().manual_seed(6)
        images = torch.rand(12, 3, 32, 32, generator=g)
        d0, v0 = -1 + 2 * torch.rand(3, 32, 32, 5, generator=g), torch.rand(12, 5, generator=g)
        va = engine.ops.l1ball_project_(v0.clone().to(DEV), 0.3)
        la = engine.DictionaryLearner(d0.clone().to(DEV), va.clone(), 0.3, 0.01, "logits")
        lb = engine.DictionaryLearner(d0.clone().to(DEV), va.clone(), 0.3, 0.01, "logits", reducer=adist.DictGradReducer())
        idx = torch.arange(12, device=DEV)
        for _ in range(3):
            la.step(net, images.to(DEV), idx); lb.step(net, images.to(DEV), idx)
        assert torch.equal(la.d, lb.d) and torch.equal(la.v, lb.v)
        batches = [[[0, 5, 7, 2, 9, 11], [1, 3, 4, 6, 8, 10]]] * 2
        kw = dict(eps=0.3, steps=2, n_atoms=5, batch_size=6, loss="logits", init_d=d0, init_v=v0, epoch_batches=batches,
                  dict_dir=str(tmp_path))
        ADIL(net, data_train=IndexedTensorDataset(images), model_name="dp", is_distributed=True, **kw)
        ADIL(net, data_train=IndexedTensorDataset(images), model_name="sp", **kw)
        a, b = torch.load(tmp_path / "ImageNet_dp.bin", map_location="cpu"), torch.load(tmp_path / "ImageNet_sp.bin", map_location="cpu")
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[3] == b[3]
    finally:
        if tdist.is_initialized():
            tdist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _two_rank_learner_run(tmp_path, env_extra):
    """Start 2 fresh child ranks of tests/dist_learn_worker.py (python -m torch.distributed.run, rendezvous on 127.0.0.1)
    and compare with the single-process learner of THIS process at the same global batches."""
    import json
    import socket
    import subprocess
    from attacks import ADIL
    from dist_learn_problem import IndexedImages, problem
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, ADIL_REDUCER_TIMING="1", **env_extra)
    env.pop("ADIL_FORCE_REDUCER", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(here, "dist_learn_worker.py"), str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    infos = [json.load(open(tmp_path / f"info_rank{k}.json")) for k in range(2)]
    d_ranks = [torch.load(tmp_path / f"d_rank{k}.pt") for k in range(2)]
    assert torch.equal(d_ranks[0], d_ranks[1])                          # the replicated dictionary never diverges
    p = problem()
    ADIL(p["net"].to(DEV), data_train=IndexedImages(p["images"]), data_val=IndexedImages(p["val"]), model_name="sp",
         dict_dir=str(tmp_path / "dicts"), **p["kw"])
    dp = torch.load(tmp_path / "dicts" / "ImageNet_dp2.bin", map_location="cpu")
    sp = torch.load(tmp_path / "dicts" / "ImageNet_sp.bin", map_location="cpu")
    assert torch.equal(dp[0], d_ranks[0])
    # the two ranks sum their partial grad_d in a different order than one pass over the global batch: fp32 rounding only
    # (median: rounding; maximum: an entry whose gradient is ~1e-8 sits in AdamW's eps regime, where a 1e-10 difference of
    # its first moment moves it by lr * 1e-10 / 1e-8)
    dd, dv = (dp[0] - sp[0]).abs(), (dp[1] - sp[1]).abs()
    assert float(dd.median()) <= 1e-6 and float(dd.max()) <= 5e-4 and float(dv.median()) <= 1e-6 and float(dv.max()) <= 5e-4
    assert dp[1].shape == sp[1].shape and list(dp[3]) == list(sp[3])     # V gathered from both ranks; fooling rates per epoch
    assert max(abs(a - b) for a, b in zip(dp[2], sp[2])) <= 1e-4 * max(1.0, max(abs(a) for a in sp[2]))
    assert abs(float(dp[4]) - float(sp[4])) < 1e-6                        # sharded validation (global stop test)
    # what the first SCALE line will say about its collective: ONE all-reduce of grad_d (P K fp32) per step,
    # bracketed start -> wait on the compute stream on every rank
    from dist_learn_problem import EPOCH_BATCHES, K as K_ATOMS
    steps = sum(len(e) for e in EPOCH_BATCHES)
    for i in infos:
        assert i["bytes_per_allreduce"] == 3 * 32 * 32 * K_ATOMS * 4, i
        assert i["allreduce_brackets"] == steps and i["allreduce_ms_start_to_wait_mean"] > 0.0, i
        assert i["allreduce_ms_start_to_wait_max"] >= i["allreduce_ms_start_to_wait_mean"]
    return infos


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per device")
def test_two_rank_rccl_learner_matches_single_process(tmp_path):
    """the RCCL path with MORE THAN ONE rank — 2 fresh child ranks, one GPU each, backend 'nccl' (no gloo,
    no shared device): learn_dictionary_distributed == learn_dictionary_a at the global batches, D bit-identical across
    ranks, and the asynchronous all_reduce_start / wait branch (engine.py: overlap with the code-row update) is the one
    that ran.  Skipped on the one-GPU boxes; the same worker runs there as the rehearsal below."""
    infos = _two_rank_learner_run(tmp_path, {})
    assert [i["backend"] for i in infos] == ["nccl", "nccl"] and all(i["world_size"] == 2 for i in infos)
    assert all(i["async_work"] for i in infos)
    assert {i["device"] for i in infos} == {"cuda:0", "cuda:1"}
    assert all(i["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for i in infos)


def test_two_rank_learner_rehearsal_on_one_gpu(tmp_path):
    """The same worker as the RCCL test on a one-GPU box: two ranks share the card and exchange through gloo
    (ADIL_DIST_BACKEND=gloo ADIL_SHARE_GPU=1; RCCL refuses two ranks on one device).  Everything but the transport is the
    shipped path: HIP kernels, ownership, the sharded validation with its max-reduced stop slot, the padded gather of V."""
    infos = _two_rank_learner_run(tmp_path, {"ADIL_DIST_BACKEND": "gloo", "ADIL_SHARE_GPU": "1"})
    assert [i["backend"] for i in infos] == ["gloo", "gloo"] and all(i["world_size"] == 2 for i in infos)


@pytest.mark.parametrize("tag,norm,optim", [("linf_adam", "linf", "adam"), ("l2_sgd", "l2", "sgd")])
def test_uappgd_baseline_matches_reference_run(tag, norm, optim, tmp_path):
    """UAPPGD (K = 1 on the ADiL kernels) replaying the reference's own run (golden G14: shuffled batch order, both norms
    and optimisers): learned perturbation, per-epoch validation fooling rates and training fooled counts."""
    from attacks import UAPPGD
    from attacks.utils import QuickAttackDataset
    z = load_golden("g14_uappgd")
    net = tinynet_from_npz(z).to(DEV)
    images, labels, val = t(z["images"]), t(z["labels"]).long(), t(z["val"])
    batches = [[list(map(int, b)) for b in e] for e in z[f"{tag}_batches"]]
    atk = UAPPGD(net, steps=int(z["steps"]), batch_size=int(z["batch_size"]), beta=float(z["beta"]),
                 step_size=float(z[f"{tag}_lr"]), norm=norm, eps=float(z[f"{tag}_eps"]), optimizer=optim, model_dir=str(tmp_path))
    attack = atk.learn_attack(QuickAttackDataset(images, labels), QuickAttackDataset(val, torch.zeros(len(val), dtype=torch.long)),
                              batches=batches)
    close(attack, z[f"{tag}_attack"], 2e-5)
    assert [float(f) for f in atk.fooling_rate] == [float(f) for f in z[f"{tag}_fooling"]]
    assert atk.train_fooled == [int(f) for f in z[f"{tag}_train_fooled"]]
    adv = atk(images.to(DEV), labels.to(DEV))                         # forward: clamp(images + attack, 0, 1) from the saved file
    close(adv, (images + t(z[f"{tag}_attack"])).clamp(0, 1), 2e-5)


def test_demo_cli_result_dumps(tmp_path, monkeypatch, capsys):
    """demo_dL_attack.py end to end on a seeded synthetic dataset: the dictionary file is named after the CLI's model
    string as upstream (trained_dicts/ImageNet_resnet.bin, adil.py:89-91) and the two result dumps carry the reference's
    file names and layout (demo_dL_attack.py:148-156: {'fooling_rate','rmse','mse','time'} -> {sub_name: [values]})."""
    import demo_dL_attack
    monkeypatch.chdir(tmp_path)
    args = demo_dL_attack.build_parser().parse_args(
        ["-m", "resnet", "-s", "4", "--synthetic", "--synthetic-classes", "3", "--trained-classes", "3", "--image-size", "64",
         "--n-atoms", "4", "--steps", "2", "--batch-size", "2", "--steps-inference", "3"])
    torch.random.manual_seed(args.seed)
    val_perf, test_perf = demo_dL_attack.main(args)
    # the clean-accuracy pass the reference runs first (demo_dL_attack.py:65-66, model_accuracy.py): over all 150 structured
    # images, of which the head was fitted to the 3 of the training split (one per class)
    said = [l for l in capsys.readouterr().out.splitlines() if l.startswith("accuracy of the the model resnet is ")]
    # What is asserted, and why 50 %: the head is a nearest-centroid fit on ONE image per class, scored by the
    # bf16 network on 64 x 64 images — 80.7 ... 100 % over the recorded boxes, i.e. a quantity with real spread that is not
    # a property of any ADiL kernel.  The check is that the fit WORKED: a head that ignores its input, or chance, scores
    # 33.3 % on three balanced classes; 50 % is out of reach for both (binomial sigma of 150 images at 1/3: 3.8 pp, so
    # 50 % is 4.3 sigma above chance) and far below every recorded run.
    assert len(said) == 1 and 50.0 < float(said[0].rsplit(" ", 1)[1]) <= 100.0
    d, v, loss_all, fooling_rate_all, val_fool = torch.load("trained_dicts/ImageNet_resnet.bin", map_location="cpu")
    assert d.shape == (3, 64, 64, 4) and v.shape == (3, 4) and len(loss_all) == 2 and len(fooling_rate_all) == 2
    out = "dict_model_ImageNet_version_constrained"
    val_dump = torch.load(os.path.join(out, "model_sampling_adil_inference_rlts_sampling_3_3_4_ce.bin"), weights_only=False)
    test_dump = torch.load(os.path.join(out, "model_adil_resultat_test_ce.bin"), weights_only=False)
    for dump, ret in ((val_dump, val_perf), (test_dump, test_perf)):
        assert set(dump) == {"fooling_rate", "rmse", "mse", "time"}
        assert list(dump["fooling_rate"]) == ["adil_atoms_4_loss_logits_"]
        assert len(dump["rmse"]["adil_atoms_4_loss_logits_"]) == 1
        a = float(dump["fooling_rate"]["adil_atoms_4_loss_logits_"][0])
        b = float(ret["fooling_rate"]["adil_atoms_4_loss_logits_"][0])
        assert a == b and 0.0 <= a <= 1.0             # structured synthetic data + fitted head: a real fooling rate, not 0 / 0
    # the unstructured stand-in (U[0,1) noise, random-init head): performance.py's correctly-classified filter keeps next to
    # nothing, the metrics are then 0 / 0 exactly as upstream's would be — the plumbing still runs
    args0 = demo_dL_attack.build_parser().parse_args(
        ["-m", "resnet", "-s", "4", "--synthetic", "--synthetic-structured", "0", "--synthetic-classes", "3", "--trained-classes", "3",
         "--image-size", "64", "--n-atoms", "4", "--steps", "2", "--batch-size", "2", "--steps-inference", "3"])
    os.remove("trained_dicts/ImageNet_resnet.bin")
    v0, t0 = demo_dL_attack.main(args0)
    f0 = float(t0["fooling_rate"]["adil_atoms_4_loss_logits_"][0])
    assert f0 != f0 or 0.0 <= f0 <= 1.0


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_starts_its_own_ranks(launcher, tmp_path):
    """`python bench.py --gpus 2` outside torchrun: the parent (which makes no GPU call) starts the two ranks itself and rank
    0 prints ONE JSON line with n_gpus = 2 — and the driver's own form, `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`, gives the same.  This box has one GPU,
    so the rehearsal shares it (ADIL_SHARE_GPU=1) over gloo (RCCL refuses two ranks on one device); rank plumbing,
    weak-scaling accounting, the per-step all-reduce of grad_d, the max-over-ranks timing and the JSON contract are what is
    checked — not a scaling number."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADIL_DIST_BACKEND="gloo", ADIL_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    bench = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--model", "resnet18",
             "--batch", "32", "--atoms", "10", "--cpu-baseline", "0"]
    if launcher == "self":
        cmd = [sys.executable] + bench
    else:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + bench
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 64 and out["value"] > 0
    assert abs(out["value"] - 2 * 32 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert "cpu_baseline" not in out and out["roofline"]["bound"] == "hbm"
    assert out["config"]["recomputed_labels_variant"]["images_per_sec"] > 0      # the reference's op sequence, timed beside the headline


def test_transfer_evaluation_data_parallel(tmp_path):
    """configs[3] is a data-parallel evaluation: the dictionary is replicated, batches are dealt to the ranks, only the final
    sums cross ranks.  Two ranks (gloo rehearsal on this one GPU) through performance.get_transfer_performance reproduce the
    reference's G15 numbers exactly like the single-process run."""
    import json
    import socket
    import subprocess
    import sys
    z = load_golden("g15_transfer")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, ADIL_DIST_BACKEND="gloo", ADIL_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = tmp_path / "transfer.json"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(here, "dist_transfer_worker.py"), str(out),
                        str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert res["world"] == 2
    for name in ("src", "t1", "t2"):
        assert abs(res["perf"][name]["fooling_rate"] - float(z[f"{name}_fooling_rate"])) <= 1e-9, name
        assert abs(res["perf"][name]["rmse"] - float(z[f"{name}_rmse"])) <= 1e-6
        assert abs(res["perf"][name]["mse"] - float(z[f"{name}_mse"])) <= 1e-3


def test_transfer_evaluation_all_six_reference_targets(tmp_path):
    """configs[3] at plumbing scale: ONE dictionary, adversaries from DDrague against the source net, scored on all six
    classifiers of the reference CLI (demo_dL_attack.py:41-53; random-init definitions from zoo, 224x224 inputs) through
    performance.get_transfer_performance.  Checks the result layout and the metric identities (no golden: the reference
    cannot build these networks offline)."""
    import performance as perf
    from attacks import ADIL
    from dl_attack_on_imagenet_amd import zoo
    names = ("resnet", "densenet", "googlenet", "inception", "mobilenet", "vgg")
    models = {n: zoo.build_classifier(n, seed=3, device=DEV) for n in names}
    g = torch.Generator().manual_seed(11)
    images = torch.rand(8, 3, 224, 224, generator=g)
    labels = models["resnet"](images.to(DEV)).argmax(-1).cpu()
    d = -1 + 2 * torch.rand(3, 224, 224, 10, generator=g)
    torch.save([d, torch.zeros(1), [], [], torch.tensor(0.)], os.path.join(tmp_path, "ImageNet_resnet.bin"))
    atk = ADIL(models["resnet"], eps=8 / 255, n_atoms=10, attack="supervised", model_name="resnet", loss="logits",
               steps_inference=4, dict_dir=str(tmp_path))
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(images, labels), batch_size=4, shuffle=False)
    out = perf.get_transfer_performance({"adil": [atk]}, models, loader, device=torch.device(DEV))["adil"]
    assert set(out) == set(names)
    rm = [out[n]["rmse"] for n in names]
    assert max(rm) - min(rm) < 1e-9 and 0 < rm[0] < 0.05          # one adversary, the same distortion for every target
    for n in names:
        assert 0.0 <= out[n]["fooling_rate"] <= 1.0 and np.isfinite(out[n]["mse"])
    assert out["resnet"]["fooling_rate"] >= max(out[n]["fooling_rate"] for n in names if n != "resnet") - 0.25


def test_graphed_learner_step_is_bit_identical():
    """DictionaryLearner.step_graphed (the whole step as one hipGraph launch, AdamW's step-dependent scalars through device
    memory) against the eager step: full-batch and minibatch schedules incl. a ragged last batch (falls back to eager),
    8 steps — D, V, both moment pairs, losses and fooled counts bit for bit."""
    from dl_attack_on_imagenet_amd import engine, ops
    from tinynet import make_tinynet
    net = make_tinynet(4).to(DEV)
    g = torch.Generator().manual_seed(9)
    images = torch.rand(20, 3, 32, 32, generator=g).to(DEV)
    d0 = (-1 + 2 * torch.rand(3, 32, 32, 6, generator=g)).to(DEV)
    v0 = ops.l1ball_project_(torch.rand(20, 6, generator=g).to(DEV), 0.3)
    full = engine.predict(net, images)
    for schedule, cached in (([list(range(20))] * 8, False),
                             ([[3, 1, 4, 15, 9, 2, 6, 5], [8, 7, 0, 19, 18, 17, 16, 14], [13, 12, 11, 10]] * 3, False),
                             ([[3, 1, 4, 15, 9, 2, 6, 5], [8, 7, 0, 19, 18, 17, 16, 14], [13, 12, 11, 10]] * 3, True)):
        a = engine.DictionaryLearner(d0.clone(), v0.clone(), 0.3, 0.01, "logits")
        b = engine.DictionaryLearner(d0.clone(), v0.clone(), 0.3, 0.01, "logits")
        for idx in schedule:
            index = torch.tensor(idx, device=DEV)
            lab = full[index] if cached else None                  # cached pseudo-labels (the learners' default): one forward less in the recording
            la, fa = a.step(net, images[index].contiguous(), index, lab)
            lb, fb = b.step_graphed(net, images[index].contiguous(), index, lab)
            assert float(la) == float(lb) and int(fa) == int(fb)
        assert b._graph is not None                                # a graph was captured and replayed
        for x, y in ((a.d, b.d), (a.v, b.v), (a.m_d, b.m_d), (a.s_d, b.s_d), (a.m_v, b.m_v), (a.s_v, b.s_v)):
            assert torch.equal(x, y)
        assert a.sched_d.t == b.sched_d.t == len(schedule)


def test_graphed_learner_free_running_replays_match_eager():
    """30 replays of the graphed step queued WITHOUT any host synchronisation while the GPU is kept busy (so the
    host is dozens of replays — several rings of pinned scalar slots — ahead of the device) must still hand every
    recorded AdamW launch the scalars of ITS step: D, V, the moments and the per-step fooled counts equal the eager loop
    bit for bit.  Before the slots were guarded by events (ops.PinnedRing) a host more than 8 replays ahead overwrote a
    slot before its copy had run.  Same for the DDrague solver's groups of three iterations."""
    from dl_attack_on_imagenet_amd import engine, ops
    from tinynet import make_tinynet
    net = make_tinynet(4).to(DEV)
    g = torch.Generator().manual_seed(19)
    images = torch.rand(16, 3, 32, 32, generator=g).to(DEV)
    index = torch.arange(16, device=DEV)
    d0 = (-1 + 2 * torch.rand(3, 32, 32, 6, generator=g)).to(DEV)
    v0 = ops.l1ball_project_(torch.rand(16, 6, generator=g).to(DEV), 0.3)
    steps = 33
    a = engine.DictionaryLearner(d0.clone(), v0.clone(), 0.3, 0.01, "logits")
    fooled_a = [a.step(net, images, index)[1] for _ in range(steps)]
    b = engine.DictionaryLearner(d0.clone(), v0.clone(), 0.3, 0.01, "logits")
    fooled_b = [b.step_graphed(net, images, index)[1] for _ in range(3)]     # two eager warm-ups + the capturing call
    busy = torch.randn(4096, 4096, device=DEV)
    torch.cuda.synchronize()
    for _ in range(60):                                   # ~100 ms of queued GPU work: the host runs far ahead below
        busy = (busy @ busy).clamp_(-1, 1)
    for _ in range(steps - 3):
        fooled_b.append(b.step_graphed(net, images, index)[1])              # no .item(), no sync
    assert b._graph is not None
    torch.cuda.synchronize()
    assert [int(f) for f in fooled_a] == [int(f) for f in fooled_b]
    for x, y in ((a.d, b.d), (a.v, b.v), (a.m_d, b.m_d), (a.s_d, b.s_d), (a.m_v, b.m_v), (a.s_v, b.s_v)):
        assert torch.equal(x, y)
    # the inference solver: 11 groups of three iterations queued behind the same kind of backlog
    dd = (-1 + 2 * torch.rand(3, 32, 32, 6, generator=g)).to(DEV)
    eager = engine.DDragueSolver(net, images[:5], dd, 0.1, "logits").run(36)
    gr = engine.DDragueSolver(net, images[:5], dd, 0.1, "logits")
    for _ in range(3):
        gr.iterate()
    gr._capture()
    for _ in range(60):
        busy = (busy @ busy).clamp_(-1, 1)
    for _ in range(11):
        gr._replay()
    torch.cuda.synchronize()
    assert torch.equal(eager.z, gr.z) and torch.equal(eager.m, gr.m) and torch.equal(eager.s, gr.s)


def test_integration_md_stub_runs_a_learning_step():
    """The ctypes stub printed in INTEGRATION.md is executed as it stands (only the library path is filled in) and drives
    one learning step through the C ABI; the result equals the product's DictionaryLearner step bit for bit.  Keeps the
    document honest about signatures and argument order."""
    import re
    from dl_attack_on_imagenet_amd import _lib, engine, ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    assert 'ctypes.CDLL("libadil_hip.so")' in code
    ns = {}
    exec(code.replace('ctypes.CDLL("libadil_hip.so")', f'ctypes.CDLL({_lib.LIBPATH!r})'), ns)
    g = torch.Generator().manual_seed(13)
    n, b, k, eps = 9, 6, 5, 0.4
    x = torch.rand(b, 3, 16, 16, generator=g).to(DEV)
    d0 = (-1 + 2 * torch.rand(3, 16, 16, k, generator=g)).to(DEV)
    v0 = ops.l1ball_project_(torch.rand(n, k, generator=g).to(DEV), eps)
    gup = torch.randn(b, 3, 16, 16, generator=g).to(DEV)                   # stands in for dLoss/d(x + Dv)
    index = torch.tensor([4, 0, 8, 2, 7, 5], device=DEV)
    # the stub, as a reference maintainer would call it
    d, v = d0.clone(), v0.clone()
    md, sd, mv, sv = torch.zeros_like(d), torch.zeros_like(d), torch.zeros_like(v), torch.zeros_like(v)
    pos = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    vp = ns["pack_codes"](v, index, pos)
    xt = ns["synth"](x, d, vp)
    gd, gv = ns["grad"](gup, d, vp)
    scal = ns["adamw_scalars"](0.01, 1)
    ns["adamw_clamp_"](d, gd, md, sd, scal, -1.0, 1.0)
    ns["adamw_l1ball_"](v, gv, pos, mv, sv, scal, eps)
    # the product
    d2, v2 = d0.clone(), v0.clone()
    vp2 = ops.pack_codes(v2, index, b)
    assert torch.equal(xt, ops.synth(x, d2, vp2, b))
    gd2, gv2 = ops.grad(gup, d2, vp2, b)
    learner = engine.DictionaryLearner(d2, v2, eps, 0.01, "logits")
    ops.pack_codes(learner.v, index, b, pos=learner.pos)
    learner.update_d(gd2); learner.update_v(gv2)
    assert torch.equal(d, learner.d) and torch.equal(v, learner.v) and int((pos != -1).sum()) == 0


def test_graphed_ddrague_is_bit_identical(tmp_path):
    """forward_supervised_DDrague replayed from a hipGraph (three iterations per launch, AdamW scalars and the stop slots in
    device memory) against the eager loop: same adversarial images bit for bit — for step counts around the group size,
    for a second and third batch served by the SAME recorded graph (ADIL keeps the solver per batch shape), and when the
    loop converges in the middle of a replayed group."""
    from attacks import ADIL
    from dl_attack_on_imagenet_amd import engine
    from tinynet import make_tinynet
    net = make_tinynet(6).to(DEV)
    g = torch.Generator().manual_seed(17)
    d = (-1 + 2 * torch.rand(3, 32, 32, 6, generator=g)).to(DEV)
    batches = [torch.rand(5, 3, 32, 32, generator=g).to(DEV) for _ in range(3)]
    for steps in (5, 6, 14, 30):
        eager = engine.solve_ddrague(net, batches[0], d, 0.1, steps, "logits")
        graphed = engine.solve_ddrague(net, batches[0], d, 0.1, steps, "logits", use_graph=True)
        assert torch.equal(eager, graphed), steps
    torch.save([d.cpu(), torch.zeros(1), [], [], torch.tensor(0.)], os.path.join(tmp_path, "ImageNet_gr.bin"))
    kw = dict(eps=0.1, n_atoms=6, attack="supervised", model_name="gr", loss="logits", steps_inference=11, dict_dir=str(tmp_path))
    a, b = ADIL(net, use_graph=False, **kw), ADIL(net, use_graph=True, **kw)
    lab = torch.zeros(5, dtype=torch.long, device=DEV)
    for x in batches:
        assert torch.equal(a(x, lab), b(x, lab))
    assert len(b._solvers) == 1 and next(iter(b._solvers.values()))._graph is not None
    # the cache is bounded (a solver holds z, m, s and a graph): other batch sizes evict the least recently used shape
    for n in (1, 2, 3, 4, 2):
        x = batches[0][:n]
        assert torch.equal(a(x, lab[:n]), b(x, lab[:n]))
    assert len(b._solvers) == ADIL._MAX_SOLVERS and (5, 3, 32, 32) not in [k[0] for k in b._solvers]
    assert list(b._solvers)[-1][0] == (2, 3, 32, 32)
    # convergence inside a replayed group: a zero dictionary direction makes every z-step a no-op from the start
    flat = make_tinynet(6).to(DEV)
    for p in flat.parameters():
        p.data.zero_()                                                    # constant logits -> zero gradient -> max|dz| = 0
    e = engine.DDragueSolver(flat, batches[0], d, 0.1, "ce").run(20)
    gr = engine.DDragueSolver(flat, batches[0], d, 0.1, "ce").run(20, use_graph=True)
    assert torch.equal(e.result()[0], gr.result()[0]) and gr.stop.converged() and e.iters <= 4 and gr.iters <= 6


def test_ddrague_codes_from_the_zstep_match_the_separate_contraction():
    """Round 4: the codes v = z D_dagger^T of every iteration after the first come out of the preceding z-step
    (adil_zstep_codes) instead of a contraction launch of their own.  Both routes compute the same fp32-grade sums in a
    different order, so the two solvers agree to accumulated rounding: adversarial images within 2e-5, identical label
    decisions, the same iteration count; the fused route never launches the separate contraction (counted)."""
    from dl_attack_on_imagenet_amd import engine, ops
    from tinynet import make_tinynet
    net = make_tinynet(6).to(DEV)
    g = torch.Generator().manual_seed(23)
    d = (-1 + 2 * torch.rand(3, 32, 32, 6, generator=g)).to(DEV)
    x = torch.rand(40, 3, 32, 32, generator=g).to(DEV)
    calls = {"grad_on_z": 0}
    real_grad = ops.grad

    def counting_grad(gq, dq, *a, **kw):
        if gq.dtype == torch.float32 and gq.shape == x.shape and dq.data_ptr() == solver.dpt.data_ptr():
            calls["grad_on_z"] += 1
        return real_grad(gq, dq, *a, **kw)

    pinv = engine.PseudoInverse(d)
    solver = engine.DDragueSolver(net, x, d, 0.1, "logits", pinv=pinv)
    assert solver._vslabs is not None
    ops.grad = counting_grad
    try:
        solver.run(12)
        adv_f, v_f = solver.result()
    finally:
        ops.grad = real_grad
    assert calls["grad_on_z"] == 0
    two = engine.DDragueSolver(net, x, d, 0.1, "logits", pinv=pinv, fuse_codes=False).run(12)
    adv_t, v_t = two.result()
    assert two._vslabs is None and solver.iters == two.iters == 12
    assert float((adv_f - adv_t).abs().max()) <= 2e-5 and float((v_f - v_t).abs().max()) <= 2e-5
    assert torch.equal(net(adv_f).argmax(-1), net(adv_t).argmax(-1))
    # no iteration at all: z = 0, the codes are zero, the result is the clamped clean batch
    fresh = engine.DDragueSolver(net, x, d, 0.1, "logits", pinv=pinv)
    adv0, v0 = fresh.result()
    assert torch.equal(adv0, x.clamp(0, 1)) and not bool(v0.any())
    # reset() re-arms the zero-codes state for the next batch
    solver.reset(x[:40])
    assert solver._vnext is None
    solver.run(12)
    assert torch.equal(solver.result()[0], adv_f)


def test_precise_head_is_entered_by_the_inference_solver_only():
    """zoo.build_classifier(head_fp32="inference") + engine.precise_head (round 4): the bf16 FusedResNet computes its logits in
    fp32 (pooling + last linear layer, fp32 weights kept under the bf16 cast) exactly inside the DDrague solver's classifier
    calls — eager and replayed from a hipGraph alike — and nowhere else: not in the learner's step, not in a plain call (what
    performance.py's scoring does).  The two heads agree to bf16 rounding of the logits."""
    from dl_attack_on_imagenet_amd import engine, ops, zoo
    kw = dict(num_classes=10, seed=3, device=DEV, dtype=torch.bfloat16, channels_last=True, fuse_bn_act=True, fuse_stem=True)
    net = zoo.build_classifier("resnet18", head_fp32="inference", **kw)
    fused = net[0]
    assert fused.head32 is not None and fused.head32.weight.dtype == torch.float32 and fused.fc.weight.dtype == torch.bfloat16
    calls = {"n": 0}
    fused.head32.register_forward_hook(lambda *a: calls.__setitem__("n", calls["n"] + 1))
    g = torch.Generator().manual_seed(12)
    x = torch.rand(8, 3, 64, 64, generator=g).to(DEV).to(torch.bfloat16)
    plain = net(x)
    assert plain.dtype == torch.bfloat16 and calls["n"] == 0
    with engine.precise_head(net):
        sharp = net(x)
        with engine.precise_head(net, False):                      # nests and restores
            assert net(x).dtype == torch.bfloat16
        assert fused.head32_on
    assert sharp.dtype == torch.float32 and calls["n"] == 1 and not fused.head32_on
    assert float((sharp - plain.float()).abs().max()) <= 2.0 ** -7 * float(sharp.abs().max()) + 1e-3   # one bf16 rounding of the logits
    d = (-1 + 2 * torch.rand(3, 64, 64, 6, generator=g)).to(DEV)
    v = ops.l1ball_project_(torch.rand(8, 6, generator=g).to(DEV), 0.1)
    learner = engine.DictionaryLearner(d.clone(), v, 0.1, 0.01, "logits")
    before = calls["n"]
    learner.step(net, x, torch.arange(8, device=DEV))
    assert calls["n"] == before                                     # the learner keeps the bf16 head
    eager = engine.DDragueSolver(net, x, d, 0.1, "logits").run(9)
    assert calls["n"] == before + 9                                 # once per inference iteration
    # replayed from a hipGraph the recorded iterations carry the fp32 head as well (the switch is read while recording).  No
    # bit-equality with the eager loop is asserted here: this classifier's library calls (MIOpen convolutions, the GEMM of
    # the head) may pick other algorithms under stream capture, and nine chaotic iterations amplify one rounding (the
    # bit-identical replay is asserted on plain-torch classifiers in test_graphed_ddrague_is_bit_identical)
    graphed = engine.DDragueSolver(net, x, d, 0.1, "logits").run(9, use_graph=True)
    adv_e, adv_g = eager.result()[0], graphed.result()[0]
    assert not fused.head32_on and graphed.iters == 9
    for adv in (adv_e, adv_g):
        assert bool(torch.isfinite(adv.float()).all()) and float(adv.min()) >= 0.0 and float(adv.max()) <= 1.0
    # both perturb the same images by D D_dagger z with |z| <= eps = 0.1 (the perturbation itself may exceed eps: quirk Q6)
    assert float((adv_e.float() - adv_g.float()).abs().max()) <= 0.6
    # a classifier without the switch is left alone
    with engine.precise_head(torch.nn.Linear(3, 2)):
        pass


@pytest.mark.parametrize("graph", [1, 0])
def test_main_cli_one_image_attack(graph, tmp_path, monkeypatch):
    """main.py end to end (the reference's one-image demo, main.py:29-100): default classifier (mobilenet_v2), the
    dictionary file named after the CLI's model string, 30 DDrague iterations — with the hipGraph replay main.py uses by
    default and with the eager loop; both must report the same labels."""
    import main as main_cli
    monkeypatch.chdir(tmp_path)
    os.makedirs("trained_dicts")
    g = torch.Generator().manual_seed(3)
    torch.save([-1 + 2 * torch.rand(3, 224, 224, 100, generator=g), torch.zeros(1), [], [], torch.tensor(0.)],
               "trained_dicts/ImageNet_mobilenet.bin")
    args = main_cli.build_parser().parse_args(["--synthetic", "--graph", str(graph), "--figure", str(tmp_path / "fig.png")])
    out = main_cli.main(args)
    assert isinstance(out, tuple) and len(out) == 2 and all(isinstance(v, int) for v in out)
    test_main_cli_one_image_attack.seen = getattr(test_main_cli_one_image_attack, "seen", {})
    test_main_cli_one_image_attack.seen[graph] = out
    if len(test_main_cli_one_image_attack.seen) == 2:
        assert test_main_cli_one_image_attack.seen[0] == test_main_cli_one_image_attack.seen[1]


@pytest.mark.parametrize("method", ["gd", "alter"])
def test_cached_pseudo_labels_match_the_golden_runs(method, tmp_path):
    """ADIL(cache_labels=True): the clean pseudo-label of an image is computed on its first visit and reused in later
    epochs (engine.LabelCache) instead of one extra classifier forward per step.  Same golden trajectories as the
    recomputing learners (G7 logits / G8: D, V, fooling rates exact), and the classifier really runs fewer forwards."""
    from attacks import ADIL
    z = load_golden("g7_learn_a" if method == "gd" else "g8_learn_b")
    net = net_on_gpu(z)
    calls = {"n": 0}
    hook = net.register_forward_hook(lambda *a: calls.__setitem__("n", calls["n"] + 1))
    tag = "logits_" if method == "gd" else ""
    common = dict(eps=float(z[f"{tag}eps"]), steps=int(z["steps"]), norm="linf", n_atoms=int(z["k"]),
                  batch_size=int(z["batch_size"]), data_val=None, step_size=float(z["step_size"]), loss="logits",
                  method=method, kappa=float(z["kappa"]), init_d=t(z[f"{tag}d0"]), epoch_batches=z[f"{tag}batches"].tolist(),
                  dict_dir=str(tmp_path))
    if method == "gd":
        common.update(init_v=t(z["logits_v0raw"]))
    else:
        common.update(init_v=torch.zeros(z["v0"].shape), steps_in=int(z["steps_in"]))
    runs = {}
    for cached in (False, True):
        calls["n"] = 0
        atk = ADIL(net, data_train=IndexedTensorDataset(t(z["images"])), model_name=f"lc{method}{int(cached)}",
                   cache_labels=cached, **common)
        runs[cached] = (torch.load(atk.model_file, map_location="cpu"), calls["n"])
    hook.remove()
    (d0, v0, l0, f0, _), n0 = runs[False]
    (d1, v1, l1, f1, _), n1 = runs[True]
    assert torch.equal(d0, d1) and torch.equal(v0, v1) and l0 == l1 and f0 == f1
    close(d1, z[f"{tag}d"], TOL_D, "D"); close(v1, z[f"{tag}v"], TOL_V, "V")
    assert list(f1) == list(z[f"{tag}fooling_rate_all"])
    steps_total = sum(len(e) for e in common["epoch_batches"][:len(l0) * (1 if method == "gd" else 2 * common.get("steps_in", 1))])
    first_epoch = len(common["epoch_batches"][0])
    assert n0 == 2 * steps_total and n1 == steps_total + first_epoch, (n0, n1, steps_total, first_epoch)


def test_classifier_batch_bucket_leaves_the_attack_unchanged(tmp_path):
    """engine.classifier_batch_bucket: the frozen classifier only sees batches rounded up to the bucket (zero rows
    appended, their logits / gradients dropped); predictions, input gradients and a whole DDrague attack on a ragged
    batch are the same as without it, and the classifier really ran at the bucket size."""
    from attacks import ADIL
    from dl_attack_on_imagenet_amd import engine
    from tinynet import make_tinynet
    net = make_tinynet(6).to(DEV)
    sizes = []
    hook = net.register_forward_hook(lambda m, inp, out: sizes.append(inp[0].shape[0]))
    g = torch.Generator().manual_seed(23)
    x = torch.rand(5, 3, 32, 32, generator=g).to(DEV)
    lab = engine.predict(net, x)
    out0, ls0, g0 = engine.input_gradient(net, x, lab, "ce", -1.0, 50.0, "mean")
    with engine.classifier_batch_bucket(8):
        sizes.clear()
        assert torch.equal(engine.predict(net, x), lab)
        out1, ls1, g1 = engine.input_gradient(net, x, lab, "ce", -1.0, 50.0, "mean")
        assert sizes == [8, 8] and out1.shape == out0.shape and g1.shape == g0.shape
    close(out1, out0, 1e-6); close(g1, g0, 1e-7 + 1e-5 * float(g0.abs().max())); close(ls1, ls0, 1e-6)
    sizes.clear()
    engine.predict(net, x)
    assert sizes == [5]                                                    # switched off again
    d = (-1 + 2 * torch.rand(3, 32, 32, 6, generator=g)).to(DEV)
    torch.save([d.cpu(), torch.zeros(1), [], [], torch.tensor(0.)], os.path.join(tmp_path, "ImageNet_bk.bin"))
    atk = ADIL(net, eps=0.1, n_atoms=6, attack="supervised", model_name="bk", loss="logits", steps_inference=12,
               dict_dir=str(tmp_path))
    plain = atk(x, lab)
    with engine.classifier_batch_bucket(8):
        sizes.clear()
        bucketed = atk(x, lab)
    assert set(sizes) == {8}
    close(bucketed, plain, 1e-6)
    assert torch.equal(engine.predict(net, bucketed), engine.predict(net, plain))
    hook.remove()


def test_val_every_writes_the_same_dictionary_file(tmp_path):
    """ADIL(val_every=0): validation (100 AdamW iterations per validation batch) after the last epoch only instead of
    after every epoch.  The reference only prints the per-epoch value and stores the last one (adil.py:199-210), so the
    dictionary file is the same — and equal to golden G7 — while the classifier runs a fraction of the forwards."""
    from attacks import ADIL
    z = load_golden("g7_learn_a")
    net = net_on_gpu(z)
    calls = {"n": 0}
    hook = net.register_forward_hook(lambda *a: calls.__setitem__("n", calls["n"] + 1))
    files, counts = [], []
    for every in (1, 0, 2):
        calls["n"] = 0
        atk = ADIL(net, eps=float(z["logits_eps"]), steps=int(z["steps"]), norm="linf", n_atoms=int(z["k"]),
                   batch_size=int(z["batch_size"]), data_train=IndexedTensorDataset(t(z["images"])),
                   data_val=IndexedTensorDataset(t(z["val"])), model_name=f"ve{every}", step_size=float(z["step_size"]),
                   loss="logits", method="gd", kappa=float(z["kappa"]), init_d=t(z["logits_d0"]), init_v=t(z["logits_v0raw"]),
                   epoch_batches=z["logits_batches"].tolist(), val_batches=z["logits_val_batches"].tolist(),
                   dict_dir=str(tmp_path), val_every=every)
        files.append(torch.load(atk.model_file, map_location="cpu"))
        counts.append(calls["n"])
    hook.remove()
    for d, v, loss_all, fool, val_fool in files[1:]:
        assert torch.equal(d, files[0][0]) and torch.equal(v, files[0][1]) and loss_all == files[0][2] and fool == files[0][3]
        assert float(val_fool) == float(files[0][4]) == float(z["logits_val_fool"])
    assert counts[1] < counts[2] < counts[0]


def test_val_every_keeps_the_seeded_shuffle_stream(tmp_path):
    """on the DEFAULT path (shuffled loaders, no injected batch order) a skipped validation still consumes the
    validation loader's draws from the global torch RNG, so a seeded run writes the same D, V, losses and stored
    validation value for val_every = 1, 2 and 0 (the end-of-run validation replays the skipped epoch's shuffle)."""
    from attacks import ADIL
    from tinynet import make_tinynet
    net = make_tinynet(5).to(DEV)
    g = torch.Generator().manual_seed(41)
    images, val = torch.rand(21, 3, 32, 32, generator=g), torch.rand(10, 3, 32, 32, generator=g)
    d0, v0 = -1 + 2 * torch.rand(3, 32, 32, 4, generator=g), torch.rand(21, 4, generator=g)
    files = []
    for every in (1, 2, 0):
        torch.manual_seed(2024)
        atk = ADIL(net, eps=0.3, steps=3, n_atoms=4, batch_size=6, data_train=IndexedTensorDataset(images),
                   data_val=IndexedTensorDataset(val), model_name=f"seeded{every}", loss="ce", init_d=d0, init_v=v0,
                   dict_dir=str(tmp_path), val_every=every)
        files.append(torch.load(atk.model_file, map_location="cpu"))
    for d, v, loss_all, fool, val_fool in files[1:]:
        assert torch.equal(d, files[0][0]) and torch.equal(v, files[0][1]) and loss_all == files[0][2] and fool == files[0][3]
        assert float(val_fool) == float(files[0][4])


def test_resident_batches_feed_the_evaluation_harness(tmp_path):
    """loader.ResidentBatches (the evaluation set resident in HBM, one gather kernel per batch) is a drop-in for the host
    DataLoader in performance.get_transfer_performance: same numbers on the G15 fixture; with shard=(rank, world) a rank
    uploads and yields only the batches performance.py deals to it (batch i -> rank i % world), placeholders elsewhere."""
    import performance as perf
    from attacks import ADIL
    from dl_attack_on_imagenet_amd import loader
    z = load_golden("g15_transfer")
    targets = {name: tinynet_from_npz(z, prefix=f"{name}.").to(DEV) for name in ("src", "t1", "t2")}
    torch.save([t(z["d"]), torch.zeros(1), [], [], torch.tensor(0.)], os.path.join(tmp_path, "ImageNet_g15.bin"))
    atk = ADIL(targets["src"], eps=float(z["eps"]), n_atoms=z["d"].shape[-1], attack="supervised", model_name="g15", loss="logits",
               steps_inference=int(z["steps"]), kappa=float(z["kappa"]), dict_dir=str(tmp_path))
    images, labels, bs = t(z["images"]), t(z["labels"]), int(z["batch_size"])
    ds = torch.utils.data.TensorDataset(images, labels)
    host = perf.get_transfer_performance({"adil": [atk]}, targets, torch.utils.data.DataLoader(ds, batch_size=bs), device=torch.device(DEV))
    res = loader.ResidentBatches(ds, labels, bs, DEV)
    assert len(res) == (len(ds) + bs - 1) // bs and len(res.dataset) == len(ds) and res.batch_size == bs
    resident = perf.get_transfer_performance({"adil": [atk]}, targets, res, device=torch.device(DEV))
    for name in targets:
        for key in ("fooling_rate", "rmse", "mse"):
            assert host["adil"][name][key] == resident["adil"][name][key], (name, key)
        assert abs(resident["adil"][name]["fooling_rate"] - float(z[f"{name}_fooling_rate"])) < 1e-9
    n = len(ds)
    for world in (2, 3):
        seen = []
        for rank in range(world):
            part = loader.ResidentBatches(ds, labels, bs, DEV, shard=(rank, world))
            got = list(part)
            assert len(got) == len(res)
            for i, (x, y) in enumerate(got):
                if i % world != rank:
                    assert x is None and y is None
                else:
                    lo, hi = i * bs, min((i + 1) * bs, n)
                    assert torch.equal(x.cpu(), images[lo:hi]) and torch.equal(y.cpu(), labels[lo:hi])
                    seen += list(range(lo, hi))
            assert len(part.images) == sum(min((i + 1) * bs, n) - i * bs for i in range(len(res)) if i % world == rank)
        assert sorted(seen) == list(range(n))


def test_bench_transfer_mode_line(tmp_path):
    """`bench.py --mode transfer` (configs[3] as a workload) at a plumbing size: one JSON line with the contract's keys, the
    six targets of the reference CLI scored, the z-step as the roofline kernel, the DDrague iteration count per batch."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "transfer", "--model", "resnet18", "--batch", "4",
                        "--atoms", "10", "--steps", "2", "--warmup", "1", "--steps-inference", "5", "--cpu-baseline", "0"],
                       capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["unit"] == "images/sec" and line["value"] > 0
    assert set(line["config"]["transfer_fooling_rates"]) == {"resnet18", "densenet121", "googlenet", "inception_v3", "mobilenet_v2", "vgg11"}
    # the roofline kernel = the launch group with the largest total time: the z-step at any real size; at this plumbing size
    # (4 images, 10 atoms) every kernel is launch latency and the two contractions of an iteration (`grad`: z D_dagger^T and
    # g D, two launches each iteration) can add up to more than the one z-step launch
    kern = line["roofline"]["kernel"]
    # round 4: the z-step also produces the next iteration's codes, so an iteration has ONE contraction launch (`grad`: g D)
    assert kern in ("zstep_codes_", "grad", "synth") and line["roofline"]["launches_timed"] == 2 * 5 + (2 if kern == "synth" else 0)
    assert "grad[z D_dagger^T]" not in line["kernels_ms_per_launch"]
    assert max(line["kernels_ms_per_launch"].values()) < 0.5, line["kernels_ms_per_launch"]   # all launch latency at 4 images x 10 atoms
    assert line["config"]["ddrague_iterations_run_per_batch"] == 5
    # two ranks (one-GPU rehearsal: gloo, shared card): performance.py deals batch i to rank i % 2, each rank keeps only its
    # own batches resident, the final sums are all-reduced; the line says which backend it ran on
    env = dict(os.environ, ADIL_DIST_BACKEND="gloo", ADIL_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "transfer", "--model", "resnet18",
                        "--batch", "4", "--atoms", "10", "--steps", "1", "--warmup", "0", "--steps-inference", "3", "--cpu-baseline", "0"],
                       capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["config"]["collective"]["backend"] == "gloo"
    assert abs(line["value"] - 2 * 4 * 1 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]


def test_bench_inference_and_learn_mode_lines(tmp_path):
    """`bench.py --mode inference` and the default learn mode at a plumbing size: one JSON line each with the contract's keys;
    the learn line carries BOTH pseudo-label policies at the top level (`value` = cached labels by default, the reference's
    recomputing sequence next to it) and `--cache-labels 0` swaps them."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--model", "resnet18", "--batch", "8", "--atoms", "10", "--steps", "3", "--warmup", "1", "--cpu-baseline", "0"]

    def line(*extra):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + list(extra), capture_output=True, text=True,
                           timeout=600, cwd=str(tmp_path))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    inf = line("--mode", "inference")
    # the z-step at any real size; at 8 images x 10 atoms every launch is latency and any of the iteration's kernels can be slowest
    assert inf["roofline"]["kernel"] in ("zstep_codes_", "grad", "synth")
    # why a set and not "the z-step" (VERDICT r3 #6): at this plumbing size no launch moves more than 30 MB — every group is
    # launch latency (tens of microseconds), and which of them is slowest is decided by the box, not by the kernels
    assert max(inf["kernels_ms_per_step"].values()) < 0.5, inf["kernels_ms_per_step"]
    assert inf["unit"] == "images/sec" and inf["n_gpus"] == 1 and inf["steps"] == 3
    # round 4: no contraction launch for z D_dagger^T any more — the z-step leaves the next iteration's codes itself
    assert set(inf["kernels_ms_per_step"]) >= {"synth", "grad", "zstep_codes_", "pack_codes"}
    assert "grad[z D_dagger^T]" not in inf["kernels_ms_per_step"] and "zstep_" not in inf["kernels_ms_per_step"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in inf["roofline"]
    learn = line()
    assert "cached" in learn["config"]["workload"] and learn["value_cached_labels"] == learn["value"]
    assert "BENCH_r02" in learn["value_definition"] and "cached" in learn["value_definition"]
    assert learn["value_reference_op_sequence_recomputed_labels"] == learn["config"]["recomputed_labels_variant"]["images_per_sec"]
    assert learn["config"]["collective"]["world_size"] == 1 and learn["config"]["collective"]["backend"] is None
    ref = line("--cache-labels", "0")
    assert "recomputed" in ref["config"]["workload"] and ref["value_reference_op_sequence_recomputed_labels"] == ref["value"]
    assert ref["value_cached_labels"] == ref["config"]["cached_labels_variant"]["images_per_sec"]


def test_resident_images_upload_through_worker_processes():
    """loader.ResidentImages(num_workers=2): the items fetched by DataLoader worker processes (what a JPEG-decoding dataset
    needs for its one-time upload) give the same resident tensor as the in-process fetch — all rows and a row subset,
    with and without the `indexed` protocol switched on."""
    from dl_attack_on_imagenet_amd import loader
    g = torch.Generator().manual_seed(8)
    ds = IndexedTensorDataset(torch.rand(37, 3, 16, 16, generator=g))
    ds.indexed = True
    a = loader.ResidentImages(ds, DEV, torch.bfloat16, chunk=8)
    b = loader.ResidentImages(ds, DEV, torch.bfloat16, chunk=8, num_workers=2)
    assert ds.indexed is True and torch.equal(a.images, b.images) and torch.equal(a.images.float().cpu(), ds.images.to(torch.bfloat16).float())
    rows = [5, 0, 36, 7, 7, 20]
    c = loader.ResidentImages(ds, DEV, torch.float32, rows=rows, chunk=4, num_workers=2)
    assert torch.equal(c.images.cpu(), ds.images[rows])
