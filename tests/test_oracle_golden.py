"""CPU: the oracle (oracle/adil_oracle.py) against every golden vector produced by the reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, t
from oracle import adil_oracle as O
from tinynet import tinynet_from_npz

torch.set_num_threads(4)


def close(a, b, tol):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float((a - b).abs().max()) if a.numel() else 0.0
    assert err <= tol, err


def test_g1_l1ball():
    z = load_golden("g1_l1ball")
    for tag in "abcde":
        close(O.project_onto_l1_ball(t(z[f"x_{tag}"]), float(z["eps"])), z[f"y_{tag}"], 1e-7)


def test_g2_constraints():
    z = load_golden("g2_constraints")
    eps = float(z["eps"])
    close(O.constraint_dict(t(z["d"]), "l2ball"), z["l2ball"], 1e-7)
    close(O.constraint_dict(t(z["d"]), "l2sphere"), z["l2sphere"], 1e-7)
    close(O.projection_v(t(z["v"]), eps, "l2"), z["pv_l2"], 1e-7)
    close(O.projection_v(t(z["v"]), eps, "linf"), z["pv_linf"], 1e-7)
    close(O.projection_d(t(z["d"]), "l2"), z["pd_l2"], 1e-7)
    close(O.projection_d(t(z["pd_linf_in"]), "linf"), z["pd_linf"], 0)


def test_g3_softshrink():
    z = load_golden("g3_softshrink")
    close(O.softshrink(t(z["x"]), float(z["lam"])), z["y"], 0)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g4_synth_grad(tag):
    z = load_golden("g4_synth_grad")
    d, v, x, idx, g = (t(z[f"{n}_{tag}"]) for n in ("d", "v", "x", "index", "g"))
    close(O.synth(x, d, v[idx]), z[f"y_{tag}"], 1e-6)
    gd, gvr = O.grad_dv(g, d, v[idx])
    gv = torch.zeros_like(v)
    gv[idx] = gvr
    close(gd, z[f"grad_d_{tag}"], 1e-5)
    close(gv, z[f"grad_v_{tag}"], 1e-4)


def test_g5_floss():
    z = load_golden("g5_floss")
    lg = t(z["logits"]).requires_grad_(True)
    val = O.f_loss(lg, t(z["labels"]), float(z["kappa"]))
    val.sum().backward()
    close(val.detach(), z["value"], 0)
    close(lg.grad, z["grad"], 0)


def test_g6_adamw_steps():
    z = load_golden("g6_adamw_steps")
    eps, lr = float(z["eps"]), float(z["lr"])
    d, v = t(z["d0"]).clone(), t(z["v0"]).clone()
    sd, sv = O.AdamWState(d, lr), O.AdamWState(v, lr)
    for step in range(z["g"].shape[0]):
        idx = t(z["index"][step])
        gd, gvr = O.grad_dv(t(z["g"][step]), d, v[idx])
        gv = torch.zeros_like(v)
        gv[idx] = gvr
        sd.step(d, gd)
        sv.step(v, gv)
        v.copy_(O.project_onto_l1_ball(v, eps))
        d.clamp_(-1, 1)
        close(d, z["d_hist"][step], 2e-6)
        close(v, z["v_hist"][step], 2e-6)
    close(sd.m, z["m_d"], 1e-6); close(sd.v, z["s_d"], 1e-6)
    close(sv.m, z["m_v"], 1e-6); close(sv.v, z["s_v"], 1e-6)
    assert sd.t == int(z["step"])


@pytest.mark.parametrize("tag", ["ce", "logits"])
def test_g7_learn_a(tag):
    z = load_golden("g7_learn_a")
    net = tinynet_from_npz(z)
    o = O.learn_dictionary_a(net, t(z["images"]), t(z[f"{tag}_d0"]), t(z[f"{tag}_v0"]), z[f"{tag}_batches"].tolist(),
                             float(z[f"{tag}_eps"]), float(z["step_size"]), tag, False, float(z["kappa"]),
                             t(z["val"]), z[f"{tag}_val_batches"].tolist())
    close(o["d"], z[f"{tag}_d"], 5e-5)
    close(o["v"], z[f"{tag}_v"], 5e-5)
    close(o["loss_all"], z[f"{tag}_loss_all"], 1e-3)
    close(o["fooling_rate_all"], z[f"{tag}_fooling_rate_all"], 0)
    close(o["val_fool"], z[f"{tag}_val_fool"], 0)
    close(O.project_onto_l1_ball(t(z[f"{tag}_v0raw"]), float(z[f"{tag}_eps"])), z[f"{tag}_v0"], 1e-7)


def test_g8_learn_b():
    z = load_golden("g8_learn_b")
    net = tinynet_from_npz(z)
    steps, steps_in = int(z["steps"]), int(z["steps_in"])
    epochs = z["batches"].tolist()
    outer = [(epochs[o * 2 * steps_in: o * 2 * steps_in + steps_in],
              epochs[o * 2 * steps_in + steps_in: (o + 1) * 2 * steps_in]) for o in range(steps // steps_in)]
    o = O.learn_dictionary_b(net, t(z["images"]), t(z["d0"]), t(z["v0"]), outer, float(z["eps"]), steps_in,
                             float(z["step_size"]), "logits", False, float(z["kappa"]))
    close(o["d"], z["d"], 5e-5); close(o["v"], z["v"], 5e-5)
    close(o["loss_all"], z["loss_all"], 1e-3); close(o["fooling_rate_all"], z["fooling_rate_all"], 0)


@pytest.mark.parametrize("tag", ["ce", "logits"])
def test_g9_ddrague(tag):
    z = load_golden("g9_ddrague")
    net = tinynet_from_npz(z)
    adv = O.forward_supervised_ddrague(net, t(z["images"]), t(z["d"]), float(z["eps"]), int(z[f"{tag}_steps"]), tag,
                                       False, float(z["kappa"]))
    close(adv, z[f"{tag}_adv"], 2e-5)
    assert torch.equal(net(adv).argmax(-1), t(z[f"{tag}_adv_labels"]))          # bit-exact label decisions
    assert float((adv - t(z["images"])).abs().max()) > float(z["eps"])          # quirk Q6 is part of the contract


@pytest.mark.parametrize("tag", ["ce", "logits"])
def test_g10_adamw_inference(tag):
    z = load_golden("g10_adamw_inference")
    net = tinynet_from_npz(z)
    args = (net, t(z["images"]), t(z["d"]), float(z["eps"]), tag, False, float(z["kappa"]), "linf")
    assert O.forward_supervised_adamw(*args, "train") == int(z[f"{tag}_count"])
    adv, tr = O.forward_supervised_adamw(*args, "attack", return_trace=True)
    close(adv, z[f"{tag}_adv"], 2e-5)
    close(tr["v"], z[f"{tag}_v"], 2e-5)
    assert tr["iters"] == int(z[f"{tag}_iters"])


@pytest.mark.parametrize("norm", ["linf", "l2"])
def test_g11_unsupervised(norm):
    z = load_golden("g11_unsupervised")
    net = tinynet_from_npz(z)
    eps = float(z["eps"])
    for u, s in zip(z[f"{norm}_u"], z[f"{norm}_v_trials"]):
        close(O.sample_sphere_from_uniform(t(u), eps, norm), s, 1e-6)
    adv, dvn = O.forward_unsupervised(net, t(z["images"]), t(z["d"]), eps, [t(s) for s in z[f"{norm}_v_trials"]])
    close(adv, z[f"{norm}_adv"], 1e-6)
    close(dvn, z[f"{norm}_dv_norm_inf"], 1e-6)


def test_g12_ista_and_metrics():
    z = load_golden("g12_ista_metrics")
    net = tinynet_from_npz(z)
    images, labels, d = t(z["images"]), t(z["labels"]), t(z["d"])
    lam, step = float(z["lam"]), float(z["step"])
    v, _ = O.learn_coding_vectors(net, images, labels, d, True, 6, float(z["lcv_lambda_l1"]), lam, 3, step)
    close(v, z["lcv_v"], 2e-5)
    od, ov, ol = O.adil_full_batch(net, images, labels, t(z["adil_d0"]), True, 4, lam, lam, 3, step)
    close(od, z["adil_d"], 5e-5); close(ov, z["adil_v"], 5e-5); close(ol, z["adil_loss"], 1e-3)
    sd, sv, _ = O.sadil(net, images, labels, t(z["sadil_d0"]), True, 2, 3, lam, lam, step)
    close(sd, z["sadil_d"], 5e-5); close(sv, z["sadil_v"], 5e-5)
    adv = t(z["metric_adv"])
    assert O.compute_fooling_rate(net, adv, images) == float(z["fooling"])
    close(O.compute_rmse(adv, images), z["rmse"], 1e-6)
    close(O.compute_mse(adv, images), z["mse"], 1e-4)
    ylab = t(z["perf_labels"])
    loader = [(images[:3], ylab[:3]), (images[3:], ylab[3:])]
    perf = O.performance(lambda x, y: (x + 0.08 * torch.sign(x - 0.5)).clamp(0, 1), net, loader)
    close(perf["fooling_rate"], z["perf_fooling_rate"], 1e-6)
    close(perf["rmse"], z["perf_rmse"], 1e-6)
    close(perf["mse"], z["perf_mse"], 1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g13_sadil_updated(tag):
    z = load_golden("g13_sadil_updated")
    net = tinynet_from_npz(z)
    d, v, loss = O.sadil_updated(net, t(z["images"]), t(z["labels"]), t(z[f"{tag}_d0"]), True, 3, 2, float(z[f"{tag}_lam"]),
                                 float(z["l2"]), float(z[f"{tag}_step"]))
    close(d, z[f"{tag}_d"], 5e-5); close(v, z[f"{tag}_v"], 5e-5); close(loss, z[f"{tag}_loss"], 1e-3)


@pytest.mark.parametrize("tag,norm,optim", [("linf_adam", "linf", "adam"), ("l2_sgd", "l2", "sgd")])
def test_g14_uappgd(tag, norm, optim):
    """The UAPPGD baseline restatement against the reference's own run (uappgd.py:70-107), shuffled batch order as recorded."""
    z = load_golden("g14_uappgd")
    net = tinynet_from_npz(z)
    batches = [[list(map(int, b)) for b in e] for e in z[f"{tag}_batches"]]
    attack, fooling, fooled = O.uappgd_learn(net, t(z["images"]), t(z["labels"]).long(), batches, float(z[f"{tag}_lr"]), norm,
                                             float(z[f"{tag}_eps"]), float(z["beta"]), optim, t(z["val"]))
    close(attack, z[f"{tag}_attack"], 1e-6)
    close(torch.stack(fooling), z[f"{tag}_fooling"], 0)
    assert [int(f) for f in fooled] == [int(f) for f in z[f"{tag}_train_fooled"]]


def test_g15_transfer_performance():
    """Transfer evaluation (performance.py:183-232) against the reference's own numbers: adversaries from DDrague
    against the source net, scored on the source and two other targets, sums divided by the dataset size."""
    z = load_golden("g15_transfer")
    targets = {name: tinynet_from_npz(z, prefix=f"{name}.") for name in ("src", "t1", "t2")}
    images, labels, d, n, bs = t(z["images"]), t(z["labels"]), t(z["d"]), len(z["images"]), int(z["batch_size"])
    perf = O.transfer_performance(
        lambda x, y: O.forward_supervised_ddrague(targets["src"], x, d, float(z["eps"]), int(z["steps"]), "logits", False,
                                                  float(z["kappa"])),
        targets, [(images[i:i + bs], labels[i:i + bs]) for i in range(0, n, bs)], n)
    for name in targets:
        close(perf[name]["fooling_rate"], z[f"{name}_fooling_rate"], 1e-6)
        close(perf[name]["rmse"], z[f"{name}_rmse"], 1e-6)
        close(perf[name]["mse"], z[f"{name}_mse"], 1e-4)


def test_g16_constraint_dict_l1_branch():
    z = load_golden("g16_constraint_l1")
    close(O.constraint_dict(t(z["d"]), "l1ball"), z["l1ball"], 1e-7)
