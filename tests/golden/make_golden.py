"""Generate the golden fixtures (tests/golden/*.npz) by RUNNING THE REFERENCE.

Build-container only: imports /root/reference through ref_import.py, drives its
functions on seeded CPU inputs with the tiny classifiers of tests/tinynet.py,
and stores inputs + the reference's outputs.  Fixtures are data only (inputs,
expected outputs, the recorded batch order); no reference source is stored.
Each fixture is also cross-checked here against oracle/adil_oracle.py so a
fixture that the oracle cannot reproduce is caught at generation time.

    cd /root/repo && python tests/golden/make_golden.py
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)

import ref_import  # noqa: E402
from tinynet import make_tinynet, state_to_npz_dict  # noqa: E402

U, A, R, PERF = ref_import.load_reference()
from oracle import adil_oracle as O  # noqa: E402  (imported AFTER the reference so names cannot mix)

torch.set_num_threads(1)          # deterministic reductions while generating
EPS = 8 / 255


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def close(a, b, tol=1e-5, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol, f"oracle/reference mismatch {what}: {err}"
    return err


class RecordingDataset(torch.utils.data.Dataset):
    """Implements the reference's `indexed` protocol (imagenet_loading.py:8-18)
    and logs which items the DataLoader asked for."""

    def __init__(self, images, labels):
        self.images, self.labels = images, labels
        self.indexed = False
        self.log = []

    def __len__(self):
        return len(self.images)

    def __getitem__(self, item):
        if item >= len(self.images):
            raise IndexError
        if self.indexed:
            self.log.append(int(item))
            return item, self.images[item], self.labels[item]
        return self.images[item], self.labels[item]


class RecordingValDataset(RecordingDataset):
    def __getitem__(self, item):
        if item >= len(self.images):
            raise IndexError
        self.log.append(int(item))
        return self.images[item], self.labels[item]


def chunk(seq, n):
    return [seq[i:i + n] for i in range(0, len(seq), n)]


@contextlib.contextmanager
def scratch_cwd():
    old = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "trained_dicts"))
        os.chdir(tmp)
        try:
            yield tmp
        finally:
            os.chdir(old)


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
        return fn(*a, **k)


# --------------------------------------------------------------------------- #
def g1_l1ball():
    g = torch.Generator().manual_seed(101)
    arrays = {}
    for tag, (n, k) in {"a": (7, 10), "b": (33, 50), "c": (5, 100), "d": (9, 64), "e": (4, 128)}.items():
        x = torch.randn(n, k, generator=g) * 0.02
        x[0] = 0.0                                    # zeros row (inside)
        x[1] = x[1] * 0.05                            # well inside the ball
        x[2] = x[2] / x[2].abs().sum() * EPS          # (numerically) on the ball
        x[3] = x[3] * 50                              # far outside -> very sparse result
        if n > 4:
            x[4, : k // 2] = 0.01                     # ties
            x[4, k // 2:] = -0.01
        if n > 5:
            x[5] = 0.0
            x[5, 3] = 1.0                             # single spike
        y = U.project_onto_l1_ball(x.clone(), EPS)
        close(O.project_onto_l1_ball(x, EPS), y, 1e-7, "l1ball")
        arrays[f"x_{tag}"], arrays[f"y_{tag}"] = x, y
    # atom-wise l1 (constraint_dict else-branch uses eps=1 on (C,H,W) slices: rows = C)
    save("g1_l1ball", eps=EPS, **arrays)


def g2_constraints():
    g = torch.Generator().manual_seed(102)
    d = torch.randn(3, 8, 8, 6, generator=g) * 0.2
    d[..., 0] *= 0.01                                  # an atom strictly inside the unit ball
    ball = U.constraint_dict(d.clone(), "l2ball")
    sphere = U.constraint_dict(d.clone(), "l2sphere")
    close(O.constraint_dict(d, "l2ball"), ball, 1e-7)
    close(O.constraint_dict(d, "l2sphere"), sphere, 1e-7)
    v = torch.randn(12, 6, generator=g) * 0.05
    v[0] *= 0.01

    class _S:  # minimal self for the unbound reference methods
        pass
    s = _S(); s.eps = EPS
    s.norm = "l2"
    pv_l2 = A.ADIL.projection_v(s, v.clone())
    pd_l2 = A.ADIL.projection_d(s, d.clone())
    s.norm = "linf"
    pv_linf = A.ADIL.projection_v(s, v.clone())
    pd_linf = A.ADIL.projection_d(s, (d * 8).clone())
    close(O.projection_v(v, EPS, "l2"), pv_l2, 1e-7)
    close(O.projection_v(v, EPS, "linf"), pv_linf, 1e-7)
    close(O.projection_d(d * 8, "linf"), pd_linf, 0)
    save("g2_constraints", eps=EPS, d=d, l2ball=ball, l2sphere=sphere, v=v, pv_l2=pv_l2, pv_linf=pv_linf,
         pd_l2=pd_l2, pd_linf_in=d * 8, pd_linf=pd_linf)


def g3_softshrink():
    g = torch.Generator().manual_seed(103)
    x = torch.randn(17, 12, generator=g) * 0.1
    x[0, 0], x[0, 1], x[0, 2] = 0.03, -0.03, 0.0
    lam = 0.03
    y = U.get_prox_l1(lam)(x)
    close(O.softshrink(x, lam), y, 0)
    save("g3_softshrink", x=x, lam=lam, y=y)


def g4_synth_grad():
    g = torch.Generator().manual_seed(104)
    out = {}
    for tag, (b, n, c, h, w, k) in {"a": (5, 9, 3, 8, 8, 4), "b": (6, 6, 3, 16, 16, 10), "c": (3, 7, 3, 12, 20, 50)}.items():
        d = -1 + 2 * torch.rand(c, h, w, k, generator=g)
        v = torch.randn(n, k, generator=g) * 0.01
        x = torch.rand(b, c, h, w, generator=g)
        index = torch.randperm(n, generator=g)[:b]
        gup = torch.randn(b, c, h, w, generator=g)
        m = A.Attack_dict_model(d.clone(), v.clone(), EPS)
        y = m(x, index, lambda t: t)
        y.backward(gup)
        gd, gv_rows = O.grad_dv(gup, d, v[index])
        close(O.synth(x, d, v[index]), y, 1e-6)
        close(gd, m.d.grad, 1e-5)
        close(gv_rows, m.v.grad[index], 1e-4)
        out.update({f"d_{tag}": d, f"v_{tag}": v, f"x_{tag}": x, f"index_{tag}": index, f"g_{tag}": gup,
                    f"y_{tag}": y, f"grad_d_{tag}": m.d.grad, f"grad_v_{tag}": m.v.grad})
    save("g4_synth_grad", **out)


def g5_floss():
    g = torch.Generator().manual_seed(105)
    logits = torch.randn(12, 10, generator=g) * 3
    labels = logits.argmax(dim=1)
    labels[3] = (labels[3] + 1) % 10               # a mis-predicted sample
    logits[4] = -logits[4].abs() - 1.0             # all logits negative -> i clamps at 0 (quirk Q5)
    labels[4] = logits[4].argmax()
    logits[5, labels[5]] += 100.0                  # margin > kappa -> not clamped from below, large value
    logits[6, labels[6]] -= 100.0                  # j - i < -kappa -> clamped at -kappa
    net = make_tinynet(1)

    class _S:
        pass
    s = _S(); s.device = torch.device("cpu"); s._targeted = False; s.kappa = 50.0
    lg = logits.clone().requires_grad_(True)
    val = A.ADIL.f_loss(s, lg, labels)
    val.sum().backward()
    lo = logits.clone().requires_grad_(True)
    vo = O.f_loss(lo, labels, 50.0)
    vo.sum().backward()
    close(vo, val, 0); close(lo.grad, lg.grad, 0)
    save("g5_floss", logits=logits, labels=labels, kappa=50.0, value=val, grad=lg.grad)


def g6_adamw_steps():
    """T steps of {AdamW(d,v) ; update_v ; update_d} driven by given upstream grads."""
    g = torch.Generator().manual_seed(106)
    c, h, w, k, n, b, T = 3, 8, 8, 6, 10, 4, 5
    d0 = -1 + 2 * torch.rand(c, h, w, k, generator=g)
    v0 = U.project_onto_l1_ball(torch.rand(n, k, generator=g), EPS)
    x = torch.rand(b, c, h, w, generator=g)
    idxs = [torch.randperm(n, generator=g)[:b] for _ in range(T)]
    gups = [torch.randn(b, c, h, w, generator=g) * (10.0 ** (-t)) for t in range(T)]
    m = A.Attack_dict_model(d0.clone(), v0.clone(), EPS)
    opt = torch.optim.AdamW(m.parameters(), lr=0.01)
    od, ov = d0.clone(), v0.clone()
    sd, sv = O.AdamWState(od, 0.01), O.AdamWState(ov, 0.01)
    d_hist, v_hist = [], []
    for t in range(T):
        opt.zero_grad()
        m(x, idxs[t], lambda z: z).backward(gups[t])
        opt.step(); m.update_v(); m.update_d()
        d_hist.append(m.d.data.clone()); v_hist.append(m.v.data.clone())
        gd, gvr = O.grad_dv(gups[t], od, ov[idxs[t]])
        gv = torch.zeros_like(ov); gv[idxs[t]] = gvr
        sd.step(od, gd); sv.step(ov, gv)
        ov.copy_(O.project_onto_l1_ball(ov, EPS)); od.clamp_(-1, 1)
        close(od, m.d.data, 2e-6, f"adamw d t={t}"); close(ov, m.v.data, 2e-6, f"adamw v t={t}")
    st = opt.state[m.d]
    sv_ = opt.state[m.v]
    save("g6_adamw_steps", eps=EPS, lr=0.01, d0=d0, v0=v0, x=x, index=torch.stack(idxs), g=torch.stack(gups),
         d_hist=torch.stack(d_hist), v_hist=torch.stack(v_hist), m_d=st["exp_avg"], s_d=st["exp_avg_sq"],
         m_v=sv_["exp_avg"], s_v=sv_["exp_avg_sq"], step=int(st["step"]))


def _learn_setup(seed, n=16, nval=8, c=3, h=16, w=16):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(n, c, h, w, generator=g)
    val = torch.rand(nval, c, h, w, generator=g)
    net = make_tinynet(seed + 1000)
    return images, val, net


def g7_learn_a():
    out = {}
    for tag, loss, eps in (("ce", "ce", 1.5), ("logits", "logits", 1.5)):
        images, val, net = _learn_setup(107)
        n, k, bs, steps, seed = images.shape[0], 4, 8, 5, 7
        labels = torch.zeros(n, dtype=torch.long)
        train_ds = RecordingDataset(images, labels)
        val_ds = RecordingValDataset(val, torch.zeros(len(val), dtype=torch.long))
        with scratch_cwd():
            torch.manual_seed(seed)
            atk = quiet(A.ADIL, net, eps=eps, steps=steps, norm="linf", n_atoms=k, batch_size=bs, data_train=train_ds,
                        data_val=val_ds, model_name=f"g7{tag}", step_size=0.01, loss=loss, method="gd", kappa=50)
            res = torch.load(atk.model_file)
        # re-derive the reference's own random draws (adil.py:148,150) from the same seed
        torch.manual_seed(seed)
        d0 = -1 + 2 * torch.rand(3, 16, 16, k)
        v0raw = torch.rand(n, k)
        v0 = U.project_onto_l1_ball(v0raw.clone(), eps)
        epochs = chunk(train_ds.log, n)
        epochs_batches = [chunk(e, bs) for e in epochs]
        vlog = val_ds.log[1:] if len(val_ds.log) % len(val) else val_ds.log
        val_epochs = [chunk(e, bs) for e in chunk(vlog, len(val))]
        o = O.learn_dictionary_a(net, images, d0, v0, epochs_batches, eps, 0.01, loss, False, 50.0, val, val_epochs)
        e1 = close(o["d"], res[0], 5e-5, "learn_a D"); e2 = close(o["v"], res[1], 5e-5, "learn_a V")
        close(o["loss_all"], res[2], 1e-3, "learn_a loss"); close(o["fooling_rate_all"], res[3], 0, "learn_a fool")
        close(o["val_fool"], float(res[4]), 0, "learn_a val")
        print(f"  g7[{tag}] oracle-vs-reference: D {e1:.2e} V {e2:.2e} fooling {res[3]} val {float(res[4])}")
        out.update({f"{tag}_d0": d0, f"{tag}_v0": v0, f"{tag}_v0raw": v0raw,
                    f"{tag}_batches": np.array(epochs_batches), f"{tag}_val_batches": np.array(val_epochs),
                    f"{tag}_d": res[0], f"{tag}_v": res[1], f"{tag}_loss_all": np.array(res[2]),
                    f"{tag}_fooling_rate_all": np.array(res[3]), f"{tag}_val_fool": float(res[4]), f"{tag}_eps": eps})
    images, val, net = _learn_setup(107)
    save("g7_learn_a", images=images, val=val, k=4, batch_size=8, steps=5, step_size=0.01, kappa=50.0,
         **state_to_npz_dict(net), **out)


def g8_learn_b():
    images, val, net = _learn_setup(108)
    n, k, bs, steps, steps_in, seed, eps, loss = images.shape[0], 4, 8, 4, 2, 11, 0.25, "logits"
    train_ds = RecordingDataset(images, torch.zeros(n, dtype=torch.long))
    val_ds = RecordingValDataset(val, torch.zeros(len(val), dtype=torch.long))
    with scratch_cwd():
        torch.manual_seed(seed)
        atk = quiet(A.ADIL, net, eps=eps, steps=steps, norm="linf", n_atoms=k, batch_size=bs, data_train=train_ds,
                    data_val=val_ds, model_name="g8", step_size=0.01, loss=loss, method="alter", steps_in=steps_in,
                    kappa=50)
        res = torch.load(atk.model_file)
    torch.manual_seed(seed)
    d0 = -1 + 2 * torch.rand(3, 16, 16, k)
    v0 = U.project_onto_l1_ball(torch.zeros(n, k), eps)
    epochs = [chunk(e, bs) for e in chunk(train_ds.log, n)]
    outer = []
    for o_ in range(steps // steps_in):
        base = o_ * 2 * steps_in
        outer.append((epochs[base: base + steps_in], epochs[base + steps_in: base + 2 * steps_in]))
    o = O.learn_dictionary_b(net, images, d0, v0, outer, eps, steps_in, 0.01, loss, False, 50.0)
    e1 = close(o["d"], res[0], 5e-5, "learn_b D"); e2 = close(o["v"], res[1], 5e-5, "learn_b V")
    close(o["loss_all"], res[2], 1e-3); close(o["fooling_rate_all"], res[3], 0)
    print(f"  g8 oracle-vs-reference: D {e1:.2e} V {e2:.2e} fooling {res[3]}")
    save("g8_learn_b", images=images, val=val, k=k, batch_size=bs, steps=steps, steps_in=steps_in, step_size=0.01,
         eps=eps, kappa=50.0, d0=d0, v0=v0, batches=np.array(epochs), d=res[0], v=res[1],
         loss_all=np.array(res[2]), fooling_rate_all=np.array(res[3]), **state_to_npz_dict(net))


def _attack_fixture(seed, k=6, b=8, eps=0.12):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(b, 3, 16, 16, generator=g)
    d = (-1 + 2 * torch.rand(3, 16, 16, k, generator=g))
    net = make_tinynet(seed + 1000)
    labels = net(images).argmax(dim=-1)
    return images, labels, d, net


def g9_ddrague():
    out = {}
    for tag, loss, steps in (("ce", "ce", 12), ("logits", "logits", 12)):
        images, labels, d, net = _attack_fixture(109)
        eps = 0.12
        with scratch_cwd():
            torch.save([d, torch.zeros(1), [], [], torch.tensor(0.)], "trained_dicts/ImageNet_g9.bin")
            atk = quiet(A.ADIL, net, eps=eps, n_atoms=d.shape[-1], attack="supervised", model_name="g9", loss=loss,
                        steps_inference=steps, kappa=50)
            adv = atk(images, labels).detach()
        oadv, tr = O.forward_supervised_ddrague(net, images, d, eps, steps, loss, False, 50.0, return_trace=True)
        e = close(oadv, adv, 2e-5, "ddrague adv")
        assert torch.equal(net(oadv).argmax(-1), net(adv).argmax(-1))
        print(f"  g9[{tag}] adv err {e:.2e}  max|adv-x| {float((adv - images).abs().max()):.4f} (eps {eps})"
              f"  fooled {int((net(adv).argmax(-1) != labels).sum())}/{len(labels)}")
        out.update({f"{tag}_adv": adv, f"{tag}_adv_labels": net(adv).argmax(-1), f"{tag}_steps": steps})
    save("g9_ddrague", images=images, labels=labels, d=d, eps=0.12, kappa=50.0, **state_to_npz_dict(net), **out)


def g10_adamw_inference():
    out = {}
    images, labels, d, net = _attack_fixture(110, k=6, b=8)
    eps = 0.3

    class _Shim(A.ADIL):
        def __init__(self, model, **kw):              # bypass learning: only the fields the method reads
            ref_import._AttackBase.__init__(self, "ADIL", model)
            self.__dict__.update(kw)

    for tag, loss in (("ce", "ce"), ("logits", "logits")):
        atk = _Shim(net, eps=eps, n_atoms=d.shape[-1], targeted=False, loss=loss, kappa=50.0, norm="linf")
        cnt = quiet(atk.forward_supervised_AdamW, images, labels, d.clone(), "train")
        adv = quiet(atk.forward_supervised_AdamW, images, labels, d.clone(), "attack").detach()
        ocnt = O.forward_supervised_adamw(net, images, d, eps, loss, False, 50.0, "linf", "train")
        oadv, tr = O.forward_supervised_adamw(net, images, d, eps, loss, False, 50.0, "linf", "attack", return_trace=True)
        assert int(cnt) == ocnt, (int(cnt), ocnt)
        e = close(oadv, adv, 2e-5, "adamw-inference adv")
        print(f"  g10[{tag}] count {int(cnt)}/{len(labels)} adv err {e:.2e} iters {tr['iters']}")
        out.update({f"{tag}_count": int(cnt), f"{tag}_adv": adv, f"{tag}_v": tr["v"], f"{tag}_iters": tr["iters"]})
    save("g10_adamw_inference", images=images, labels=labels, d=d, eps=eps, kappa=50.0, **state_to_npz_dict(net), **out)


def g11_unsupervised():
    images, labels, d, net = _attack_fixture(111, k=6, b=8)
    eps, trials = 0.2, 4
    out = {}
    for norm in ("linf", "l2"):
        with scratch_cwd():
            torch.save([d, torch.zeros(1), [], [], torch.tensor(0.)], "trained_dicts/ImageNet_g11.bin")
            atk = quiet(A.ADIL, net, eps=eps, n_atoms=d.shape[-1], attack="unsupervised", model_name="g11",
                        trials=trials, norm=norm)
            samples, us = [], []
            orig = atk.sample_sphere

            def rec(n_samples, _orig=orig):
                state = torch.get_rng_state()
                s = _orig(n_samples)
                torch.set_rng_state(state)            # replay the draw to capture the underlying uniforms
                u = torch.rand(n_samples, d.shape[-1], 1)[:, :, 0] if norm == "linf" else torch.rand(n_samples, d.shape[-1])
                samples.append(s.clone()); us.append(u)
                return s
            atk.sample_sphere = rec
            torch.manual_seed(5)
            adv, dv_norm = atk(images, labels)
        for s, u in zip(samples, us):
            close(O.sample_sphere_from_uniform(u, eps, norm), s, 1e-6, "sample_sphere")
        oadv, onorm = O.forward_unsupervised(net, images, d, eps, samples)
        close(oadv, adv, 1e-6, "unsupervised adv"); close(onorm, dv_norm, 1e-6)
        out.update({f"{norm}_v_trials": torch.stack(samples), f"{norm}_u": torch.stack(us), f"{norm}_adv": adv,
                    f"{norm}_dv_norm_inf": np.array(dv_norm)})
    save("g11_unsupervised", images=images, labels=labels, d=d, eps=eps, **state_to_npz_dict(net), **out)


def g12_ista_and_metrics():
    g = torch.Generator().manual_seed(112)
    n, k = 6, 4
    images = torch.rand(n, 3, 16, 16, generator=g)
    net = make_tinynet(1112)
    labels = net(images).argmax(-1)
    d = U.constraint_dict(torch.randn(3, 16, 16, k, generator=g), "l2ball")
    ds = U.QuickAttackDataset(images, labels)
    out = {}
    # learn_coding_vectors
    v = quiet(R.learn_coding_vectors, ds, net, targeted=True, niter=6, lambda_l1=2.0, lambda_l2=0.05, batch_size=3,
              step_size=torch.tensor(0.05), n_atom=k, dictionary=d.clone())
    ov, _ = O.learn_coding_vectors(net, images, labels, d, True, 6, 2.0, 0.05, 3, 0.05)
    e = close(ov, v, 2e-5, "learn_coding_vectors")
    print(f"  g12 lcv err {e:.2e} nnz {int((v != 0).sum())}/{v.numel()}")
    out["lcv_v"] = v
    # adil full batch, fixed initial D passed through `dictionary`? (that freezes D) -> use the seeded-internal init
    torch.manual_seed(21)
    dd, vv, la = quiet(R.adil, ds, net, targeted=True, niter=4, lambdaCoding=0.05, l2_fool=0.05, batchsize=3,
                       step_size=0.05, n_atom=k)
    torch.manual_seed(21)
    d0 = U.constraint_dict(torch.randn(3, 16, 16, k), "l2ball")
    od, ovv, ola = O.adil_full_batch(net, images, labels, d0, True, 4, 0.05, 0.05, 3, 0.05)
    e1 = close(od, dd, 5e-5, "adil D"); e2 = close(ovv, vv, 5e-5, "adil V")
    print(f"  g12 adil err D {e1:.2e} V {e2:.2e} loss {la}")
    out.update(adil_d0=d0, adil_d=dd.detach(), adil_v=vv.detach(), adil_loss=np.array(la))
    # sadil
    with scratch_cwd():
        torch.manual_seed(22)
        sd_, sv_, _ = quiet(R.sadil, ds, net, targeted=True, nepochs=2, batchsize=3, lambdaCoding=0.05, l2_fool=0.05,
                            stepsize=0.05, n_atom=k, model_file="sadil.bin")
    torch.manual_seed(22)
    sd0 = U.constraint_dict(torch.randn(3, 16, 16, k), "l2ball")
    osd, osv, _ = O.sadil(net, images, labels, sd0, True, 2, 3, 0.05, 0.05, 0.05)
    e1 = close(osd, sd_, 5e-5, "sadil D"); e2 = close(osv, sv_, 5e-5, "sadil V")
    print(f"  g12 sadil err D {e1:.2e} V {e2:.2e}")
    out.update(sadil_d0=sd0, sadil_d=sd_.detach(), sadil_v=sv_.detach())
    # evaluation metrics + performance() with a fixed-perturbation attack
    adv = (images + 0.05 * torch.sign(torch.randn(images.shape, generator=g))).clamp(0, 1)
    out.update(metric_adv=adv, fooling=PERF.compute_fooling_rate(net, adv, images), rmse=PERF.compute_rmse(adv, images),
               mse=PERF.compute_mse(adv, images))
    assert out["fooling"] == O.compute_fooling_rate(net, adv, images)
    close(O.compute_rmse(adv, images), out["rmse"], 1e-6); close(O.compute_mse(adv, images), out["mse"], 1e-4)

    class _FixedAttack:
        device = torch.device("cpu")

        def __call__(self, x, y):
            return (x + 0.08 * torch.sign(x - 0.5)).clamp(0, 1)
    ylab = labels.clone(); ylab[0] = (ylab[0] + 1) % 10          # one sample filtered out as mis-classified
    loader = [(images[:3], ylab[:3]), (images[3:], ylab[3:])]
    perf = quiet(PERF.performance, _FixedAttack(), net, loader)
    operf = O.performance(_FixedAttack(), net, loader)
    for key in ("fooling_rate", "rmse", "mse"):
        close(operf[key], float(perf[key]), 1e-6, key)
    out.update(perf_labels=ylab, perf_fooling_rate=float(perf["fooling_rate"]), perf_rmse=float(perf["rmse"]),
               perf_mse=float(perf["mse"]))
    save("g12_ista_metrics", images=images, labels=labels, d=d, lam=0.05, lcv_lambda_l1=2.0, step=0.05, **state_to_npz_dict(net), **out)


def g13_sadil_updated():
    g = torch.Generator().manual_seed(113)
    n, k = 6, 4
    images = torch.rand(n, 3, 16, 16, generator=g)
    net = make_tinynet(1113)
    labels = net(images).argmax(-1)
    ds = U.QuickAttackDataset(images, labels)
    out = {}
    for tag, step, lam in (("a", 0.05, 0.05), ("b", 0.5, 0.3)):      # b: the line searches actually trigger
        with scratch_cwd():
            torch.manual_seed(31)
            d_, v_ = quiet(R.sadil_updated, ds, net, targeted=True, nepochs=3, batchsize=2, lambdaCoding=lam, l2_fool=0.05,
                           stepsize=step, n_atom=k, model_file="su.bin")
            saved = torch.load("su.bin")
        torch.manual_seed(31)
        d0 = U.constraint_dict(torch.randn(3, 16, 16, k), "l2ball")
        od, ov, ol = O.sadil_updated(net, images, labels, d0, True, 3, 2, lam, 0.05, step)
        e1 = close(od, d_, 5e-5, "sadil_updated D"); e2 = close(ov, v_, 5e-5, "sadil_updated V")
        close(ol, saved[4], 1e-3, "sadil_updated loss")
        print(f"  g13[{tag}] err D {e1:.2e} V {e2:.2e} loss {saved[4]}")
        out.update({f"{tag}_d0": d0, f"{tag}_d": d_.detach(), f"{tag}_v": v_.detach(), f"{tag}_loss": np.array(saved[4]),
                    f"{tag}_step": step, f"{tag}_lam": lam})
    save("g13_sadil_updated", images=images, labels=labels, l2=0.05, **state_to_npz_dict(net), **out)


def g14_uappgd():
    """UAPPGD baseline (uappgd.py:70-107): the universal perturbation and per-epoch validation fooling rates, for both
    norms / optimisers; the DataLoader's shuffled order is recorded through the dataset and handed to the oracle."""
    import importlib
    UAP = importlib.import_module("attacks.attacks_classes.uappgd")
    assert os.path.realpath(UAP.__file__).startswith(os.path.realpath(ref_import.REFERENCE_ROOT))

    class XY(torch.utils.data.Dataset):
        def __init__(self, images, labels):
            self.images, self.labels, self.log = images, labels, []

        def __len__(self):
            return len(self.images)

        def __getitem__(self, item):
            if item >= len(self.images):
                raise IndexError
            self.log.append(int(item))
            return self.images[item], self.labels[item]

    out = {}
    images, val, net = _learn_setup(114)
    with torch.no_grad():
        labels = net(images).argmax(-1)
    n, bs, steps = images.shape[0], 8, 3
    for tag, norm, eps, optim, lr, seed in (("linf_adam", "linf", 0.1, "adam", 0.01, 5), ("l2_sgd", "l2", 2.0, "sgd", 0.5, 6)):
        train_ds, val_ds = XY(images, labels), XY(val, torch.zeros(len(val), dtype=torch.long))
        with scratch_cwd():
            os.makedirs("dict_model_ImageNet_version_constrained")
            torch.manual_seed(seed)
            atk = quiet(UAP.UAPPGD, net, data_train=train_ds, data_val=val_ds, steps=steps, batch_size=bs, beta=9,
                        step_size=lr, norm=norm, eps=eps, optimizer=optim)
            attack, fr = torch.load(atk.model_name)
        log = train_ds.log[1:]                                   # the first access is the shape probe `dataset[0]` (uappgd.py:77)
        assert len(log) == steps * n
        epochs_batches = [chunk(e, bs) for e in chunk(log, n)]
        o_attack, o_fr, o_fooled = O.uappgd_learn(net, images, labels, epochs_batches, lr, norm, eps, 9.0, optim, val)
        e1 = close(o_attack, attack.detach(), 1e-6, "uappgd attack")
        close(torch.stack(o_fr), torch.stack([torch.as_tensor(f) for f in fr]), 0, "uappgd fooling")
        print(f"  g14[{tag}] oracle-vs-reference: attack {e1:.2e} |attack|max {float(attack.abs().max()):.3f} val fooling {[float(f) for f in fr]}")
        out.update({f"{tag}_batches": np.array(epochs_batches), f"{tag}_attack": attack.detach(),
                    f"{tag}_fooling": np.array([float(f) for f in fr]), f"{tag}_eps": eps, f"{tag}_lr": lr,
                    f"{tag}_train_fooled": np.array(o_fooled)})
    save("g14_uappgd", images=images, val=val, labels=labels, batch_size=bs, steps=steps, beta=9.0, **state_to_npz_dict(net), **out)


def g15_transfer():
    """Transfer evaluation (performance.py:183-232): ONE dictionary, adversaries from the reference's own ADIL
    (DDrague inference against the source net), scored on the source and two other toy targets."""
    g = torch.Generator().manual_seed(115)
    n, k, eps, steps = 14, 6, 0.12, 8
    images = torch.rand(n, 3, 16, 16, generator=g)
    d = -1 + 2 * torch.rand(3, 16, 16, k, generator=g)
    src = make_tinynet(1115)
    targets = {"src": src, "t1": make_tinynet(2115), "t2": make_tinynet(3115)}
    labels = src(images).argmax(-1)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(images, labels), batch_size=5, shuffle=False)
    with scratch_cwd():
        torch.save([d, torch.zeros(1), [], [], torch.tensor(0.)], "trained_dicts/ImageNet_g15.bin")
        atk = quiet(A.ADIL, src, eps=eps, n_atoms=k, attack="supervised", model_name="g15", loss="logits",
                    steps_inference=steps, kappa=50)
        perf = quiet(PERF.get_transfer_performance, {"adil": [atk], "none": []}, targets, loader)
    assert set(perf) == {"adil", "none"} and all(np.isnan(v["mse"]) for v in perf["none"].values())
    operf = O.transfer_performance(lambda x, y: O.forward_supervised_ddrague(src, x, d, eps, steps, "logits", False, 50.0),
                                   targets, [(images[i:i + 5], labels[i:i + 5]) for i in range(0, n, 5)], n)
    out = {}
    for name in targets:
        for key in ("fooling_rate", "rmse", "mse"):
            e = close(operf[name][key], float(perf["adil"][name][key]), 1e-6 if key != "mse" else 1e-4, f"{name}.{key}")
            out[f"{name}_{key}"] = float(perf["adil"][name][key])
        print(f"  g15[{name}] fooling {out[name + '_fooling_rate']:.4f} rmse {out[name + '_rmse']:.5f} mse {out[name + '_mse']:.4f}")
    nets = {}
    for name, net in targets.items():
        nets.update(state_to_npz_dict(net, prefix=f"{name}."))
    save("g15_transfer", images=images, labels=labels, d=d, eps=eps, steps=steps, kappa=50.0, batch_size=5, **nets, **out)


def g16_constraint_l1():
    """constraint_dict's third branch (utils.py:55-56): every (channel, atom) row of H*W pixels onto the l1 ball of
    radius 1.  No caller upstream; pinned all the same."""
    g = torch.Generator().manual_seed(116)
    d = torch.randn(3, 12, 10, 5, generator=g) * 0.05          # rows of 120 pixels, l1 norm ~ 4.8: outside the ball
    d[..., 0] *= 0.05                                           # atom 0: every row inside the ball (untouched)
    d[1, :, :, 2] = 0.0                                         # a zero row
    d[2, 0, :4, 3] = 0.7                                        # ties among the largest magnitudes
    ref = U.constraint_dict(d.clone(), "l1ball")
    close(O.constraint_dict(d, "l1ball"), ref, 1e-7)
    save("g16_constraint_l1", d=d, l1ball=ref)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (g1_l1ball, g2_constraints, g3_softshrink, g4_synth_grad, g5_floss, g6_adamw_steps, g7_learn_a,
               g8_learn_b, g9_ddrague, g10_adamw_inference, g11_unsupervised, g12_ista_and_metrics, g13_sadil_updated,
               g14_uappgd, g15_transfer, g16_constraint_l1):
        if only and fn.__name__.split("_")[0] not in only:
            continue
        print(fn.__name__)
        fn()
