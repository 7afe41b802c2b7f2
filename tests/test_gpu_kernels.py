"""GPU parity: every C-ABI kernel (through ctypes, dl_attack_on_imagenet_amd.ops) against the oracle on the same
seeded inputs and against the golden vectors.  fp32 paths: tolerances are absolute fp32 rounding bounds written
per test; bf16 stream paths: compared with an fp32 oracle evaluated on the bf16-rounded inputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, t
from oracle import adil_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from dl_attack_on_imagenet_amd import ops as _ops
    return _ops


def close(a, b, tol, what=""):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float((a - b).abs().max()) if a.numel() else 0.0
    assert err <= tol, (what, err, tol)


def test_library_loaded_and_abi():
    from dl_attack_on_imagenet_amd import _lib
    lib = _lib.load()
    assert lib.adil_abi_version() == _lib.ABI_VERSION
    assert lib.adil_max_atoms() >= 128


# ----------------------------------------------------------------------------- projections / prox
def test_l1ball_golden():
    z = load_golden("g1_l1ball")
    for tag in "abcde":
        x = t(z[f"x_{tag}"], DEV).contiguous()
        y = ops().l1ball_project_(x.clone(), float(z["eps"]))
        close(y, z[f"y_{tag}"], 2e-7, f"l1ball {tag}")


@pytest.mark.parametrize("n,k", [(1, 1), (3, 2), (257, 10), (1000, 50), (513, 64), (300, 65), (200, 100), (64, 128)])
def test_l1ball_random(n, k):
    g = torch.Generator().manual_seed(n * 131 + k)
    x = torch.randn(n, k, generator=g) * 0.03
    x[::7] *= 0.01                        # rows inside the ball
    x[1::11] = 0.0                        # zero rows
    eps = 8 / 255
    y = ops().l1ball_project_(x.to(DEV), eps)
    ref = O.project_onto_l1_ball(x, eps)
    close(y, ref, 3e-7)
    l1 = y.abs().sum(1).cpu()
    assert float(l1.max()) <= eps * (1 + 1e-5)
    # idempotence (size-independent property)
    close(ops().l1ball_project_(y.clone(), eps), y, 3e-7)


def test_l2ball_and_constraints_golden():
    z = load_golden("g2_constraints")
    eps = float(z["eps"])
    close(ops().l2ball_project_(t(z["v"], DEV).clone(), eps), z["pv_l2"], 1e-7)
    close(ops().l1ball_project_(t(z["v"], DEV).clone(), eps), z["pv_linf"], 2e-7)
    close(ops().atom_l2_project_(t(z["d"], DEV).clone(), sphere=False), z["l2ball"], 2e-7)
    close(ops().atom_l2_project_(t(z["d"], DEV).clone(), sphere=True), z["l2sphere"], 2e-7)


def test_softshrink_golden_and_ista():
    z = load_golden("g3_softshrink")
    close(ops().ista_step_(t(z["x"], DEV).clone(), None, 0.0, float(z["lam"])), z["y"], 0)
    g = torch.Generator().manual_seed(3)
    v, gr = torch.randn(37, 50, generator=g) * 0.1, torch.randn(37, 50, generator=g)
    out = ops().ista_step_(v.to(DEV), gr.to(DEV), 0.05, 0.02)
    close(out, O.softshrink(v - 0.05 * gr, 0.02), 1e-7)


# ----------------------------------------------------------------------------- contractions
def _oracle_synth(x, d, vrows, delta_clamp=None, pixel_clamp=False):
    dv = (vrows @ O.dict_matrix(d).t()).reshape(x.shape)
    if delta_clamp is not None:
        dv = dv.clamp(-delta_clamp, delta_clamp)
    out = x + dv
    return out.clamp(0, 1) if pixel_clamp else out


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_synth_grad_golden(tag):
    z = load_golden("g4_synth_grad")
    d, v, x, idx, g = (t(z[f"{n}_{tag}"], DEV) for n in ("d", "v", "x", "index", "g"))
    b = x.shape[0]
    vp = ops().pack_codes(v, idx, b)
    close(ops().synth(x, d, vp, b), z[f"y_{tag}"], 2e-6)
    gd, gvb = ops().grad(g, d, vp, b)
    close(gd, z[f"grad_d_{tag}"], 2e-5)
    gv = torch.zeros_like(v)
    gv[idx] = gvb
    close(gv, z[f"grad_v_{tag}"], 2e-4)
    # the autograd wrapper must give the same thing (this is what Attack_dict_model.forward uses)
    dd, vv = d.clone().requires_grad_(True), v.clone().requires_grad_(True)
    ops().dict_synth(x, dd, vv, idx).backward(g)
    close(dd.grad, z[f"grad_d_{tag}"], 2e-5)
    close(vv.grad, z[f"grad_v_{tag}"], 2e-4)


SHAPES = [  # (B, C, H, W, K): ragged batch / pixel tails, K = 1 .. 128
    (1, 3, 4, 4, 1), (5, 3, 7, 9, 3), (33, 3, 16, 16, 10), (64, 3, 20, 12, 50), (31, 1, 13, 17, 64),
    (40, 3, 8, 8, 100), (17, 3, 10, 10, 128), (96, 3, 32, 32, 16),
    (600, 3, 8, 8, 50),        # more rows than one grad_v launch holds (row chunks of 256 fp32 / 512 bf16)
]


@pytest.mark.parametrize("shape", SHAPES)
def test_synth_random_f32(shape):
    b, c, h, w, k = shape
    g = torch.Generator().manual_seed(sum(shape))
    d = -1 + 2 * torch.rand(c, h, w, k, generator=g)
    v = torch.randn(b + 3, k, generator=g) * 0.02
    idx = torch.randperm(b + 3, generator=g)[:b]
    x = torch.rand(b, c, h, w, generator=g)
    vp = ops().pack_codes(v.to(DEV), idx.to(DEV), b)
    close(vp[:b, :k], v[idx], 0)
    assert float(vp[b:].abs().sum()) == 0 and float(vp[:, k:].abs().sum()) == 0
    out = ops().synth(x.to(DEV), d.to(DEV), vp, b)
    close(out, _oracle_synth(x, d, v[idx]), 1e-5 * max(1, k ** 0.5))
    out = ops().synth(x.to(DEV), d.to(DEV), vp, b, delta_clamp=0.01, pixel_clamp=True)
    close(out, _oracle_synth(x, d, v[idx], 0.01, True), 1e-5 * max(1, k ** 0.5))
    only = ops().synth(None, d.to(DEV), vp, b, out_shape=x.shape, out_dtype=torch.float32)
    close(only, _oracle_synth(torch.zeros_like(x), d, v[idx]), 1e-5 * max(1, k ** 0.5))


@pytest.mark.parametrize("shape", SHAPES)
def test_grad_random_f32(shape):
    b, c, h, w, k = shape
    gen = torch.Generator().manual_seed(sum(shape) + 1)
    d = -1 + 2 * torch.rand(c, h, w, k, generator=gen)
    v = torch.randn(b, k, generator=gen) * 0.02
    g = torch.randn(b, c, h, w, generator=gen)
    vp = ops().pack_codes(v.to(DEV), None, b)
    gd, gvb = ops().grad(g.to(DEV), d.to(DEV), vp, b)
    rd, rv = O.grad_dv(g.double(), d.double(), v.double())
    p = c * h * w
    close(gd, rd, 2e-6 * b ** 0.5 * 4)                 # |g|~1, |v|~0.02, sum over B terms
    close(gvb, rv, 3e-6 * p ** 0.5 * 4)                # |g|~1, |d|<=1, sum over P terms
    # want_d / want_v individually, and accumulation into grad_d
    gd2, none = ops().grad(g.to(DEV), d.to(DEV), vp, b, want_v=False)
    assert none is None
    close(gd2, gd, 2e-6 * b ** 0.5 * 4)                # separate kernel, different (fixed) summation order than the fused pass
    acc = gd.clone()
    ops().grad(g.to(DEV), d.to(DEV), vp, b, want_v=False, grad_d=acc, accumulate_d=True)
    close(acc, 2 * gd, 2e-6 * b ** 0.5 * 4)
    none, gv2 = ops().grad(g.to(DEV), d.to(DEV), None, b, want_d=False)
    close(gv2, gvb, 3e-6 * p ** 0.5 * 4)


@pytest.mark.parametrize("shape", [(33, 3, 16, 16, 10), (64, 3, 20, 12, 50), (40, 3, 8, 8, 100)])
def test_synth_grad_bf16_streams(shape):
    """bf16 image streams: x / g arrive in bf16, D and V are rounded to bf16 as MFMA operands, accumulation is
    fp32.  Compared with the fp64 oracle evaluated on the bf16-rounded operands."""
    b, c, h, w, k = shape
    gen = torch.Generator().manual_seed(sum(shape) + 2)
    d = -1 + 2 * torch.rand(c, h, w, k, generator=gen)
    v = torch.randn(b, k, generator=gen) * 0.02
    x = torch.rand(b, c, h, w, generator=gen).bfloat16()
    g = torch.randn(b, c, h, w, generator=gen).bfloat16()
    vp = ops().pack_codes(v.to(DEV), None, b)
    out = ops().synth(x.to(DEV), d.to(DEV), vp, b)
    assert out.dtype == torch.bfloat16
    dq, vq = d.bfloat16().float(), v.bfloat16().float()
    ref = _oracle_synth(x.float(), dq, vq)
    close(out.float(), ref.bfloat16().float(), 2 ** -7)     # one bf16 ulp at magnitude <= 2
    gd, gvb = ops().grad(g.to(DEV), d.to(DEV), vp, b)
    rd, rv = O.grad_dv(g.double(), dq.double(), vq.double())
    close(gd, rd, 1e-5 * b ** 0.5 * 4)
    close(gvb, rv, 1e-5 * (c * h * w) ** 0.5 * 4)
    # and the bf16 operand rounding itself stays within bf16 resolution of the fp32 result
    rd32, rv32 = O.grad_dv(g.double(), d.double(), v.double())
    close(gd, rd32, 2 ** -7 * 0.02 * b ** 0.5 * 4)
    close(gvb, rv32, 2 ** -7 * (c * h * w) ** 0.5 * 4)


def test_linearity_full_size():
    """Size-independent property at the BASELINE image size (P = 150528): the adjoint identity
    <g, synth(0,D,V)> == <grad_d, D-direction> ... here: <g, D v> == <grad_vb, v> == <grad_d, 1 (x) v> checks."""
    b, k = 48, 50
    gen = torch.Generator().manual_seed(9)
    d = (-1 + 2 * torch.rand(3, 224, 224, k, generator=gen)).to(DEV)
    v = (torch.randn(b, k, generator=gen) * 0.02).to(DEV)
    g = torch.randn(b, 3, 224, 224, generator=gen).to(DEV)
    vp = ops().pack_codes(v, None, b)
    dv = ops().synth(None, d, vp, b, out_shape=g.shape, out_dtype=torch.float32)
    gd, gvb = ops().grad(g, d, vp, b)
    lhs = float((g.double() * dv.double()).sum())
    assert abs(lhs - float((gvb.double() * v.double()).sum())) <= 1e-4 * abs(lhs) + 1e-3
    assert abs(lhs - float((gd.double() * d.double()).sum())) <= 1e-4 * abs(lhs) + 1e-3
    # and against a torch fp64 matmul on the device
    ref_dv = (v.double() @ d.reshape(-1, k).double().t()).reshape(g.shape)
    close(dv, ref_dv, 2e-5)
    close(gvb, g.reshape(b, -1).double() @ d.reshape(-1, k).double(), 2e-2)
    close(gd.reshape(-1, k), g.reshape(b, -1).double().t() @ v.double(), 2e-5)


@pytest.mark.parametrize("b,k,dt", [(544, 50, torch.bfloat16), (192, 100, torch.bfloat16), (96, 128, torch.float32),
                                    (130, 10, torch.bfloat16), (512, 100, torch.bfloat16), (300, 100, torch.float32)])
def test_full_size_configs(b, k, dt):
    """BASELINE image size (P = 150528) at the atom counts of the other configs (K = 10 / 50 / 100 / 128), ragged
    batch sizes, against fp64 matmuls on the device (operands rounded as the kernels round them)."""
    gen = torch.Generator().manual_seed(b + k)
    d = (-1 + 2 * torch.rand(3, 224, 224, k, generator=gen)).to(DEV)
    v = (torch.randn(b, k, generator=gen) * 0.02).to(DEV)
    x = torch.rand(b, 3, 224, 224, generator=gen).to(DEV).to(dt)
    g = torch.randn(b, 3, 224, 224, generator=gen).to(DEV).to(dt)
    vp = ops().pack_codes(v, None, b)
    dq, vq = (d.bfloat16().double(), v.bfloat16().double()) if dt == torch.bfloat16 else (d.double(), v.double())
    dm = dq.reshape(-1, k)
    out = ops().synth(x, d, vp, b)
    ref = (x.double().reshape(b, -1) + vq @ dm.t()).reshape(x.shape)
    close(out.double(), ref.to(dt).double(), 2 ** -7 if dt == torch.bfloat16 else 2e-5)
    gd, gvb = ops().grad(g, d, vp, b)
    g2 = g.double().reshape(b, -1)
    close(gd.reshape(-1, k), g2.t() @ vq, 5e-5 * b ** 0.5)
    close(gvb, g2 @ dm, 3e-2)
    # run-to-run bitwise reproducibility (no float atomics anywhere)
    gd2, gvb2 = ops().grad(g, d, vp, b)
    assert torch.equal(gd, gd2) and torch.equal(gvb, gvb2)


@pytest.mark.parametrize("b,k,dt", [(70, 50, torch.bfloat16), (70, 50, torch.float32), (96, 128, torch.float32),
                                    (33, 128, torch.bfloat16), (1, 7, torch.float32)])
def test_synth_guard_rows_and_store_hazard(b, k, dt):
    """Full image size, ragged batch: (1) rows >= B of the output allocation stay untouched (the aligned interior
    clips rows through the buffer descriptor's num_records), (2) repeated launches agree bit for bit with each other
    and with fp64 — the fp32 path once lost rows to a >64-bit store-data hazard that only showed under load."""
    gen = torch.Generator().manual_seed(7 * b + k)
    P = 150528
    d = (-1 + 2 * torch.rand(1, 1, P, k, generator=gen)).to(DEV)
    v = (torch.randn(b, k, generator=gen) * 0.02).to(DEV)
    x = torch.rand(b, 1, 1, P, generator=gen).to(DEV).to(dt)
    buf = torch.full((b + 40, 1, 1, P), 7.0, device=DEV, dtype=dt)
    vp = ops().pack_codes(v, None, b)
    dq, vq = (d.bfloat16().double(), v.bfloat16().double()) if dt == torch.bfloat16 else (d.double(), v.double())
    ref = (x.double().reshape(b, -1) + vq @ dq.reshape(-1, k).t()).reshape(x.shape)
    first = None
    for _ in range(4):
        out = ops().synth(x, d, vp, b, out=buf[:b])
        assert bool((buf[b:] == 7.0).all()), "rows beyond the batch were written"
        close(out.double(), ref.to(dt).double(), 2 ** -7 if dt == torch.bfloat16 else 2e-5)
        first = out.clone() if first is None else first
        assert torch.equal(first, out)


# ----------------------------------------------------------------------------- optimiser
def test_adamw_steps_golden():
    """T steps of {AdamW(d,v); l1-ball(v); clamp(d)} against the reference's torch.optim.AdamW trajectory (G6)."""
    z = load_golden("g6_adamw_steps")
    eps, lr = float(z["eps"]), float(z["lr"])
    d, v = t(z["d0"], DEV).clone(), t(z["v0"], DEV).clone()
    md, sd, mv, sv = torch.zeros_like(d), torch.zeros_like(d), torch.zeros_like(v), torch.zeros_like(v)
    sched_d, sched_v = ops().AdamWSchedule(lr), ops().AdamWSchedule(lr)
    pos = torch.empty(v.shape[0], dtype=torch.int32, device=DEV)
    for step in range(z["g"].shape[0]):
        idx = t(z["index"][step], DEV)
        b = idx.numel()
        vp = ops().pack_codes(v, idx, b)
        gd, gvb = ops().grad(t(z["g"][step], DEV), d, vp, b)
        pos.fill_(-1)
        pos[idx] = torch.arange(b, dtype=torch.int32, device=DEV)
        ops().adamw_clamp_(d, gd, md, sd, sched_d.next(), -1.0, 1.0)
        ops().adamw_l1ball_(v, gvb, pos, mv, sv, sched_v.next(), eps)
        close(d, z["d_hist"][step], 5e-6, f"d step {step}")
        close(v, z["v_hist"][step], 5e-6, f"v step {step}")
    close(md, z["m_d"], 1e-5); close(sd, z["s_d"], 1e-5)
    close(mv, z["m_v"], 1e-5); close(sv, z["s_v"], 1e-5)


@pytest.mark.parametrize("n", [1, 3, 4, 1023, 150528 * 3 + 2])
@pytest.mark.parametrize("gdtype", [torch.float32, torch.bfloat16])
def test_adamw_clamp_vs_torch(n, gdtype):
    gen = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=gen)
    gs = [torch.randn(n, generator=gen).to(gdtype) for _ in range(3)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=0.01)
    p = p0.clone().to(DEV)
    # device buffers from a 16-byte aligned allocation
    m, s = torch.zeros_like(p), torch.zeros_like(p)
    sched = ops().AdamWSchedule(0.01)
    delta = torch.zeros(1, device=DEV)
    for g in gs:
        prev = ref.data.clone()
        ref.grad = g.float()
        opt.step()
        ref.data.clamp_(-0.5, 0.5)
        delta.zero_()
        ops().adamw_clamp_(p, g.to(DEV), m, s, sched.next(), -0.5, 0.5, max_abs_delta=delta)
        close(p, ref.data, 2e-6)
        assert abs(float(delta) - float((ref.data - prev).abs().max())) <= 2e-6


def test_gram_pinv():
    gen = torch.Generator().manual_seed(5)
    for (c, h, w, k) in [(3, 16, 16, 6), (3, 30, 21, 50), (3, 9, 9, 100), (3, 12, 12, 127), (3, 16, 16, 128)]:
        d = (-1 + 2 * torch.rand(c, h, w, k, generator=gen))
        dtd, dtd_inv, d_drg = O.gram_pinv(d.double())
        gm = ops().gram(d.to(DEV))
        close(gm, dtd, 1e-5 * (c * h * w) ** 0.5)
        out = ops().dict_rightmul(d.to(DEV), dtd_inv.float().to(DEV))      # K = 127/128: > 64 KB of dynamic LDS
        close(out.reshape(-1, k), d_drg.reshape(k, -1).t(), 1e-5)
        # the K x K inverse on the device (fp64 Gauss-Jordan in LDS) against the fp64 host inverse
        inv = ops().spd_inverse(dtd.float().to(DEV))
        ref = torch.linalg.inv(dtd.float().double())
        close(inv, ref, 2e-6 * float(ref.abs().max()))
        assert float((inv.double().cpu() @ dtd.float().double() - torch.eye(k, dtype=torch.float64)).abs().max()) < 1e-4


def test_pseudo_inverse_k128_ddrague():
    """ADIL_MAX_ATOMS = 128 through the default inference path (PseudoInverse -> solve_ddrague)."""
    from dl_attack_on_imagenet_amd import engine
    from tinynet import make_tinynet
    gen = torch.Generator().manual_seed(12)
    d = -1 + 2 * torch.rand(3, 16, 16, 128, generator=gen)
    x = torch.rand(6, 3, 16, 16, generator=gen)
    net = make_tinynet(2)
    adv_o = O.forward_supervised_ddrague(net, x, d, 0.1, 4, "ce")
    adv = engine.solve_ddrague(net.to(DEV), x.to(DEV), d.to(DEV), 0.1, 4, "ce")
    close(adv, adv_o, 5e-4)


@pytest.mark.parametrize("src_dt,dst_dt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                           (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
def test_gather_images(src_dt, dst_dt):
    gen = torch.Generator().manual_seed(8)
    src = torch.rand(37, 3, 20, 12, generator=gen).to(src_dt)
    index = torch.tensor([5, 0, 36, 5, 17, 2, 2, 30, 11])                # repeats allowed, any order
    got = ops().gather_images(src.to(DEV), index.to(DEV), dtype=dst_dt)
    assert got.dtype == dst_dt and got.shape == (9, 3, 20, 12)
    assert torch.equal(got.cpu(), src[index].to(dst_dt))                  # bit-exact (RNE conversion, like .to())
    out = torch.empty(37, 3, 20, 12, dtype=dst_dt, device=DEV)
    ops().gather_images(src.to(DEV), None, out=out)                       # identity gather = pure cast
    assert torch.equal(out.cpu(), src.to(dst_dt))
    big = torch.rand(70, 3, 224, 224, generator=gen)                      # BASELINE image size, more rows than one wave of blocks
    idx = torch.randperm(70, generator=gen)[:64]
    assert torch.equal(ops().gather_images(big.to(DEV), idx.to(DEV), dtype=dst_dt).cpu(), big[idx].to(dst_dt))


def test_pack_codes_slot_table_round_trip():
    """adil_pack_codes writes pos[index[b]] = b; adil_adamw_l1ball consumes it and hands it back all -1 — equal to the
    fill + scatter it replaces, step after step."""
    gen = torch.Generator().manual_seed(9)
    n, k, eps = 50, 10, 0.3
    v = O.project_onto_l1_ball(torch.rand(n, k, generator=gen), eps)
    va, vb = v.clone().to(DEV), v.clone().to(DEV)
    ma, sa, mb, sb = (torch.zeros(n, k, device=DEV) for _ in range(4))
    pos = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    sch_a, sch_b = ops().AdamWSchedule(0.01), ops().AdamWSchedule(0.01)
    for step in range(4):
        idx = torch.randperm(n, generator=gen)[:13].to(DEV)
        gvb = torch.randn(13, k, generator=gen).to(DEV)
        vp = ops().pack_codes(va, idx, 13, pos=pos)
        ref = torch.full((n,), -1, dtype=torch.int32, device=DEV)
        ref[idx] = torch.arange(13, dtype=torch.int32, device=DEV)
        assert torch.equal(pos, ref)
        assert torch.equal(vp[:13, :k], va[idx])
        ops().adamw_l1ball_(va, gvb, pos, ma, sa, sch_a.next(), eps, reset_pos=True)
        assert int((pos != -1).sum()) == 0
        ops().adamw_l1ball_(vb, gvb, ref, mb, sb, sch_b.next(), eps)
        assert torch.equal(va, vb)
    ops().adamw_l1ball_(va, None, pos, ma, sa, sch_a.next(), eps, reset_pos=True)     # no row of the batch is ours
    gv0 = torch.zeros(n, k, device=DEV)
    ops().adamw_l1ball_(vb, gv0, None, mb, sb, sch_b.next(), eps)
    assert torch.equal(va, vb)


@pytest.mark.parametrize("shape,dt", [((9, 3, 31, 17), torch.float32), ((9, 3, 31, 17), torch.bfloat16),
                                      ((16, 3, 224, 224), torch.float32), ((16, 3, 224, 224), torch.bfloat16),
                                      ((5, 3, 8, 6), torch.bfloat16)])
def test_image_metrics(shape, dt):
    """Per-image sum (adv - x)^2 and sum x^2 (performance.py:249-266) against fp64 sums of the same (stream-rounded) values:
    the 16-byte-load path (full-size images) and the element-wise path (odd sizes / rows that do not start on 16 bytes)."""
    gen = torch.Generator().manual_seed(6)
    x = torch.rand(*shape, generator=gen).to(dt)
    adv = (x.float() + 0.05 * torch.randn(shape, generator=gen)).clamp(0, 1).to(dt)
    se, sn = ops().image_metrics(adv.to(DEV), x.to(DEV))
    ref_e = ((adv.double() - x.double()) ** 2).sum(dim=[1, 2, 3])
    ref_n = (x.double() ** 2).sum(dim=[1, 2, 3])
    close(se, ref_e, 2e-6 * float(ref_e.max()) + 1e-6)
    close(sn, ref_n, 2e-6 * float(ref_n.max()) + 1e-6)
    se2, sn2 = ops().image_metrics(adv.to(DEV), x.to(DEV))                        # bitwise reproducible
    assert torch.equal(se, se2) and torch.equal(sn, sn2)


def test_errors_are_loud():
    o = ops()
    with pytest.raises(RuntimeError):
        o.l1ball_project_(torch.zeros(4, 4), 0.1)                      # CPU tensor: no fallback
    with pytest.raises(Exception):
        o.l1ball_project_(torch.zeros(4, 300, device=DEV), 0.1)        # K > max atoms -> ADIL_EINVAL
    with pytest.raises(TypeError):
        o.synth(torch.zeros(2, 3, 4, 4, device=DEV, dtype=torch.float16), torch.zeros(3, 4, 4, 2, device=DEV),
                torch.zeros(32, 16, device=DEV), 2)
    # ABI 6 arguments, straight through the C ABI: inconsistent slab descriptions are refused (ADIL_EINVAL = -1), never run
    from ctypes import c_void_p
    from dl_attack_on_imagenet_amd import _lib
    lib = _lib.load()
    v = torch.zeros(8, 4, device=DEV)
    vp = torch.zeros(32, 16, device=DEV)
    pos = torch.full((8,), -1, dtype=torch.int32, device=DEV)
    slabs = torch.zeros(3 * 32 * 4, device=DEV)
    P, st = (lambda t: c_void_p(0 if t is None else t.data_ptr())), c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.adil_pack_codes(P(None), P(None), 8, 4, P(vp), P(None), P(None), 0, P(None), 3, 32, st) == -1    # slabs announced, none given
    assert lib.adil_pack_codes(P(None), P(None), 8, 4, P(vp), P(pos), P(None), 0, P(slabs), 3, 32, st) == -1   # slab source has no slot table
    assert lib.adil_pack_codes(P(None), P(None), 8, 4, P(vp), P(None), P(None), 0, P(slabs), 3, 4, st) == -1    # fewer slab rows than batch rows
    assert lib.adil_pack_codes(P(None), P(None), 8, 4, P(vp), P(None), P(None), 0, P(None), 0, 0, st) == -1     # neither v nor slabs
    m, s_ = torch.zeros_like(v), torch.zeros_like(v)
    args = (8, 4, 0.99, 0.9, 0.999, 1e-8, 0.1, 0.03, 0.5, P(None), P(None), 0.0, P(None), P(None))
    assert lib.adil_adamw_l1ball(P(v), P(None), P(None), 0, P(m), P(s_), *args, P(None), 3, 32, st) == -1        # nslabs without slabs
    assert lib.adil_adamw_l1ball(P(v), P(None), P(None), 0, P(m), P(s_), *args, P(slabs), 3, 4, st) == -1        # no slot table: one slab row per code row
    assert lib.adil_adamw_l1ball(P(v), P(None), P(None), 0, P(m), P(s_), *args, P(None), 0, 0, st) == -1         # no gradient source at all
    assert lib.adil_atom_l1ball_project(P(None), 3, 16, 4, 1.0, st) == -1
    assert lib.adil_grad_code_rows(0) == 0 and lib.adil_grad_code_rows(50) == 64 and lib.adil_grad_code_rows(100) == 128
    torch.cuda.synchronize()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [False, True])
def test_affine_act_epilogue(dt, cl):
    """Fused eval-BatchNorm + residual + ReLU epilogue (frozen-classifier helper) vs plain torch ops, fwd and bwd."""
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(4, 16, 6, 10, generator=gen).to(DEV).to(dt)
    res = torch.randn(4, 16, 6, 10, generator=gen).to(DEV).to(dt)
    if cl:
        x, res = x.contiguous(memory_format=torch.channels_last), res.contiguous(memory_format=torch.channels_last)
    scale = (torch.rand(16, generator=gen) + 0.5).to(DEV)
    shift = torch.randn(16, generator=gen).to(DEV)
    for use_res, relu in ((True, True), (False, True), (False, False)):
        xa = x.clone().requires_grad_(True)
        ra = res.clone().requires_grad_(True)
        y = ops().affine_act(xa, scale, shift, res=ra if use_res else None, relu=relu)
        xb = x.clone().float().requires_grad_(True)
        rb = res.clone().float().requires_grad_(True)
        ref = xb * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if use_res:
            ref = ref + rb
        if relu:
            ref = torch.relu(ref)
        tol = 1e-5 if dt == torch.float32 else 3e-2
        close(y.float(), ref, tol)
        gout = torch.randn(y.shape, generator=gen).to(DEV)
        y.backward(gout.to(dt).contiguous(memory_format=torch.channels_last) if cl else gout.to(dt))
        ref.backward(gout.to(dt).float())
        mask = (y.float() == 0) & relu                      # ReLU boundary elements may round to either side in bf16
        close(torch.where(mask, torch.zeros_like(xb.grad), xa.grad.float() - xb.grad), torch.zeros_like(xb.grad), tol)
        if use_res:
            close(torch.where(mask, torch.zeros_like(rb.grad), ra.grad.float() - rb.grad), torch.zeros_like(rb.grad), tol)


def test_fused_resnet_matches_plain():
    from dl_attack_on_imagenet_amd import zoo
    plain = zoo.build_classifier("resnet18", num_classes=10, seed=4, device=DEV)
    g = torch.Generator().manual_seed(1)
    for m in plain.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    fused = torch.nn.Sequential(plain[0], zoo.FusedResNet(plain[1])).to(DEV)
    for m in fused.modules():
        if isinstance(m, zoo._ConvAffine):
            m.scale, m.shift = m.scale.float().to(DEV), m.shift.float().to(DEV)
    x = torch.rand(4, 3, 64, 64, generator=g).to(DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = plain(xa), fused(xb)
    close(yb, ya, 1e-4 * float(ya.abs().max()))
    ga, = torch.autograd.grad(ya.sum(), xa)
    gb, = torch.autograd.grad(yb.sum(), xb)
    assert float((ga - gb).norm() / ga.norm()) < 1e-3


@pytest.mark.parametrize("shape", [(5, 3, 7, 9, 3), (33, 3, 16, 16, 10), (64, 3, 32, 32, 50), (40, 3, 8, 8, 100)])
def test_zstep_fused_vs_unfused(shape):
    """K8: gz = gv D_dagger formed inside the kernel + AdamW(z) + clamp + max|dz| against the explicit sequence."""
    b, c, h, w, k = shape
    gen = torch.Generator().manual_seed(sum(shape) + 7)
    dpt = torch.randn(c, h, w, k, generator=gen) * 0.1
    gv = torch.randn(b, k, generator=gen)
    z0 = torch.randn(b, c, h, w, generator=gen) * 0.01
    eps = 0.02
    zr, st = z0.clone(), O.AdamWState(z0, 1e-2)
    zf = z0.clone().to(DEV)
    mf, sf = torch.zeros_like(zf), torch.zeros_like(zf)
    sched = ops().AdamWSchedule(1e-2)
    delta = torch.zeros(1, device=DEV)
    for it in range(3):
        gz = (gv.double() @ dpt.reshape(-1, k).double().t()).float().reshape(z0.shape) * (0.5 ** it)
        prev = zr.clone()
        st.step(zr, gz)
        zr.clamp_(-eps, eps)
        delta.zero_()
        ops().zstep_(zf, mf, sf, dpt.to(DEV), ops().pack_codes((gv * 0.5 ** it).to(DEV), None, b), b, sched.next(), -eps, eps,
                     max_abs_delta=delta)
        # the first AdamW steps are ~ lr*g/(|g|+1e-8): elements with |gz| ~ 1e-7 amplify the fp32-vs-fp64 rounding of
        # gz, so the bound is 5e-5 on the max and 1e-7 on the mean
        close(zf, zr, 5e-5, f"z it {it}")
        assert float((zf.cpu() - zr).abs().mean()) <= 1e-7
        assert abs(float(delta) - float((zr - prev).abs().max())) <= 5e-5
    close(mf, st.m, 1e-5); close(sf, st.v, 1e-5)


@pytest.mark.parametrize("shape", [(512, 3, 32, 32, 50), (500, 3, 16, 16, 64), (33, 3, 16, 16, 10), (300, 3, 16, 24, 100),
                                   (600, 3, 16, 16, 50), (70, 3, 16, 8, 112), (16, 3, 224, 224, 50), (512, 3, 224, 224, 100)])
def test_zstep_codes_fused_with_next_codes(shape):
    """adil_zstep_codes (ABI 7): the z-step that also contracts the updated z with D_dagger^T.  (i) z, m, s and max|dz| are
    the bits of adil_zstep on the same inputs (same arithmetic, other workgroup shape); (ii) the codes pack_codes sums from
    its slabs equal z_new D_dagger^T of an fp64 matmul to fp32-grade tolerance (the fixed-order slab sums make them
    bitwise reproducible); (iii) a launch the device-side stop test skips leaves z AND the slabs untouched.  One / two
    blocks per wave, ragged rows, a second row range of workgroups (600 rows), all three atom tilings, one and several
    slices per workgroup (224 x 224: 1176 slices), and the inference bench's own size at the reference's atom count
    (512 images, 100 atoms: two row ranges of workgroups, 236 slabs)."""
    b, c, h, w, k = shape
    o = ops()
    p = c * h * w
    gen = torch.Generator().manual_seed(sum(shape) + 11)
    dpt = (torch.randn(c, h, w, k, generator=gen) * 0.1).to(DEV)
    gv = torch.randn(b, k, generator=gen).to(DEV)
    z0 = (torch.randn(b, c, h, w, generator=gen) * 0.01).to(DEV)
    eps = 0.02
    nbytes = o.zstep_codes_slab_bytes(b, p, k)
    assert nbytes > 0
    slabs = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    za, zb = z0.clone(), z0.clone()
    ma, sa, mb, sb = (torch.zeros_like(z0) for _ in range(4))
    sched_a, sched_b = o.AdamWSchedule(1e-2), o.AdamWSchedule(1e-2)
    stop_a, stop_b = o.StopTest(DEV, 1e-6), o.StopTest(DEV, 1e-6)
    for it in range(3):
        gvp = o.pack_codes(gv * 0.5 ** it, None, b)
        o.zstep_(za, ma, sa, dpt, gvp, b, sched_a.next(), -eps, eps, stop=stop_a)
        vnext = o.zstep_codes_(zb, mb, sb, dpt, gvp, b, sched_b.next(), -eps, eps, slabs, stop=stop_b)
        assert torch.equal(za, zb) and torch.equal(ma, mb) and torch.equal(sa, sb), it
        assert torch.equal(stop_a.slots, stop_b.slots), it
        assert isinstance(vnext, o.SlabGrad) and vnext.shape == (b, k) and vnext.nslabs >= 1
        codes = o.pack_codes(vnext, None, b)
        ref = zb.double().reshape(b, p) @ dpt.double().reshape(p, k)
        close(codes[:b, :k], ref, 2e-6 * float(ref.abs().max()), f"codes it {it}")       # ~30 fp32 ulps of the largest entry
        assert not bool(codes[b:].any()) and not bool(codes[:, k:].any())
        assert torch.equal(codes, o.pack_codes(vnext, None, b))
    # the contraction launch it replaces (another summation order, the same fp32-grade maths)
    _, dense = o.grad(zb, dpt, None, b, want_d=False)
    close(codes[:b, :k], dense, 4e-6 * float(ref.abs().max()), "vs adil_grad")
    # a skipped launch: previous slot below the threshold -> nothing moves, the slabs keep the converged z's codes
    stop_b.slots[(stop_b.t + 2) % 3] = 0.0
    keep_z, keep_slabs = zb.clone(), slabs.clone()
    o.zstep_codes_(zb, mb, sb, dpt, o.pack_codes(gv, None, b), b, sched_b.next(), -eps, eps, slabs, stop=stop_b)
    assert torch.equal(zb, keep_z) and torch.equal(slabs, keep_slabs) and stop_b.converged()


def test_zstep_codes_unsupported_shapes_are_refused():
    """Shapes outside the fused kernel (pixel count not a multiple of 128, K > 112) report 0 slab bytes, and the entry
    point itself answers ADIL_EINVAL instead of running; DDragueSolver then keeps the two-launch route."""
    from dl_attack_on_imagenet_amd import engine
    o = ops()
    assert o.zstep_codes_slab_bytes(5, 3 * 7 * 9, 3) == 0 and o.zstep_codes_slab_bytes(40, 3 * 64 * 64, 128) == 0
    assert o.zstep_codes_slab_bytes(40, 3 * 64 * 64, 112) > 0
    z = torch.zeros(5, 3, 7, 9, device=DEV)
    with pytest.raises(ValueError):
        o.zstep_codes_(z, z.clone(), z.clone(), torch.zeros(3, 7, 9, 3, device=DEV), o.pack_codes(torch.zeros(5, 3, device=DEV), None, 5),
                       5, o.AdamWSchedule(1e-2).next(), -1.0, 1.0, torch.empty(1024, dtype=torch.uint8, device=DEV))
    from tinynet import make_tinynet
    net = make_tinynet(3).to(DEV)
    gen = torch.Generator().manual_seed(2)
    d = (-1 + 2 * torch.rand(3, 7, 9, 3, generator=gen)).to(DEV)
    solver = engine.DDragueSolver(net, torch.rand(5, 3, 7, 9, generator=gen).to(DEV), d, 0.1, "ce")
    assert solver._vslabs is None
    solver.run(3)
    assert solver.result()[0].shape == (5, 3, 7, 9)


@pytest.mark.parametrize("shape", [(5, 3, 7, 9, 3), (33, 3, 16, 16, 10), (64, 3, 20, 12, 50), (40, 3, 8, 8, 100), (70, 3, 32, 32, 128)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_synth_fp8_operands(shape, dt):
    """adil_synth_fp8 (configs[4]: fp8 D.V MFMA) against the oracle's restatement of the quantised contraction (operands
    rounded to OCP e4m3 after scaling, exact products, fp32 accumulation): agreement to fp32 accumulation error / one
    ulp of the stream type; and against the exact fp32 synthesis within what 3 mantissa bits per operand allow."""
    b, c, h, w, k = shape
    eps = 8 / 255
    gen = torch.Generator().manual_seed(sum(shape) + 7)
    d = -1 + 2 * torch.rand(c, h, w, k, generator=gen)
    v = O.project_onto_l1_ball(torch.randn(b, k, generator=gen) * 0.02, eps)            # |v| <= eps, rows in the l1 ball
    x = torch.rand(b, c, h, w, generator=gen).to(dt).float()
    vp = ops().pack_codes(v.to(DEV), None, b)
    out = ops().synth(x.to(DEV).to(dt), d.to(DEV), vp, b, fp8_absmax=eps)
    assert out.dtype == dt
    want = O.synth_fp8(x, d, v, eps)
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 0.0          # outputs in [0, 1 + eps]: half a bf16 ulp at 1.0 = 2^-8
    close(out.float(), want, 5e-6 + ulp, "fp8 synth vs quantised oracle")
    exact = _oracle_synth(x, d, v)
    dv_err = float((want - exact).abs().max())
    assert dv_err <= 0.15 * eps, dv_err                         # the price of e4m3 operands, well inside the budget
    clamped = ops().synth(x.to(DEV).to(dt), d.to(DEV), vp, b, fp8_absmax=eps, delta_clamp=0.004, pixel_clamp=True)
    wantc = (x.double() + (want.double() - x.double()).clamp(-0.004, 0.004)).clamp(0, 1)
    close(clamped.float(), wantc, 5e-6 + ulp)


def test_synth_fp8_full_size_against_the_quantised_oracle_and_the_persistent_copy():
    """configs[4] at the size it names (VERDICT r3 #4a): 3 x 224 x 224 images, 100 atoms, 96 rows, against the ORACLE's
    restatement of the quantised contraction (O.synth_fp8), not against another kernel.  And the persistent fp8 copy of the
    dictionary (round 4): dict_to_fp8 and the copy adamw_clamp_ maintains are the bytes the on-the-fly conversion forms, so
    adil_synth_fp8_packed returns adil_synth_fp8's result bit for bit — before and after an AdamW step."""
    o = ops()
    b, k, eps = 96, 100, 8 / 255
    gen = torch.Generator().manual_seed(31)
    d = (-1 + 2 * torch.rand(3, 224, 224, k, generator=gen))
    v = O.project_onto_l1_ball(torch.randn(b, k, generator=gen) * 0.02, eps)
    x = torch.rand(b, 3, 224, 224, generator=gen).to(torch.bfloat16)
    dg, vp = d.to(DEV), o.pack_codes(v.to(DEV), None, b)
    out = o.synth(x.to(DEV), dg, vp, b, fp8_absmax=eps)
    want = O.synth_fp8(x.float(), d, v, eps)
    close(out.float(), want, 5e-6 + 2.0 ** -7, "fp8 synth, full size, vs quantised oracle")
    assert float((want - _oracle_synth(x.float(), d, v)).abs().max()) <= 0.15 * eps
    d8 = o.dict_to_fp8(dg)
    assert d8.dtype == torch.uint8 and d8.shape == dg.shape
    assert torch.equal(out, o.synth(x.to(DEV), dg, vp, b, fp8_absmax=eps, d_fp8=d8))
    out32 = o.synth(x.to(DEV).float(), dg, vp, b, fp8_absmax=eps, delta_clamp=0.004, pixel_clamp=True)
    assert torch.equal(out32, o.synth(x.to(DEV).float(), dg, vp, b, fp8_absmax=eps, delta_clamp=0.004, pixel_clamp=True, d_fp8=d8))
    # one AdamW + clamp step with the copy maintained inside the launch == the step without it, then re-quantised
    g = torch.randn(d.shape, generator=gen).to(DEV)
    pa, pb = dg.clone(), dg.clone()
    ma, sa, mb, sb = (torch.zeros_like(dg) for _ in range(4))
    h = o.AdamWSchedule(0.01).next()
    o.adamw_clamp_(pa, g, ma, sa, h, -1.0, 1.0)
    o.adamw_clamp_(pb, g, mb, sb, h, -1.0, 1.0, p_fp8=d8)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(sa, sb) and torch.equal(d8, o.dict_to_fp8(pa))
    # K = 50 is not a multiple of 4: the packed route refuses, the on-the-fly route serves
    assert not o.fp8_dict_supported(torch.empty(3, 32, 32, 50)) and o.fp8_dict_supported(torch.empty(3, 32, 32, 100))


def test_fp8_synth_learner_vit_b16():
    """configs[4] path: ViT-B/16 (197 tokens), 100 atoms, bf16 streams, the synthesis contraction on fp8 MFMAs, through
    DictionaryLearner.step — judged on what the attack is about: fooled counts against the bf16-operand learner on the
    same inputs (free-running: within 2 of 16 images at every step, teacher-forced from the bf16 learner's state
    as well), and the perturbations it synthesises within 0.15 eps of the bf16-operand ones."""
    from dl_attack_on_imagenet_amd import engine, zoo
    b, k, eps, T = 16, 100, 8 / 255, 6
    gen = torch.Generator().manual_seed(5)
    images = torch.rand(b, 3, 224, 224, generator=gen).to(DEV).to(torch.bfloat16)
    d0 = (-1 + 2 * torch.rand(3, 224, 224, k, generator=gen)).to(DEV)
    v0 = ops().l1ball_project_(torch.rand(b, k, generator=gen).to(DEV), eps)
    model = zoo.build_classifier("vit_b_16", seed=1, device=DEV, dtype=torch.bfloat16)
    index = torch.arange(b, device=DEV)
    ref = engine.DictionaryLearner(d0.clone(), v0.clone(), eps, 0.01, "logits")
    fp8 = engine.DictionaryLearner(d0.clone(), v0.clone(), eps, 0.01, "logits", fp8_synth=True)
    forced = engine.DictionaryLearner(d0.clone(), v0.clone(), eps, 0.01, "logits", fp8_synth=True)
    f_ref, f_fp8, f_forced, dv_err = [], [], [], 0.0
    for _ in range(T):
        for a, bb in ((forced.d, ref.d), (forced.v, ref.v), (forced.m_d, ref.m_d), (forced.s_d, ref.s_d),
                      (forced.m_v, ref.m_v), (forced.s_v, ref.s_v)):
            a.copy_(bb)
        forced.sched_d.t, forced.sched_v.t = ref.sched_d.t, ref.sched_v.t
        forced.sync_fp8_copy()                                   # d was overwritten from outside: its fp8 copy follows
        vp = ops().pack_codes(ref.v, None, b)
        dv_bf = ops().synth(None, ref.d, vp, b, out_shape=images.shape, out_dtype=torch.float32)
        dv_f8 = ops().synth(None, ref.d, vp, b, out_shape=images.shape, out_dtype=torch.float32, fp8_absmax=eps)
        dv_err = max(dv_err, float((dv_bf - dv_f8).abs().max()))
        f_forced.append(int(forced.step(model, images, index)[1]))
        f_ref.append(int(ref.step(model, images, index)[1]))
        f_fp8.append(int(fp8.step(model, images, index)[1]))
    print("fp8 ViT-B/16: fooled bf16 %s fp8 %s teacher-forced fp8 %s, max |dv_fp8 - dv_bf16| = %.2e (eps %.2e)" %
          (f_ref, f_fp8, f_forced, dv_err, eps))
    assert f_ref[-1] >= 8                                       # the attack works on this target
    assert max(abs(a - c) for a, c in zip(f_ref, f_fp8)) <= 2
    assert max(abs(a - c) for a, c in zip(f_ref, f_forced)) <= 2
    assert dv_err <= 0.15 * eps
    assert float(fp8.d.abs().max()) <= 1.0 and float(fp8.v.abs().sum(1).max()) <= eps * (1 + 1e-5)


def test_device_side_stop_test_is_exact():
    """ops.StopTest: after the iteration whose max|delta| falls below the threshold, later launches change nothing, so
    polling the flag every n-th iteration ends on exactly the iterate the reference's per-iteration `break` returns
    (adil.py:559, :614) — including when many launches follow the converged one (the slots rotate through stale values).
    Driven with 3 real gradients and then zeros: AdamW's momentum decays 0.9 x per step until the step is below 2e-3."""
    gen = torch.Generator().manual_seed(3)
    n, k, T, thr = 9, 6, 60, 2e-3
    grads = [(torch.randn(n, k, generator=gen) if t < 3 else torch.zeros(n, k)).to(DEV) for t in range(T)]
    v = torch.zeros(n, k, device=DEV); m, s = torch.zeros_like(v), torch.zeros_like(v)
    sched, delta = ops().AdamWSchedule(1e-2), torch.zeros(1, device=DEV)
    ref_iters = 0
    for t in range(T):                                          # reference semantics: test after every iteration
        ref_iters += 1
        delta.zero_()
        ops().adamw_l1ball_(v, grads[t], None, m, s, sched.next(), -1.0, max_abs_delta=delta)
        if float(delta) < thr:
            break
    v_ref = v.clone()
    assert 8 < ref_iters < T - 20, ref_iters                    # converges mid-run, with >= 20 launches to spare
    for poll in (1, 4, 7, 1000):
        v = torch.zeros(n, k, device=DEV); m, s = torch.zeros_like(v), torch.zeros_like(v)
        sched, stop = ops().AdamWSchedule(1e-2), ops().StopTest(DEV, thr)
        launched = 0
        for t in range(T):
            launched += 1
            ops().adamw_l1ball_(v, grads[t], None, m, s, sched.next(), -1.0, stop=stop)
            if (t + 1) % poll == 0 and stop.converged():
                break
        assert torch.equal(v, v_ref), poll                      # bit-identical to the per-iteration break
        assert stop.converged() and launched >= ref_iters, poll
        assert launched == (T if poll == 1000 else -(-ref_iters // poll) * poll), (poll, launched, ref_iters)


@pytest.mark.parametrize("b,k,hw", [(512, 50, 64), (512, 10, 64), (500, 50, 64), (1024, 50, 64), (512, 64, 32),
                                    (512, 50, 224), (1024, 33, 224)])
def test_grad_fused_whole_workgroups(b, k, hw):
    """The fused single pass at the benchmark's workgroup shape (bf16 streams, 512-row workgroups, K <= 64): one, a few
    and many tiles per workgroup (hw = 32 / 64 / 224), ragged rows inside the last block (500), a second accumulating
    row chunk (1024), both atom-tile counts, compact slab rows (K not a multiple of 32) — against fp64 matmuls on the
    device with operands rounded as the kernel rounds them, against the two single-output kernels (different code, same
    maths), and bitwise reproducible."""
    gen = torch.Generator().manual_seed(b + k + hw)
    d = (-1 + 2 * torch.rand(3, hw, hw, k, generator=gen)).to(DEV)
    v = (torch.randn(b, k, generator=gen) * 0.02).to(DEV)
    g = torch.randn(b, 3, hw, hw, generator=gen).to(DEV).to(torch.bfloat16)
    vp = ops().pack_codes(v, None, b)
    gd, gvb = ops().grad(g, d, vp, b)
    dq, vq = d.bfloat16().double().reshape(-1, k), v.bfloat16().double()
    g2 = g.double().reshape(b, -1)
    p = 3 * hw * hw
    close(gd.reshape(-1, k), g2.t() @ vq, 5e-5 * b ** 0.5)
    close(gvb, g2 @ dq, 1e-4 * p ** 0.5)
    gd_only, _ = ops().grad(g, d, vp, b, want_v=False)
    _, gv_only = ops().grad(g, d, None, b, want_d=False)
    close(gd, gd_only, 2e-6 * b ** 0.5 * 4)
    close(gvb, gv_only, 3e-6 * p ** 0.5 * 4)
    gd2, gvb2 = ops().grad(g, d, vp, b)
    assert torch.equal(gd, gd2) and torch.equal(gvb, gvb2)
    acc = gd.clone()
    ops().grad(g, d, vp, b, want_v=True, grad_d=acc, accumulate_d=True)
    close(acc, 2 * gd, 2e-6 * b ** 0.5 * 8)


@pytest.mark.parametrize("b,k,hw", [(256, 50, 64), (512, 50, 64), (500, 10, 64), (70, 64, 32), (33, 33, 16), (300, 50, 224)])
def test_grad_fused_fp32_presplit_kernel(b, k, hw):
    """The fp32 fused pass with every operand split once (grad_fused_f32_kernel: 32-pixel tiles, image and dictionary
    planes in LDS, pre-split codes): one launch and an accumulating second chunk (512 / 500 / 300 rows), ragged rows,
    one / a few / many tiles per workgroup, both atom-tile counts — against fp64 matmuls at fp32-grade tolerances,
    against the two single-output kernels, accumulation into an existing grad_d, bitwise reproducible."""
    gen = torch.Generator().manual_seed(b + k + hw + 1)
    d = (-1 + 2 * torch.rand(3, hw, hw, k, generator=gen)).to(DEV)
    v = (torch.randn(b, k, generator=gen) * 0.02).to(DEV)
    g = torch.randn(b, 3, hw, hw, generator=gen).to(DEV)
    vp = ops().pack_codes(v, None, b)
    gd, gvb = ops().grad(g, d, vp, b)
    p = 3 * hw * hw
    g2 = g.double().reshape(b, -1)
    close(gd.reshape(-1, k), g2.t() @ v.double(), 2e-6 * b ** 0.5 * 4)
    close(gvb, g2 @ d.double().reshape(-1, k), 3e-6 * p ** 0.5 * 4)
    gd_only, _ = ops().grad(g, d, vp, b, want_v=False)
    _, gv_only = ops().grad(g, d, None, b, want_d=False)
    close(gd, gd_only, 2e-6 * b ** 0.5 * 4)
    close(gvb, gv_only, 3e-6 * p ** 0.5 * 4)
    gd2, gvb2 = ops().grad(g, d, vp, b)
    assert torch.equal(gd, gd2) and torch.equal(gvb, gvb2)
    acc = gd.clone()
    ops().grad(g, d, vp, b, grad_d=acc, accumulate_d=True)
    close(acc, 2 * gd, 2e-6 * b ** 0.5 * 8)


@pytest.mark.parametrize("b,k,hw", [(600, 50, 64), (512, 64, 32), (300, 33, 64), (200, 50, 32), (97, 10, 32), (5, 50, 224)])
def test_grad_v_fp32_presplit_kernel(b, k, hw):
    """v = z D_dagger^T / dL/dv = g D for fp32 streams (grad_v_f32_kernel: planes split once on the way to LDS, 512 rows
    per launch): 16 / 8 / 4-wave workgroups, waves beyond the batch, a second chunk (600 rows), ragged rows and atoms —
    against an fp64 matmul at fp32-grade tolerance, bitwise reproducible."""
    gen = torch.Generator().manual_seed(3 * b + k + hw)
    d = (-1 + 2 * torch.rand(3, hw, hw, k, generator=gen)).to(DEV)
    g = torch.randn(b, 3, hw, hw, generator=gen).to(DEV)
    p = 3 * hw * hw
    _, gv = ops().grad(g, d, None, b, want_d=False)
    assert gv.shape == (b, k)
    close(gv, g.double().reshape(b, -1) @ d.double().reshape(-1, k), 3e-6 * p ** 0.5 * 4)
    _, gv2 = ops().grad(g, d, None, b, want_d=False)
    assert torch.equal(gv, gv2)


@pytest.mark.parametrize("b,k,c,h,w", [(512, 100, 3, 32, 32), (300, 65, 3, 16, 24), (700, 128, 3, 16, 16), (1024, 100, 3, 16, 16),
                                       (257, 100, 3, 9, 7), (200, 100, 3, 16, 16)])
def test_grad_bf16_many_atoms(b, k, c, h, w):
    """K > 64 on bf16 streams: up to 256 rows one fused launch; beyond that grad_d through LDS in 512-row launches (the
    grad_d half of the fused kernel; the second chunk accumulates) + the grad_v kernel.  Ragged rows / pixels / atoms,
    grad_d alone, accumulation into an existing grad_d, against fp64 matmuls on the bf16-rounded operands, reproducible."""
    gen = torch.Generator().manual_seed(b + k + h)
    d = (-1 + 2 * torch.rand(c, h, w, k, generator=gen)).to(DEV)
    v = (torch.randn(b, k, generator=gen) * 0.02).to(DEV)
    g = torch.randn(b, c, h, w, generator=gen).to(DEV).bfloat16()
    p = c * h * w
    vp = ops().pack_codes(v, None, b)
    gd, gvb = ops().grad(g, d, vp, b)
    g2 = g.double().reshape(b, p)
    rd = g2.t() @ v.bfloat16().double()
    rv = g2 @ d.bfloat16().double().reshape(p, k)
    close(gd.reshape(p, k), rd, 1e-5 * b ** 0.5 * 4)
    close(gvb, rv, 1e-5 * p ** 0.5 * 4)
    gd_only, none = ops().grad(g, d, vp, b, want_v=False)
    assert none is None
    close(gd_only.reshape(p, k), rd, 1e-5 * b ** 0.5 * 4)
    gd2, gvb2 = ops().grad(g, d, vp, b)
    assert torch.equal(gd, gd2) and torch.equal(gvb, gvb2)
    acc = gd.clone()
    ops().grad(g, d, vp, b, grad_d=acc, accumulate_d=True)
    close(acc, 2 * gd, 1e-5 * b ** 0.5 * 8)
    acc = gd_only.clone()
    ops().grad(g, d, vp, b, want_v=False, grad_d=acc, accumulate_d=True)
    close(acc, 2 * gd_only, 1e-5 * b ** 0.5 * 8)


@pytest.mark.parametrize("b,k,c,h,w", [(512, 100, 3, 32, 32), (300, 65, 3, 16, 24), (130, 128, 3, 16, 16), (40, 100, 3, 8, 8),
                                       (257, 100, 3, 9, 7)])
def test_grad_fp32_many_atoms(b, k, c, h, w):
    """K > 64 on fp32 streams: grad_d alone through the pre-split kernel (grad_fused_f32_kernel without its grad_v half,
    256 rows per launch, the second chunk accumulates) and grad_v through grad_v_f32_kernel<4> — or the generic kernels
    when the rows are not whole 32-pixel tiles (9x7).  fp32-grade tolerances against fp64 matmuls, reproducible."""
    gen = torch.Generator().manual_seed(b + k + h + 5)
    d = (-1 + 2 * torch.rand(c, h, w, k, generator=gen)).to(DEV)
    v = (torch.randn(b, k, generator=gen) * 0.02).to(DEV)
    g = torch.randn(b, c, h, w, generator=gen).to(DEV)
    p = c * h * w
    vp = ops().pack_codes(v, None, b)
    gd, gvb = ops().grad(g, d, vp, b)
    g2 = g.double().reshape(b, p)
    rd, rv = g2.t() @ v.double(), g2 @ d.double().reshape(p, k)
    close(gd.reshape(p, k), rd, 2e-6 * b ** 0.5 * 4)
    close(gvb, rv, 3e-6 * p ** 0.5 * 4)
    gd_only, _ = ops().grad(g, d, vp, b, want_v=False)
    _, gv_only = ops().grad(g, d, None, b, want_d=False)
    assert torch.equal(gd, gd_only) and torch.equal(gvb, gv_only)
    gd2, gvb2 = ops().grad(g, d, vp, b)
    assert torch.equal(gd, gd2) and torch.equal(gvb, gvb2)
    acc = gd.clone()
    ops().grad(g, d, vp, b, grad_d=acc, accumulate_d=True)
    close(acc, 2 * gd, 2e-6 * b ** 0.5 * 8)


@pytest.mark.parametrize("k", [10, 50, 100])
def test_gram_full_size(k):
    """D^T D at the BASELINE image size on the MFMA kernel (gram_mfma_kernel: the transposed D tile is both operands):
    fp32-grade against an fp64 matmul, symmetric, bitwise reproducible."""
    gen = torch.Generator().manual_seed(k)
    d = (-1 + 2 * torch.rand(3, 224, 224, k, generator=gen)).to(DEV)
    gm = ops().gram(d)
    d2 = d.double().reshape(-1, k)
    ref = d2.t() @ d2
    close(gm, ref, 3e-6 * float(ref.abs().max()))
    close(gm, gm.t(), 3e-6 * float(ref.abs().max()))
    assert torch.equal(gm, ops().gram(d))


# ----------------------------------------------------------------------------- ABI 6: helper launches folded into neighbours
@pytest.mark.parametrize("b,k,hw,dt", [(512, 50, 56, torch.bfloat16), (500, 50, 64, torch.bfloat16), (37, 10, 56, torch.bfloat16),
                                       (256, 50, 56, torch.float32), (70, 64, 32, torch.float32), (5, 3, 9, torch.float32),
                                       (512, 100, 32, torch.bfloat16), (256, 100, 32, torch.float32), (97, 128, 24, torch.bfloat16)])
def test_transposed_codes_and_deferred_slab_reduction_are_bit_neutral(b, k, hw, dt):
    """ABI 6: (i) pack_codes' transposed copy is exactly vp^T in the stream dtype and grad_d computed from it equals the
    grad_d of the call that transposes for itself, bit for bit; (ii) a grad_v left as per-workgroup partial sums
    (ops.SlabGrad) and summed inside the consumer — pack_codes, adamw_l1ball_ — gives the bits of the dense route (the
    library's own reduce kernel uses the same slab_sum order); slab counts with and without a tail of the 32-wide rounds
    (56x56: 147 slabs = 4 rounds + 19; 64x64: 192 = 6 rounds; 9x9: 4)."""
    o = ops()
    g0 = torch.Generator().manual_seed(b * 7 + k + hw)
    n = b + 9
    d = (-1 + 2 * torch.rand(3, hw, hw, k, generator=g0)).to(DEV)
    v = (torch.randn(n, k, generator=g0) * 0.01).to(DEV)
    index = torch.randperm(n, generator=g0)[:b].to(DEV)
    gup = torch.randn(b, 3, hw, hw, generator=g0).to(DEV).to(dt).contiguous()
    vp = o.pack_codes(v, index, b)
    vp2, vpt = o.pack_codes(v, index, b, transposed=dt)
    assert torch.equal(vp, vp2)
    rows = vpt.shape[0]
    assert vpt.dtype == dt and rows >= vp.shape[1] and vpt.shape[1] == vp.shape[0]
    assert torch.equal(vpt[:vp.shape[1]].float(), vp.t().to(dt).float()) and not bool(vpt[vp.shape[1]:].any())
    gd_a, gv_a = o.grad(gup, d, vp, b)
    gd_b, gv_b = o.grad(gup, d, vp, b, vpt=vpt)
    assert torch.equal(gd_a, gd_b) and torch.equal(gv_a, gv_b)
    # deferred reduction, fused pass and grad_v-only pass
    for want_d in (True, False):
        _, dense = o.grad(gup, d, vp if want_d else None, b, want_d=want_d)
        _, lazy = o.grad(gup, d, vp if want_d else None, b, want_d=want_d, defer_v=True)
        if isinstance(lazy, o.SlabGrad):
            assert lazy.shape == (b, k) and lazy.nslabs > 0
            packed = o.pack_codes(lazy, None, b)
            assert torch.equal(packed, o.pack_codes(dense, None, b))
            _, lazy = o.grad(gup, d, vp if want_d else None, b, want_d=want_d, defer_v=True)    # same workspace, same values
            p2, p2t = o.pack_codes(lazy, None, b, transposed=torch.float32)
            assert torch.equal(p2, packed) and torch.equal(p2t[:packed.shape[1]], packed.t())
        else:
            assert torch.equal(lazy, dense)                       # more than one row chunk: reduced inside adil_grad
        # through AdamW + projection on all n rows (slot table from pack_codes)
        res = []
        for defer in (False, True):
            vv, m, s = v.clone(), torch.zeros_like(v), torch.zeros_like(v)
            pos = torch.full((n,), -1, dtype=torch.int32, device=DEV)
            sched = o.AdamWSchedule(0.01)
            for _ in range(2):
                o.pack_codes(vv, index, b, pos=pos)
                _, gv = o.grad(gup, d, vp if want_d else None, b, want_d=want_d, defer_v=defer)
                o.adamw_l1ball_(vv, gv, pos, m, s, sched.next(), 8 / 255, reset_pos=True)
            res.append((vv, m, s))
            assert int((pos != -1).sum()) == 0
        for x, y in zip(*res):
            assert torch.equal(x, y)
    # no slot table: row n <-> gradient row n (forward_supervised_AdamW's call)
    _, lazy = o.grad(gup, d, None, b, want_d=False, defer_v=True)
    _, dense = o.grad(gup, d, None, b, want_d=False)
    out = []
    for gsrc in (dense, lazy):
        vv = v[:b].clone()
        m, s = torch.zeros_like(vv), torch.zeros_like(vv)
        if gsrc is lazy and isinstance(lazy, o.SlabGrad):
            _, gsrc = o.grad(gup, d, None, b, want_d=False, defer_v=True)
        o.adamw_l1ball_(vv, gsrc, None, m, s, o.AdamWSchedule(0.01).next(), 8 / 255)
        out.append(vv)
    assert torch.equal(out[0], out[1])


def test_stale_deferred_gradient_is_refused():
    """a SlabGrad points into the per-stream scratch of the ops.grad call that produced it; once that scratch has
    been handed out again (another ops.grad / gram / atom_norms on the stream) its consumers must refuse it instead of
    summing whatever the later call left there.  Slabs in a caller-owned buffer (zstep_codes_) carry no such limit."""
    o = ops()
    g0 = torch.Generator().manual_seed(4)
    b, k = 64, 10
    d = (-1 + 2 * torch.rand(3, 32, 32, k, generator=g0)).to(DEV)
    gup = torch.randn(b, 3, 32, 32, generator=g0).to(DEV)
    _, lazy = o.grad(gup, d, None, b, want_d=False, defer_v=True)
    assert isinstance(lazy, o.SlabGrad)
    fresh = o.pack_codes(lazy, None, b)                          # consumed right away: fine, and repeatable
    assert torch.equal(fresh, o.pack_codes(lazy, None, b))
    o.gram(d)                                                    # the scratch is handed out again
    with pytest.raises(RuntimeError, match="stale SlabGrad"):
        o.pack_codes(lazy, None, b)
    v, m, s = torch.zeros(b, k, device=DEV), torch.zeros(b, k, device=DEV), torch.zeros(b, k, device=DEV)
    with pytest.raises(RuntimeError, match="stale SlabGrad"):
        o.adamw_l1ball_(v, lazy, None, m, s, o.AdamWSchedule(0.01).next(), 0.1)
    _, again = o.grad(gup, d, None, b, want_d=False, defer_v=True)
    assert torch.equal(o.pack_codes(again, None, b), fresh)


def test_deferred_reduction_sums_match_the_oracle_at_full_size():
    """Full-size check of the consumer-side reduction (ResNet-sized images, 512 bf16 rows, 50 atoms: 236 slabs): the codes
    pack_codes forms from the slabs equal g D of the fp64 oracle on the bf16-rounded operands."""
    o = ops()
    g0 = torch.Generator().manual_seed(3)
    b, k = 512, 50
    d = (-1 + 2 * torch.rand(3, 224, 224, k, generator=g0)).to(DEV)
    gup = (torch.randn(b, 3, 224, 224, generator=g0) * 1e-3).to(DEV).to(torch.bfloat16)
    _, lazy = o.grad(gup, d, None, b, want_d=False, defer_v=True)
    assert isinstance(lazy, o.SlabGrad) and lazy.nslabs > 200
    got = o.pack_codes(lazy, None, b)[:b, :k]
    ref = gup.double().flatten(1) @ d.to(torch.bfloat16).double().reshape(-1, k)
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-7


def test_constraint_dict_l1_branch_golden_and_full_size():
    """constraint_dict's else-branch (utils.py:55-56; no caller upstream): every (channel, atom) row of H*W pixels onto the
    l1 ball of radius 1 — fixture G16 (the reference's own output) and a full-size dictionary against the oracle's
    sort-based projection, plus the size-independent properties: row l1 norms <= 1, idempotence, untouched inside rows."""
    from dl_attack_on_imagenet_amd.attacks import utils as U
    z = load_golden("g16_constraint_l1")
    d = t(z["d"], DEV).contiguous()
    out = U.constraint_dict(d.clone(), "l1ball")
    close(out, z["l1ball"], 2e-7, "G16")
    g0 = torch.Generator().manual_seed(16)
    big = torch.randn(3, 224, 224, 7, generator=g0) * 1e-4          # rows of 50176 pixels, l1 norm ~ 4
    big[..., 0] *= 0.1                                               # atom 0 inside the ball
    got = U.constraint_dict(big.to(DEV), "l1ball")
    ref = O.constraint_dict(big, "l1ball")
    close(got, ref, 2e-8, "full size")      # the oracle's fp32 cumsum over 50176 sorted values carries ~2e-9 into its threshold
    l1 = got.abs().sum(dim=(1, 2)).cpu()
    assert float(l1.max()) <= 1.0 + 1e-5 and torch.equal(got[..., 0].cpu(), big[..., 0])
    close(U.constraint_dict(got.clone(), "l1ball"), got, 2e-8, "idempotent")
