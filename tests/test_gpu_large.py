"""Large batches: 32-bit overflow guard for the row/byte arithmetic of the contraction kernels.  At 4096+ fp32 images of
3x224x224 the image tensor is 2.47 GB (> 2^31 bytes, element count 6.2e8) — far below what 288 GB of HBM invites.  The
expected values come from torch matmuls on the same device (fp32, a few rows checked in fp64)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from dl_attack_on_imagenet_amd import ops as o
    return o


@pytest.mark.parametrize("dtype,b", [(torch.float32, 4099), (torch.bfloat16, 8195)])
def test_contractions_beyond_2gb(dtype, b):
    k, hw = 10, 224
    p = 3 * hw * hw
    assert b * p * torch.empty((), dtype=dtype).element_size() > 2 ** 31
    gen = torch.Generator().manual_seed(b)
    d = (-1 + 2 * torch.rand(3, hw, hw, k, generator=gen)).to(DEV)
    v = (torch.randn(b, k, generator=gen) * 0.01).to(DEV)
    x = torch.rand(b, 3, hw, hw, device=DEV, dtype=dtype)          # device-side fill: 2.5 GB through the host is slow
    vp = ops().pack_codes(v, None, b)
    out = ops().synth(x, d, vp, b)
    tol = 2e-6 if dtype == torch.float32 else 8e-3
    d2 = d.reshape(p, k)
    for rows in (slice(0, 3), slice(b // 2 - 1, b // 2 + 2), slice(b - 3, b)):       # first, middle (past 2^31 bytes), last rows
        ref = x[rows].reshape(-1, p).double() + v[rows].double() @ d2.double().t()
        assert (out[rows].reshape(-1, p).double() - ref).abs().max().item() <= tol
    del out
    g = x                                                          # any stream will do as the gradient
    gd, gvb = ops().grad(g, d, vp, b)
    sel = torch.tensor([0, 1, b // 2, b - 2, b - 1], device=DEV)
    # the bf16-stream kernels feed the matrix pipe bf16 operands: the expected values use the operands as rounded
    rnd = (lambda t: t.to(torch.bfloat16).double()) if dtype == torch.bfloat16 else (lambda t: t.double())
    ref_v = g[sel].reshape(len(sel), p).double() @ rnd(d2)
    assert (gvb[sel].double() - ref_v).abs().max().item() <= 3e-6 * p ** 0.5 * 4
    # grad_d = g^T v: compare on a slice of pixels (all rows contribute to each)
    px = torch.arange(0, p, 4099, device=DEV)
    ref_d = g.reshape(b, p)[:, px].double().t() @ rnd(v)
    assert (gd.reshape(p, k)[px].double() - ref_d).abs().max().item() <= 2e-6 * b ** 0.5 * 4
    if dtype == torch.float32:                                     # z-step on the same sizes: rows past 2^31 bytes must move
        z = torch.zeros_like(x)
        m, s = torch.zeros_like(x), torch.zeros_like(x)
        from dl_attack_on_imagenet_amd import engine
        pinv = engine.PseudoInverse(d)
        hz = ops().AdamWSchedule(1e-2).next()
        gv = ops().pack_codes(torch.randn(b, k, generator=gen).to(DEV), None, b)
        ops().zstep_(z, m, s, pinv.d_pinv_t, gv, b, hz, -8 / 255, 8 / 255)
        moved = (z.reshape(b, -1).abs().amax(dim=1) > 0)
        assert bool(moved.all())
