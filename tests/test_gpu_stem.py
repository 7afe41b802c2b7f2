"""GPU parity of the hand-written frozen-ResNet kernels (csrc/adil_stem.hip, csrc/adil_convs.hip) against plain PyTorch fp32 references of
the same ops on the same (bf16-rounded) operands.  These kernels sit on either side of the ADiL hot path (they
consume x_adv and produce dLoss/dx_adv); tolerances are bf16 output rounding (2^-8 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def ops():
    from dl_attack_on_imagenet_amd import ops as o
    return o


def _stem_params(seed=0):
    gen = torch.Generator().manual_seed(seed)
    w = (torch.randn(64, 3, 7, 7, generator=gen) * 0.05).bfloat16().float()
    scale = (0.5 + torch.rand(64, generator=gen)).float()
    shift = (torch.randn(64, generator=gen) * 0.2).float()
    return w, scale, shift


def _ref_conv_stage(x, w, scale, shift):
    """y1 = relu(bn(conv(normalize(x)))) in fp32 on the bf16-rounded normalised input, rounded to bf16 at the end."""
    mean = torch.tensor(MEAN, device=x.device).reshape(1, 3, 1, 1)
    inv = (1.0 / torch.tensor(STD, device=x.device)).reshape(1, 3, 1, 1)
    xn = ((x.float() - mean) * inv).bfloat16().float()
    y = F.conv2d(xn, w, stride=2, padding=3)
    return torch.relu(y * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1))


@pytest.mark.parametrize("b,h,w_", [(2, 64, 64), (3, 32, 96), (1, 40, 24), (2, 224, 224)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_stem_conv_fwd(b, h, w_, dt):
    o = ops()
    lib = __import__("dl_attack_on_imagenet_amd._lib", fromlist=["x"])
    w, scale, shift = _stem_params()
    gen = torch.Generator().manual_seed(b * h + w_)
    x = torch.rand(b, 3, h, w_, generator=gen).to(DEV).to(dt)
    w, scale, shift = w.to(DEV), scale.to(DEV), shift.to(DEV)     # keep the device copies alive across the launch
    wf, wb = o.pack_stem_weights(w)
    y1 = torch.empty(b, h // 2, w_ // 2, 64, dtype=torch.bfloat16, device=DEV)
    rc = lib.load().adil_stem_conv_fwd(o._ptr(x), o.stream_dtype_code(dt), o._ptr(wf), *MEAN, *[1.0 / s for s in STD],
                                       o._ptr(scale), o._ptr(shift), o._ptr(y1), b, h, w_, o._stream())
    assert rc == 0
    ref = _ref_conv_stage(x, w, scale, shift).permute(0, 2, 3, 1)
    err = (y1.float() - ref).abs()
    tol = 2 ** -7 * ref.abs() + 2e-2          # bf16 output rounding + fp32 accumulation-order differences
    assert bool((err <= tol).all()), float((err - tol).max())
    assert float(err.mean()) < 2e-3


@pytest.mark.parametrize("b,oh,ow,c", [(2, 32, 32, 64), (3, 17, 23, 64), (1, 112, 112, 64), (2, 8, 6, 16)])
def test_maxpool_fwd_and_fused_bwd(b, oh, ow, c):
    """maxpool values bit-exact against torch; routing (first maximum, ties included) + ReLU mask + BN scale against
    torch autograd on the same tensor."""
    o = ops()
    lib = __import__("dl_attack_on_imagenet_amd._lib", fromlist=["x"]).load()
    gen = torch.Generator().manual_seed(oh * ow)
    y1 = torch.relu(torch.randn(b, oh, ow, c, generator=gen)).bfloat16().to(DEV)      # NHWC, many zeros, many ties
    ph, pw = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
    p = torch.empty(b, ph, pw, c, dtype=torch.bfloat16, device=DEV)
    idx = torch.empty(b, ph, pw, c, dtype=torch.uint8, device=DEV)
    assert lib.adil_maxpool_fwd(o._ptr(y1), o._ptr(p), o._ptr(idx), b, oh, ow, c, o._stream()) == 0
    t = y1.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    pref = F.max_pool2d(t, 3, 2, 1)
    assert torch.equal(p.float().permute(0, 3, 1, 2), pref.detach())
    g = torch.randn(b, ph, pw, c, generator=gen).bfloat16().to(DEV)
    scale = (0.5 + torch.rand(c, generator=gen)).to(DEV)
    gy = torch.empty(b, oh, ow, c, dtype=torch.bfloat16, device=DEV)
    assert lib.adil_stem_pool_bwd(o._ptr(g), o._ptr(idx), o._ptr(p), o._ptr(scale), o._ptr(gy), b, oh, ow, c, o._stream()) == 0
    pref.backward(g.float().permute(0, 3, 1, 2))
    ref = (t.grad * (t.detach() > 0) * scale.reshape(1, -1, 1, 1)).permute(0, 2, 3, 1)
    err = (gy.float() - ref).abs()
    assert bool((err <= 2 ** -7 * ref.abs() + 1e-6).all()), float(err.max())


@pytest.mark.parametrize("b,h,w_", [(2, 64, 64), (1, 32, 96), (2, 40, 24), (1, 224, 224)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_stem_conv_bwd(b, h, w_, dt):
    o = ops()
    lib = __import__("dl_attack_on_imagenet_amd._lib", fromlist=["x"]).load()
    w, _, _ = _stem_params(1)
    gen = torch.Generator().manual_seed(h + 3 * w_)
    gy = (torch.randn(b, h // 2, w_ // 2, 64, generator=gen) * 0.1).bfloat16().to(DEV)
    w = w.to(DEV)
    wf, wb = o.pack_stem_weights(w)
    gx = torch.empty(b, 3, h, w_, dtype=dt, device=DEV)
    inv = [1.0 / s for s in STD]
    assert lib.adil_stem_conv_bwd(o._ptr(gy), o._ptr(wb), *inv, o._ptr(gx), o.stream_dtype_code(dt), b, h, w_, o._stream()) == 0
    xin = torch.zeros(b, 3, h, w_, device=DEV, requires_grad=True)
    F.conv2d(xin, w, stride=2, padding=3).backward(gy.float().permute(0, 3, 1, 2))
    ref = xin.grad * torch.tensor(inv, device=DEV).reshape(1, 3, 1, 1)
    err = (gx.float() - ref).abs()
    tol = (2 ** -7 if dt == torch.bfloat16 else 1e-5) * ref.abs() + 1e-3
    assert bool((err <= tol).all()), float((err - tol).max())


def _bf16_depth_bound(layers: int) -> float:
    """Mean |logit error| of a bf16-activation network against its fp32 twin, RELATIVE to the rms logit, that the tests
    below allow.  Model: every activation tensor is stored once in bf16, i.e. with a relative rounding error uniform in
    +-2^-9 (rms 2^-9 / sqrt(3)); `layers` such roundings in series, propagated with unit relative gain, add in quadrature:
    rms relative error sqrt(layers) 2^-9 / sqrt(3), mean |.| = 0.8 of that.  The bound 2 * 2^-9 * sqrt(layers) is 4.3 x
    the model — room for a relative gain above 1 in a random-weight network and for the library's algorithm choices, while a
    broken epilogue (a wrong scale / shift, a dropped residual) is off by the logits' own size, i.e. 1 in these units.
    Recorded: ResNet-18 0.0046, ResNet-50 0.0060 against bounds of 0.0175 and 0.0284."""
    return 2.0 * 2.0 ** -9 * layers ** 0.5


def test_fused_stem_resnet_matches_unfused():
    """ResNet-18 with the stem kernels: against the fp32 network (same weights) its logits and input gradient are as
    accurate as the bf16 conv + epilogue path's (both differ from fp32 only by bf16 rounding of activations)."""
    from dl_attack_on_imagenet_amd import zoo
    ref = zoo.build_classifier("resnet18", num_classes=10, seed=3, device=DEV, dtype=torch.float32)
    kw = dict(num_classes=10, seed=3, device=DEV, dtype=torch.bfloat16, channels_last=True, fuse_bn_act=True)
    m0 = zoo.build_classifier("resnet18", **kw)
    m1 = zoo.build_classifier("resnet18", fuse_stem=True, **kw)
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(8, 3, 64, 64, generator=gen).to(DEV).bfloat16()
    xr, x0, x1 = (x.float().requires_grad_(True), x.clone().requires_grad_(True), x.clone().requires_grad_(True))
    lr, l0, l1 = ref(xr), m0(x0).float(), m1(x1).float()
    e0, e1 = float((l0 - lr).abs().mean()), float((l1 - lr).abs().mean())
    rms = float(lr.square().mean().sqrt())
    print("resnet18 logit error vs fp32: plain bf16 %.4f fused %.4f, rms(logits) %.4f" % (e0, e1, rms))
    assert e1 <= _bf16_depth_bound(20) * rms, (e0, e1, rms)   # measured 0.0054 at rms 1.17: 0.26 of the bound (plain: 0.0059)
    (gr,) = torch.autograd.grad(lr.square().sum(), xr)
    (g0,) = torch.autograd.grad(l0.square().sum(), x0)
    (g1,) = torch.autograd.grad(l1.square().sum(), x1)
    assert g1.dtype == x.dtype and g1.shape == x.shape
    cos = lambda a, b: float(F.cosine_similarity(a.float().flatten(), b.float().flatten(), dim=0))
    c0, c1 = cos(g0, gr), cos(g1, gr)
    assert c1 >= c0 - 0.02, (c0, c1)


@pytest.mark.parametrize("m,k,n", [(128, 64, 64), (300, 64, 256), (1000, 256, 64), (129, 512, 128), (4096, 128, 512),
                                   (512 * 49, 2048, 512)])
@pytest.mark.parametrize("relu,has_res", [(True, False), (True, True), (False, False)])
def test_pointwise_conv_fused_epilogue(m, k, n, relu, has_res):
    """adil_pw_conv_fwd against an fp32 matmul + affine + residual + ReLU on the same bf16 operands."""
    o = ops()
    lib = __import__("dl_attack_on_imagenet_amd._lib", fromlist=["x"]).load()
    gen = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=gen).bfloat16().to(DEV)
    w = (torch.randn(n, k, generator=gen) / k ** 0.5).bfloat16().to(DEV)
    scale = (0.5 + torch.rand(n, generator=gen)).to(DEV)
    shift = (torch.randn(n, generator=gen) * 0.3).to(DEV)
    res = torch.randn(m, n, generator=gen).bfloat16().to(DEV) if has_res else None
    y = torch.full((m + 3, n), 7.0, dtype=torch.bfloat16, device=DEV)
    pre = None
    if k <= 512 and has_res:                       # also exercise the fused prologue x' = relu(x * pscale + pshift)
        pre = ((0.5 + torch.rand(k, generator=gen)).to(DEV), (torch.randn(k, generator=gen) * 0.3).to(DEV))
    assert lib.adil_pw_conv_fwd(o._ptr(x), o._ptr(w), o._ptr(scale), o._ptr(shift), o._ptr(res), o._ptr(y), m, k, n, int(relu),
                                o._ptr(pre[0] if pre else None), o._ptr(pre[1] if pre else None), 0, 0, o._stream()) == 0
    xe = torch.relu(x.float() * pre[0] + pre[1]).bfloat16().float() if pre else x.float()
    ref = (xe @ w.float().t()) * scale + shift
    if has_res:
        ref = ref + res.float()
    if relu:
        ref = torch.relu(ref)
    err = (y[:m].float() - ref).abs()
    assert bool((err <= 2 ** -7 * ref.abs() + 2e-3).all()), float(err.max())
    assert bool((y[m:] == 7.0).all())


@pytest.mark.parametrize("m,k,n", [(128, 64, 64), (300, 64, 256), (1000, 256, 64), (129, 128, 512), (4096, 512, 128),
                                   (512 * 49, 512, 2048)])
@pytest.mark.parametrize("relu,has_res,has_g2", [(True, False, False), (True, True, True), (False, False, False),
                                                  (True, True, False)])
def test_pointwise_conv_fused_backward(m, k, n, relu, has_res, has_g2):
    """adil_pw_conv_bwd against fp32: gres = (g+g2)*[y>0], gx = (gres*scale) @ W on the same bf16 operands."""
    o = ops()
    lib = __import__("dl_attack_on_imagenet_amd._lib", fromlist=["x"]).load()
    gen = torch.Generator().manual_seed(m + 2 * k + n)
    g = torch.randn(m, n, generator=gen).bfloat16().to(DEV)
    g2 = torch.randn(m, n, generator=gen).bfloat16().to(DEV) if has_g2 else None
    y = torch.relu(torch.randn(m, n, generator=gen)).bfloat16().to(DEV)
    w = (torch.randn(n, k, generator=gen) / n ** 0.5).bfloat16().to(DEV)
    wt = w.t().contiguous()
    scale = (0.5 + torch.rand(n, generator=gen)).to(DEV)
    gx = torch.full((m + 2, k), 7.0, dtype=torch.bfloat16, device=DEV)
    gres = torch.full((m + 2, n), 7.0, dtype=torch.bfloat16, device=DEV) if has_res else None
    xin, ps, pb = None, None, None
    if has_g2:                                     # also exercise the fused epilogue gx *= [xin*ps+pb > 0] * ps
        xin = torch.randn(m, k, generator=gen).bfloat16().to(DEV)
        ps, pb = (0.5 + torch.rand(k, generator=gen)).to(DEV), (torch.randn(k, generator=gen) * 0.3).to(DEV)
    assert lib.adil_pw_conv_bwd(o._ptr(g), o._ptr(g2), o._ptr(y), o._ptr(scale), o._ptr(wt), o._ptr(gx), o._ptr(gres), m, k, n,
                                int(relu), o._ptr(xin), o._ptr(ps), o._ptr(pb), None, 0, 0, o._stream()) == 0
    v = g.float() + (g2.float() if has_g2 else 0.0)
    if relu:
        v = v * (y > 0)
    if has_res:
        assert bool(((gres[:m].float() - v).abs() <= 2 ** -7 * v.abs() + 1e-6).all())
        assert bool((gres[m:] == 7.0).all())
    gz = (v.bfloat16().float() if has_res or True else v) * scale      # the kernel scales the fp32 value, rounds once
    ref = (v * scale).bfloat16().float() @ w.float()
    if xin is not None:
        ref = ref.bfloat16().float() * ((xin.float() * ps + pb) > 0) * ps     # the kernel rounds gx once before the epilogue
    err = (gx[:m].float() - ref).abs()
    assert bool((err <= 2 ** -6 * ref.abs() + 3e-2).all()), float(err.max())
    assert float(err.mean()) < 4e-3
    assert bool((gx[m:] == 7.0).all())


def test_fused_resnet50_gradient_matches_fp32():
    """ResNet-50 (bottlenecks: fused pointwise forward/backward kernels, twin residual gradients, stem kernels) against
    the fp32 network with the same weights: input gradient direction and logits."""
    from dl_attack_on_imagenet_amd import zoo
    ref = zoo.build_classifier("resnet50", num_classes=10, seed=5, device=DEV, dtype=torch.float32)
    kw = dict(num_classes=10, seed=5, device=DEV, dtype=torch.bfloat16, channels_last=True)
    m0 = zoo.build_classifier("resnet50", **kw)                                     # plain bf16 torch path
    m1 = zoo.build_classifier("resnet50", fuse_bn_act=True, fuse_stem=True, **kw)
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(4, 3, 64, 64, generator=gen).to(DEV).bfloat16()
    xr, x0, x1 = (x.float().requires_grad_(True), x.clone().requires_grad_(True), x.clone().requires_grad_(True))
    lr, l0, l1 = ref(xr), m0(x0).float(), m1(x1).float()
    e0, e1 = float((l0 - lr).abs().mean().detach()), float((l1 - lr).abs().mean().detach())
    rms = float(lr.square().mean().sqrt())
    print("resnet50 logit error vs fp32: plain bf16 %.4f fused %.4f, rms(logits) %.4f" % (e0, e1, rms))
    # ADVICE r3 / VERDICT r3 #6: an ABSOLUTE bound from a model of bf16 storage, not a ratio to the plain bf16 network (whose
    # own error moves with the MIOpen algorithms a box picks: 0.059 ... 0.088 recorded, which made a ratio test flaky)
    assert e1 <= _bf16_depth_bound(53) * rms, (e0, e1, rms)   # measured 0.0905 at rms 15.1: 0.21 of the bound
    (gr,) = torch.autograd.grad(lr.square().sum(), xr)
    (g0,) = torch.autograd.grad(l0.square().sum(), x0)
    (g1,) = torch.autograd.grad(l1.square().sum(), x1)
    cos = lambda a, b: float(F.cosine_similarity(a.float().flatten(), b.float().flatten(), dim=0))
    c0, c1 = cos(g0, gr), cos(g1, gr)
    assert c1 >= c0 - 0.02, (c0, c1)
    n0, n1, nr = float(g0.float().norm()), float(g1.float().norm()), float(gr.norm())
    assert abs(n1 - nr) <= 2.0 * abs(n0 - nr) + 0.05 * nr, (n0, n1, nr)


def _pack3x3(w):
    """(N,C,3,3) -> forward layout [N][9][C] and input-gradient layout [C][9][N] (taps flipped), bf16."""
    n, c = w.shape[:2]
    fwd = w.permute(0, 2, 3, 1).reshape(n, 9, c).contiguous().bfloat16()
    bwd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(c, 9, n).contiguous().bfloat16()
    return fwd, bwd


@pytest.mark.parametrize("b,h,w_,c,n", [(2, 14, 14, 64, 64), (1, 7, 7, 128, 128), (2, 56, 56, 64, 64), (1, 28, 28, 128, 128),
                                        (3, 5, 9, 64, 128), (2, 14, 14, 256, 256), (1, 3, 3, 192, 64)])
def test_conv3x3_forward_and_input_gradient(b, h, w_, c, n):
    """adil_conv3x3 against torch's fp32 conv2d (same bf16 operands), forward and (with the flipped/transposed
    packing) input gradient; ragged pixel counts, image edges and image-to-image boundaries included."""
    o = ops()
    lib = __import__("dl_attack_on_imagenet_amd._lib", fromlist=["x"]).load()
    gen = torch.Generator().manual_seed(b * h * w_ + c + n)
    x = torch.randn(b, h, w_, c, generator=gen).bfloat16().to(DEV)
    wt = (torch.randn(n, c, 3, 3, generator=gen) / (9 * c) ** 0.5).bfloat16().float().to(DEV)
    wf, wb = _pack3x3(wt)
    y = torch.full((b * h * w_ + 5, n), 7.0, dtype=torch.bfloat16, device=DEV)
    assert lib.adil_conv3x3(o._ptr(x), o._ptr(wf), o._ptr(y), b, h, w_, c, n, o._stream()) == 0
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1).reshape(-1, n)
    err = (y[:b * h * w_].float() - ref).abs()
    assert bool((err <= 2 ** -7 * ref.abs() + 2e-3).all()), float(err.max())
    assert bool((y[b * h * w_:] == 7.0).all())
    g = torch.randn(b, h, w_, n, generator=gen).bfloat16().to(DEV)
    gx = torch.empty(b * h * w_, c, dtype=torch.bfloat16, device=DEV)
    assert lib.adil_conv3x3(o._ptr(g), o._ptr(wb), o._ptr(gx), b, h, w_, n, c, o._stream()) == 0
    xin = torch.zeros(b, c, h, w_, device=DEV, requires_grad=True)
    F.conv2d(xin, wt, padding=1).backward(g.float().permute(0, 3, 1, 2))
    gref = xin.grad.permute(0, 2, 3, 1).reshape(-1, c)
    gerr = (gx.float() - gref).abs()
    assert bool((gerr <= 2 ** -7 * gref.abs() + 2e-3).all()), float(gerr.max())


@pytest.mark.parametrize("b,oh,ow,k,n", [(2, 14, 14, 256, 512), (3, 7, 5, 64, 128), (1, 28, 28, 256, 512)])
def test_pointwise_conv_stride2_gather_and_compact_gradient(b, oh, ow, k, n):
    """Stride-2 1x1 convolution read through the in-kernel gather (forward), and the stride-2 gradient added into a
    full-resolution backward without materialising the zero-upsampled tensor."""
    o = ops()
    lib = __import__("dl_attack_on_imagenet_amd._lib", fromlist=["x"]).load()
    gen = torch.Generator().manual_seed(b + oh + ow + k + n)
    h, w_ = 2 * oh, 2 * ow
    xfull = torch.randn(b, h, w_, k, generator=gen).bfloat16().to(DEV)                 # NHWC
    w = (torch.randn(n, k, generator=gen) / k ** 0.5).bfloat16().to(DEV)
    scale = (0.5 + torch.rand(n, generator=gen)).to(DEV); shift = (torch.randn(n, generator=gen) * 0.3).to(DEV)
    m = b * oh * ow
    y = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    assert lib.adil_pw_conv_fwd(o._ptr(xfull), o._ptr(w), o._ptr(scale), o._ptr(shift), None, o._ptr(y), m, k, n, 0, None, None,
                                ow, oh * ow, o._stream()) == 0
    ref = (xfull[:, ::2, ::2].reshape(m, k).float() @ w.float().t()) * scale + shift
    assert bool(((y.float() - ref).abs() <= 2 ** -7 * ref.abs() + 2e-3).all())
    # backward of a full-resolution layer (channels n2 -> k2) receiving g, g2 and a stride-2 g3
    n2, k2 = k, 64
    mf = b * h * w_
    g = torch.randn(mf, n2, generator=gen).bfloat16().to(DEV); g2 = torch.randn(mf, n2, generator=gen).bfloat16().to(DEV)
    g3 = torch.randn(b, oh, ow, n2, generator=gen).bfloat16().to(DEV)
    yy = torch.relu(torch.randn(mf, n2, generator=gen)).bfloat16().to(DEV)
    wt = (torch.randn(k2, n2, generator=gen) / n2 ** 0.5).bfloat16().to(DEV)
    sc2 = (0.5 + torch.rand(n2, generator=gen)).to(DEV)
    gx = torch.empty(mf, k2, dtype=torch.bfloat16, device=DEV); gres = torch.empty(mf, n2, dtype=torch.bfloat16, device=DEV)
    assert lib.adil_pw_conv_bwd(o._ptr(g), o._ptr(g2), o._ptr(yy), o._ptr(sc2), o._ptr(wt), o._ptr(gx), o._ptr(gres), mf, k2, n2, 1,
                                None, None, None, o._ptr(g3), ow, oh * ow, o._stream()) == 0
    up = torch.zeros(b, h, w_, n2, device=DEV)
    up[:, ::2, ::2] = g3.float()
    v = (g.float() + g2.float() + up.reshape(mf, n2)) * (yy > 0)
    assert bool(((gres.float() - v).abs() <= 2 ** -7 * v.abs() + 1e-6).all())
    gref = (v * sc2).bfloat16().float() @ wt.float().t()
    gerr = (gx.float() - gref).abs()
    assert bool((gerr <= 2 ** -6 * gref.abs() + 3e-2).all()) and float(gerr.mean()) < 4e-3


def test_fused_resnet_accepts_empty_batch():
    """An empty batch (performance.py's correctly-classified filter can leave none) passes through the fused classifier
    like it does through plain torch modules."""
    from dl_attack_on_imagenet_amd import zoo
    m = zoo.build_classifier("resnet50", num_classes=10, seed=1, device=DEV, dtype=torch.bfloat16, channels_last=True,
                             fuse_bn_act=True, fuse_stem=True)
    out = m(torch.zeros(0, 3, 64, 64, device=DEV, dtype=torch.bfloat16))
    assert out.shape == (0, 10)
