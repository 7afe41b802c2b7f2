"""Parity at the BENCHMARKED configurations (BASELINE.json configs[0..4]), on the GPU.

* configs[0] (resnet18, 32 images of 3x224x224, 10 atoms, 20 inner iterations, fp32): the HIP learner against the
  ORACLE RUN ON THE SAME CLASSIFIER BACKEND (oracle code on cuda tensors, the identical `gpu_model` object), which
  isolates the hand-written kernels from MIOpen-vs-MKL differences of the frozen network: fooled counts must be
  EQUAL AT EVERY ITERATION (bit-exact label decisions), perturbations / codes within a stated fp32 bound.  The
  CPU-oracle leg (different conv library under the classifier) is kept as a second, looser assertion.
* configs[1] path (ResNet-50 through zoo.FusedResNet, bf16 image streams): fooling counts and final attack success
  rate of the bf16 product path against the fp32 oracle on the same inputs and seeds.
* configs[2] / configs[4] classifiers (DenseNet-121, ViT-B/16 at 197 tokens) through DictionaryLearner.step.

The numbers each test prints are copied into profiles/r02_parity_configs.md and quoted in DESIGN.md §2."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _note(name, payload):
    """Keep the measured parity numbers next to the other GPU-run artefacts (gpurun_out/ is merged back)."""
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "parity_configs.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **payload}) + "\n")
    except OSError:
        pass
    print(name, json.dumps(payload))


def _oracle_run(O, model, images, d0, v0, T, eps, batches, loss="logits", dev="cpu"):
    """T epochs of learn_dictionary_a's hot loop with the oracle on `dev` tensors. Returns d, v, fooled per step."""
    d, v = d0.clone().to(dev), v0.clone().to(dev)
    sd, sv = O.AdamWState(d, 0.01), O.AdamWState(v, 0.01)
    x = images.to(dev)
    fooled, losses = [], []
    for _ in range(T):
        for idx in batches:
            index = torch.as_tensor(idx, dtype=torch.int64, device=dev)
            ls, fl = O.learn_step_a(model, x[index], index, d, v, sd, sv, eps, loss, -1.0, 50.0)
            fooled.append(int(fl)); losses.append(float(ls))
    return d, v, fooled, losses


def _hip_run(engine, model, images, d0, v0, T, eps, batches, loss="logits", dtype=torch.float32):
    learner = engine.DictionaryLearner(d0.clone().to(DEV), v0.clone().to(DEV), eps, 0.01, loss, False, 50.0)
    x = images.to(DEV).to(dtype).contiguous()
    fooled, losses = [], []
    for _ in range(T):
        for idx in batches:
            index = torch.as_tensor(idx, dtype=torch.int64, device=DEV)
            ls, fl = learner.step(model, x[index].contiguous(), index)
            fooled.append(int(fl)); losses.append(float(ls))
    return learner.d, learner.v, fooled, losses


def _delta(d, v):
    k = d.shape[-1]
    return v.double() @ d.reshape(-1, k).double().t()


# --------------------------------------------------------------------------------------------------------------- #
def test_config1_same_backend_fooled_counts_exact():
    """configs[0]: HIP kernels vs the oracle on the same classifier backend.  Bounds (fp32, stated):
    fooled counts equal at every one of the 20 iterations; loss within 1e-4 relative; max |V_hip - V_oracle| <= 2e-4
    (l1 radius 0.0314); max |D v_hip - D v_oracle| <= 1e-3 (perturbation budget 0.0314).  The fraction of dictionary
    entries that end more than 1e-3 apart is REPORTED, not bounded: a first AdamW step is lr*sign(g), so entries whose
    gradient is ~0 flip by 2*lr on last-bit differences of g and never meet again."""
    from dl_attack_on_imagenet_amd import engine, zoo
    from oracle import adil_oracle as O
    n, k, T, eps = 32, 10, 20, 8 / 255
    g = torch.Generator().manual_seed(21)
    images = torch.rand(n, 3, 224, 224, generator=g)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    gpu_model = zoo.build_classifier("resnet18", seed=5, device=DEV)
    batches = [list(range(n))]
    do, vo, fo, lo = _oracle_run(O, gpu_model, images, d0, v0, T, eps, batches, dev=DEV)
    dh, vh, fh, lh = _hip_run(engine, gpu_model, images, d0, v0, T, eps, batches)
    dd = (dh - do).abs()
    e_v = float((vh - vo).abs().max())
    e_dv = float((_delta(dh, vh) - _delta(do, vo)).abs().max())
    rel_loss = max(abs(a - b) / max(1.0, abs(a)) for a, b in zip(lo, lh))
    _note("config1_same_backend", dict(fooled_oracle=fo, fooled_hip=fh, max_dV=e_v, max_dDv=e_dv, loss_rel=rel_loss,
                                       dD_max=float(dd.max()), dD_median=float(dd.median()),
                                       frac_dD_gt_1e3=float((dd > 1e-3).float().mean())))
    assert fh == fo                                        # bit-exact label decisions, all 20 iterations
    assert fo[-1] > fo[0]                                  # the attack actually progresses on this workload
    assert rel_loss <= 1e-4
    assert e_v <= 2e-4
    assert e_dv <= 1e-3


def test_config1_cpu_oracle_leg():
    """configs[0] against the CPU oracle (the reference's own CPU-runnable case): the classifier now runs on MKL vs
    MIOpen, so only what the attack is about is asserted — one step from the identical state tight, fooled counts
    within one image per iteration and equal at the end, loss within 2 %."""
    from dl_attack_on_imagenet_amd import engine, zoo
    from oracle import adil_oracle as O
    n, k, T, eps = 32, 10, 20, 8 / 255
    g = torch.Generator().manual_seed(21)
    images = torch.rand(n, 3, 224, 224, generator=g)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    cpu_model = zoo.build_classifier("resnet18", seed=5)
    gpu_model = zoo.build_classifier("resnet18", seed=5, device=DEV)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    batches = [list(range(n))]
    d1o, v1o, _, _ = _oracle_run(O, cpu_model, images, d0, v0, 1, eps, batches)
    d1h, v1h, _, _ = _hip_run(engine, gpu_model, images, d0, v0, 1, eps, batches)
    dd1 = (d1h.cpu() - d1o).abs()
    assert float(dd1.median()) <= 1e-6 and float((dd1 > 1e-4).float().mean()) <= 1e-3
    assert float((v1h.cpu() - v1o).abs().max()) <= 1e-5
    do, vo, fo, lo = _oracle_run(O, cpu_model, images, d0, v0, T, eps, batches)
    dh, vh, fh, lh = _hip_run(engine, gpu_model, images, d0, v0, T, eps, batches)
    dd = (dh.cpu() - do).abs()
    _note("config1_cpu_oracle", dict(fooled_cpu=fo, fooled_hip=fh, max_dV=float((vh.cpu() - vo).abs().max()),
                                     max_dDv=float((_delta(dh.cpu(), vh.cpu()) - _delta(do, vo)).abs().max()),
                                     dD_max=float(dd.max()), frac_dD_gt_1e3=float((dd > 1e-3).float().mean()),
                                     step1_dD_median=float(dd1.median())))
    assert max(abs(a - b) for a, b in zip(fo, fh)) <= 1 and fo[-1] == fh[-1]
    assert max(abs(a - b) for a, b in zip(lo, lh)) <= 2e-2 * max(abs(a) for a in lo)


def test_config2_bf16_fused_resnet50_asr_vs_fp32_oracle():
    """configs[1] path at a size the fp32 oracle finishes in seconds: ResNet-50 through zoo.FusedResNet (stem / pointwise
    / 3x3 kernels), bf16 image streams, 50 atoms, 256 images in batches of 64, 10 epochs = 40 steps — against the fp32
    oracle (plain fp32 ResNet-50, oracle maths) on the same inputs, seeds and batch order.
    bf16 changes the classifier's activations (8 mantissa bits), so label decisions are NOT expected bit-exact here;
    the bound is on what north_star asks of the throughput configuration: attack success rate.  Stated tolerance:
    per-epoch fooling rate within 3 pp, final-epoch ASR within 1.5 pp (256 images: 1 image = 0.39 pp)."""
    from dl_attack_on_imagenet_amd import engine, zoo
    from oracle import adil_oracle as O
    n, bsz, k, epochs, eps = 256, 64, 50, 10, 8 / 255
    g = torch.Generator().manual_seed(33)
    images = torch.rand(n, 3, 224, 224, generator=g)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(n, k, generator=g), eps)
    batches = [list(range(s, s + bsz)) for s in range(0, n, bsz)]
    ref_model = zoo.build_classifier("resnet50", seed=0, device=DEV)                    # fp32, plain PyTorch modules
    fast_model = zoo.build_classifier("resnet50", seed=0, device=DEV, dtype=torch.bfloat16, channels_last=True,
                                      fuse_bn_act=True, fuse_stem=True)
    with torch.no_grad():
        lab32 = ref_model(images.to(DEV)).argmax(-1)
        lab16 = fast_model(images.to(DEV).to(torch.bfloat16)).argmax(-1)
    agree = float((lab32 == lab16).float().mean())
    do, vo, fo, _ = _oracle_run(O, ref_model, images, d0, v0, epochs, eps, batches, dev=DEV)
    dh, vh, fh, _ = _hip_run(engine, fast_model, images, d0, v0, epochs, eps, batches, dtype=torch.bfloat16)
    per = len(batches)
    rate_o = [sum(fo[e * per:(e + 1) * per]) / n for e in range(epochs)]
    rate_h = [sum(fh[e * per:(e + 1) * per]) / n for e in range(epochs)]

    def asr(model, d, v, dtype):                            # performance.py:238-246 on the learned (D, V)
        x = images.to(DEV)
        adv = (x + (v @ d.reshape(-1, k).t()).reshape(x.shape)).to(dtype)
        with torch.no_grad():
            return float((model(adv).argmax(-1) != model(x.to(dtype)).argmax(-1)).float().mean())
    asr_o, asr_h = asr(ref_model, do, vo, torch.float32), asr(fast_model, dh, vh, torch.bfloat16)
    asr_cross = asr(ref_model, dh, vh, torch.float32)       # the bf16-learned dictionary judged by the fp32 network
    _note("config2_bf16_asr", dict(clean_label_agreement=agree, fooling_rate_fp32_oracle=rate_o, fooling_rate_bf16_hip=rate_h,
                                   asr_fp32_oracle=asr_o, asr_bf16_hip=asr_h, asr_bf16_dict_on_fp32_net=asr_cross,
                                   max_dV=float((vh - vo).abs().max())))
    assert rate_o[-1] > rate_o[0] + 0.2                     # a working attack, not a flat line
    assert max(abs(a - b) for a, b in zip(rate_o, rate_h)) <= 0.03
    assert abs(asr_o - asr_h) <= 0.015
    assert abs(asr_o - asr_cross) <= 0.015


@pytest.mark.parametrize("name,k,b", [("densenet121", 50, 16), ("vit_b_16", 100, 16)])
def test_other_classifiers_through_the_learner(name, k, b):
    """configs[2] / configs[4] classifiers (DenseNet-121; ViT-B/16 = 197 tokens) through DictionaryLearner.step.
    fp32: one step from the identical state against the oracle on the same backend (tight), then 3 more steps with
    equal fooled counts.  bf16 streams + bf16 classifier: 4 steps, the invariants of the update (|D| <= 1,
    ||v||_1 <= eps, finite) and fooled counts within 2 images of the fp32 run."""
    from dl_attack_on_imagenet_amd import engine, zoo
    from oracle import adil_oracle as O
    eps = 8 / 255
    g = torch.Generator().manual_seed(5)
    images = torch.rand(b, 3, 224, 224, generator=g)
    d0 = -1 + 2 * torch.rand(3, 224, 224, k, generator=g)
    v0 = O.project_onto_l1_ball(torch.rand(b, k, generator=g), eps)
    model = zoo.build_classifier(name, seed=1, device=DEV)
    batches = [list(range(b))]
    d1o, v1o, f1o, _ = _oracle_run(O, model, images, d0, v0, 1, eps, batches, dev=DEV)
    d1h, v1h, f1h, _ = _hip_run(engine, model, images, d0, v0, 1, eps, batches)
    dd = (d1h - d1o).abs()
    assert f1o == f1h
    assert float(dd.median()) <= 1e-6 and float((dd > 1e-4).float().mean()) <= 1e-3
    assert float((v1h - v1o).abs().max()) <= 1e-5
    _, _, fo, _ = _oracle_run(O, model, images, d0, v0, 4, eps, batches, dev=DEV)
    dh, vh, fh, _ = _hip_run(engine, model, images, d0, v0, 4, eps, batches)
    model16 = zoo.build_classifier(name, seed=1, device=DEV, dtype=torch.bfloat16)
    d16, v16, f16, _ = _hip_run(engine, model16, images, d0, v0, 4, eps, batches, dtype=torch.bfloat16)
    _note(f"learner_{name}", dict(fooled_oracle=fo, fooled_hip=fh, fooled_bf16=f16, step1_dD_median=float(dd.median())))
    assert fo == fh
    assert torch.isfinite(d16).all() and torch.isfinite(v16).all()
    assert float(d16.abs().max()) <= 1.0 and float(v16.abs().sum(1).max()) <= eps * (1 + 1e-5)
    assert max(abs(a - c) for a, c in zip(fh, f16)) <= 2
