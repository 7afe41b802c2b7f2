"""Shared machinery of the config-level parity tests: ONE classifier evaluation per iteration, handed to both sides.

A teacher-forced step: the HIP learner is put into the oracle's exact state (D, V, both AdamW moment pairs, counters);
both synthesise x + D v[index] (compared first); the classifier runs ONCE, on the product's synthesised batch, and the
same logits / input gradient g go to the oracle's update (oracle.apply_gradient_a) and to the product's
(DictionaryLearner.backward + update_v + update_d).  What is compared is the kernels' arithmetic alone: MIOpen's
run-to-run backward noise — which used to sit inside the comparison because each side called the classifier itself —
is gone.  Three checks per step, each isolating one kernel group:

  synth     x + D v                          HIP vs oracle (fp32: absolute; bf16 streams: one bf16 ulp of the result + fp32 accumulation noise)
  gradient  grad_d = g^T V, grad_v = g D     HIP vs the fp64 evaluation of the oracle's contraction on the identical g,
                                             relative to the largest entry
  update    AdamW + clamp / + l1 projection  HIP kernels fed the ORACLE's gradient vs the oracle's update: identical inputs,
                                             so any difference is the update kernels' own arithmetic
and the composite (HIP gradient -> HIP update vs oracle gradient -> oracle update), which additionally contains AdamW's
conditioning: the step is step_size * m_hat / (sqrt(v_hat) + 1e-8), so on an entry whose gradient history is below ~1e-6
a difference of 1e-11 between two correctly rounded contractions is multiplied by up to lr / 1e-8 = 1e6.  The composite
is therefore bounded (a) on the well-conditioned entries (sqrt(v_hat) >= 1e-6) by 1e-5 and (b) everywhere by the
first-order amplification bound computed from the measured gradient difference — no fractions, no 2*lr escape."""
import torch


def bf16_result_error(a: torch.Tensor, b: torch.Tensor, x: torch.Tensor, abs_terms: torch.Tensor):
    """How far two bf16 results a, b of x + sum_k v_k D_k are apart, in units of what two CORRECT evaluations may differ by:
    one bf16 ulp of the result (an fp32 sum that lands on the other side of a rounding boundary) plus the fp32
    accumulation noise 2^-20 * (|x| + sum_k |v_k D_k|) (the order of the K + 1 additions; it dominates where the sum
    cancels to ~0, e.g. a clamped pixel x = 0 with a perturbation of 3e-8 formed from terms of 1e-4).  The binade comes
    from frexp (exact; a device log2 is not correctly rounded at powers of two).  Returns (max ratio, worst element)."""
    a32, b32 = a.float(), b.float()
    mag = torch.maximum(a32.abs(), b32.abs()).clamp_min(2.0 ** -120)
    _, e = torch.frexp(mag)                                         # mag = m * 2^e, m in [0.5, 1)
    ulp = torch.ldexp(torch.ones_like(mag), e - 8)                  # bf16: 8 significant bits -> spacing 2^(e-1-7)
    allowed = ulp + 2.0 ** -20 * (x.float().abs() + abs_terms)
    ratio = (a32 - b32).abs() / allowed
    i = int(ratio.flatten().argmax())
    worst = dict(index=i, ratio=float(ratio.flatten()[i]), a=float(a32.flatten()[i]), b=float(b32.flatten()[i]),
                 x=float(x.float().flatten()[i]), sum_abs_terms=float(abs_terms.flatten()[i]))
    return worst["ratio"], worst


def force_state(learner, d, v, sd, sv):
    learner.d.copy_(d); learner.v.copy_(v)
    learner.m_d.copy_(sd.m); learner.s_d.copy_(sd.v); learner.m_v.copy_(sv.m); learner.s_v.copy_(sv.v)
    learner.sched_d.t, learner.sched_v.t = sd.t, sv.t


def shared_gradient_step(O, engine, model, learner, twin, x_stream, index, labels, d, v, sd, sv, eps, loss, kappa=50.0):
    """One teacher-forced step with a shared classifier evaluation.  (d, v, sd, sv) is the oracle's state (updated in
    place); `learner` and `twin` (two DictionaryLearners) are forced into it first: `learner` takes the product's whole
    step, `twin` applies the product's update kernels to the ORACLE's gradient.  x_stream: the batch in the product's
    stream dtype; the oracle sees the same values widened to fp32 and, for bf16 streams, the contraction operands rounded
    to bf16 exactly as the kernels round them (D and the batch's code rows), with x + D v rounded once to bf16.
    Returns the step's differences."""
    ops = engine.ops
    bf16 = x_stream.dtype == torch.bfloat16
    b, k = x_stream.shape[0], d.shape[-1]
    force_state(learner, d, v, sd, sv)
    force_state(twin, d, v, sd, sv)
    rows = v[index]
    dop = d.to(torch.bfloat16).float() if bf16 else d
    vop = rows.to(torch.bfloat16).float() if bf16 else rows
    # (1) synthesis
    xt_o = O.synth(x_stream.float(), dop, vop)
    xt_h, codes = learner.synthesize(x_stream, index)
    synth_worst = None
    if bf16:
        abs_terms = O.synth(torch.zeros_like(xt_o), dop.abs(), vop.abs())      # sum_k |v_k D_k| per pixel
        synth_err, synth_worst = bf16_result_error(xt_h, xt_o.to(torch.bfloat16), x_stream, abs_terms)
        del abs_terms
    else:
        synth_err = float((xt_h - xt_o).abs().max())
    # (2) the classifier, ONCE (on the product's batch); the label decisions on the oracle's own batch, forward only
    out, ls, g = engine.input_gradient(model, xt_h, labels, loss, -1.0, kappa, "sum")
    fooled = int((out.argmax(-1) != labels).sum())
    with torch.no_grad():
        out_o = engine._classify(model, xt_o.to(x_stream.dtype)).float()
    fooled_o = int((out_o.argmax(-1) != labels).sum())
    # images on which the two synthesised batches (equal up to the bound above) are classified differently, and how close
    # to the decision boundary they sit: top-2 margin, the smaller of the two batches' values
    differ = out.float().argmax(-1) != out_o.argmax(-1)
    n_differ, differ_margin = int(differ.sum()), 0.0
    if n_differ:
        t_h, t_o = out.float().topk(2, dim=1).values, out_o.topk(2, dim=1).values
        differ_margin = float(torch.minimum(t_h[:, 0] - t_h[:, 1], t_o[:, 0] - t_o[:, 1])[differ].max())
    # (3) the gradient contractions on the SAME g
    gd_o, gv_o = O.grad_dv(g.float(), dop, vop)                    # the oracle's own fp32 contraction (feeds its update)
    gd_x, gv_x = O.grad_dv(g.double(), dop.double(), vop.double())  # ... and the exact one the kernels are measured against
    gd_h, gvb = learner.backward(g, codes)
    gv_h = ops.pack_codes(gvb, None, b)[:b, :k] if isinstance(gvb, ops.SlabGrad) else gvb
    e_gd = float((gd_h.double() - gd_x).abs().max() / gd_x.abs().max().clamp_min(1e-30))
    e_gv = float((gv_h.double() - gv_x).abs().max() / gv_x.abs().max().clamp_min(1e-30))
    abs_gd = float((gd_h - gd_o).abs().max())
    del gd_x, gv_x
    # (4) the update kernels on the ORACLE's gradient (twin) and the product's full step (learner)
    twin.synthesize(x_stream, index)                               # fills the batch-slot table update_v consumes
    twin.update_v(gv_o.contiguous())
    twin.update_d(gd_o.contiguous())
    learner.update_v(gvb)
    learner.update_d(gd_h)
    # (5) the oracle's update from the same g
    O.apply_gradient_a(g.float(), index, d, v, sd, sv, eps, d_operand=dop, v_operand=vop)
    bc2 = 1.0 - sd.b2 ** sd.t
    vhat = sd.v.sqrt() / bc2 ** 0.5
    well = vhat >= 1e-6
    dd = (learner.d - d).abs()
    amplification = sd.lr / sd.eps          # |d(step)/d(grad)| at its largest: step = lr g / (|g| + eps) at t = 1, |g| << eps
    return dict(synth=synth_err, synth_worst=synth_worst, fooled=fooled, fooled_on_oracle_synth=fooled_o, loss=float(ls),
                label_decisions_differ=n_differ, differing_margin=differ_margin,
                grad_d_rel=e_gd, grad_v_rel=e_gv,
                update_dD=float((twin.d - d).abs().max()), update_dV=float((twin.v - v).abs().max()),
                dV=float((learner.v - v).abs().max()), dD=float(dd.max()),
                dD_well_conditioned=float(dd[well].max()) if bool(well.any()) else 0.0,
                dD_bound=4.0 * amplification * abs_gd + 1e-7, frac_well_conditioned=float(well.float().mean()))


def worst_of(records, skip=("fooled", "fooled_on_oracle_synth", "loss", "frac_well_conditioned", "synth_worst")):
    out = {}
    for r in records:
        for key, val in r.items():
            if key not in skip:
                out[key] = max(out.get(key, 0.0), val)
    if records and records[0].get("synth_worst") is not None:
        out["synth_worst_element"] = max((r["synth_worst"] for r in records), key=lambda w: w["ratio"])
    out["frac_well_conditioned_min"] = min(r["frac_well_conditioned"] for r in records)
    out["contraction_reference"] = "fp64"          # (marker for tests/parity_report.py: earlier runs compared with the fp32 GEMM)
    return out
