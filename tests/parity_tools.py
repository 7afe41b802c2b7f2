"""Shared machinery of the config-level parity tests: ONE classifier evaluation per iteration, handed to both sides.

A teacher-forced step: the HIP learner is put into the oracle's exact state (D, V, both AdamW moment pairs, counters);
both synthesise x + D v[index] (compared first); the classifier runs ONCE, on the product's synthesised batch, and the
same logits / input gradient g go to the oracle's update (oracle.apply_gradient_a) and to the product's
(DictionaryLearner.backward + update_v + update_d).  What is compared is the kernels' arithmetic alone: MIOpen's
run-to-run backward noise — which used to sit inside the comparison because each side called the classifier itself —
is gone.  Three checks per step, each isolating one kernel group:

  synth     x + D v                          HIP vs oracle (fp32: absolute; bf16 streams: in bf16 ulps at the operands' scale)
  gradient  grad_d = g^T V, grad_v = g D     HIP vs oracle on the identical g, relative to the largest entry
  update    AdamW + clamp / + l1 projection  HIP kernels fed the ORACLE's gradient vs the oracle's update: identical inputs,
                                             so any difference is the update kernels' own arithmetic
and the composite (HIP gradient -> HIP update vs oracle gradient -> oracle update), which additionally contains AdamW's
conditioning: the step is step_size * m_hat / (sqrt(v_hat) + 1e-8), so on an entry whose gradient history is below ~1e-6
a difference of 1e-11 between two correctly rounded contractions is multiplied by up to lr / 1e-8 = 1e6.  The composite
is therefore bounded (a) on the well-conditioned entries (sqrt(v_hat) >= 1e-6) by 1e-5 and (b) everywhere by the
first-order amplification bound computed from the measured gradient difference — no fractions, no 2*lr escape."""
import torch


def bf16_scale_ulps(a: torch.Tensor, b: torch.Tensor, *operands: torch.Tensor):
    """max |a - b| in units of one bf16 ulp at the scale of the LARGEST of |a|, |b| and the operands that formed them
    (x + delta can cancel to ~0, where an ulp of the result itself says nothing about the arithmetic).  The binade comes
    from frexp (exact; a device log2 is not correctly rounded at powers of two).  Returns (ratio, description of the
    worst element)."""
    a32, b32 = a.float(), b.float()
    scale = torch.maximum(a32.abs(), b32.abs())
    for o in operands:
        scale = torch.maximum(scale, o.float().abs())
    _, e = torch.frexp(scale.clamp_min(2.0 ** -120))              # scale = m * 2^e, m in [0.5, 1)
    ulp = torch.ldexp(torch.ones_like(scale), e - 8)              # bf16: 8 significant bits -> spacing 2^(e-1-7)
    ratio = (a32 - b32).abs() / ulp
    i = int(ratio.flatten().argmax())
    worst = dict(index=i, ulps=float(ratio.flatten()[i]), a=float(a32.flatten()[i]), b=float(b32.flatten()[i]),
                 operands=[float(o.float().flatten()[i]) for o in operands])
    return worst["ulps"], worst


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
        synth_err, synth_worst = bf16_scale_ulps(xt_h, xt_o.to(torch.bfloat16), x_stream)
        synth_worst["oracle_before_rounding"] = float(xt_o.flatten()[synth_worst["index"]])
    else:
        synth_err = float((xt_h - xt_o).abs().max())
    # (2) the classifier, ONCE (on the product's batch); the label decisions on the oracle's own batch, forward only
    out, ls, g = engine.input_gradient(model, xt_h, labels, loss, -1.0, kappa, "sum")
    fooled = int((out.argmax(-1) != labels).sum())
    fooled_o = int((engine.predict(model, xt_o.to(x_stream.dtype)) != labels).sum())
    # (3) the gradient contractions on the SAME g
    gd_o, gv_o = O.grad_dv(g.float(), dop, vop)
    gd_h, gvb = learner.backward(g, codes)
    gv_h = ops.pack_codes(gvb, None, b)[:b, :k] if isinstance(gvb, ops.SlabGrad) else gvb
    e_gd = float((gd_h - gd_o).abs().max() / gd_o.abs().max().clamp_min(1e-30))
    e_gv = float((gv_h - gv_o).abs().max() / gv_o.abs().max().clamp_min(1e-30))
    abs_gd = float((gd_h - gd_o).abs().max())
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
        out["synth_worst_element"] = max((r["synth_worst"] for r in records), key=lambda w: w["ulps"])
    out["frac_well_conditioned_min"] = min(r["frac_well_conditioned"] for r in records)
    return out
