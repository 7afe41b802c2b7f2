"""Shared machinery of the config-level parity tests: ONE classifier evaluation per iteration, handed to both sides.

A teacher-forced step: the HIP learner is put into the oracle's exact state (D, V, both AdamW moment pairs, counters);
both synthesise x + D v[index] (compared first); the classifier runs ONCE, on the product's synthesised batch, and the
same logits / input gradient g go to the oracle's update (oracle.apply_gradient_a) and to the product's
(DictionaryLearner.backward + update_v + update_d): what is compared is the kernels' arithmetic alone — MIOpen's
run-to-run backward noise, which AdamW turns into visible differences of single near-zero-gradient entries, is not in
the comparison any more."""
import torch


def bf16_ulp_distance(a: torch.Tensor, b: torch.Tensor) -> int:
    """Largest distance, in units in the last place, between two bf16 tensors (finite values)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7fff), i)                   # sign-magnitude -> monotonic integers (+-0 coincide)
    return int((key(a) - key(b)).abs().max())


def force_state(learner, d, v, sd, sv):
    learner.d.copy_(d); learner.v.copy_(v)
    learner.m_d.copy_(sd.m); learner.s_d.copy_(sd.v); learner.m_v.copy_(sv.m); learner.s_v.copy_(sv.v)
    learner.sched_d.t, learner.sched_v.t = sd.t, sv.t


def shared_gradient_step(O, engine, model, learner, x_stream, index, labels, d, v, sd, sv, eps, loss, kappa=50.0):
    """One teacher-forced step with a shared classifier evaluation.  (d, v, sd, sv) is the oracle's state (updated in
    place); `learner` is forced into it first.  x_stream: the batch in the product's stream dtype (fp32 or bf16); the
    oracle sees the same values widened to fp32 and, for bf16 streams, the contraction operands rounded to bf16 exactly
    as the kernels round them (D and the batch's code rows), with x + D v rounded once to bf16.
    Returns a dict of the step's differences."""
    bf16 = x_stream.dtype == torch.bfloat16
    force_state(learner, d, v, sd, sv)
    rows = v[index]
    dop = d.to(torch.bfloat16).float() if bf16 else None
    vop = rows.to(torch.bfloat16).float() if bf16 else None
    # (1) synthesis
    xt_o = O.synth(x_stream.float(), d if dop is None else dop, rows if vop is None else vop)
    xt_h, codes = learner.synthesize(x_stream, index)
    if bf16:
        synth_err = bf16_ulp_distance(xt_h, xt_o.to(torch.bfloat16))            # in bf16 ulps
    else:
        synth_err = float((xt_h - xt_o).abs().max())
    # (2) the classifier, ONCE; plus the label decisions on the oracle's own synthesised batch (forward only)
    out, ls, g = engine.input_gradient(model, xt_h, labels, loss, -1.0, kappa, "sum")
    fooled = int((out.argmax(-1) != labels).sum())
    fooled_o = int((engine.predict(model, xt_o.to(x_stream.dtype)) != labels).sum())
    # (3) both updates from the SAME g
    O.apply_gradient_a(g.float(), index, d, v, sd, sv, eps, d_operand=dop, v_operand=vop)
    gd, gvb = learner.backward(g, codes)
    learner.update_v(gvb)
    learner.update_d(gd)
    return dict(synth=synth_err, fooled=fooled, fooled_on_oracle_synth=fooled_o, loss=float(ls),
                dD=float((learner.d - d).abs().max()), dV=float((learner.v - v).abs().max()),
                dmD=float((learner.m_d - sd.m).abs().max()), dsD=float((learner.s_d - sd.v).abs().max()),
                dmV=float((learner.m_v - sv.m).abs().max()), dsV=float((learner.s_v - sv.v).abs().max()))
