"""ADiL solvers on the HIP kernels: the reference's hot loops with the PyTorch-op
sequences (tensordot + autograd + optim.AdamW + sort-based projection) replaced
by fused kernels.  The frozen classifier's forward/backward stays in PyTorch-ROCm.

Reference loops (file:line in flavie-yuan-liu/DL_attack_on_ImageNet):
  DictionaryLearner.step          learn_dictionary_a hot loop      adil.py:168-191
  DictionaryLearner.step_codes/d  learn_dictionary_b V / D steps   adil.py:268-311
  solve_codes_adamw               forward_supervised_AdamW         adil.py:569-623
  solve_ddrague                   forward_supervised_DDrague       adil.py:508-567
  attack_unsupervised             forward_unsupervised             adil.py:460-506
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import ops
from .dist import DictGradReducer

Tensor = torch.Tensor
STOP_POLL = 4        # the solvers look at the device-side stop flag (one host sync) every STOP_POLL iterations


# --------------------------------------------------------------------------- #
# losses on the classifier logits (tiny B x n_classes work, stays in torch)
# --------------------------------------------------------------------------- #
def margin_loss(outputs: Tensor, labels: Tensor, kappa: float, targeted: bool = False) -> Tensor:
    """CW-style margin of ADIL.f_loss (adil.py:103-112).  The label logit is zeroed (not masked to -inf) before the
    max — reference quirk Q5 — and the reference always evaluates the untargeted branch (`self._targeted`)."""
    col = labels.unsqueeze(1)
    other = outputs.scatter(1, col, 0.0).max(dim=1).values
    own = outputs.gather(1, col).squeeze(1)
    return (other - own).clamp(min=-kappa) if targeted else (own - other).clamp(min=-kappa)


def attack_loss(outputs: Tensor, labels: Tensor, loss: str, coeff: float, kappa: float, ce_reduction: str) -> Tensor:
    outputs = outputs.float()
    if loss == "ce":
        return coeff * F.cross_entropy(outputs, labels, reduction=ce_reduction)
    if loss == "logits":
        return margin_loss(outputs, labels, kappa).sum()
    raise ValueError(f"unknown loss {loss!r} (expected 'ce' or 'logits')")


# --------------------------------------------------------------------------- #
# the frozen classifier, called at a few batch sizes only
# --------------------------------------------------------------------------- #
_BATCH_BUCKET = 0


class classifier_batch_bucket:
    """Round the batch the frozen classifier sees up to a multiple of `multiple` (rows of zeros appended, their logits
    and input gradients dropped), so that it runs at a handful of batch sizes instead of every size a caller produces.

    Why: on this stack MIOpen goes through its find / compile step for every convolution configuration it has not
    seen, i.e. for every NEW batch size: 0.6-1.0 s (MobileNetV2, fp32) to 2.2-3.2 s (ResNet-50, bf16) for the first
    forward + backward at a size against 5-8 ms afterwards (tools/exp_ragged_batches.py, FINDINGS.md 23) — and
    performance.py's correctly-classified filter (performance.py:163-165) hands the attack a different number of
    images per batch.  The attack's own arithmetic is untouched: the loss, its mean, the stop test and every ADiL
    kernel see the real rows only; an eval-mode classifier treats the rows of a batch independently.
    Use as a context manager (`with classifier_batch_bucket(20): ...`) or call `.set()` / `.clear()`; 0 / 1 = off."""

    def __init__(self, multiple: int):
        self.multiple, self._prev = int(multiple), 0

    def set(self):
        global _BATCH_BUCKET
        self._prev, _BATCH_BUCKET = _BATCH_BUCKET, self.multiple
        return self

    def clear(self):
        global _BATCH_BUCKET
        _BATCH_BUCKET = self._prev

    __enter__ = set

    def __exit__(self, *exc):
        self.clear()


def _padded_rows(x: Tensor) -> Tensor:
    n = x.shape[0]
    if _BATCH_BUCKET <= 1 or n == 0 or n % _BATCH_BUCKET == 0:
        return x
    pad = x.new_zeros((_BATCH_BUCKET - n % _BATCH_BUCKET,) + tuple(x.shape[1:]))
    return torch.cat([x, pad])


def _classify(model, x: Tensor) -> Tensor:
    """Logits of the rows of x (the classifier itself may see zero rows appended, see classifier_batch_bucket)."""
    xp = _padded_rows(x)
    out = model(xp)
    return out if xp is x else out[:x.shape[0]]


class precise_head:
    """Context manager: while it is in force, classifiers that carry an optional higher-precision head (zoo.FusedResNet built
    with head_fp32="inference") use it.  The DDrague inference solver wraps its classifier calls in it; any other model is
    left alone."""

    def __init__(self, model, enabled: bool = True):
        self.parts = [m for m in model.modules() if hasattr(m, "precise_head")] if hasattr(model, "modules") else []
        self.enabled, self.prev = enabled, []

    def __enter__(self):
        self.prev = [m.precise_head(self.enabled) for m in self.parts]
        return self

    def __exit__(self, *exc):
        for m, p in zip(self.parts, self.prev):
            m.precise_head(p)


@torch.no_grad()
def predict(model, x: Tensor) -> Tensor:
    return _classify(model, x).argmax(dim=-1)


def input_gradient(model, xt: Tensor, labels: Tensor, loss: str, coeff: float, kappa: float,
                   ce_reduction: str) -> Tuple[Tensor, Tensor, Tensor]:
    """One classifier forward + backward at xt.  Only dLoss/dxt is requested, so autograd skips the frozen
    classifier's weight gradients (the reference computes and discards them, quirk Q8)."""
    xt = xt.detach().requires_grad_(True)
    with torch.enable_grad():
        out = _classify(model, xt)
        ls = attack_loss(out, labels, loss, coeff, kappa, ce_reduction)
        (g,) = torch.autograd.grad(ls, xt)
    return out.detach(), ls.detach(), g.contiguous()


def _flat_images(x: Tensor) -> Tensor:
    if x.dim() != 4:
        raise ValueError(f"images must be (B,C,H,W), got {tuple(x.shape)}")
    return x.contiguous()


class LabelCache:
    """Clean pseudo-labels per image of a resident dataset, computed on the first visit and reused afterwards.

    The learners of the reference recompute `model(x).argmax` for the same clean images in every epoch (adil.py:172,
    268, 295).  The classifier is frozen and in eval mode, so the label is a constant of the image: after the first
    epoch the second forward of every step is pure recomputation (a third of the step's classifier time).  The one way the
    cached value could differ from the reference's — a library rounding an image's logits differently in a different
    batch and flipping a near-tie — was measured (round 3, tools/exp_label_stability.py, profiles/r03_label_stability.md):
    10 000 images re-labelled in 3 shuffled batch orders incl. a ragged last batch, bf16 FusedResNet-50 and plain fp32
    ResNet-50: 0 labels changed.  So it is the DEFAULT of the learners (`ADIL(cache_labels=True)`); cache_labels=False
    is the reference's op sequence.  Which rows are known is tracked on the host (the batch order is
    host data), so a lookup costs no synchronisation."""

    def __init__(self, n: int, device):
        self.labels = torch.full((n,), -1, dtype=torch.int64, device=device)
        self._known = [False] * n

    def get(self, model, x: Tensor, index: Tensor, rows) -> Tensor:
        """Labels of the batch `x` = resident rows `rows` (host ints; `index` is the same on the device)."""
        rows = [int(r) for r in rows]
        if all(self._known[r] for r in rows):
            return self.labels[index]
        lab = predict(model, x)
        self.labels[index] = lab
        for r in rows:
            self._known[r] = True
        return lab


# --------------------------------------------------------------------------- #
class DictionaryLearner:
    """State + fused update of the learnable pair (D, V) of Attack_dict_model (adil.py:16-35).

    d: (C,H,W,K) fp32, v: (N,K) fp32 (updated in place), each with AdamW moments.
    `lr_d`/`lr_v` default to the single-optimiser setting of learn_dictionary_a (adil.py:154).
    With a DictGradReducer the rows of v are the LOCAL shard and grad_d is summed over ranks
    (one all-reduce per step) before the identical AdamW update on every rank."""

    def __init__(self, d: Tensor, v: Tensor, eps: float, step_size: float = 0.01, loss: str = "ce",
                 targeted: bool = False, kappa: float = 50.0, lr_d: Optional[float] = None,
                 lr_v: Optional[float] = None, reducer: Optional[DictGradReducer] = None, fp8_synth: bool = False):
        self.d = ops._dev(d, "d", torch.float32)
        self.v = ops._dev(v, "v", torch.float32)
        self.eps, self.loss, self.kappa = float(eps), loss, float(kappa)
        self.coeff = 1.0 if targeted else -1.0
        self.m_d, self.s_d = torch.zeros_like(d), torch.zeros_like(d)
        self.m_v, self.s_v = torch.zeros_like(v), torch.zeros_like(v)
        self.sched_d = ops.AdamWSchedule(step_size if lr_d is None else lr_d)
        self.sched_v = ops.AdamWSchedule(step_size if lr_v is None else lr_v)
        self.pos = torch.full((v.shape[0],), -1, dtype=torch.int32, device=v.device)
        self.grad_d = torch.empty_like(d)
        self.reducer = reducer
        # configs[4]: the D.V contraction of the synthesis on fp8 MFMAs.  Legal here because every row of v lives in
        # the l1 ball of radius eps after update_v (so |v| <= eps bounds the code scale) and |d| <= 1 after update_d.
        self.fp8_absmax = float(eps) if fp8_synth else None
        # ... and on a PERSISTENT fp8 copy of D (round 4): the AdamW + clamp launch keeps it current (one more byte per
        # element written), the synthesis reads it instead of the fp32 master (a quarter of the dictionary bytes)
        self.d_fp8 = ops.dict_to_fp8(self.d) if (fp8_synth and ops.fp8_dict_supported(self.d)) else None
        self._graph = None                       # (graph, x, index, loss, fooled, batch size, labels) once `use_graph` captured a step
        self._graph_warm = 0
        self._dyn_d = self._dyn_v = None
        self._pending = None                     # handle of the step's all-reduce between forward_backward and update_d

    def sync_fp8_copy(self) -> None:
        """Re-derive the persistent fp8 copy from the fp32 master — after `d` was overwritten from outside (a warm start, a
        test forcing a state); update_d keeps it current by itself."""
        if self.d_fp8 is not None:
            ops.dict_to_fp8(self.d, out=self.d_fp8)

    # -- pieces ------------------------------------------------------------- #
    def synthesize(self, x: Tensor, index: Tensor, want_d: bool = True, want_v: bool = True):
        """K1: x + D v[index] (adil.py:25-26).  Returns (xt, codes) with `codes` what backward() needs: the gather of
        the batch's code rows also records their batch slots in `pos` (consumed + reset by update_v) and, for a D-step,
        writes the transposed copy in the stream dtype that the grad_d contraction reads."""
        b = x.shape[0]
        vp = ops.pack_codes(self.v, index, b, pos=self.pos if want_v else None, transposed=x.dtype if want_d else None)
        vpt = None
        if want_d:
            vp, vpt = vp
        xt = ops.synth(_flat_images(x), self.d, vp, b, fp8_absmax=self.fp8_absmax, d_fp8=self.d_fp8)
        return xt, (vp, vpt, b)

    def backward(self, g: Tensor, codes, want_d: bool = True, want_v: bool = True):
        """K2 + K3 for the upstream gradient g = dLoss/d(x + D v): one pass over g (adil.py:185 through the tensordot).
        grad_v stays in the kernel's per-workgroup partial sums, which update_v sums itself; with a reducer the step's ONE
        collective is started here and waited for in update_d: the update of the code rows, which does not depend on the
        reduced gradient, runs while RCCL moves grad_d over xGMI on its own stream."""
        vp, vpt, b = codes
        gd, gvb = ops.grad(g, self.d, vp, b, want_d=want_d, want_v=want_v, grad_d=self.grad_d if want_d else None,
                           vpt=vpt, defer_v=True)
        self._pending = None
        if want_d and self.reducer is not None:
            self._pending = self.reducer.all_reduce_start(gd)
        return gd, gvb

    def forward_backward(self, model, x: Tensor, index: Tensor, labels: Tensor, want_d: bool, want_v: bool):
        if x.shape[0] == 0:
            return self._empty_batch(x, want_d)
        xt, codes = self.synthesize(x, index, want_d, want_v)                            # K1
        out, ls, g = input_gradient(model, xt, labels, self.loss, self.coeff, self.kappa, "sum")
        fooled = (out.argmax(dim=-1) != labels).sum()                                    # adil.py:177
        gd, gvb = self.backward(g, codes, want_d, want_v)                                # K2 + K3
        return ls, fooled, gd, gvb

    def _empty_batch(self, x: Tensor, want_d: bool):
        """This rank owns no image of the current global batch (ragged shards): it contributes a zero grad_d and still
        joins the step's all-reduce, so every rank issues the same collectives in the same order."""
        zero = torch.zeros((), dtype=torch.float32, device=self.d.device)
        gd = None
        self._pending = None
        if want_d:
            gd = self.grad_d.zero_()
            if self.reducer is not None:
                self._pending = self.reducer.all_reduce_start(gd)
        return zero, zero.to(torch.int64), gd, None

    @staticmethod
    def _next_scalars(sched, dyn):
        """Advance the AdamW step counter.  With device-resident scalars (graphs) the two step-dependent values are copied
        to the device first — except while a graph is being captured: the copy must stay OUTSIDE the graph (a recorded
        host-to-device copy would re-read one fixed pinned slot on every replay); step_graphed issues it before replay."""
        if dyn is None or torch.cuda.is_current_stream_capturing():
            return sched.next()
        return sched.next_to_device()

    def update_d(self, gd: Tensor) -> None:
        if self._pending is not None:
            self._pending.wait()                                 # stream-ordered: the host does not block
            self._pending = None
        h = self._next_scalars(self.sched_d, self._dyn_d)
        ops.adamw_clamp_(self.d, gd, self.m_d, self.s_d, h, -1.0, 1.0, dyn=self._dyn_d, p_fp8=self.d_fp8)    # K4: step + update_d

    def update_v(self, gvb: Optional[Tensor]) -> None:
        if self.v.shape[0] == 0:
            return
        h = self._next_scalars(self.sched_v, self._dyn_v)
        ops.adamw_l1ball_(self.v, gvb, self.pos, self.m_v, self.s_v, h, self.eps, reset_pos=True,      # K5
                          dyn=self._dyn_v)

    # -- the whole step as ONE hipGraph launch (launch-bound configurations) ------------------------------------ #
    def step_graphed(self, model, x: Tensor, index: Tensor, labels: Optional[Tensor] = None):
        """`step()` replayed from a hipGraph: the ~200 launches of a step (classifier forward twice, backward, the five
        ADiL kernels) become one graph launch — for launch-bound uses (small crops, tiny classifiers; configs[0] itself,
        resnet18 on 32 images of 224x224, turned out GPU-bound: 7.2 ms either way).  The first two calls run eagerly (library autotuning must not
        happen under capture), the third captures the step for this batch size and replays it; later calls copy x / index
        into the graph's static inputs and replay.  AdamW's step-dependent scalars reach the recorded launches through
        device memory (`dyn_scalars`).  `labels`: the cached clean pseudo-labels of the batch (engine.LabelCache); the
        recording then holds one classifier forward less.  A different batch size, a change between given and recomputed
        labels, a reducer (the collective is not captured) or an empty batch falls back to the eager step.  Results are
        bit-identical to `step()` (tests/test_gpu_adil.py)."""
        index = index.to(device=self.v.device, dtype=torch.int64)
        b = x.shape[0]
        if self.reducer is not None or b == 0 or (self._graph is not None and
                                                  (self._graph[5] != b or (self._graph[6] is None) != (labels is None))):
            return self.step(model, x, index, labels)
        if self._graph is None:
            if self._graph_warm < 2:
                self._graph_warm += 1
                return self.step(model, x, index, labels)
            self._dyn_d = self.sched_d.enable_device_scalars(self.d.device)
            self._dyn_v = self.sched_v.enable_device_scalars(self.v.device)
            gx, gi = x.clone(), index.clone()
            gl = labels.clone() if labels is not None else None
            t_d, t_v = self.sched_d.t, self.sched_v.t
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                ls, fooled = self.step(model, gx, gi, gl)
            self.sched_d.t, self.sched_v.t = t_d, t_v            # capture records, it does not execute: no step was taken
            self._graph = (graph, gx, gi, ls, fooled, b, gl)
        graph, gx, gi, ls, fooled, _, gl = self._graph
        gx.copy_(x)
        gi.copy_(index)
        if gl is not None:
            gl.copy_(labels)
        self.sched_d.next_to_device()                            # this step's scalars, ordered before the replay
        self.sched_v.next_to_device()
        graph.replay()
        return ls.clone(), fooled.clone()

    # -- reference loops ---------------------------------------------------- #
    def step(self, model, x: Tensor, index: Tensor, labels: Optional[Tensor] = None):
        """learn_dictionary_a hot-loop body (adil.py:168-191). Returns (loss, #fooled) as 0-d device tensors."""
        index = index.to(device=self.v.device, dtype=torch.int64)
        if labels is None and x.shape[0]:
            labels = predict(model, x)                                                   # adil.py:172
        ls, fooled, gd, gvb = self.forward_backward(model, x, index, labels, True, True)
        self.update_v(gvb)                                       # overlaps with the all-reduce of grad_d (if any)
        self.update_d(gd)
        return ls, fooled

    def step_codes(self, model, x: Tensor, index: Tensor, labels: Optional[Tensor] = None):
        """V-step of learn_dictionary_b (adil.py:268-287)."""
        index = index.to(device=self.v.device, dtype=torch.int64)
        if labels is None:
            labels = predict(model, x)
        ls, fooled, _, gvb = self.forward_backward(model, x, index, labels, False, True)
        self.update_v(gvb)
        return ls, fooled

    def step_dictionary(self, model, x: Tensor, index: Tensor, labels: Optional[Tensor] = None):
        """D-step of learn_dictionary_b (adil.py:295-311)."""
        index = index.to(device=self.v.device, dtype=torch.int64)
        if labels is None:
            labels = predict(model, x)
        ls, fooled, gd, _ = self.forward_backward(model, x, index, labels, True, False)
        self.update_d(gd)
        return ls, fooled


# --------------------------------------------------------------------------- #
def solve_codes_adamw(model, images: Tensor, d: Tensor, eps: float, loss: str = "ce", targeted: bool = False,
                      kappa: float = 50.0, norm: str = "linf", mode: str = "train", max_iter: int = 100,
                      labels: Optional[Tensor] = None, return_codes: bool = False, mean_over: Optional[int] = None,
                      reducer: Optional[DictGradReducer] = None):
    """forward_supervised_AdamW (adil.py:569-623): per-image codes with D fixed.
    mode 'train' -> fooled count (0-d tensor); otherwise clamp(images + D proj(v), 0, 1).
    mean_over: size of the GLOBAL batch when `images` is one rank's shard of it — the reference's mean-reduced CE
    (adil.py:578) then divides by that size, so a sharded validation solves the same problem as an unsharded one.
    reducer (with mean_over): the stop test of adil.py:614 is on max|dv| of the WHOLE batch, so the shards' stop slots
    are max-reduced over the ranks after every iteration (one float; the kernels of the next iteration read the global
    value) — every rank, also one that owns no image of this batch, runs the same number of iterations and issues
    the same collectives."""
    images = _flat_images(images)
    b = images.shape[0]
    p, k = ops.dict_shape(d)
    coeff = 1.0 if targeted else -1.0
    ce_reduction = "mean"
    if mean_over is not None and mean_over != b:
        coeff, ce_reduction = coeff / float(mean_over), "sum"
    if b == 0:
        if reducer is not None:
            _idle_rank_stop_loop(images.device, reducer, int(max_iter))
        if mode == "train":
            return torch.zeros((), dtype=torch.int64, device=images.device)
        return images.clone()
    v = torch.zeros(b, k, dtype=torch.float32, device=images.device)
    m, s = torch.zeros_like(v), torch.zeros_like(v)
    sched = ops.AdamWSchedule(1e-2)
    if labels is None:
        labels = predict(model, images)                                                  # adil.py:598 (constant)
    stop = ops.StopTest(images.device, 1e-6)                                             # adil.py:614, on the device
    iters = 0
    for it in range(int(max_iter)):
        iters += 1
        vp = ops.pack_codes(v, None, b)
        xt = ops.synth(images, d, vp, b)
        _, _, g = input_gradient(model, xt, labels, loss, coeff, kappa, ce_reduction)
        _, gvb = ops.grad(g, d, None, b, want_d=False, want_v=True, defer_v=True)
        ops.adamw_l1ball_(v, gvb, None, m, s, sched.next(), eps, stop=stop)                # adil.py:609-610
        if reducer is not None:
            reducer.max_(stop.last_slot())                     # the batch's max|dv|, not the shard's
        if (it + 1) % STOP_POLL == 0 and stop.converged():     # launches after the converged one are no-ops
            break
    vproj = v.clone()
    if norm == "l2":
        ops.l2ball_project_(vproj, eps)                                                  # adil.py:617 -> :626-629
    else:
        ops.l1ball_project_(vproj, eps)
    vp = ops.pack_codes(vproj, None, b)
    if mode == "train":
        xt = ops.synth(images, d, vp, b)
        res = (predict(model, xt) != labels).sum()                                       # adil.py:619-620
    else:
        res = ops.synth(images, d, vp, b, pixel_clamp=True)                              # adil.py:622-623
    if return_codes:
        return res, dict(v=v, iters=iters, labels=labels)
    return res


def _idle_rank_stop_loop(device, reducer: DictGradReducer, max_iter: int) -> None:
    """A rank that owns no image of a sharded validation batch: it has nothing to solve but must pair up with the other
    ranks' per-iteration max-reduction of the stop slot, and leave the loop at the same poll point as they do."""
    stop = ops.StopTest(device, 1e-6)
    for it in range(max_iter):
        stop.idle_iteration()
        reducer.max_(stop.last_slot())
        if (it + 1) % STOP_POLL == 0 and stop.converged():
            break


class PseudoInverse:
    """D_dagger^T = D (DtD)^-1^T stored P x K like D (adil.py:523-525).  The K x K inverse is a tiny
    dense solve and stays in torch; the two P-sized contractions are HIP kernels."""

    def __init__(self, d: Tensor):
        self.d = ops._dev(d, "d", torch.float32)
        self.gram = ops.gram(d)
        self.gram_inv = ops.spd_inverse(self.gram)         # `dtd.inverse()` (adil.py:524) on the device, fp64 inside
        self.d_pinv_t = ops.dict_rightmul(d, self.gram_inv)


class DDragueSolver:
    """State of forward_supervised_DDrague (adil.py:508-567) on one batch: z (B,C,H,W) fp32 with its AdamW(1e-2)
    moments; the perturbation is D D_dagger z; z is clamped to +-eps (the perturbation itself is not — quirk Q6).
    `iterate()` is the loop body (adil.py:539-559), `result()` the output (adil.py:563-567), `run(steps)` the loop with
    its stop test, optionally replayed from a hipGraph; `reset(images)` re-arms the same buffers for the next batch of
    the same shape (so a captured graph serves every batch of an evaluation)."""

    def __init__(self, model, images: Tensor, d: Tensor, eps: float, loss: str = "ce", targeted: bool = False,
                 kappa: float = 50.0, pinv: Optional[PseudoInverse] = None, labels: Optional[Tensor] = None,
                 fuse_codes: bool = True):
        self.model, self.images, self.d = model, _flat_images(images).clone(), d
        self.b = self.images.shape[0]
        self.eps, self.loss, self.kappa = float(eps), loss, float(kappa)
        self.coeff = 1.0 if targeted else -1.0
        self.dpt = (pinv if pinv is not None else PseudoInverse(d)).d_pinv_t
        self.z = torch.zeros_like(self.images, dtype=torch.float32)
        self.m, self.s = torch.zeros_like(self.z), torch.zeros_like(self.z)
        self.sched = ops.AdamWSchedule(1e-2)
        self.labels = (predict(model, self.images) if labels is None else labels).clone()   # adil.py:539 (constant)
        self.stop = ops.StopTest(self.images.device, 1e-6)                               # adil.py:559, on the device
        self.iters = 0
        self._graph = None
        # The codes v = z D_dagger^T of the NEXT iteration come out of the z-step itself (ops.zstep_codes_, round 4): the
        # kernel contracts the z tile it has just updated against the D_dagger slice it holds in LDS anyway, and leaves
        # per-workgroup partial sums in this buffer for pack_codes — the separate contraction launch and its second pass
        # over z are gone from the loop.  `_vnext`: None = z is still all zero (so are its codes), else the SlabGrad of
        # the last z-step.  Shapes the fused kernel does not take (ragged pixel count, K > 112) keep the two launches.
        p, k = ops.dict_shape(d)
        nbytes = ops.zstep_codes_slab_bytes(self.b, p, k) if (fuse_codes and self.b > 0) else 0
        self._vslabs = torch.empty(nbytes, dtype=torch.uint8, device=self.z.device) if nbytes else None
        self._vnext = None

    def reset(self, images: Tensor, labels: Optional[Tensor] = None) -> "DDragueSolver":
        """Same buffers, next batch (same shape and dtype): z, moments, step counter and stop slots back to their start."""
        self.images.copy_(_flat_images(images))
        self.labels.copy_(predict(self.model, self.images) if labels is None else labels)
        for t in (self.z, self.m, self.s):
            t.zero_()
        self.sched.t, self.iters = 0, 0
        self.stop.reset()
        self._vnext = None
        return self

    def codes(self, defer: bool = False):
        """v = z D_dagger^T (K6, adil.py:542) as its own contraction.  defer: for pack_codes only — possibly still as
        partial sums (ops.SlabGrad)."""
        _, vcode = ops.grad(self.z, self.dpt, None, self.b, want_d=False, want_v=True, defer_v=defer)
        return vcode

    def packed_codes(self) -> Tensor:
        """The codes of the current z, packed for the synthesis: from the last z-step's slabs when it produced them."""
        if self._vslabs is None:
            return ops.pack_codes(self.codes(defer=True), None, self.b)
        if self._vnext is None:                                  # no z-step yet: z = 0 (adil.py:530), so v = 0
            p, k = ops.dict_shape(self.d)
            return torch.zeros(ops._round_up(self.b, 32), ops._round_up(k, 16), dtype=torch.float32, device=self.z.device)
        return ops.pack_codes(self._vnext, None, self.b)

    def iterate(self, dyn: Optional[Tensor] = None) -> None:
        """One iteration.  The stop test of adil.py:559 runs on the device (ops.StopTest): after the iteration whose
        max|dz| falls below 1e-6 the z-step launches do nothing; `self.stop.converged()` polls it."""
        b = self.b
        self.iters += 1
        vp = self.packed_codes()                                                         # adil.py:542
        xt = ops.synth(self.images, self.d, vp, b)                                       # adil.py:543-544
        with precise_head(self.model):                                                   # fp32 logits where the classifier offers them
            _, _, g = input_gradient(self.model, xt, self.labels, self.loss, self.coeff, self.kappa, "mean")
        _, gv = ops.grad(g, self.d, None, b, want_d=False, want_v=True, defer_v=True)    # dL/dv = g D (summed by pack_codes)
        # dL/dz = (dL/dv) D_dagger is formed inside the kernel and consumed by AdamW(z) + clamp: never materialised (K8)
        gvp = ops.pack_codes(gv, None, b)
        if self._vslabs is not None:
            self._vnext = ops.zstep_codes_(self.z, self.m, self.s, self.dpt, gvp, b, self.sched.next(), -self.eps, self.eps,
                                           self._vslabs, stop=self.stop, dyn=dyn)
        else:
            ops.zstep_(self.z, self.m, self.s, self.dpt, gvp, b, self.sched.next(), -self.eps, self.eps, stop=self.stop,
                       dyn=dyn)

    # -- three iterations as ONE hipGraph launch ------------------------------------------------------------------ #
    def _capture(self) -> None:
        """Record iterations t, t+1, t+2 (t a multiple of 3: the stop slots rotate with period 3, so a group of three is
        what can be replayed verbatim).  AdamW's step-dependent scalars come from a [3][2] device buffer that is refreshed
        before every replay; capture records and does not execute, so the counters are put back."""
        dev = self.z.device
        self._dyn = torch.zeros(3, 2, dtype=torch.float32, device=dev)
        self._dyn_ring = ops.PinnedRing((3, 2), torch.float32)   # slots guarded by events: safe however far the host runs ahead
        t_sched, t_stop, iters = self.sched.t, self.stop.t, self.iters
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                for j in range(3):
                    self.iterate(dyn=self._dyn[j])
        finally:                                                 # recorded or failed: nothing was executed
            self.sched.t, self.stop.t, self.iters = t_sched, t_stop, iters
        self._graph = graph

    def _replay(self) -> None:
        hs = [self.sched.next() for _ in range(3)]
        self._dyn_ring.push(self._dyn, [(h.step_size, h.bc2_sqrt) for h in hs])     # ordered before the replay on the same stream
        self.stop.t += 3
        self.iters += 3
        self._graph.replay()

    def run(self, steps: int, use_graph: bool = False) -> "DDragueSolver":
        """The loop of adil.py:534-559.  use_graph: after three eager iterations (library warm-up; aligns the slot
        rotation) the iterations run three at a time from a hipGraph — one launch instead of ~600 — which is what a
        launch-bound attack needs (one image through mobilenet_v2: 7.2 ms per eager iteration whatever the batch size).
        The device-side stop test makes post-convergence launches no-ops, so the result is the eager loop's bit for bit."""
        steps, it = int(steps), 0
        if use_graph and steps >= 6:
            while it < 3:
                self.iterate()
                it += 1
            if self._graph is None:
                try:
                    self._capture()
                except RuntimeError as e:
                    # a classifier whose forward / backward cannot be recorded (a library workspace allocation or an implicit
                    # synchronisation under capture): the attack still runs, eagerly (main.py defaults to --graph 1)
                    import warnings
                    warnings.warn(f"hipGraph capture of the DDrague iterations failed ({str(e).splitlines()[0]}); "
                                  "running the eager loop instead", RuntimeWarning)
                    self._graph = False
            while self._graph and it + 3 <= steps and not self.stop.converged():
                self._replay()
                it += 3
        while it < steps:
            self.iterate()
            it += 1
            if it % STOP_POLL == 0 and self.stop.converged():                            # adil.py:559
                break
        return self

    def result(self) -> Tuple[Tensor, Tensor]:
        vp = self.packed_codes()
        adv = ops.synth(self.images, self.d, vp, self.b, pixel_clamp=True)               # adil.py:563-567
        return adv, vp[:self.b, :ops.dict_shape(self.d)[1]].contiguous()


def solve_ddrague(model, images: Tensor, d: Tensor, eps: float, steps_inference: int = 30, loss: str = "ce",
                  targeted: bool = False, kappa: float = 50.0, pinv: Optional[PseudoInverse] = None,
                  labels: Optional[Tensor] = None, return_trace: bool = False, use_graph: bool = False,
                  fuse_codes: bool = True):
    """forward_supervised_DDrague (adil.py:508-567): optimise z (B,C,H,W) with AdamW(1e-2), the perturbation
    being D D_dagger z; z is clamped to +-eps (the perturbation itself is not — quirk Q6)."""
    solver = DDragueSolver(model, images, d, eps, loss, targeted, kappa, pinv, labels, fuse_codes).run(steps_inference,
                                                                                                       use_graph)
    adv, vcode = solver.result()
    if return_trace:
        return adv, dict(z=solver.z, v=vcode, iters=solver.iters, labels=solver.labels)
    return adv


@torch.no_grad()
def attack_unsupervised(model, images: Tensor, d: Tensor, eps: float, v_trials: Sequence[Tensor]):
    """forward_unsupervised (adil.py:460-506) with the per-image Python loop (adil.py:480-484) replaced by one
    synthesis launch per trial.  v_trials[t] is the (B,K) sample of trial t.  Returns (adv_best, dv_norm_inf
    of the last trial) like the reference (quirk Q12)."""
    images = _flat_images(images)
    b = images.shape[0]
    dev = images.device
    pre = predict(model, images)
    fooling_flag = torch.zeros(b, dtype=torch.bool, device=dev)
    mse_best_no_fool = torch.full((b,), float("inf"), device=dev)
    adv_best = images.clone()
    dv_norm_inf = None
    for trial, vt in enumerate(v_trials):
        vp = ops.pack_codes(vt.to(device=dev, dtype=torch.float32).contiguous(), None, b)
        adv = ops.synth(images, d, vp, b, delta_clamp=eps, pixel_clamp=True)             # adil.py:481-484
        if trial == len(v_trials) - 1:           # only the LAST trial's norms are returned (quirk Q12): one extra launch, once
            dv = ops.synth(None, d, vp, b, out_shape=images.shape, out_dtype=torch.float32, delta_clamp=eps)
            dv_norm_inf = dv.abs().flatten(1).max(dim=1).values
        fooling = predict(model, adv) != pre
        mse, _ = ops.image_metrics(adv, images)
        # keep-best bookkeeping of adil.py:494-504, vectorised
        first = (~fooling_flag) & fooling
        same = (fooling_flag & fooling) | ((~fooling_flag) & (~fooling))
        better = same & (mse < mse_best_no_fool)
        take = first | better
        mse_best_no_fool = torch.where(better, mse, mse_best_no_fool)
        fooling_flag = fooling_flag | first
        adv_best = torch.where(take.view(-1, 1, 1, 1), adv, adv_best)
    return adv_best, dv_norm_inf
