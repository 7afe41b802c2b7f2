"""Tensor-level wrappers over the C ABI (include/adil_hip.h).

PyTorch is used for device memory and streams only: every function takes CUDA
(ROCm) tensors, passes raw device pointers + the current HIP stream through
ctypes, and raises if handed anything the kernels cannot run on.  There is no
CPU path.
"""
from __future__ import annotations

import math
from ctypes import byref, c_int, c_void_p
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib

Tensor = torch.Tensor
_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}
_workspaces = {}


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def stream_dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dtype]
    except KeyError:
        raise TypeError(f"ADiL image streams must be float32 or bfloat16, got {dtype}") from None


def _dev(t: Tensor, name: str, dtype: Optional[torch.dtype] = None) -> Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/ROCm tensor: the ADiL hot path runs on HIP kernels only "
                           "(no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _ptr(t: Optional[Tensor]) -> c_void_p:
    return c_void_p(0 if t is None else t.data_ptr())


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _workspace(device: torch.device, nbytes: int) -> Tensor:
    """Scratch slab of the reductions (grad / gram / atom norms): one per (device, stream), because kernels of two
    streams may run concurrently and the caching allocator only orders a buffer's reuse on its allocation stream."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    # every hand-out starts a new "generation" of this scratch: a SlabGrad that still points into it is stale from here
    # on (the contract used to live in a docstring only)
    _generations[key] = _generations.get(key, 0) + 1
    ws._adil_key = key
    return ws


_generations = {}


def _workspace_generation(ws: Tensor):
    key = getattr(ws, "_adil_key", None)
    return (key, _generations.get(key)) if key is not None else None


def dict_shape(d: Tensor) -> Tuple[int, int]:
    """(P, K) of a dictionary stored as the reference's (C,H,W,K) tensor (adil.py:20)."""
    k = d.shape[-1]
    return d.numel() // k, k


# --------------------------------------------------------------------------- #
@dataclass
class AdamWScalars:
    decay: float
    b1: float
    b2: float
    eps: float
    step_size: float
    bc2_sqrt: float


class AdamWSchedule:
    """Host-side step counter of one torch.optim.AdamW param (defaults of adil.py:154: betas (0.9,0.999),
    eps 1e-8, weight_decay 1e-2).  The scalars are formed in double exactly as torch forms them."""

    def __init__(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        self.lr, self.b1, self.b2, self.eps, self.wd = float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
        self.t = 0

    def next(self) -> AdamWScalars:
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        return AdamWScalars(1.0 - self.lr * self.wd, self.b1, self.b2, self.eps, self.lr / bc1, math.sqrt(bc2))

    # -- hipGraph replay: the two step-dependent scalars live in device memory and are refreshed before each replay -- #
    def enable_device_scalars(self, device) -> Tensor:
        """A 2-float device buffer {step_size, bc2_sqrt} (the `dyn_scalars` of the AdamW entry points) fed through a ring
        of pinned host slots (PinnedRing), so that a launch recorded in a graph picks up the scalars of the step it is
        replayed for."""
        self.dyn = torch.zeros(2, dtype=torch.float32, device=device)
        self._ring = PinnedRing((2,), torch.float32)
        return self.dyn

    def next_to_device(self) -> AdamWScalars:
        """next(), and a stream-ordered copy of its step-dependent scalars into the device buffer."""
        h = self.next()
        self._ring.push(self.dyn, (h.step_size, h.bc2_sqrt))
        return h


class PinnedRing:
    """Small host values that recorded (hipGraph) launches read from device memory, refreshed before every replay.

    A stream-ordered host-to-device copy reads its pinned source when the GPU REACHES it, not when it was queued, so the
    source must not be rewritten before then: every slot carries an event recorded behind its copy, and a slot is only
    reused after that event has completed (a host wait that happens only when the host is a whole ring of replays ahead
    of the GPU).  Without it a free-running replay loop more than `slots` steps ahead handed the recorded AdamW kernels
    the scalars of a LATER step."""

    def __init__(self, shape, dtype, slots: int = 8):
        self.host = torch.zeros((slots,) + tuple(shape), dtype=dtype).pin_memory()
        self.events = [None] * slots
        self.i = 0

    def push(self, dst: Tensor, values) -> None:
        """dst <- values (nested sequence of the slot's shape), ordered on the current stream before whatever follows."""
        k = self.i % len(self.events)
        self.i += 1
        if self.events[k] is None:
            self.events[k] = torch.cuda.Event()
        else:
            self.events[k].synchronize()                         # the copy that last read this slot has run
        self.host[k].copy_(torch.as_tensor(values, dtype=self.host.dtype))
        dst.copy_(self.host[k], non_blocking=True)
        self.events[k].record()


class StopTest:
    """Device-side stop test of the inference solvers (`if max|x - x_old| < 1e-6: break`, adil.py:559, :614).

    Three rotating fp32 slots: iteration t accumulates its max|delta| into slot t%3, does nothing at all if slot
    (t-1)%3 — the previous iteration's maximum — is already below the threshold, and clears slot (t+1)%3 for its
    successor.  Once an iteration converges every later launch is therefore a no-op: the host can poll `converged()`
    (a device->host sync) only every few iterations and still stop on exactly the iterate the reference breaks at."""

    def __init__(self, device, threshold: float = 1e-6):
        self.slots = torch.tensor([0.0, 0.0, 3.0e38], dtype=torch.float32, device=device)
        self.threshold, self.t = float(threshold), 0

    def _slot(self, i: int) -> c_void_p:
        return c_void_p(self.slots.data_ptr() + 4 * (i % 3))

    def launch_args(self):
        """(max_abs_delta, skip_if_below, skip_threshold, clear) of the launch of iteration t; advances t."""
        t = self.t
        self.t += 1
        return self._slot(t), self._slot(t + 2), self.threshold, self._slot(t + 1)

    def reset(self) -> None:
        self.slots.copy_(torch.tensor([0.0, 0.0, 3.0e38], dtype=torch.float32))
        self.t = 0

    def last_slot(self) -> Tensor:
        """1-element view of the slot the last launched iteration accumulated its max|delta| into (for a max-reduction
        over ranks when the iterate is sharded: the next launch then tests the global value)."""
        i = (self.t - 1) % 3
        return self.slots[i:i + 1]

    def idle_iteration(self) -> None:
        """What a launch does to the slots when it has no rows: nothing to accumulate, clear the successor's slot."""
        self.slots[(self.t + 1) % 3] = 0.0
        self.t += 1

    def converged(self) -> bool:
        """True once the last launched iteration (or an earlier one) moved nothing by the threshold or more."""
        return self.t > 0 and float(self.slots[(self.t - 1) % 3]) < self.threshold


# --------------------------------------------------------------------------- #
class SlabGrad:
    """A (B,K) code gradient (or code matrix) that is still the per-workgroup partial sums ("slabs") of the ops.grad
    call that produced it, inside that call's workspace: `nslabs` slabs of [rows][K] fp32 at `ptr`.  Its consumers —
    adamw_l1ball_ and pack_codes — sum the slabs inside their own launch (same fixed order as the library's reduce
    kernel, identical bits), so no reduction launch sits between producer and consumer.  Valid until the workspace of
    the current stream is used again (the next ops.grad / gram / atom_norms call): consume it right away."""
    __slots__ = ("ws", "ptr", "nslabs", "rows", "batch", "k", "stamp")

    def __init__(self, ws: Tensor, ptr: int, nslabs: int, rows: int, batch: int, k: int):
        self.ws, self.ptr, self.nslabs, self.rows, self.batch, self.k = ws, ptr, nslabs, rows, batch, k
        # slabs inside the shared per-stream scratch are only valid until that scratch is handed out again: remember its
        # generation (None for slabs in a buffer the caller owns, e.g. zstep_codes_)
        self.stamp = _workspace_generation(ws)

    @property
    def shape(self):
        return (self.batch, self.k)

    def check_fresh(self) -> None:
        if self.stamp is not None and _generations.get(self.stamp[0]) != self.stamp[1]:
            raise RuntimeError("stale SlabGrad: the scratch workspace it points into has been used by another ops.grad / gram / "
                               "atom_norms call on this stream since it was produced — consume a deferred gradient "
                               "(adamw_l1ball_ / pack_codes) before the next such call")


def _slab_args(src):
    if isinstance(src, SlabGrad):
        src.check_fresh()
        return c_void_p(src.ptr), src.nslabs, src.rows
    return c_void_p(0), 0, 0


def pack_codes(v, index: Optional[Tensor], batch: Optional[int] = None, pos: Optional[Tensor] = None,
               transposed: Optional[torch.dtype] = None):
    """vp [roundup(B,32)][roundup(K,16)] = zero-padded v[index] (adil.py:25 `self.v[index, :]`).
    pos (int32, one entry per row of v, all -1): receives pos[index[b]] = b for adamw_l1ball_(..., reset_pos=True).
    v may be a SlabGrad (the rows are then summed from the producer's slabs inside this launch; index / pos unused).
    transposed = dtype of the gradient stream ops.grad will see: also returns vpt [code_rows(K)][roundup(B,32)] in that
    dtype (fp32 / bf16), to be handed to ops.grad(vpt=...) — returns (vp, vpt) then."""
    lib = _lib.load()
    slabs = isinstance(v, SlabGrad)
    if slabs:
        if index is not None or pos is not None:
            raise ValueError("pack_codes: a SlabGrad source has its rows in batch order (no index / pos)")
        k, dev = v.k, v.ws.device
        b = v.batch if batch is None else batch
        if b != v.batch:
            raise ValueError("pack_codes: batch does not match the SlabGrad")
    else:
        _dev(v, "v", torch.float32)
        k, dev = v.shape[1], v.device
        if index is not None:
            index = _dev(index.to(device=v.device, dtype=torch.int64), "index")
            b = index.numel()
        else:
            b = v.shape[0] if batch is None else batch
    if pos is not None:
        _dev(pos, "pos", torch.int32)
        if pos.numel() != v.shape[0]:
            raise ValueError("pos must have one entry per row of v")
    vp = torch.empty(_round_up(b, 32), _round_up(k, 16), dtype=torch.float32, device=dev)
    vpt, code = None, 0
    if transposed is not None:
        code = stream_dtype_code(transposed)
        vpt = torch.empty(lib.adil_grad_code_rows(k), _round_up(b, 32), dtype=transposed, device=dev)
    sp, ns, sr = _slab_args(v)
    _lib.check(lib.adil_pack_codes(None if slabs else _ptr(v), _ptr(index), b, k, _ptr(vp), _ptr(pos), _ptr(vpt), code,
                                   sp, ns, sr, _stream()), "adil_pack_codes")
    return vp if transposed is None else (vp, vpt)


def gather_images(src: Tensor, index: Optional[Tensor], out: Optional[Tensor] = None,
                  dtype: Optional[torch.dtype] = None) -> Tensor:
    """out[b] = src[index[b]] converted to `dtype` — one batch of a dataset resident in HBM (src: (R,C,H,W)).
    index None: rows 0..B-1 of src into `out` (a pure cast, used when the dataset is uploaded)."""
    lib = _lib.load()
    _dev(src, "src")
    if index is not None:
        index = _dev(index.to(device=src.device, dtype=torch.int64), "index")
        b = index.numel()
    else:
        b = src.shape[0] if out is None else out.shape[0]
    if out is None:
        out = torch.empty((b,) + tuple(src.shape[1:]), dtype=src.dtype if dtype is None else dtype, device=src.device)
    _dev(out, "out")
    p = src[0].numel() if src.shape[0] else 0
    if b == 0:
        return out
    if out.numel() != b * p or p % 8:
        raise ValueError("gather_images: out must be (B,) + src.shape[1:] and the image size a multiple of 8")
    if index is None and b > src.shape[0]:
        raise ValueError("gather_images: more output rows than source rows")
    _lib.check(lib.adil_gather_images(_ptr(src), stream_dtype_code(src.dtype), _ptr(index), _ptr(out),
                                      stream_dtype_code(out.dtype), b, p, _stream()), "adil_gather_images")
    return out


def fp8_dict_supported(d: Tensor) -> bool:
    """Shapes adil_synth_fp8_packed takes: whole 128-pixel slices, atoms in groups of four."""
    p, k = dict_shape(d)
    return p % 128 == 0 and k % 4 == 0 and p <= (1 << 23)


def dict_to_fp8(d: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """The persistent fp8 copy of a dictionary: bytes e4m3(256 d), same shape, uint8 (adil_dict_to_fp8)."""
    lib = _lib.load()
    _dev(d, "d", torch.float32)
    if out is None:
        out = torch.empty(d.shape, dtype=torch.uint8, device=d.device)
    _dev(out, "out", torch.uint8)
    if out.numel() != d.numel() or d.numel() % 4:
        raise ValueError("dict_to_fp8: same number of elements, a multiple of 4")
    _lib.check(lib.adil_dict_to_fp8(_ptr(d), d.numel(), _ptr(out), _stream()), "adil_dict_to_fp8")
    return out


def synth(x: Optional[Tensor], d: Tensor, vp: Tensor, batch: int, *, out: Optional[Tensor] = None,
          out_shape=None, out_dtype=None, delta_clamp: float = -1.0, pixel_clamp: bool = False,
          fp8_absmax: Optional[float] = None, d_fp8: Optional[Tensor] = None) -> Tensor:
    """out = x + vp D^T with optional +-delta_clamp on the perturbation and [0,1] pixel clamp.
    fp8_absmax: contract with fp8 (e4m3) operands instead (adil_synth_fp8); the value is a bound on |vp| (the l1 radius
    eps for projected codes) and |d| <= 1 is assumed (the invariant update_d maintains).
    d_fp8 (with fp8_absmax): the persistent fp8 copy of d (dict_to_fp8 / adamw_clamp_(..., p_fp8=)): the kernel then reads
    that — a quarter of the dictionary bytes — instead of converting the fp32 master; same result bit for bit."""
    lib = _lib.load()
    _dev(d, "d", torch.float32)
    _dev(vp, "vp", torch.float32)
    p, k = dict_shape(d)
    if x is not None:
        _dev(x, "x")
        if x.numel() != batch * p:
            raise ValueError(f"x has {x.numel()} elements, expected B*P = {batch}*{p}")
        out_shape, out_dtype = x.shape, x.dtype
    if out is None:
        out = torch.empty(out_shape, dtype=out_dtype, device=d.device)
    _dev(out, "out", out_dtype)
    if out.numel() != batch * p or vp.shape != (_round_up(batch, 32), _round_up(k, 16)):
        raise ValueError("synth: operand shapes do not match (B, P, K)")
    if fp8_absmax is not None:
        if not fp8_absmax > 0:
            raise ValueError("fp8_absmax must be a positive bound on |vp|")
        if d_fp8 is not None:
            _dev(d_fp8, "d_fp8", torch.uint8)
            if d_fp8.numel() != d.numel() or not fp8_dict_supported(d):
                raise ValueError("synth: d_fp8 must be the fp8 copy of d (P a multiple of 128, K a multiple of 4)")
            _lib.check(lib.adil_synth_fp8_packed(_ptr(x), _ptr(d_fp8), _ptr(vp), _ptr(out), batch, p, k,
                                                 stream_dtype_code(out.dtype), float(fp8_absmax), float(delta_clamp),
                                                 int(bool(pixel_clamp)), _stream()), "adil_synth_fp8_packed")
            return out
        _lib.check(lib.adil_synth_fp8(_ptr(x), _ptr(d), _ptr(vp), _ptr(out), batch, p, k, stream_dtype_code(out.dtype),
                                      float(fp8_absmax), float(delta_clamp), int(bool(pixel_clamp)), _stream()),
                   "adil_synth_fp8")
        return out
    _lib.check(lib.adil_synth(_ptr(x), _ptr(d), _ptr(vp), _ptr(out), batch, p, k, stream_dtype_code(out.dtype),
                              float(delta_clamp), int(bool(pixel_clamp)), _stream()), "adil_synth")
    return out


def grad(g: Tensor, d: Tensor, vp: Optional[Tensor], batch: int, *, want_d: bool = True, want_v: bool = True,
         grad_d: Optional[Tensor] = None, accumulate_d: bool = False, vpt: Optional[Tensor] = None,
         defer_v: bool = False):
    """(grad_d (C,H,W,K), grad_vb (B,K)) of the synthesis for upstream gradient g (B x P).
    vpt: the transposed codes of pack_codes(..., transposed=g.dtype) (saves the launch that makes them here).
    defer_v: the caller hands grad_vb straight to adamw_l1ball_ / pack_codes: when the kernels allow it (one row chunk)
    the second result is a SlabGrad — the reduction then happens inside the consumer's launch — else the dense tensor."""
    lib = _lib.load()
    _dev(g, "g")
    _dev(d, "d", torch.float32)
    p, k = dict_shape(d)
    if g.numel() != batch * p:
        raise ValueError(f"g has {g.numel()} elements, expected B*P = {batch}*{p}")
    gvb = None
    ws, ws_bytes = None, 0
    if want_d:
        _dev(vp, "vp", torch.float32)
        if vp.shape != (_round_up(batch, 32), _round_up(k, 16)):
            raise ValueError("grad: vp shape does not match (B, K)")
        if grad_d is None:
            grad_d = torch.empty_like(d)
            accumulate_d = False
        _dev(grad_d, "grad_d", torch.float32)
        if grad_d.shape != d.shape:
            raise ValueError("grad_d must have the dictionary's shape")
        if vpt is not None:
            _dev(vpt, "vpt", g.dtype)
            if vpt.shape != (lib.adil_grad_code_rows(k), _round_up(batch, 32)):
                raise ValueError("grad: vpt shape does not match (code_rows(K), roundup(B,32))")
    else:
        grad_d, vpt = None, None
    if want_v:
        gvb = torch.empty(batch, k, dtype=torch.float32, device=d.device)
    ws_bytes = lib.adil_grad_workspace_bytes(batch, p, k)
    ws = _workspace(d.device, ws_bytes)
    nslabs = c_int(0)
    _lib.check(lib.adil_grad(_ptr(g), _ptr(d), _ptr(vp), _ptr(vpt), _ptr(grad_d), _ptr(gvb), batch, p, k,
                             stream_dtype_code(g.dtype), int(bool(accumulate_d)), _ptr(ws), ws_bytes,
                             byref(nslabs) if (defer_v and want_v) else None, _stream()), "adil_grad")
    if nslabs.value > 0:
        gvb = SlabGrad(ws, ws.data_ptr() + lib.adil_grad_slab_offset(batch, p, k), nslabs.value, _round_up(batch, 32),
                       batch, k)
    return grad_d, gvb


def adamw_clamp_(p: Tensor, g: Tensor, m: Tensor, s: Tensor, h: AdamWScalars, lo: float, hi: float,
                 max_abs_delta: Optional[Tensor] = None, dyn: Optional[Tensor] = None, p_fp8: Optional[Tensor] = None) -> None:
    """In-place fused AdamW + clamp[lo,hi] on a flat fp32 parameter (adil.py:186,188 / :554-555).
    p_fp8: also refresh the persistent fp8 copy of p (dict_to_fp8) inside the same launch."""
    lib = _lib.load()
    if p_fp8 is not None:
        for name, t in (("p", p), ("m", m), ("s", s)):
            _dev(t, name, torch.float32)
        _dev(g, "g")
        _dev(p_fp8, "p_fp8", torch.uint8)
        if not (p.numel() == g.numel() == m.numel() == s.numel() == p_fp8.numel()) or p.numel() % 4:
            raise ValueError("adamw_clamp_: size mismatch (with p_fp8 the size must be a multiple of 4)")
        if max_abs_delta is not None:
            _dev(max_abs_delta, "max_abs_delta", torch.float32)
        _lib.check(lib.adil_adamw_clamp_fp8(_ptr(p), _ptr(g), stream_dtype_code(g.dtype), _ptr(m), _ptr(s), p.numel(),
                                            h.decay, h.b1, h.b2, h.eps, h.step_size, h.bc2_sqrt, float(lo), float(hi),
                                            _ptr(max_abs_delta), _ptr(dyn), _ptr(p_fp8), _stream()), "adil_adamw_clamp_fp8")
        return
    for name, t in (("p", p), ("m", m), ("s", s)):
        _dev(t, name, torch.float32)
    _dev(g, "g")
    if not (p.numel() == g.numel() == m.numel() == s.numel()):
        raise ValueError("adamw_clamp_: size mismatch")
    if max_abs_delta is not None:
        _dev(max_abs_delta, "max_abs_delta", torch.float32)
    _lib.check(lib.adil_adamw_clamp(_ptr(p), _ptr(g), stream_dtype_code(g.dtype), _ptr(m), _ptr(s), p.numel(),
                                    h.decay, h.b1, h.b2, h.eps, h.step_size, h.bc2_sqrt, float(lo), float(hi),
                                    _ptr(max_abs_delta), _ptr(dyn), _stream()), "adil_adamw_clamp")


def _stop_args(max_abs_delta: Optional[Tensor], stop: Optional[StopTest]):
    if stop is not None:
        if max_abs_delta is not None:
            raise ValueError("pass either max_abs_delta or stop")
        return stop.launch_args()
    if max_abs_delta is not None:
        _dev(max_abs_delta, "max_abs_delta", torch.float32)
    return _ptr(max_abs_delta), c_void_p(0), 0.0, c_void_p(0)


def zstep_(z: Tensor, m: Tensor, s: Tensor, dpinv_t: Tensor, gvp: Tensor, batch: int, h: AdamWScalars, lo: float,
           hi: float, max_abs_delta: Optional[Tensor] = None, stop: Optional[StopTest] = None,
           dyn: Optional[Tensor] = None) -> None:
    """In-place fused DDrague step: gz = gvp D_dagger formed on the fly, AdamW(z), clamp, max|dz| (adil.py:551-559)."""
    lib = _lib.load()
    for name, t in (("z", z), ("m", m), ("s", s), ("dpinv_t", dpinv_t), ("gvp", gvp)):
        _dev(t, name, torch.float32)
    p, k = dict_shape(dpinv_t)
    if not (z.numel() == m.numel() == s.numel() == batch * p) or gvp.shape != (_round_up(batch, 32), _round_up(k, 16)):
        raise ValueError("zstep_: operand shapes do not match (B, P, K)")
    dmax, skip, thr, clear = _stop_args(max_abs_delta, stop)
    _lib.check(lib.adil_zstep(_ptr(z), _ptr(m), _ptr(s), _ptr(dpinv_t), _ptr(gvp), batch, p, k, h.decay, h.b1, h.b2, h.eps,
                              h.step_size, h.bc2_sqrt, float(lo), float(hi), dmax, skip, thr, clear, _ptr(dyn), _stream()),
               "adil_zstep")


def zstep_codes_slab_bytes(batch: int, p: int, k: int) -> int:
    """Bytes of the slab buffer zstep_codes_ needs for this shape; 0 = the fused kernel does not take it (whole 128-pixel
    slices, K <= 112): use zstep_ and a separate ops.grad(z, D_dagger^T) then."""
    return int(_lib.load().adil_zstep_codes_slab_bytes(int(batch), int(p), int(k)))


def zstep_codes_(z: Tensor, m: Tensor, s: Tensor, dpinv_t: Tensor, gvp: Tensor, batch: int, h: AdamWScalars, lo: float,
                 hi: float, slabs: Tensor, stop: Optional[StopTest] = None, dyn: Optional[Tensor] = None) -> "SlabGrad":
    """zstep_ that also contracts the freshly updated z with D_dagger^T: returns the NEXT iteration's codes
    v' = z_new D_dagger^T (adil.py:542) as a SlabGrad living in `slabs` (a uint8 buffer of zstep_codes_slab_bytes that
    the caller owns — it must outlive the SlabGrad, which pack_codes sums).  A launch skipped by the device-side stop
    test leaves the slabs as they were, i.e. the codes of the converged z."""
    lib = _lib.load()
    for name, t in (("z", z), ("m", m), ("s", s), ("dpinv_t", dpinv_t), ("gvp", gvp)):
        _dev(t, name, torch.float32)
    _dev(slabs, "slabs", torch.uint8)
    p, k = dict_shape(dpinv_t)
    if not (z.numel() == m.numel() == s.numel() == batch * p) or gvp.shape != (_round_up(batch, 32), _round_up(k, 16)):
        raise ValueError("zstep_codes_: operand shapes do not match (B, P, K)")
    need = zstep_codes_slab_bytes(batch, p, k)
    if need == 0 or slabs.numel() < need:
        raise ValueError("zstep_codes_: shape not supported by the fused kernel, or slab buffer too small")
    dmax, skip, thr, clear = _stop_args(None, stop)
    nslabs = c_int(0)
    _lib.check(lib.adil_zstep_codes(_ptr(z), _ptr(m), _ptr(s), _ptr(dpinv_t), _ptr(gvp), batch, p, k, h.decay, h.b1, h.b2,
                                    h.eps, h.step_size, h.bc2_sqrt, float(lo), float(hi), dmax, skip, thr, clear, _ptr(dyn),
                                    _ptr(slabs), slabs.numel(), byref(nslabs), _stream()), "adil_zstep_codes")
    return SlabGrad(slabs, slabs.data_ptr(), nslabs.value, _round_up(batch, 32), batch, k)


def adamw_l1ball_(v: Tensor, grad_vb, pos: Optional[Tensor], m: Tensor, s: Tensor, h: AdamWScalars,
                  radius: float, max_abs_delta: Optional[Tensor] = None, reset_pos: bool = False,
                  stop: Optional[StopTest] = None, dyn: Optional[Tensor] = None) -> None:
    """In-place AdamW on ALL rows of v (zero gradient outside the batch) + l1-ball projection
    (adil.py:186-187; radius < 0 skips the projection).  pos is the batch-slot table written by pack_codes; with
    reset_pos the kernel hands it back all -1.  grad_vb None = no row of this v is in the batch (pos all -1); a
    SlabGrad (ops.grad(..., defer_v=True)) is summed inside the launch."""
    lib = _lib.load()
    for name, t in (("v", v), ("m", m), ("s", s)):
        _dev(t, name, torch.float32)
    n, k = v.shape
    slabs = isinstance(grad_vb, SlabGrad)
    if grad_vb is not None:
        if not slabs:
            _dev(grad_vb, "grad_vb", torch.float32)
        if grad_vb.shape[1] != k:
            raise ValueError("adamw_l1ball_: shape mismatch")
    if pos is not None:
        _dev(pos, "pos", torch.int32)
        if pos.numel() != n:
            raise ValueError("pos must have one entry per row of v")
    elif grad_vb is None or grad_vb.shape[0] != n:
        raise ValueError("without pos, grad_vb must have one row per row of v")
    if m.shape != v.shape or s.shape != v.shape:
        raise ValueError("adamw_l1ball_: shape mismatch")
    dmax, skip, thr, clear = _stop_args(max_abs_delta, stop)
    sp, ns, sr = _slab_args(grad_vb)
    _lib.check(lib.adil_adamw_l1ball(_ptr(v), None if slabs else _ptr(grad_vb), _ptr(pos), int(bool(reset_pos)), _ptr(m),
                                     _ptr(s), n, k, h.decay, h.b1, h.b2, h.eps, h.step_size, h.bc2_sqrt, float(radius),
                                     dmax, skip, thr, clear, _ptr(dyn), sp, ns, sr, _stream()), "adil_adamw_l1ball")


def l1ball_project_(x: Tensor, radius: float) -> Tensor:
    """In-place row-wise projection onto the l1 ball (utils.py:21-41). x is (N, ...) fp32; rows are flattened."""
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    n = x.shape[0]
    k = x.numel() // max(n, 1)
    _lib.check(lib.adil_l1ball_project(_ptr(x), n, k, float(radius), _stream()), "adil_l1ball_project")
    return x


def l2ball_project_(x: Tensor, radius: float) -> Tensor:
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    n = x.shape[0]
    _lib.check(lib.adil_l2ball_project(_ptr(x), n, x.numel() // max(n, 1), float(radius), _stream()),
               "adil_l2ball_project")
    return x


def ista_step_(v: Tensor, g: Optional[Tensor], step: float, lam: float) -> Tensor:
    """v = softshrink(v - step*g, lam) in place (adil_regularized.py:570-573)."""
    lib = _lib.load()
    _dev(v, "v", torch.float32)
    if g is not None:
        _dev(g, "g", torch.float32)
        if g.numel() != v.numel():
            raise ValueError("ista_step_: size mismatch")
    _lib.check(lib.adil_ista_step(_ptr(v), _ptr(g), v.numel(), float(step), float(lam), _stream()), "adil_ista_step")
    return v


def atom_norms(d: Tensor) -> Tensor:
    lib = _lib.load()
    _dev(d, "d", torch.float32)
    p, k = dict_shape(d)
    nbytes = lib.adil_atom_workspace_bytes(p, k)
    ws = _workspace(d.device, nbytes)
    norms = torch.empty(k, dtype=torch.float32, device=d.device)
    _lib.check(lib.adil_atom_norms(_ptr(d), p, k, _ptr(norms), _ptr(ws), nbytes, _stream()), "adil_atom_norms")
    return norms


def atom_l2_project_(d: Tensor, sphere: bool = False, radius: float = 1.0) -> Tensor:
    """constraint_dict 'l2ball' / 'l2sphere' in place (utils.py:44-54); radius != 1 projects every atom onto the
    l2 ball / sphere of that radius (UAPPGD.project with its single atom, uappgd.py:60-66)."""
    lib = _lib.load()
    norms = atom_norms(d)
    if radius != 1.0:
        norms.div_(float(radius))                                     # d / max(|d|/r, 1) = r d / max(|d|, r)
    p, k = dict_shape(d)
    _lib.check(lib.adil_atom_scale(_ptr(d), p, k, _ptr(norms), int(bool(sphere)), _stream()), "adil_atom_scale")
    return d


def atom_l1_project_(d: Tensor, radius: float = 1.0) -> Tensor:
    """constraint_dict's l1 branch in place (utils.py:55-56): every (channel, atom) row of H*W pixels of the (C,H,W,K)
    dictionary onto the l1 ball of `radius`."""
    lib = _lib.load()
    _dev(d, "d", torch.float32)
    if d.dim() != 4:
        raise ValueError("atom_l1_project_: the dictionary must be (C,H,W,K)")
    c, h, w, k = d.shape
    _lib.check(lib.adil_atom_l1ball_project(_ptr(d), c, h * w, k, float(radius), _stream()), "adil_atom_l1ball_project")
    return d


def gram(d: Tensor) -> Tensor:
    """D^T D (K x K) (adil.py:523)."""
    lib = _lib.load()
    _dev(d, "d", torch.float32)
    p, k = dict_shape(d)
    nbytes = lib.adil_gram_workspace_bytes(p, k)
    ws = _workspace(d.device, nbytes)
    out = torch.empty(k, k, dtype=torch.float32, device=d.device)
    _lib.check(lib.adil_gram(_ptr(d), p, k, _ptr(out), _ptr(ws), nbytes, _stream()), "adil_gram")
    return out


def spd_inverse(a: Tensor) -> Tensor:
    """Inverse of the K x K Gram matrix on the device (`dtd.inverse()`, adil.py:524): no host round trip."""
    lib = _lib.load()
    _dev(a, "a", torch.float32)
    k = a.shape[0]
    if a.shape != (k, k):
        raise ValueError("spd_inverse: square matrix expected")
    out = torch.empty_like(a)
    _lib.check(lib.adil_spd_inverse(_ptr(a), k, _ptr(out), _stream()), "adil_spd_inverse")
    return out


def dict_rightmul(d: Tensor, mat: Tensor) -> Tensor:
    """D M^T as a (C,H,W,K) tensor; with M = (DtD)^-1 this is D_dagger^T (adil.py:525)."""
    lib = _lib.load()
    _dev(d, "d", torch.float32)
    mat = _dev(mat.contiguous(), "mat", torch.float32)           # K x K: LAPACK inverses come back column-major
    p, k = dict_shape(d)
    if mat.shape != (k, k):
        raise ValueError("mat must be K x K")
    out = torch.empty_like(d)
    _lib.check(lib.adil_dict_rightmul(_ptr(d), _ptr(mat), p, k, _ptr(out), _stream()), "adil_dict_rightmul")
    return out


def image_metrics(adv: Tensor, x: Tensor) -> Tuple[Tensor, Tensor]:
    """Per-image sum (adv-x)^2 and sum x^2 (performance.py:249-266)."""
    lib = _lib.load()
    _dev(adv, "adv")
    _dev(x, "x", adv.dtype)
    b = x.shape[0]
    if b == 0:                                                   # empty batch (e.g. nothing correctly classified): empty sums
        z = torch.zeros(0, dtype=torch.float32, device=x.device)
        return z, z.clone()
    p = x.numel() // b
    se = torch.empty(b, dtype=torch.float32, device=x.device)
    sn = torch.empty(b, dtype=torch.float32, device=x.device)
    _lib.check(lib.adil_image_metrics(_ptr(adv), _ptr(x), b, p, stream_dtype_code(x.dtype), _ptr(se), _ptr(sn),
                                      _stream()), "adil_image_metrics")
    return se, sn


# --------------------------------------------------------------------------- #
def _channel_inner(t: Tensor) -> int:
    """Stride pattern of a 4-d activation: 1 for channels_last storage, H*W for contiguous NCHW."""
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous():
        return 1
    if t.is_contiguous():
        return t.shape[2] * t.shape[3] if t.dim() == 4 else 1
    raise ValueError("activation must be contiguous (NCHW) or channels_last")


class AffineActFunction(torch.autograd.Function):
    """y = act(x * scale[c] + shift[c] (+ res)) in one pass; backward gives the INPUT gradients only (the classifier is
    frozen).  Replaces eval-BatchNorm + residual add + ReLU of a frozen conv block (see zoo.fuse_bn_act_)."""

    @staticmethod
    def forward(ctx, x, res, scale, shift, relu):
        lib = _lib.load()
        _dev(scale, "scale", torch.float32)
        _dev(shift, "shift", torch.float32)
        inner = _channel_inner(x)
        if res is not None and res.shape != x.shape:
            raise ValueError(f"affine_act: residual shape {tuple(res.shape)} != activation shape {tuple(x.shape)}")
        if res is not None and (_channel_inner_or_none(res) != inner or res.dtype != x.dtype):
            res = res.to(x.dtype).contiguous(memory_format=torch.channels_last if inner == 1 else torch.contiguous_format)
        y = torch.empty_like(x)
        if x.numel() > 0:                                             # empty batches pass through like plain torch modules
            _lib.check(lib.adil_affine_act_fwd(_ptr(x), _ptr(res), _ptr(scale), _ptr(shift), _ptr(y), x.numel(), x.shape[1],
                                               inner, int(relu), stream_dtype_code(x.dtype), _stream()),
                       "adil_affine_act_fwd")
        ctx.save_for_backward(y if relu else None, scale)
        ctx.meta = (inner, bool(relu), res is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        y, scale = ctx.saved_tensors
        inner, relu, has_res = ctx.meta
        ref = y if y is not None else g
        if _channel_inner_or_none(g) != inner:
            g = g.contiguous(memory_format=torch.channels_last if inner == 1 else torch.contiguous_format)
        gx = torch.empty_like(g)
        gres = torch.empty_like(g) if has_res else None
        if g.numel() == 0:
            return gx, gres, None, None, None
        _lib.check(lib.adil_affine_act_bwd(_ptr(g), _ptr(y), _ptr(scale), _ptr(gx), _ptr(gres), g.numel(), g.shape[1], inner,
                                           int(relu), stream_dtype_code(g.dtype), _stream()), "adil_affine_act_bwd")
        return gx, gres, None, None, None


def _channel_inner_or_none(t: Tensor):
    try:
        return _channel_inner(t)
    except ValueError:
        return None


def affine_act(x: Tensor, scale: Tensor, shift: Tensor, res: Optional[Tensor] = None, relu: bool = True) -> Tensor:
    return AffineActFunction.apply(x, res, scale, shift, relu)


# --------------------------------------------------------------------------- #
class PointwiseConvFunction(torch.autograd.Function):
    """y = act(conv1x1(x') * scale + shift (+ res)) on channels_last bf16 tensors as ONE kernel (adil_pw_conv_fwd: the
    GEMM with the epilogue applied to its accumulators), and its input gradient as ONE kernel (adil_pw_conv_bwd: the
    epilogue backward applied to the GEMM operand on its way into LDS).
    `pre=(pscale, pshift)`: x is the RAW output of the previous (library) convolution and x' = relu(x*pscale+pshift)
    is formed inside the kernels — that layer's BatchNorm + ReLU makes no HBM pass of its own, forward or backward —
    otherwise x' = x.
    `outputs` = 1: y.  2: (y, y) — a residual block hands one to its main path and one to its skip path, so that the two
    gradients arrive here separately and are summed inside the backward kernel instead of by an autograd add kernel.
    3: (y, y, y[:, :, ::2, ::2]) — the third is the (strided, uncopied) input of the next block's stride-2 downsample
    convolution; its gradient comes back on the stride-2 grid and is added in the kernel, never zero-upsampled.
    `stride2=True`: x is such a [:, :, ::2, ::2] view (offset 0) of a channels_last tensor; the kernel gathers."""

    @staticmethod
    def forward(ctx, x, res, w2d, wt2d, scale, shift, relu, outputs, pscale, pshift, stride2):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        b, cin, h, w = x.shape                                            # h, w: OUTPUT grid (= input grid unless stride2)
        cout = w2d.shape[0]
        sub_w, sub_hw = 0, 0
        if stride2:
            full = (4 * h * w * cin, 1, 4 * w * cin, 2 * cin)             # strides of t[:, :, ::2, ::2], t channels_last
            if tuple(x.stride()) == full:
                x2, sub_w, sub_hw = x, w, h * w                           # read in place through the gather
            else:
                x2 = x.permute(0, 2, 3, 1).contiguous()
        else:
            x2 = x.permute(0, 2, 3, 1)
            if not x2.is_contiguous():
                x2 = x2.contiguous()
        r2 = None
        if res is not None:
            r2 = res.permute(0, 2, 3, 1)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        y = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device)
        if b > 0:                                                     # empty batches pass through like plain torch modules
            _lib.check(lib.adil_pw_conv_fwd(_ptr(x2), _ptr(w2d), _ptr(scale), _ptr(shift), _ptr(r2), _ptr(y), b * h * w, cin,
                                            cout, int(relu), _ptr(pscale), _ptr(pshift), sub_w, sub_hw, _stream()),
                       "adil_pw_conv_fwd")
        if pscale is not None and sub_w:
            raise RuntimeError("prologue and stride-2 gather are not combined")
        ctx.save_for_backward(y if relu else None, scale, wt2d, x2 if pscale is not None else None, pscale, pshift)
        ctx.meta = (bool(relu), res is not None, cin)
        out = y.permute(0, 3, 1, 2)
        if outputs == 1:
            return out
        if outputs == 2:
            return out, out.view_as(out)
        return out, out.view_as(out), out[:, :, ::2, ::2]

    @staticmethod
    def backward(ctx, g, g_twin=None, g_sub=None):
        lib = _lib.load()
        y, scale, wt2d, xin, pscale, pshift = ctx.saved_tensors
        relu, has_res, cin = ctx.meta
        if g is None:
            g, g_twin = g_twin, None
        if g is None:
            if g_sub is not None:
                raise RuntimeError("a stride-2 gradient arrived without a full-resolution one")
            return (None,) * 11

        def nhwc(t):
            t2 = t.permute(0, 2, 3, 1)
            return t2 if (t2.is_contiguous() and t2.dtype == torch.bfloat16) else t2.to(torch.bfloat16).contiguous()

        g2 = nhwc(g)
        gt = nhwc(g_twin) if g_twin is not None else None
        b, h, w, cout = g2.shape
        g3, sub_w, sub_hw = None, 0, 0
        if g_sub is not None:
            g3 = nhwc(g_sub)
            sub_w, sub_hw = g3.shape[2], g3.shape[1] * g3.shape[2]
            if (h, w) != (2 * g3.shape[1], 2 * g3.shape[2]):
                raise RuntimeError("stride-2 gradient does not match an even full-resolution grid")
        gx = torch.empty((b, h, w, cin), dtype=torch.bfloat16, device=g2.device)
        gres = torch.empty_like(g2) if has_res else None
        if b == 0:
            return (gx.permute(0, 3, 1, 2), gres.permute(0, 3, 1, 2) if has_res else None) + (None,) * 9
        _lib.check(lib.adil_pw_conv_bwd(_ptr(g2), _ptr(gt), _ptr(y), _ptr(scale), _ptr(wt2d), _ptr(gx), _ptr(gres), b * h * w,
                                        cin, cout, int(relu), _ptr(xin), _ptr(pscale), _ptr(pshift), _ptr(g3), sub_w, sub_hw,
                                        _stream()), "adil_pw_conv_bwd")
        return (gx.permute(0, 3, 1, 2), gres.permute(0, 3, 1, 2) if has_res else None) + (None,) * 9


def pointwise_conv_affine(x: Tensor, w2d: Tensor, wt2d: Tensor, scale: Tensor, shift: Tensor, res: Optional[Tensor] = None,
                          relu: bool = True, outputs: int = 1, pre: Optional[Tuple[Tensor, Tensor]] = None,
                          stride2: bool = False):
    pscale, pshift = pre if pre is not None else (None, None)
    return PointwiseConvFunction.apply(x, res, w2d, wt2d, scale, shift, relu, outputs, pscale, pshift, stride2)


# --------------------------------------------------------------------------- #
def pack_conv3x3_weights(weight: Tensor) -> Tuple[Tensor, Tensor]:
    """(N,C,3,3) conv weight -> the two bf16 layouts of adil_conv3x3 (include/adil_hip.h), both 2-D:
    forward [N][9*C] (w[n][c][kh][kw] at [n][kh*3+kw][c]) and input gradient [C][9*N] (taps flipped, channels swapped)."""
    n, c, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise ValueError(f"expected a 3x3 convolution weight, got {tuple(weight.shape)}")
    w = weight.detach().float()
    fwd = w.permute(0, 2, 3, 1).reshape(n, 9 * c).to(torch.bfloat16).contiguous()
    bwd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(c, 9 * n).to(torch.bfloat16).contiguous()
    return fwd, bwd


class Conv3x3Function(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 convolution on channels_last bf16 tensors (adil_conv3x3), raw output; the input gradient
    is the same kernel on the flipped / transposed weights.  No weight gradient: the network is frozen."""

    @staticmethod
    def forward(ctx, x, wp_fwd, wp_bwd):
        lib = _lib.load()
        b, c, h, w = x.shape
        n = wp_fwd.shape[0]
        x2 = x.permute(0, 2, 3, 1)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y = torch.empty((b, h, w, n), dtype=torch.bfloat16, device=x.device)
        if b > 0:
            _lib.check(lib.adil_conv3x3(_ptr(x2), _ptr(wp_fwd), _ptr(y), b, h, w, c, n, _stream()), "adil_conv3x3")
        ctx.save_for_backward(wp_bwd)
        ctx.meta = (c,)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (wp_bwd,) = ctx.saved_tensors
        (c,) = ctx.meta
        g2 = g.permute(0, 2, 3, 1)
        if not (g2.is_contiguous() and g2.dtype == torch.bfloat16):
            g2 = g2.to(torch.bfloat16).contiguous()
        b, h, w, n = g2.shape
        gx = torch.empty((b, h, w, c), dtype=torch.bfloat16, device=g2.device)
        if b > 0:
            _lib.check(lib.adil_conv3x3(_ptr(g2), _ptr(wp_bwd), _ptr(gx), b, h, w, n, c, _stream()), "adil_conv3x3")
        return gx.permute(0, 3, 1, 2), None, None


def conv3x3(x: Tensor, wp_fwd: Tensor, wp_bwd: Tensor) -> Tensor:
    return Conv3x3Function.apply(x, wp_fwd, wp_bwd)


# --------------------------------------------------------------------------- #
def pack_stem_weights(weight: Tensor) -> Tuple[Tensor, Tensor]:
    """(64,3,7,7) conv weight -> the two bf16 layouts of include/adil_hip.h: w_fwd [64][7][8][4], w_bwd [4][49][64]."""
    if tuple(weight.shape) != (64, 3, 7, 7):
        raise ValueError(f"stem kernels are written for a (64,3,7,7) convolution, got {tuple(weight.shape)}")
    w = weight.detach().float()
    wf = torch.zeros(64, 7, 8, 4, dtype=torch.float32, device=w.device)
    wf[:, :, :7, :3] = w.permute(0, 2, 3, 1)                       # [co][kh][kw][ci]
    wb = torch.zeros(4, 49, 64, dtype=torch.float32, device=w.device)
    wb[:3] = w.permute(1, 2, 3, 0).reshape(3, 49, 64)               # [ci][kh*7+kw][co]
    # 2-D on purpose: nn.Module.to(memory_format=channels_last) re-strides every 4-D buffer
    return wf.to(torch.bfloat16).reshape(64, 224).contiguous(), wb.to(torch.bfloat16).reshape(4, 49 * 64).contiguous()


class StemFunction(torch.autograd.Function):
    """Normalize -> conv7x7/2 -> BatchNorm(eval) -> ReLU -> maxpool3x3/2 of a frozen ResNet as four HIP kernels
    (csrc/adil_stem.hip), forward and input gradient.  x: (B,3,H,W) fp32/bf16 contiguous; returns the pooled
    activation as a (B,64,H/4,W/4) bf16 tensor in channels_last storage."""

    @staticmethod
    def forward(ctx, x, w_fwd, w_bwd, scale, shift, mean, inv_std):
        lib = _lib.load()
        x = _dev(x.contiguous(), "x")
        _dev(w_fwd, "w_fwd", torch.bfloat16), _dev(w_bwd, "w_bwd", torch.bfloat16)
        _dev(scale, "scale", torch.float32), _dev(shift, "shift", torch.float32)
        b, _, h, w = x.shape
        if h % 2 or w % 2:
            raise ValueError("stem kernels need even H and W")
        oh, ow = h // 2, w // 2
        ph, pw = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
        y1 = torch.empty((b, oh, ow, 64), dtype=torch.bfloat16, device=x.device)
        if b == 0:                                                    # empty batches pass through like plain torch modules
            ctx.meta = (0, h, w, x.dtype, tuple(inv_std))
            return torch.empty((0, ph, pw, 64), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
        _lib.check(lib.adil_stem_conv_fwd(_ptr(x), stream_dtype_code(x.dtype), _ptr(w_fwd), *mean, *inv_std, _ptr(scale),
                                          _ptr(shift), _ptr(y1), b, h, w, _stream()), "adil_stem_conv_fwd")
        p = torch.empty((b, ph, pw, 64), dtype=torch.bfloat16, device=x.device)
        idx = torch.empty((b, ph, pw, 64), dtype=torch.uint8, device=x.device)
        _lib.check(lib.adil_maxpool_fwd(_ptr(y1), _ptr(p), _ptr(idx), b, oh, ow, 64, _stream()), "adil_maxpool_fwd")
        ctx.save_for_backward(p, idx, w_bwd, scale)
        ctx.meta = (b, h, w, x.dtype, tuple(inv_std))
        return p.permute(0, 3, 1, 2)                                 # logical NCHW, channels_last storage

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        b, h, w, xdtype, inv_std = ctx.meta
        if b == 0:
            return (torch.empty((0, 3, h, w), dtype=xdtype, device=g.device),) + (None,) * 6
        p, idx, w_bwd, scale = ctx.saved_tensors
        oh, ow = h // 2, w // 2
        g = g.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()   # NHWC (a no-op for channels_last gradients)
        gy = torch.empty((b, oh, ow, 64), dtype=torch.bfloat16, device=g.device)
        _lib.check(lib.adil_stem_pool_bwd(_ptr(g), _ptr(idx), _ptr(p), _ptr(scale), _ptr(gy), b, oh, ow, 64, _stream()),
                   "adil_stem_pool_bwd")
        gx = torch.empty((b, 3, h, w), dtype=xdtype, device=g.device)
        _lib.check(lib.adil_stem_conv_bwd(_ptr(gy), _ptr(w_bwd), *inv_std, _ptr(gx), stream_dtype_code(xdtype), b, h, w,
                                          _stream()), "adil_stem_conv_bwd")
        return gx, None, None, None, None, None, None


def resnet_stem(x: Tensor, w_fwd: Tensor, w_bwd: Tensor, scale: Tensor, shift: Tensor, mean, inv_std) -> Tensor:
    return StemFunction.apply(x, w_fwd, w_bwd, scale, shift, tuple(float(m) for m in mean), tuple(float(s) for s in inv_std))


# --------------------------------------------------------------------------- #
class DictSynthFunction(torch.autograd.Function):
    """x + D v[index] as a differentiable op (the tensordot of adil.py:25 and its autograd backward).
    grad wrt v is dense (N,K) with zero rows outside `index`, exactly what autograd produces."""

    @staticmethod
    def forward(ctx, x, d, v, index):
        b = x.shape[0]
        vp = pack_codes(v.detach(), index, b)
        out = synth(x.detach().contiguous(), d.detach().contiguous(), vp, b)
        ctx.save_for_backward(d.detach(), vp, index if index is not None else torch.empty(0))
        ctx.meta = (b, v.shape, index is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        d, vp, index = ctx.saved_tensors
        b, vshape, has_index = ctx.meta
        need_x, need_d, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        g = g.contiguous()
        grad_d, gvb = None, None
        if need_d or need_v:
            grad_d, gvb = grad(g, d.contiguous(), vp, b, want_d=need_d, want_v=need_v)
        grad_v = None
        if need_v:
            grad_v = torch.zeros(vshape, dtype=torch.float32, device=g.device)
            if has_index:
                grad_v.index_add_(0, index.to(g.device), gvb)
            else:
                grad_v[:b] = gvb
        return (g if need_x else None), grad_d, grad_v, None


def dict_synth(x: Tensor, d: Tensor, v: Tensor, index: Optional[Tensor]) -> Tensor:
    if index is not None and not isinstance(index, torch.Tensor):
        index = torch.as_tensor(list(index), dtype=torch.int64)
    if index is not None:
        index = index.to(device=x.device, dtype=torch.int64)
    return DictSynthFunction.apply(x, d, v, index)
