"""GPU micro-benchmark of the hand-written kernels at the BASELINE shape (not part of the product)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_attack_on_imagenet_amd import ops

dev = torch.device("cuda")
B = int(os.environ.get("B", 512)); K = int(os.environ.get("K", 50)); S = int(os.environ.get("S", 224))
P = 3 * S * S
only = os.environ.get("ONLY", "")

def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

g0 = torch.Generator().manual_seed(0)
d = (-1 + 2 * torch.rand(3, S, S, K, generator=g0)).to(dev)
v = (torch.randn(B, K, generator=g0) * 0.01).to(dev)
vp = ops.pack_codes(v, None, B)
print(f"B={B} K={K} P={P}")
for dt, s in ((torch.bfloat16, 2), (torch.float32, 4)):
    x = torch.rand(B, 3, S, S, generator=g0).to(dev).to(dt)
    g = torch.randn(B, 3, S, S, generator=g0).to(dev).to(dt)
    out = torch.empty_like(x); gd = torch.empty_like(d)
    rows = []
    t = timeit(lambda: ops.synth(x, d, vp, B, out=out)); rows.append(("synth", t, 2 * B * P * s + P * K * 4))
    t = timeit(lambda: ops.grad(g, d, vp, B, grad_d=gd)); rows.append(("grad d+v (stand-alone: + transpose + reduce launches)", t, B * P * s + 2 * P * K * 4))
    # the learning step's own sequence (engine.DictionaryLearner): the transposed codes come from pack_codes, the slabs
    # of grad_v are summed inside adamw_l1ball_
    vv, mv, sv = v.clone(), torch.zeros_like(v), torch.zeros_like(v)
    pos = torch.full((B,), -1, dtype=torch.int32, device=dev)
    idx = torch.arange(B, device=dev)
    hv = ops.AdamWSchedule(0.01).next()
    def step_sequence():
        vq, vqt = ops.pack_codes(vv, idx, B, pos=pos, transposed=dt)
        _, gv = ops.grad(g, d, vq, B, grad_d=gd, vpt=vqt, defer_v=True)
        ops.adamw_l1ball_(vv, gv, pos, mv, sv, hv, 8 / 255, reset_pos=True)
    t = timeit(step_sequence); rows.append(("pack + grad d+v + adamw_l1ball (step sequence)", t, B * P * s + 2 * P * K * 4 + 9 * B * K * 4))
    vq, vqt = ops.pack_codes(vv, idx, B, transposed=dt)
    t = timeit(lambda: ops.grad(g, d, vq, B, grad_d=gd, vpt=vqt, defer_v=True)); rows.append(("grad d+v (in-step form: one launch)", t, B * P * s + 2 * P * K * 4))
    t = timeit(lambda: ops.grad(g, d, vp, B, want_v=False, grad_d=gd)); rows.append(("grad d only", t, B * P * s + P * K * 4))
    t = timeit(lambda: ops.grad(g, d, None, B, want_d=False)); rows.append(("grad v only", t, B * P * s + P * K * 4))
    t = timeit(lambda: out.copy_(x)); rows.append(("torch copy (ref)", t, 2 * B * P * s))
    for name, t, byt in rows:
        print(f"{str(dt):16s} {name:52s} {t*1e3:9.1f} us   {byt/t/1e6:8.1f} GB/s algorithmic", flush=True)
m, sq = torch.zeros_like(d), torch.zeros_like(d)
h = ops.AdamWSchedule(0.01).next()
gd = torch.randn_like(d)
t = timeit(lambda: ops.adamw_clamp_(d, gd, m, sq, h, -1.0, 1.0)); print(f"adamw_clamp(D)  {t*1e3:9.1f} us  {7*P*K*4/t/1e6:8.1f} GB/s")

# ---- one inference iteration of forward_supervised_DDrague (dictionary path only, fp32 z as the reference)
from dl_attack_on_imagenet_amd import engine
x32 = torch.rand(B, 3, S, S, generator=g0).to(dev)
g32 = torch.randn(B, 3, S, S, generator=g0).to(dev)
z = (torch.randn(B, 3, S, S, generator=g0) * 0.01).to(dev)
mz, sz = torch.zeros_like(z), torch.zeros_like(z)
pinv = engine.PseudoInverse(d)
dpt = pinv.d_pinv_t
hz = ops.AdamWSchedule(1e-2).next()
def iteration():
    _, vc = ops.grad(z, dpt, None, B, want_d=False, defer_v=True)   # v = z D_dagger^T (summed by pack_codes)
    xt = ops.synth(x32, d, ops.pack_codes(vc, None, B), B)      # x + D v
    _, gv = ops.grad(g32, d, None, B, want_d=False, defer_v=True)   # dL/dv = g D
    ops.zstep_(z, mz, sz, dpt, ops.pack_codes(gv, None, B), B, hz, -8 / 255, 8 / 255)
t = timeit(iteration, n=10)
alg = 9 * B * P * 4 + 4 * P * K * 4
print(f"DDrague iteration (fp32, dictionary path) {t*1e3:9.1f} us  {alg/t/1e6:8.1f} GB/s algorithmic ({alg/1e9:.2f} GB)")
gvp = ops.pack_codes(torch.randn(B, K, generator=g0).to(dev) * 0.01, None, B)
t = timeit(lambda: ops.zstep_(z, mz, sz, dpt, gvp, B, hz, -8 / 255, 8 / 255), n=10)
print(f"z-step (fp32) {t*1e3:9.1f} us  {(6 * B * P * 4 + P * K * 4)/t/1e6:8.1f} GB/s algorithmic")
t = timeit(lambda: engine.PseudoInverse(d), n=5); print(f"Gram + inverse + D_dagger (once per attack call) {t*1e3:9.1f} us")

# ---- round 4: the z-step that also leaves the next iteration's codes (adil_zstep_codes): no contraction launch for z D_dagger^T
nbytes = ops.zstep_codes_slab_bytes(B, P, K)
if nbytes:
    slabs = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    state = {"v": None}
    def iteration_fused():
        vpk = ops.pack_codes(state["v"], None, B) if state["v"] is not None else torch.zeros_like(gvp)
        xt = ops.synth(x32, d, vpk, B)
        _, gv = ops.grad(g32, d, None, B, want_d=False, defer_v=True)
        state["v"] = ops.zstep_codes_(z, mz, sz, dpt, ops.pack_codes(gv, None, B), B, hz, -8 / 255, 8 / 255, slabs)
    t = timeit(iteration_fused, n=10)
    alg = 8 * B * P * 4 + 3 * P * K * 4
    print(f"DDrague iteration, codes from the z-step (fp32, dictionary path) {t*1e3:9.1f} us  {alg/t/1e6:8.1f} GB/s algorithmic ({alg/1e9:.2f} GB)")
    t = timeit(lambda: ops.zstep_codes_(z, mz, sz, dpt, gvp, B, hz, -8 / 255, 8 / 255, slabs), n=10)
    print(f"z-step + next codes (fp32) {t*1e3:9.1f} us  {(6 * B * P * 4 + P * K * 4)/t/1e6:8.1f} GB/s algorithmic")
    t = timeit(lambda: ops.pack_codes(state["v"], None, B), n=10)
    print(f"pack_codes from the z-step's {state['v'].nslabs} slabs {t*1e3:9.1f} us")
    t = timeit(lambda: ops.grad(z, dpt, None, B, want_d=False, defer_v=True), n=10)
    print(f"z D_dagger^T as its own launch (what the fusion removes) {t*1e3:9.1f} us  {(B * P * 4 + P * K * 4)/t/1e6:8.1f} GB/s algorithmic")
