"""ADiL hot-path benchmark (driver contract: `python bench.py --gpus N --steps K --warmup W`, one JSON line).

Workload = BASELINE.json configs[1]: ADiL vs ResNet-50, 512 images per GPU, 50 atoms, bf16 image streams.
A "step" is ONE pass of the learning hot path over one batch (the loop body of learn_dictionary_a,
adil.py:168-191): clean pseudo-label forward, perturbation synthesis x + D v, classifier forward + backward,
grad_d / grad_v in one pass over dLoss/dx, AdamW(D)+clamp, AdamW(all rows of V)+l1-ball projection, and for
N > 1 one RCCL all-reduce(SUM) of grad_d.  The default K = 100 steps are the config's "100 inner iters".
Every step synthesises and evaluates B adversarial images per GPU, so value = N*B*K / seconds.
Inputs are synthetic (seeded U[0,1) images, seeded random-init ResNet-50) and resident in HBM before timing.

Extra objects on the JSON line: `roofline` for the dominant hand-written kernel (HIP events recorded on the
launch stream inside the timed region; algorithmic bytes per launch from DESIGN.md) and `cpu_baseline` (the CPU
oracle timed on a bounded sample of the same workload on this box's host cores; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--model", default="resnet50")
    p.add_argument("--batch", type=int, default=512, help="images per GPU (weak scaling)")
    p.add_argument("--atoms", type=int, default=50)
    p.add_argument("--image-size", type=int, default=224)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--loss", default="logits", choices=["logits", "ce"])
    p.add_argument("--channels-last", type=int, default=1)
    p.add_argument("--fold-bn", type=int, default=1,
                   help="fold the frozen classifier's eval-mode BatchNorms into its convolutions (same function)")
    p.add_argument("--fuse-bn-act", type=int, default=1,
                   help="ResNets: run eval-BatchNorm + residual + ReLU as one epilogue kernel per convolution "
                        "(zoo.FusedResNet; conv weights untouched; supersedes --fold-bn)")
    p.add_argument("--fuse-stem", type=int, default=1,
                   help="ResNets with --fuse-bn-act, bf16: Normalize + conv1 + bn1 + ReLU + maxpool (forward and input "
                        "gradient) as the hand-written stem kernels (csrc/adil_stem.hip)")
    p.add_argument("--pad-cin", type=int, default=8,
                   help="zero-pad the first conv's 3 input channels to this width (0 = off); see zoo.ChannelPaddedConv")
    p.add_argument("--cache-labels", type=int, default=0,
                   help="1: compute the (constant) clean pseudo-labels once instead of every step (reference quirk Q4)")
    p.add_argument("--cpu-baseline", type=int, default=1)
    p.add_argument("--cpu-batch", type=int, default=16)
    p.add_argument("--cpu-steps", type=int, default=2)
    return p.parse_args()


class KernelTimer:
    """Brackets every launch group of the hand-written kernels with HIP events on the launch stream."""

    GROUPS = ("pack_codes", "synth", "grad", "adamw_clamp_", "adamw_l1ball_")

    def __init__(self, ops):
        self.ops, self.enabled, self.records, self.empty = ops, False, {g: [] for g in self.GROUPS}, []
        for name in self.GROUPS:
            setattr(ops, name, self._wrap(name, getattr(ops, name)))

    def _wrap(self, name, fn):
        def timed(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.records[name].append((e0, e1))
            if name == "synth":                                  # an EMPTY bracket, recorded the same way on the same
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # stream: what
                c0.record()                                      # two back-to-back event markers cost by themselves
                c1.record()
                self.empty.append((c0, c1))
            return out
        return timed

    def empty_bracket_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.empty) / len(self.empty) if self.empty else 0.0

    def summary_ms(self, calibrated=False):
        """Mean bracket time per launch group (this is what the JSON reports: it agrees with rocprofv3's kernel
        duration to a few %, see profiles/).  `calibrated` subtracts the mean empty-bracket time — reported only as
        a bound on what the event markers themselves may contribute."""
        off = self.empty_bracket_ms() if calibrated else 0.0
        out = {}
        for name, evs in self.records.items():
            if evs:
                out[name] = max(sum(a.elapsed_time(b) for a, b in evs) / len(evs) - off, 0.0)
        return out


def algorithmic_bytes(B, P, K, N, s):
    """Per launch group, SURVEY.md §8(d): s = bytes/element of the image streams, D / V master fp32."""
    return {
        "synth": 2 * B * P * s + P * K * 4 + B * K * 4,            # read x, write x+Dv, read D (+ codes)
        "grad": B * P * s + P * K * 4 + P * K * 4 + B * K * 8,     # read g, read D, write grad_d (+ codes, grad_v)
        "adamw_clamp_": 7 * P * K * 4,                             # read p,g,m,s; write p,m,s
        "adamw_l1ball_": 7 * N * K * 4 + B * K * 4,
        "pack_codes": 2 * B * K * 4,
    }


def cpu_baseline(args, P_shape):
    """The oracle's learn_step_a (op-for-op the reference's sequence) on the host cores, bounded sample."""
    from oracle import adil_oracle as O
    from dl_attack_on_imagenet_amd import zoo
    threads = torch.get_num_threads()
    b, k = args.cpu_batch, args.atoms
    model = zoo.build_classifier(args.model, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(b, *P_shape, generator=g)
    d = -1 + 2 * torch.rand(*P_shape, k, generator=g)
    v = O.project_onto_l1_ball(torch.rand(b, k, generator=g), 8 / 255)
    od, ov = O.AdamWState(d, 0.01), O.AdamWState(v, 0.01)
    index = torch.arange(b)
    O.learn_step_a(model, x, index, d, v, od, ov, 8 / 255, args.loss, -1.0, 50.0)      # warm-up (allocators, MKL)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        O.learn_step_a(model, x, index, d, v, od, ov, 8 / 255, args.loss, -1.0, 50.0)
    dt = time.perf_counter() - t0
    return {"value": b * args.cpu_steps / dt, "unit": "adversarial images/sec", "cores": threads, "kind": "port",
            "sample": f"{args.cpu_steps} learning steps of the CPU oracle (oracle/adil_oracle.py learn_step_a, fp32, "
                      f"torch {torch.__version__} CPU, {threads} threads of {os.cpu_count()} logical cores), "
                      f"{args.model}, batch {b}, {k} atoms, {P_shape[1]}x{P_shape[2]} images; {dt:.1f} s"}


def main():
    args = parse_args()
    from dl_attack_on_imagenet_amd import dist as adist
    rank, world, local_rank = adist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ADiL hot path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch N>1 with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from dl_attack_on_imagenet_amd import engine, ops, zoo
    timer = KernelTimer(ops)
    sdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    s_bytes = 2 if args.dtype == "bf16" else 4
    B, K, S = args.batch, args.atoms, args.image_size
    shape = (3, S, S)
    P = 3 * S * S
    eps = 8 / 255

    model = zoo.build_classifier(args.model, seed=0, device=dev, dtype=sdtype, channels_last=bool(args.channels_last),
                                 fold_bn=bool(args.fold_bn), pad_input_channels=args.pad_cin,
                                 fuse_bn_act=bool(args.fuse_bn_act),
                                 fuse_stem=bool(args.fuse_stem and args.fuse_bn_act and args.dtype == "bf16"
                                                and zoo.canonical_name(args.model).startswith("resnet")))
    gen = torch.Generator().manual_seed(1000 + rank)                 # each rank owns different images (weak scaling)
    x = torch.rand(B, *shape, generator=gen).to(dev).to(sdtype).contiguous()
    gd0 = torch.Generator().manual_seed(7)                           # the same D0 on every rank
    d = (-1 + 2 * torch.rand(*shape, K, generator=gd0)).to(dev)
    v = ops.l1ball_project_(torch.rand(B, K, generator=gen).to(dev), eps)
    # ADIL_FORCE_REDUCER=1 exercises the RCCL path (process group, all-reduce of grad_d) even with one rank
    force = os.environ.get("ADIL_FORCE_REDUCER") == "1" and torch.distributed.is_initialized()
    reducer = adist.DictGradReducer() if (world > 1 or force) else None
    learner = engine.DictionaryLearner(d, v, eps, 0.01, args.loss, False, 50.0, reducer=reducer)
    index = torch.arange(B, device=dev)
    labels = engine.predict(model, x) if args.cache_labels else None

    def step():
        return learner.step(model, x, index, labels)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    timer.enabled = True
    t0 = time.perf_counter()
    fooled = None
    for _ in range(args.steps):
        _, fooled = step()
    sync()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax)

    fool_rate = float(fooled) / B
    kern_ms = timer.summary_ms()
    alg = algorithmic_bytes(B, P, K, B, s_bytes)
    dom = max((k for k in kern_ms if k in alg), key=lambda k: kern_ms[k])
    achieved = alg[dom] / (kern_ms[dom] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")      # filled from a separate rocprofv3 --pmc run
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom)
        except Exception:
            traffic = None
    dict_ms = sum(kern_ms.values())
    out = {
        "metric": "adversarial images/sec (ADiL learning step, classifier included)",
        "value": world * B * args.steps / elapsed,
        "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"ADiL learn_dictionary_a step vs {args.model}, {B} images/GPU, {K} atoms, "
                               f"{S}x{S}, {args.dtype} image streams + fp32 D/V master, loss={args.loss}, "
                               f"{'cached' if args.cache_labels else 'recomputed'} pseudo-labels (2 fwd + 1 bwd)",
                   "classifier": f"random-init {args.model}, frozen, eval; channels_last={args.channels_last}, "
                                 f"bn_act_epilogue_fused={args.fuse_bn_act}, bn_folded={args.fold_bn}, "
                                 f"first_conv_cin_padded_to={args.pad_cin} (unused with stem kernels), stem_kernels={args.fuse_stem}, "
                                 f"pointwise_convs=fused GEMM+epilogue kernels (fwd and input gradient), "
                                 f"stride1_3x3_convs=adil_conv3x3",
                   "global_batch": world * B, "atoms": K, "inner_iters": args.steps,
                   "parallelism": f"dp{world}: images+codes sharded, D replicated, 1 all-reduce(grad_d)/step",
                   "train_fooling_rate_last_step": fool_rate},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": kern_ms[dom],
                     "empty_event_bracket_ms": timer.empty_bracket_ms()},
        "kernels_ms_per_step": kern_ms,
        "dictionary_path_ms_per_step": dict_ms,
        "dictionary_path_algorithmic_GBps": sum(alg[k] for k in kern_ms if k in alg) / (dict_ms * 1e-3) / 1e9,
    }
    if rank == 0 and world == 1 and args.cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, shape)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
