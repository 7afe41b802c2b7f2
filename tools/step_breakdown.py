"""Per-step GPU time breakdown from a rocprofv3 kernel trace of bench.py (steady-state step between two synth launches)."""
import collections, csv, glob, re, sys
path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'synth_mfma_kernel' in r['Kernel_Name']]
OURS = ("synth", "grad_", "adamw", "pack_codes", "l1ball", "zstep", "gather_images", "transpose_codes", "spd_inverse",
        "gram", "rightmul", "image_metrics")
def short(n):
    n = n.strip('"')
    m0 = re.match(r'(?:void )?(?:\(anonymous namespace\)::)?(\w+_kernel)\b', n)
    if m0 and any(k in m0.group(1) for k in OURS): return m0.group(1)      # "adamw_clamp_kernel" is not a ReLU clamp
    for key in ('stem_conv_fwd_kernel', 'stem_conv_bwd_kernel', 'stem_pool_bwd_kernel', 'maxpool_fwd_kernel',
                'pw_conv_fwd_kernel', 'pw_conv_bwd_kernel', 'conv3x3_kernel'):
        if key in n: return key
    if n.startswith('Cijk_') or 'Cijk_' in n: return 'hipBLASLt GEMM (Cijk_*)'
    for key, lab in (('batch_norm', 'batch_norm'), ('conv_bwd_data', 'conv_bwd_data'), ('igemm_bwd', 'conv_bwd_data'), ('conv_fwd', 'conv_fwd'),
                     ('igemm_fwd', 'conv_fwd'), ('direct_copy', 'copy'), ('CUDAFunctor_add', 'add'), ('SubTensorOp', 'miopen_subtensor(zero/bias)'),
                     ('threshold', 'relu_bwd'), ('clamp', 'relu/clamp'), ('max_pool', 'maxpool'), ('affine_act_fwd', 'affine_act_fwd'),
                     ('affine_act_bwd', 'affine_act_bwd'), ('batched_transpose', 'miopen_transpose'), ('fillBuffer', 'memset')):
        if key in n: return lab
    m = re.match(r'(?:void )?([\w:]+)', n); return (m.group(1) if m else n)[:44]
# optional argv[2]: which synth launch opens the step, counted from the end (default 3: the step before the last complete
# one).  bench.py's learn mode ends with `steps` extra steps of the cached-label variant, so the HEADLINE step of a run
# with --steps 20 is e.g. argv[2] = 23.
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = idx[-back], idx[-back + 1]
seg = rows[a:b]
wall = int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print(f"step: {len(seg)} kernels, wall {wall/1e6:.2f} ms, busy {busy/1e6:.2f} ms")
agg, cnt = collections.Counter(), collections.Counter()
for r in seg:
    k = short(r['Kernel_Name']); agg[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[k] += 1
for k, v in agg.most_common(18): print(f"{k:44s} {cnt[k]:5d} {v/1e6:8.3f} ms")
big = sorted(seg, key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']), reverse=True)[:8]
for r in big: print(f"  {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us  {short(r['Kernel_Name'])}  grid {r['Grid_Size_X']}")
