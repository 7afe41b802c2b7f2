#!/bin/bash
# round 4, seventh GPU pass: cooperative slab sums (parity + timing), anatomy of a bad bf16 run over seeds
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "transposed or deferred or stale or pack_codes or l1ball or adamw or stop or zstep" > $out/r4g_kernels.log 2>&1 || { tail -40 $out/r4g_kernels.log; exit 1; }
tail -2 $out/r4g_kernels.log
timeout -k 10 900 python -m pytest tests/test_gpu_adil.py -x -q -k "golden or ddrague or learn or graphed or smoke" > $out/r4g_adil.log 2>&1 || { tail -40 $out/r4g_adil.log; exit 1; }
tail -2 $out/r4g_adil.log
timeout -k 10 300 python tools/bench_kernels.py > $out/r4g_micro_k50.log 2>&1 || { tail -20 $out/r4g_micro_k50.log; exit 1; }
grep -h "pack\|DDrague" $out/r4g_micro_k50.log
timeout -k 10 900 python tests/experiments/exp_asr_gap4.py > $out/r4g_asr_gap4.json 2> $out/r4g_asr_gap4.err || { tail -30 $out/r4g_asr_gap4.err; exit 1; }
cat $out/r4g_asr_gap4.json
echo r4g done
