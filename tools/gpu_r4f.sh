#!/bin/bash
# round 4, sixth GPU pass: pipelined grad_v_f32 kernel (parity + timing), anatomy of a bad bf16 run (experiment 4)
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "grad_v_fp32 or many_atoms or fused_fp32 or transposed or gram or pseudo" > $out/r4f_kernels.log 2>&1 || { tail -40 $out/r4f_kernels.log; exit 1; }
tail -2 $out/r4f_kernels.log
timeout -k 10 300 python tools/bench_kernels.py > $out/r4f_micro_k50.log 2>&1 || { tail -20 $out/r4f_micro_k50.log; exit 1; }
K=100 timeout -k 10 300 python tools/bench_kernels.py > $out/r4f_micro_k100.log 2>&1 || { tail -20 $out/r4f_micro_k100.log; exit 1; }
grep -h "grad v only\|z D_dagger" $out/r4f_micro_k50.log $out/r4f_micro_k100.log
timeout -k 10 900 python tests/experiments/exp_asr_gap4.py > $out/r4f_asr_gap4.json 2> $out/r4f_asr_gap4.err || { tail -30 $out/r4f_asr_gap4.err; exit 1; }
cat $out/r4f_asr_gap4.json
echo r4f done
