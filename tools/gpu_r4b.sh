#!/bin/bash
# round 4, second GPU pass: tile-order of the atom-split pairs, K = 50 regression check, the ASR-gap experiment
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "grad or full_size or transposed or deferred" > $out/r4b_kernels.log 2>&1 || { tail -40 $out/r4b_kernels.log; exit 1; }
tail -2 $out/r4b_kernels.log
ONLY=1 timeout -k 10 300 python tools/bench_kernels.py > $out/r4b_micro_k50.log 2>&1 || { tail -20 $out/r4b_micro_k50.log; exit 1; }
K=100 timeout -k 10 300 python tools/bench_kernels.py > $out/r4b_micro_k100.log 2>&1 || { tail -20 $out/r4b_micro_k100.log; exit 1; }
grep -h "grad d+v\|grad v only" $out/r4b_micro_k50.log $out/r4b_micro_k100.log
timeout -k 10 1000 python tests/experiments/exp_asr_gap.py > $out/r4b_asr_gap.json 2> $out/r4b_asr_gap.err || { tail -30 $out/r4b_asr_gap.err; exit 1; }
cat $out/r4b_asr_gap.json
echo r4b done
