#!/bin/bash
# round 4, first GPU pass: the new kernels' parity tests, then micro-benchmarks, then two bench lines
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "zstep or grad or full_size or transposed or deferred or library" > $out/r4a_kernels.log 2>&1 || { tail -40 $out/r4a_kernels.log; exit 1; }
tail -3 $out/r4a_kernels.log
timeout -k 10 900 python -m pytest tests/test_gpu_adil.py -x -q -k "ddrague or golden or stop or smoke or bench_inference" > $out/r4a_adil.log 2>&1 || { tail -40 $out/r4a_adil.log; exit 1; }
tail -3 $out/r4a_adil.log
timeout -k 10 300 python tools/bench_kernels.py > $out/r4a_micro_k50.log 2>&1 || { tail -20 $out/r4a_micro_k50.log; exit 1; }
K=100 timeout -k 10 300 python tools/bench_kernels.py > $out/r4a_micro_k100.log 2>&1 || { tail -20 $out/r4a_micro_k100.log; exit 1; }
K=100 ADIL_GRAD_ATOM_SPLIT=0 timeout -k 10 300 python tools/bench_kernels.py > $out/r4a_micro_k100_nosplit.log 2>&1 || { tail -20 $out/r4a_micro_k100_nosplit.log; exit 1; }
timeout -k 10 400 python bench.py --mode inference --steps 20 --warmup 3 --cpu-baseline 0 > $out/r4a_bench_inference.json 2> $out/r4a_bench_inference.err || { tail -20 $out/r4a_bench_inference.err; exit 1; }
timeout -k 10 400 python bench.py --mode inference --atoms 100 --steps 20 --warmup 3 --cpu-baseline 0 > $out/r4a_bench_inference_k100.json 2> $out/r4a_bench_inference_k100.err || { tail -20 $out/r4a_bench_inference_k100.err; exit 1; }
timeout -k 10 400 python bench.py --atoms 100 --steps 20 --warmup 5 --cpu-baseline 0 > $out/r4a_bench_learn_k100.json 2> $out/r4a_bench_learn_k100.err || { tail -20 $out/r4a_bench_learn_k100.err; exit 1; }
grep -h "DDrague\|z-step\|pack_codes from\|z D_dagger\|grad d+v (in-step" $out/r4a_micro_k50.log $out/r4a_micro_k100.log $out/r4a_micro_k100_nosplit.log
echo r4a done
