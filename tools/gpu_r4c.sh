#!/bin/bash
# round 4, third GPU pass: fp8 persistent copy, stale-slab guard, two-rank rehearsal with the collective's record,
# classifier logit-error figures, ASR experiment 2
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fp8 or stale or adamw or synth" > $out/r4c_kernels.log 2>&1 || { tail -40 $out/r4c_kernels.log; exit 1; }
tail -2 $out/r4c_kernels.log
timeout -k 10 900 python -m pytest tests/test_gpu_adil.py -x -q -k "two_rank or single_rank or learn_dictionary or bench_transfer" > $out/r4c_adil.log 2>&1 || { tail -40 $out/r4c_adil.log; exit 1; }
tail -2 $out/r4c_adil.log
timeout -k 10 600 python -m pytest tests/test_gpu_stem.py -x -q -s -k "matches_unfused or gradient_matches_fp32" > $out/r4c_stem.log 2>&1 || { tail -40 $out/r4c_stem.log; exit 1; }
grep "logit error" $out/r4c_stem.log
K=100 timeout -k 10 300 python tools/bench_kernels.py > $out/r4c_micro_k100.log 2>&1 || { tail -20 $out/r4c_micro_k100.log; exit 1; }
timeout -k 10 400 python bench.py --atoms 100 --fp8-synth 1 --steps 20 --warmup 5 --cpu-baseline 0 > $out/r4c_bench_learn_k100_fp8.json 2> $out/r4c_bench_learn_k100_fp8.err || { tail -20 $out/r4c_bench_learn_k100_fp8.err; exit 1; }
timeout -k 10 400 python bench.py --atoms 100 --steps 20 --warmup 5 --cpu-baseline 0 > $out/r4c_bench_learn_k100.json 2> $out/r4c_bench_learn_k100.err || { tail -20 $out/r4c_bench_learn_k100.err; exit 1; }
python - <<'PY'
import json
for f in ("r4c_bench_learn_k100_fp8.json", "r4c_bench_learn_k100.json"):
    d = json.load(open("gpurun_out/" + f))
    print(f, round(d["value"]), {k: round(v * 1e3, 1) for k, v in d["kernels_ms_per_step"].items()})
PY
timeout -k 10 1000 python tests/experiments/exp_asr_gap2.py > $out/r4c_asr_gap2.json 2> $out/r4c_asr_gap2.err || { tail -30 $out/r4c_asr_gap2.err; exit 1; }
cat $out/r4c_asr_gap2.json
echo r4c done
