#!/bin/bash
# rocprofv3 evidence of one round, run on the GPU box from the repo root:  bash tools/run_profiles.sh r02
# kernel-trace/stats runs and the two PMC passes are separate invocations (counters are never combined with traces
# beyond --kernel-trace); the program itself follows `--` (no env/bash hop).
set -o pipefail
tag=${1:-r02}
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_learn -- python3 $root/bench.py --steps 20 --warmup 5 --cpu-baseline 0 > $out/prof_${tag}_learn.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_inference -- python3 $root/bench.py --mode inference --steps 20 --warmup 3 --cpu-baseline 0 > $out/prof_${tag}_inference.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_${tag}_FETCH_SIZE -- python3 $root/tools/bench_kernels.py > $out/pmc_${tag}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_${tag}_WRITE_SIZE -- python3 $root/tools/bench_kernels.py > $out/pmc_${tag}_write.log 2>&1 || exit 1
cd $root
python3 tools/prof_summary.py $out/prof_${tag}_learn $out/${tag}_learn_stats.md "bench.py (learn, ResNet-50, 512 images, 50 atoms, bf16 streams), rocprofv3 --kernel-trace --stats" > /dev/null
python3 tools/prof_summary.py $out/prof_${tag}_inference $out/${tag}_inference_stats.md "bench.py --mode inference (DDrague iteration, ResNet-50, 512 images, 50 atoms, bf16 streams, fp32 z), rocprofv3 --kernel-trace --stats" > /dev/null
python3 tools/step_breakdown.py $out/prof_${tag}_learn 23 > $out/${tag}_learn_step_breakdown.txt
python3 tools/step_breakdown.py $out/prof_${tag}_learn 3 > $out/${tag}_learn_step_breakdown_cached_labels.txt
python3 tools/pmc_traffic.py $out/pmc_${tag}_FETCH_SIZE $out/pmc_${tag}_WRITE_SIZE $out/${tag}_hbm_traffic.json > $out/${tag}_hbm_traffic.txt
echo profiles done
