#!/bin/bash
# rocprofv3 evidence of one round, run on the GPU box from the repo root:  bash tools/run_profiles.sh r03
# kernel-trace/stats runs and the PMC passes are separate invocations (counters are never combined with traces beyond
# --kernel-trace); the program itself follows `--` (no env/bash hop).  Every table comes from the SAME tree: the hash of
# the kernel sources is written next to them and into hbm_traffic*.json (bench.py reports roofline.traffic only for it).
set -o pipefail
tag=${1:-r03}
phase=${2:-all}          # stats | pmc | mfma | sq | lines | all   (one gpurun call holds 20 minutes: stats and pmc+lines fit one each)
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out
mkdir -p $out
python3 -c "import sys; sys.path.insert(0, '$root'); from dl_attack_on_imagenet_amd.build import source_hash; print(source_hash())" > $out/${tag}_kernel_source_hash.txt || exit 1
cd /tmp
stats() {   # name, description, bench arguments...
    name=$1; desc=$2; shift 2
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_$name -- python3 $root/bench.py "$@" --cpu-baseline 0 > $out/prof_${tag}_$name.log 2>&1 || return 1
    python3 $root/tools/prof_summary.py $out/prof_${tag}_$name $out/${tag}_${name}_bench_stats.md "$desc, rocprofv3 --kernel-trace --stats" > /dev/null
}
if [ "$phase" = "stats" ] || [ "$phase" = "all" ]; then
stats learn "bench.py (learn, ResNet-50, 512 images, 50 atoms, bf16 streams)" --steps 20 --warmup 5 || exit 1
stats inference "bench.py --mode inference (DDrague iteration, ResNet-50, 512 images, 50 atoms, bf16 streams, fp32 z)" --mode inference --steps 20 --warmup 3 || exit 1
stats learn_k100 "bench.py --atoms 100 (learn, ResNet-50, 512 images, 100 atoms = the reference's n_atoms, bf16 streams)" --atoms 100 --steps 20 --warmup 5 || exit 1
stats inference_k100 "bench.py --mode inference --atoms 100 (DDrague iteration, 100 atoms)" --mode inference --atoms 100 --steps 20 --warmup 3 || exit 1
stats transfer "bench.py --mode transfer (configs[3]: full attack(x, y) of 100 DDrague iterations, 100 atoms, against ResNet-50 + six targets scored, 512 images per batch)" --mode transfer --steps 2 --warmup 1 || exit 1
cd $root
python3 tools/step_breakdown.py $out/prof_${tag}_learn 23 > $out/${tag}_learn_step_breakdown.txt
python3 tools/step_breakdown.py $out/prof_${tag}_learn 3 > $out/${tag}_learn_step_breakdown_recomputed_labels.txt
# gpurun merges at most 64 MiB back: the per-launch traces (tens of MB each) are condensed above; the per-kernel stats
# CSVs of rocprofv3 itself stay next to the tables
find $out -path "*prof_${tag}_*" -name "*kernel_trace.csv" -delete
fi
if [ "$phase" = "pmc" ] || [ "$phase" = "all" ]; then
cd /tmp
for K in 50 100; do
  for C in FETCH_SIZE WRITE_SIZE; do
    K=$K rocprofv3 --pmc $C --kernel-trace --output-format csv -d $out/pmc_${tag}_k${K}_$C -- python3 $root/tools/bench_kernels.py > $out/pmc_${tag}_k${K}_$C.log 2>&1 || exit 1
  done
done
cd $root
K=50 python3 tools/pmc_traffic.py $out/pmc_${tag}_k50_FETCH_SIZE $out/pmc_${tag}_k50_WRITE_SIZE $out/${tag}_hbm_traffic.json > $out/${tag}_hbm_traffic.txt
K=100 python3 tools/pmc_traffic.py $out/pmc_${tag}_k100_FETCH_SIZE $out/pmc_${tag}_k100_WRITE_SIZE $out/${tag}_hbm_traffic_k100.json > $out/${tag}_hbm_traffic_k100.txt
cp $out/${tag}_hbm_traffic.json profiles/hbm_traffic.json        # on the GPU box: the bench lines below then carry roofline.traffic
cp $out/${tag}_hbm_traffic_k100.json profiles/hbm_traffic_k100.json
fi
if [ "$phase" = "mfma" ] || [ "$phase" = "all" ]; then
# matrix-pipe utilisation: one SQ + GRBM pass per atom count (counters only, with --kernel-trace for the durations)
cd /tmp
rocprofv3 -L > $out/${tag}_counters_avail.txt 2>&1
ctrs="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for K in 50 100; do
  K=$K rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/pmc_${tag}_k${K}_mfma -- python3 $root/tools/bench_kernels.py > $out/pmc_${tag}_k${K}_mfma.log 2>&1 || exit 1
done
cd $root
K=50 python3 tools/pmc_mfma.py $out/pmc_${tag}_k50_mfma $out/${tag}_mfma_utilisation.json > $out/${tag}_mfma_utilisation_k50.md || exit 1
K=100 python3 tools/pmc_mfma.py $out/pmc_${tag}_k100_mfma $out/${tag}_mfma_utilisation_k100.json > $out/${tag}_mfma_utilisation_k100.md || exit 1
cp $out/${tag}_mfma_utilisation.json profiles/mfma_utilisation.json
cp $out/${tag}_mfma_utilisation_k100.json profiles/mfma_utilisation_k100.json
find $out -path "*pmc_${tag}_k*_mfma*" -name "*kernel_trace.csv" -delete
fi
if [ "$phase" = "sq" ] || [ "$phase" = "all" ]; then
# LDS bank conflicts + where the wave cycles go: one pass of 8 SQ counters per atom count
cd /tmp
sq="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
for K in 50 100; do
  K=$K rocprofv3 --pmc $sq --kernel-trace --output-format csv -d $out/pmc_${tag}_k${K}_sq -- python3 $root/tools/bench_kernels.py > $out/pmc_${tag}_k${K}_sq.log 2>&1 || exit 1
done
cd $root
K=50 python3 tools/pmc_sq.py $out/pmc_${tag}_k50_sq > $out/${tag}_lds_conflicts_and_stalls_k50.md || exit 1
K=100 python3 tools/pmc_sq.py $out/pmc_${tag}_k100_sq > $out/${tag}_lds_conflicts_and_stalls_k100.md || exit 1
find $out -path "*pmc_${tag}_k*_sq*" -name "*kernel_trace.csv" -delete
fi
if [ "$phase" = "lines" ] || [ "$phase" = "all" ]; then
cd $root
python3 bench.py > $out/${tag}_bench_line.json 2> $out/${tag}_bench_line.err || exit 1
python3 bench.py --mode inference > $out/${tag}_bench_line_inference.json 2> $out/${tag}_bench_line_inference.err || exit 1
python3 bench.py --mode transfer > $out/${tag}_bench_line_transfer.json 2> $out/${tag}_bench_line_transfer.err || exit 1
fi
echo profiles $phase done
