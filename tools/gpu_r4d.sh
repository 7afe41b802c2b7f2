#!/bin/bash
# round 4, fourth GPU pass: z-step code contraction on 16x16x32 MFMAs, XCD placement probe, K = 100 traffic counters
set -o pipefail
out=gpurun_out
root=$(pwd)
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "zstep" > $out/r4d_kernels.log 2>&1 || { tail -40 $out/r4d_kernels.log; exit 1; }
tail -2 $out/r4d_kernels.log
timeout -k 10 900 python -m pytest tests/test_gpu_adil.py -x -q -k "ddrague or golden or stop" > $out/r4d_adil.log 2>&1 || { tail -40 $out/r4d_adil.log; exit 1; }
tail -2 $out/r4d_adil.log
hipcc --offload-arch=gfx950 -O2 tools/exp/xcc_map.hip -o /tmp/xcc_map > /dev/null 2>&1 && timeout -k 5 60 /tmp/xcc_map > $out/r4d_xcc_map.txt 2>&1
cat $out/r4d_xcc_map.txt
timeout -k 10 300 python tools/bench_kernels.py > $out/r4d_micro_k50.log 2>&1 || { tail -20 $out/r4d_micro_k50.log; exit 1; }
K=100 timeout -k 10 300 python tools/bench_kernels.py > $out/r4d_micro_k100.log 2>&1 || { tail -20 $out/r4d_micro_k100.log; exit 1; }
grep -h "z-step\|DDrague" $out/r4d_micro_k50.log $out/r4d_micro_k100.log
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  K=100 timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $root/$out/pmc_r4d_k100_$C -- python3 $root/tools/bench_kernels.py > $root/$out/pmc_r4d_k100_$C.log 2>&1 || { tail -5 $root/$out/pmc_r4d_k100_$C.log; exit 1; }
done
cd $root
K=100 python3 tools/pmc_traffic.py $out/pmc_r4d_k100_FETCH_SIZE $out/pmc_r4d_k100_WRITE_SIZE $out/r4d_hbm_traffic_k100.json > $out/r4d_hbm_traffic_k100.txt
grep -i "grad_fused\|grad_v_mfma\|zstep\|grad_v_f32" $out/r4d_hbm_traffic_k100.txt
find $out -path "*pmc_r4d_*" -name "*.csv" -size +20M -delete
echo r4d done
