#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1100 python tests/experiments/exp_asr_gap6.py > $out/r4j_asr_gap5.json 2> $out/r4j_asr_gap5.err || { tail -30 $out/r4j_asr_gap5.err; exit 1; }
cat $out/r4j_asr_gap5.json
echo r4i done
