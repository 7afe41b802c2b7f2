#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1100 python tests/experiments/exp_asr_gap6.py > $out/r4j_asr_gap6.json 2> $out/r4j_asr_gap6.err || { tail -30 $out/r4j_asr_gap6.err; exit 1; }
cat $out/r4j_asr_gap6.json
echo r4j done
