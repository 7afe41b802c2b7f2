#!/bin/bash
# round 4, fifth GPU pass: the distribution of the ASR over independent pipeline runs (experiment 3)
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1150 python tests/experiments/exp_asr_gap3.py > $out/r4e_asr_gap3.json 2> $out/r4e_asr_gap3.err || { tail -30 $out/r4e_asr_gap3.err; exit 1; }
cat $out/r4e_asr_gap3.json
echo r4e done
