#!/bin/bash
# does a library switch change the bf16 configuration's ASR?  seed 6033 (95.6 % by default on two boxes), 2048 images
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
VARIANTS=default timeout -k 10 500 python tests/experiments/exp_asr_gap6.py > $out/r4l_default.json 2> $out/r4l_default.err || { tail -20 $out/r4l_default.err; exit 1; }
cat $out/r4l_default.json; grep "ms per step" $out/r4l_default.err | tail -1
VARIANTS=default MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout -k 10 500 python tests/experiments/exp_asr_gap6.py > $out/r4l_nowino.json 2> $out/r4l_nowino.err || { tail -20 $out/r4l_nowino.err; exit 1; }
cat $out/r4l_nowino.json; grep "ms per step" $out/r4l_nowino.err | tail -1
VARIANTS=default MIOPEN_DEBUG_CONV_WINOGRAD=0 MIOPEN_DEBUG_CONV_DIRECT=0 timeout -k 10 500 python tests/experiments/exp_asr_gap6.py > $out/r4l_igemm.json 2> $out/r4l_igemm.err || { tail -20 $out/r4l_igemm.err; exit 1; }
cat $out/r4l_igemm.json; grep "ms per step" $out/r4l_igemm.err | tail -1
echo r4l done
