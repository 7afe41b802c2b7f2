#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests/test_gpu_parity_configs.py tests/test_gpu_adil.py -x -q -k "asr_parity or transfer or bench_ or ddrague or demo or main_cli or graphed" --durations=5 > $out/r4k_tests.log 2>&1; rc=$?
tail -12 $out/r4k_tests.log
exit $rc
