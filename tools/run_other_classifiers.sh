#!/bin/bash
# One bench line per classifier of the reference CLI + the BASELINE.json ones (learn mode, bf16 streams), round tag in $1.
set -o pipefail
tag=${1:-r03}
out=gpurun_out
mkdir -p $out
: > $out/${tag}_other_classifiers.jsonl
run() {   # model batch atoms extra...
    m=$1; b=$2; k=$3; shift 3
    python3 bench.py --model $m --batch $b --atoms $k --steps 6 --warmup 2 --cpu-baseline 0 "$@" 2> $out/${tag}_other_$m.err | grep "^{" >> $out/${tag}_other_classifiers.jsonl || return 1
}
run densenet121 512 50 || exit 1
run vit_b_16 512 100 || exit 1
run vit_b_16 1024 100 --fp8-synth 1 || exit 1
run resnet18 512 10 || exit 1
run mobilenet 512 50 || exit 1
run googlenet 256 50 || exit 1
run inception 256 50 || exit 1
run vgg 256 50 || exit 1
echo other classifiers done
