#!/bin/bash
# The same variants inside bench.py (the images are evicted by the classifier between two synth launches, unlike in the
# micro-benchmark where a 154 MB batch can stay in the 256 MB Infinity Cache): kernel brackets of both modes.
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/dl_attack_on_imagenet_amd/lib
for v in ${VARIANTS:-0_0 0_2 2_2 18_18 0_0 0_2}; do
  echo "== load_store aux $v"
  ADIL_HIP_LIBRARY=$lib/exp/libadil_$v.so python3 $root/bench.py --steps 20 --warmup 5 --cpu-baseline 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('learn', round(j['value']), round(j['ms_per_step'],3), {k: round(v*1e3,1) for k,v in j['kernels_ms_per_step'].items()})" || exit 1
  ADIL_HIP_LIBRARY=$lib/exp/libadil_$v.so python3 $root/bench.py --mode inference --steps 20 --warmup 3 --cpu-baseline 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('inference', round(j['value']), round(j['ms_per_step'],3), {k: round(v*1e3,1) for k,v in j['kernels_ms_per_step'].items()})" || exit 1
done
