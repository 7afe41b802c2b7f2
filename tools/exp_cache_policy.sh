#!/bin/bash
# Cache-policy experiment for the streaming buffer accesses of synth / z-step (aux bits of the raw buffer
# instructions: 1 = sc0, 2 = nt, 16 = sc1).  `bash tools/exp_cache_policy.sh build` cross-compiles one library per
# (load, store) policy into dl_attack_on_imagenet_amd/lib/exp/ (here, no GPU needed); `... run` on the GPU box times
# tools/bench_kernels.py with each of them through ADIL_HIP_LIBRARY.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/dl_attack_on_imagenet_amd/lib
variants="0:0 2:2 0:2 2:0 0:18 18:18 1:0"
if [ "$1" = build ]; then
  python3 -m dl_attack_on_imagenet_amd.build > /dev/null
  for v in $variants; do
    ld=${v%%:*}; st=${v##*:}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DADIL_AUX_LD=$ld -DADIL_AUX_ST=$st -c $root/dl_attack_on_imagenet_amd/csrc/adil_contract.hip \
      -o $lib/exp/contract_${ld}_${st}.o -I $root/include -I $root/dl_attack_on_imagenet_amd/csrc &
  done
  wait
  for v in $variants; do
    ld=${v%%:*}; st=${v##*:}
    hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/exp/libadil_${ld}_${st}.so $lib/exp/contract_${ld}_${st}.o $lib/adil_convs.o $lib/adil_stem.o $lib/adil_update.o
  done
  ls -la $lib/exp/*.so
else
  for v in $variants; do
    ld=${v%%:*}; st=${v##*:}
    echo "== load aux $ld, store aux $st"
    ADIL_HIP_LIBRARY=$lib/exp/libadil_${ld}_${st}.so python3 $root/tools/bench_kernels.py 2>/dev/null | grep -E "synth|DDrague|z-step" || exit 1
  done
fi
