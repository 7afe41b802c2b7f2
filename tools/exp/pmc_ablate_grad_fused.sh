#!/bin/bash
# HBM bytes per launch of the fused-gradient ablation variants (two separate PMC passes), K = 50 and K = 100
set -e
root=$PWD; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $out/pmc_r4y_$C -- $root/tools/exp/bin/ablate_grad_fused 3 > $out/pmc_r4y_$C.log 2>&1
done
cd $root
python3 - <<'PY'
import csv, glob, os, re
from collections import defaultdict
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    path = sorted(glob.glob(f"gpurun_out/pmc_r4y_{C}/**/*counter_collection.csv", recursive=True))[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != C: continue
        m = re.search(r"ablate_kernel<(\d+)>", r["Kernel_Name"])
        if not m: continue
        key = (int(m.group(1)), int(r["Grid_Size"]) if "Grid_Size" in r else 0)
        a = acc[key]; a[0] += float(r["Counter_Value"]); a[1] += 1
    res[C] = {k: v[0] / v[1] for k, v in acc.items()}
print("| ABL flags | grid | fetch MB (2 x FETCH_SIZE KiB) | write MB | total MB |")
print("|---|---|---|---|---|")
for k in sorted(res["FETCH_SIZE"]):
    f = 2 * res["FETCH_SIZE"][k] * 1024 / 1e6; w = res["WRITE_SIZE"].get(k, 0) * 1024 / 1e6
    print(f"| {k[0]} | {k[1]} | {f:.1f} | {w:.1f} | {f + w:.1f} |")
PY
