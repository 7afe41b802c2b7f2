"""LDS bank conflicts and the wave-cycle split of the hand-written kernels from ONE rocprofv3 SQ pass of tools/bench_kernels.py.

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 tools/bench_kernels.py
    python tools/pmc_sq.py gpurun_out/pmc_sq > profiles/rNN_lds_conflicts_and_stalls.md

As MI355X_MICROARCH.md defines them: SQ_LDS_BANK_CONFLICT = extra LDS-array cycles lost to conflicts, SQ_LDS_IDX_ACTIVE = all
LDS-array cycles; SQ_WAIT_ANY = a wave parked at s_waitcnt / a barrier, SQ_WAIT_INST_ANY = issue stalls (MFMA read-after-write,
busy pipe), SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and add up to about SQ_WAVE_CYCLES (all in quad-cycles)."""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_mfma import short


def main():
    d = sys.argv[1]
    path = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
    per = defaultdict(lambda: defaultdict(dict))
    for r in csv.DictReader(open(path)):
        s = short(r["Kernel_Name"])
        if s is not None:
            per[s][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    print(f"B={os.environ.get('B', 512)} K={os.environ.get('K', 50)}\n")
    print("| kernel | launches | LDS bank conflicts (share of LDS-array cycles) | waves parked (s_waitcnt / barrier) | issue stalls | issuing | "
          "of which VALU (incl. MFMA issue) | of which LDS |")
    print("|---|---|---|---|---|---|---|---|")
    for k in sorted(per):
        disp = list(per[k].values())
        disp = disp[len(disp) // 4:] if len(disp) >= 8 else disp

        def tot(c):
            return sum(x.get(c, 0.0) for x in disp)
        wc = tot("SQ_WAVE_CYCLES")
        if wc <= 0:
            continue
        idx = tot("SQ_LDS_IDX_ACTIVE")
        conf = f"{tot('SQ_LDS_BANK_CONFLICT') / idx * 100:.1f} %" if idx > 0 else "no LDS"
        print(f"| `{k}` | {len(disp)} | {conf} | {tot('SQ_WAIT_ANY') / wc * 100:.0f} % | {tot('SQ_WAIT_INST_ANY') / wc * 100:.0f} % | "
              f"{tot('SQ_ACTIVE_INST_ANY') / wc * 100:.0f} % | {tot('SQ_ACTIVE_INST_VALU') / wc * 100:.0f} % | {tot('SQ_ACTIVE_INST_LDS') / wc * 100:.0f} % |")


if __name__ == "__main__":
    main()
