"""Matrix-pipe utilisation of the hand-written kernels from ONE rocprofv3 PMC pass of tools/bench_kernels.py.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE [SQ_INSTS_MFMA ...] --kernel-trace --output-format csv \
        -d gpurun_out/pmc_mfma -- python3 tools/bench_kernels.py
    python tools/pmc_mfma.py gpurun_out/pmc_mfma [profiles/mfma_utilisation.json] > profiles/rNN_mfma_utilisation.md

The optional JSON (per launch group of bench.py, with the hash of the kernel sources it was measured on) is what
`bench.py` reports as `roofline.mfma_utilisation` — like `roofline.traffic`, only for the build it was collected from.

Units as MI355X_MICROARCH.md states them: SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles, summed over every SIMD of the
chip (= 32 x the number of v_mfma_f32_32x32x16_bf16 wave-instructions); GRBM_GUI_ACTIVE is the sum over the 8 XCDs, so
the effective clock of a dispatch is GRBM_GUI_ACTIVE / 8 / duration (reads high on dispatches well below 0.3 ms).
utilisation = busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the fraction of the matrix pipes' cycles that were busy
at the clock the kernel actually ran at; `of dense peak` prices the same MFMAs against 2.5 PFLOP/s at 2.4 GHz."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

SIMDS = 256 * 4
PEAK_TFLOPS = 2500.0


def short(name):
    m = re.search(r"(synth_mfma_kernel|grad_fused_mfma_kernel|grad_fused_f32_kernel|grad_v_f32_kernel|grad_d_mfma_kernel|"
                  r"grad_v_mfma_kernel|zstep_mfma_kernel|zstep_codes_kernel|gram_mfma_kernel|dict_rightmul_mfma_kernel)<([^>]*)", name)
    if not m:
        return None
    args = m.group(2).replace("unsigned short", "bf16").replace("(bool)1", "1").replace("(bool)0", "0").replace("true", "1").replace("false", "0")
    return f"{m.group(1)}<{args}>"


def main():
    d = sys.argv[1]
    path = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    dur = {}
    if rows and "Start_Timestamp" in rows[0]:
        for r in rows:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    else:
        kt = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
        for r in csv.DictReader(open(kt)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    per = defaultdict(lambda: defaultdict(dict))           # kernel -> dispatch -> counter -> value
    for r in rows:
        s = short(r["Kernel_Name"])
        if s is not None:
            per[s][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    counters = sorted({c for k in per.values() for v in k.values() for c in v})
    print(f"counters in this pass: {', '.join(counters)}; B={os.environ.get('B', 512)} K={os.environ.get('K', 50)}\n")
    print("| kernel | launches | avg µs (this pass) | MFMA busy cycles / launch | = 32x32x16 MFMAs / launch | effective clock | "
          "matrix-pipe utilisation | MFMA TFLOP/s (of 2 500 dense bf16) |")
    print("|---|---|---|---|---|---|---|---|")
    utils = {}
    for k in sorted(per):
        disp = [x for x in per[k] if x in dur and "SQ_VALU_MFMA_BUSY_CYCLES" in per[k][x]]
        disp = disp[len(disp) // 4:] if len(disp) >= 8 else disp          # drop warm-up launches
        if not disp:
            continue
        n = len(disp)
        t = sum(dur[x] for x in disp) / n
        busy = sum(per[k][x]["SQ_VALU_MFMA_BUSY_CYCLES"] for x in disp) / n
        gui = sum(per[k][x].get("GRBM_GUI_ACTIVE", 0.0) for x in disp) / n
        clock = gui / 8 / t if gui else float("nan")
        util = busy / (SIMDS * gui / 8) if gui else float("nan")
        mfmas = busy / 32
        tflops = mfmas * 32 * 32 * 16 * 2 / t / 1e12
        utils[k] = util
        print(f"| `{k}` | {n} | {t * 1e6:.1f} | {busy / 1e6:.2f} M | {mfmas / 1e6:.3f} M | {clock / 1e9:.2f} GHz | "
              f"{util * 100:.1f} % | {tflops:.0f} ({tflops / PEAK_TFLOPS * 100:.0f} %) |")
    if len(sys.argv) > 2:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from dl_attack_on_imagenet_amd.build import source_hash
        groups = {"synth": "synth_mfma_kernel<bf16", "grad": "grad_fused_mfma_kernel<bf16", "zstep_": "zstep_mfma_kernel", "zstep_codes_": "zstep_codes_kernel",
                  "grad[z D_dagger^T]": "grad_v_f32_kernel"}
        rec = {"_source": {"counters": path, "kernel_source_hash": source_hash(), "atoms": int(os.environ.get("K", 50)),
                           "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)"},
               "_kernels": utils}
        for g, prefix in groups.items():
            sel = [u for k, u in sorted(utils.items()) if k.startswith(prefix) and u == u]
            if sel:
                rec[g] = sel[0]
        json.dump(rec, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
