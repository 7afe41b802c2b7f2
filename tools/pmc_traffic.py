"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM traffic of the hand-written kernels.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/bench_kernels.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 tools/bench_kernels.py
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/hbm_traffic.json

Units / corrections as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced read stream, so it is doubled; WRITE_SIZE is exact.
The run also contains torch's copy kernel of a known size, printed as a calibration line."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(dirname, counter):
    path = sorted(glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True))[0]
    acc = defaultdict(lambda: [0.0, 0])
    prev = ""
    for r in sorted((r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"])):
        name = r["Kernel_Name"]
        if "grad_v_reduce_kernel" in name:        # one reduce kernel serves every producer: keep them apart by the launch before
            name = "grad_v_reduce_kernel after " + prev
        else:
            prev = name
        a = acc[name]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, path


def short(name):
    if name.startswith("grad_v_reduce_kernel after "):
        return "grad_v_reduce after " + (short(name[len("grad_v_reduce_kernel after "):]) or "?")
    m = re.search(r"(synth_mfma_kernel|grad_fused_mfma_kernel|grad_fused_f32_kernel|grad_v_f32_kernel|grad_d_mfma_kernel|grad_v_mfma_kernel|grad_v_reduce_kernel|adamw_clamp_kernel|"
                  r"adamw_l1ball_kernel|pack_codes_kernel|transpose_codes_kernel|zstep_mfma_kernel|zstep_codes_kernel|gather_images_kernel)<([^>]*)", name)
    if m:
        return f"{m.group(1)}<{m.group(2).split('>')[0]}>"
    if "direct_copy_kernel" in name or "copy" in name.lower():
        return "torch_copy"
    return None


def main():
    fdir, wdir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, fp = per_kernel(fdir, "FETCH_SIZE")
    write, wp = per_kernel(wdir, "WRITE_SIZE")
    rows = {}
    for name in set(fetch) | set(write):
        s = short(name)
        if s is None:
            continue
        f_kib, w_kib = fetch.get(name, 0.0), write.get(name, 0.0)
        rows[s] = {"fetch_bytes_corrected": 2 * f_kib * 1024, "write_bytes": w_kib * 1024,
                   "hbm_bytes": 2 * f_kib * 1024 + w_kib * 1024, "raw_FETCH_SIZE_KiB": f_kib, "raw_WRITE_SIZE_KiB": w_kib}
    # the learning step's grad launch group = the fused kernel (ABI 6: the code transpose rides in pack_codes, the slab
    # reduction in adamw_l1ball; stand-alone ops.grad calls of the micro-benchmark still launch both helpers, and the
    # reduce launches that follow the fused kernel there are attributed to it)
    groups = {"synth": [k for k in rows if k.startswith("synth_mfma")],
              # (the last template argument false = the grad_d-only instantiation the micro-benchmark also launches: not the step's pass)
              "grad": [k for k in rows if k.startswith("grad_fused_mfma") and not k.endswith("false>")],
              "adamw_clamp_": [k for k in rows if k.startswith("adamw_clamp")],
              "adamw_l1ball_": [k for k in rows if k.startswith("adamw_l1ball")],
              "zstep_": [k for k in rows if k.startswith("zstep_mfma")],
              "zstep_codes_": [k for k in rows if k.startswith("zstep_codes")],
              "grad[z D_dagger^T]": [k for k in rows if k.startswith(("grad_v_mfma_kernel<float", "grad_v_f32_kernel"))],
              "pack_codes": [k for k in rows if k.startswith("pack_codes")]}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dl_attack_on_imagenet_amd.build import source_hash
    result = {"_source": {"fetch": fp, "write": wp, "correction": "hbm = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950)",
                          # the build these counters were measured on: bench.py reports roofline.traffic only for this hash
                          "kernel_source_hash": source_hash(), "atoms": int(os.environ.get("K", 50)),
                          "batch": int(os.environ.get("B", 512))},
              "_kernels": rows}
    for g, ks in groups.items():
        # bf16 instantiations only (the bench workload): template arg 't' = unsigned short
        sel = [k for k in ks if "<unsigned short" in k or "<" not in k or
               g in ("adamw_clamp_", "adamw_l1ball_", "pack_codes", "zstep_", "zstep_codes_", "grad[z D_dagger^T]")]
        if sel:
            result[g] = sum(rows[k]["hbm_bytes"] for k in sel)
    json.dump(result, open(out, "w"), indent=1)
    for k, v in sorted(rows.items()):
        print(f"{k:70s} fetch(x2) {v['fetch_bytes_corrected'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:9.1f} MB")
    print({k: v for k, v in result.items() if not k.startswith("_")})


if __name__ == "__main__":
    main()
