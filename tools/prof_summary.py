"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` run into a short table.
usage: python tools/prof_summary.py <dir with *_kernel_stats.csv> [out.md] [title]"""
import csv
import glob
import os
import re
import sys


OURS = ("synth", "grad_", "adamw", "pack_codes", "l1ball", "l2ball", "ista", "atom_", "gram", "rightmul", "image_metrics",
        "sum_partials", "fused", "zstep", "gather_images", "spd_inverse", "transpose_codes", "stem_", "maxpool_fwd",
        "pw_conv", "conv3x3", "affine_act")


def short(name: str) -> str:
    name = name.strip('"')
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void\s+)?((?:\w+::)*)(\w+)[<(]", name)
    if m and m.group(2).endswith("_kernel") and any(s in m.group(2) for s in OURS):
        return (m.group(1) + m.group(2))[:70]        # this repo's kernels first: "adamw_clamp_kernel" is not at::clamp
    if "kernel_grouped_conv_bwd_data" in name: return "ck::grouped_conv_bwd_data_xdl"
    if "kernel_grouped_conv_fwd" in name: return "ck::grouped_conv_fwd_xdl"
    if "batch_norm_elementwise_backward_eval" in name: return "at::batch_norm_elementwise_backward_eval"
    if "direct_copy_kernel" in name: return "at::direct_copy_kernel (layout/dtype copies)"
    for key in ("threshold_kernel", "CUDAFunctor_add", "max_pool", "avg_pool", "clamp", "fill"):
        if key in name: return "at::" + key
    if m: return (m.group(1) + m.group(2))[:70]
    return re.sub(r"\(.*", "", name)[:70]


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    title = sys.argv[3] if len(sys.argv) > 3 else d
    path = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0, float("inf"), 0.0])
        a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]); a[2] = min(a[2], float(r["MinNs"])); a[3] = max(a[3], float(r["MaxNs"]))
    tot = sum(a[1] for a in agg.values())
    lines = [f"# {title}", "", f"source: `{os.path.basename(path)}` (rocprofv3 --kernel-trace --stats), total GPU kernel time {tot / 1e6:.1f} ms", "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    ours = lambda k: k.endswith("_kernel") and any(s in k for s in OURS)
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    shown = [kv for kv in items if ours(kv[0])] + [kv for kv in items if not ours(kv[0])][:12]
    for k, a in shown:
        tag = "**" if ours(k) else ""
        lines.append(f"| {tag}{k}{tag} | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / tot:.2f} |")
    text = "\n".join(lines) + "\n"
    print(text)
    if out:
        open(out, "w").write(text)


if __name__ == "__main__":
    main()
