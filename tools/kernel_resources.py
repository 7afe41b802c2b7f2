"""Per-kernel register / LDS / occupancy table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py dl_attack_on_imagenet_amd/csrc/adil_contract.hip [filter-substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-I", "include", "-I", "dl_attack_on_imagenet_amd/csrc", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": m.group(2)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
if "error" in out and not rows:
    print(out)
    sys.exit(1)
demangle = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':90s} vgpr agpr spill occ scratch")
for r, d in zip(rows, demangle):
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    if flt in d:
        print(f"{d[:90]:90s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('spill', 0):5d} {r.get('occ', 0):3d} {r.get('scratch', 0):7d}")
