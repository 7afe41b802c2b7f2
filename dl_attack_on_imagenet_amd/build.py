"""Build libadil_hip.so (gfx950) in-tree with hipcc.  `python -m dl_attack_on_imagenet_amd.build`"""
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIBPATH = os.path.join(LIBDIR, "libadil_hip.so")
ARCH = "gfx950"


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources and headers: identifies the build a measurement was taken on
    (profiles/hbm_traffic.json records it; bench.py reports `roofline.traffic` only for a matching build)."""
    import hashlib
    h = hashlib.sha256()
    for path in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(s) > t for s in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    """Compile every csrc/*.hip for gfx950 into lib/libadil_hip.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-I", os.path.join(ROOT, "include"), "-I", CSRC]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIBPATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIBPATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIBPATH)
