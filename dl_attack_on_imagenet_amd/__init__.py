"""MI355X-native ADiL (Adversarial Dictionary Learning) attack engine.

Host side (Python, PyTorch-ROCm for device memory / streams / the frozen
classifier) over a C-ABI shared library of hand-written HIP kernels for gfx950
(`csrc/`, declared in `include/adil_hip.h`).  There is NO CPU fallback: every op
in `ops` raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
