"""ctypes binding of the C ABI declared in include/adil_hip.h.

The library is loaded lazily; if it is missing or a symbol is absent this module
RAISES — the product has no fallback path."""
import ctypes
import os
from ctypes import c_float, c_int, c_size_t, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get("ADIL_HIP_LIBRARY") or os.path.join(_PKG, "lib", "libadil_hip.so")

# name -> (restype, argtypes); mirrors include/adil_hip.h one to one
SIGNATURES = {
    "adil_abi_version": (c_int, []),
    "adil_max_atoms": (c_int, []),
    "adil_grad_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "adil_grad_code_rows": (c_int, [c_int]),
    "adil_grad_slab_offset": (c_size_t, [c_int, c_int, c_int]),
    "adil_pack_codes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                c_void_p]),
    "adil_gather_images": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "adil_spd_inverse": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "adil_synth": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "adil_synth_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
                               c_void_p]),
    "adil_dict_to_fp8": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "adil_adamw_clamp_fp8": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float,
                                     c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "adil_synth_fp8_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                                      c_int, c_void_p]),
    "adil_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                          c_void_p, c_size_t, c_void_p, c_void_p]),
    "adil_adamw_clamp": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float,
                                 c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "adil_zstep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                           c_float, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_float, c_void_p,
                           c_void_p, c_void_p]),
    "adil_zstep_codes_slab_bytes": (c_size_t, [c_int, c_int, c_int]),
    "adil_zstep_codes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                                 c_float, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_float, c_void_p,
                                 c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "adil_adamw_l1ball": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_float,
                                  c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                  c_void_p, c_int, c_int, c_void_p]),
    "adil_atom_l1ball_project": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "adil_l1ball_project": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p]),
    "adil_l2ball_project": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p]),
    "adil_ista_step": (c_int, [c_void_p, c_void_p, c_size_t, c_float, c_float, c_void_p]),
    "adil_atom_workspace_bytes": (c_size_t, [c_int, c_int]),
    "adil_atom_norms": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "adil_atom_scale": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "adil_gram_workspace_bytes": (c_size_t, [c_int, c_int]),
    "adil_gram": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "adil_dict_rightmul": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "adil_image_metrics": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "adil_affine_act_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int,
                                    c_void_p]),
    "adil_affine_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int,
                                    c_void_p]),
    "adil_stem_conv_fwd": (c_int, [c_void_p, c_int, c_void_p, c_float, c_float, c_float, c_float, c_float, c_float,
                                   c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "adil_maxpool_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "adil_stem_pool_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "adil_pw_conv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "adil_pw_conv_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "adil_conv3x3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "adil_stem_conv_bwd": (c_int, [c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p]),
}

ABI_VERSION = 7
_lib = None


class AdilLibraryError(RuntimeError):
    pass


def load(path: str = None) -> ctypes.CDLL:
    """Load libadil_hip.so and bind every declared symbol. Raises AdilLibraryError on any problem."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIBPATH
    # PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7, same as /opt/rocm's).  Importing torch
    # first makes the dynamic loader bind our NEEDED libamdhip64.so.7 to the runtime torch already loaded, so
    # streams and device pointers handed over by torch belong to the SAME HIP runtime instance.  Loaded the other
    # way round the process ends up with two runtimes and every launch fails with hipErrorNoDevice.
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise AdilLibraryError(
            f"HIP kernel library not found at {p}. Build it with `python -m dl_attack_on_imagenet_amd.build` "
            "(needs hipcc). There is no CPU fallback for the ADiL hot path.")
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:
        raise AdilLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise AdilLibraryError(f"{p} does not export `{name}` (stale build?)") from e
        fn.restype, fn.argtypes = res, args
    if lib.adil_abi_version() != ABI_VERSION:
        raise AdilLibraryError(f"ABI mismatch: library {lib.adil_abi_version()} != binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        names = {-1: "ADIL_EINVAL (bad argument)", -2: "ADIL_EWORKSPACE (workspace too small)"}
        raise AdilLibraryError(f"{what} failed: {names.get(rc, f'hipError_t {rc}')}")
