"""Loader of the in-tree HIP shared library (C ABI, include/baspacho_amd.h)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BSP_LIB_PATH: developer override, e.g. a BSP_KTRACE=1 trace build kept beside the product library)
LIB_PATH = os.environ.get("BSP_LIB_PATH") or os.path.join(_HERE, "libbaspacho_amd.so")

_lib = None


def build(force=False):
    """compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)"""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force:
        import shutil
        shutil.rmtree(os.path.join(_HERE, "_build"), ignore_errors=True)
    subprocess.check_call(["bash", script])
    return LIB_PATH


def kernel_source_sha16():
    """sha256 (16 hex digits) over the native sources of the library, in a fixed order: what
    bench.py compares with the hash recorded next to committed rocprofv3 PMC numbers, so that
    counters of an older build are never reported beside a newer build's timings"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def load():
    """Load the library.  There is no fallback: without the HIP library the product is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "baspacho_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (no CPU fallback exists)" % LIB_PATH)
    # torch ships its own libamdhip64 (same soname): import it first so that this library and
    # torch share ONE HIP runtime and device pointers are interchangeable.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a requirement of the C ABI
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.bsp_last_error.restype = ctypes.c_char_p
    lib.bsp_version.restype = ctypes.c_char_p
    for name in ["bsp_order", "bsp_data_size", "bsp_num_spans", "bsp_num_lumps",
                 "bsp_can_factor_up_to_span", "bsp_span_vector_offset"]:
        getattr(lib, name).restype = ctypes.c_int64
    lib.bsp_factor_flops.restype = ctypes.c_double
    _lib = lib
    return lib
