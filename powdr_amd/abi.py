"""ctypes binding of libpowdr_gpu.so — the same symbols and struct layouts the
reference's Rust FFI declares in /root/reference/openvm/src/cuda_abi.rs:8-169.

Fails loudly (ImportError) if the shared library has not been built; there is no
fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# (POWDR_LIB_DIR: another build of the same library — the sanitizer build of tools/asan_cpu_suite.sh; a developer hook, not a fallback)
_LIBPATH = Path(os.environ.get("POWDR_LIB_DIR") or Path(__file__).resolve().parent / "lib") / "libpowdr_gpu.so"


class OriginalAir(C.Structure):  # cuda_abi.rs:66-73 — 24 bytes
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("buffer", C.c_void_p), ("row_block_size", C.c_int32)]


class Subst(C.Structure):  # cuda_abi.rs:75-86 — 16 bytes
    _fields_ = [("air_index", C.c_int32), ("col", C.c_int32), ("row", C.c_int32), ("apc_col", C.c_int32)]


class ExprSpan(C.Structure):  # cuda_abi.rs:162-169 — 8 bytes
    _fields_ = [("off", C.c_uint32), ("len", C.c_uint32)]


class DerivedExprSpec(C.Structure):  # cuda_abi.rs:88-95 — 16 bytes
    _fields_ = [("col_base", C.c_uint64), ("span", ExprSpan)]


class DevInteraction(C.Structure):  # cuda_abi.rs:149-160 — 12 bytes
    _fields_ = [("bus_id", C.c_uint32), ("num_args", C.c_uint32), ("args_index_off", C.c_uint32)]


assert C.sizeof(OriginalAir) == 24 and OriginalAir.buffer.offset == 8
assert C.sizeof(Subst) == 16 and C.sizeof(ExprSpan) == 8
assert C.sizeof(DerivedExprSpec) == 16 and C.sizeof(DevInteraction) == 12

# every symbol include/powdr_gpu.h declares
ABI_SYMBOLS = [
    "_apc_tracegen", "_apc_apply_derived_expr", "_apc_apply_bus",
    "powdr_apc_apply_derived_expr_cols", "powdr_apc_apply_bus_cols",
    "powdr_gpu_set_stream", "powdr_gpu_get_stream", "powdr_gpu_timing_enable",
    "powdr_gpu_timing_report", "powdr_gpu_version",
]


def _load():
    # PyTorch-ROCm wheels bundle their own libamdhip64; if libpowdr_gpu.so pulled in /opt/rocm's copy
    # first, the process would hold two HIP runtimes and the second one reports hipErrorNoDevice.
    # Importing torch first makes the loader resolve our DT_NEEDED libamdhip64 to the copy torch
    # loaded, so device pointers and streams are shared. (Rust callers link /opt/rocm directly.)
    import torch  # noqa: F401

    if not _LIBPATH.exists():
        raise ImportError(
            f"{_LIBPATH} is missing: build it with `python -m powdr_amd.build` "
            "(the product has no CPU fallback)")
    lib = C.CDLL(str(_LIBPATH))
    vp, sz, i32, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
    lib._apc_tracegen.argtypes = [vp, sz, vp, vp, sz, i32]
    lib._apc_tracegen.restype = i32
    lib._apc_apply_derived_expr.argtypes = [vp, sz, i32, vp, sz, vp]
    lib._apc_apply_derived_expr.restype = i32
    lib._apc_apply_bus.argtypes = [vp, i32, vp, sz, vp, sz, vp, sz, u32, vp, sz, u32, vp, u32, u32, u32, vp]
    lib._apc_apply_bus.restype = i32
    lib.powdr_gpu_set_stream.argtypes = [vp]
    lib.powdr_gpu_get_stream.restype = vp
    lib.powdr_gpu_timing_enable.argtypes = [i32]
    lib.powdr_gpu_timing_report.argtypes = [C.c_char_p, sz]
    lib.powdr_gpu_timing_report.restype = sz
    lib.powdr_gpu_version.restype = C.c_char_p
    lib.powdr_gpu_call_stats.argtypes = [vp, i32]
    lib.powdr_gpu_call_stats.restype = None
    return lib


lib = _load()


class HipError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        raise HipError(f"{what} failed with hipError {rc}")


CALL_STATS = ("gather_sparse_jobs", "gather_whole_jobs", "gather_chunk_jobs", "gather_calls", "bus_fast_interactions",
              "bus_interpreted_interactions", "bus_binned_windows", "bus_direct_calls", "bus_xbc_calls", "jit_launches",
              "interpreter_launches")


def call_stats(reset: bool = False) -> dict:
    """powdr_gpu_call_stats of the calling thread as {name: count}."""
    a = (C.c_uint64 * 16)()
    lib.powdr_gpu_call_stats(a, 1 if reset else 0)
    return {n: int(a[i]) for i, n in enumerate(CALL_STATS)}


def timing_report() -> dict:
    """{kernel name: (launch count, total ms)} since the last powdr_gpu_timing_enable(1)."""
    n = lib.powdr_gpu_timing_report(None, 0)
    buf = C.create_string_buffer(n + 16)
    lib.powdr_gpu_timing_report(buf, n + 16)
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.rsplit(" ", 2)
        out[name] = (int(cnt), float(ms))
    return out
