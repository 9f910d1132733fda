"""Thin Python callers of the three APC trace-generation entry points.

They do what the reference's safe wrappers do (`cuda_abi::apc_tracegen`,
`apc_apply_derived_expr`, `apc_apply_bus`, /root/reference/openvm/src/cuda_abi.rs:97-223):
take host tables, upload them (`to_device()`), call the C ABI with raw device
pointers. Device memory is torch tensors (int32 storage of BabyBear words);
torch is plumbing only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import contextlib

import numpy as np
import torch

from . import abi


def _dev(a: np.ndarray, device) -> torch.Tensor:
    """Upload a host table as raw bytes (like MemCopyH2D::to_device)."""
    a = np.ascontiguousarray(a)
    if a.nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(device)


@dataclass
class DeviceMatrix:
    """Column-major device matrix of BabyBear words (DeviceMatrix<BabyBear>)."""

    buf: torch.Tensor  # int32, len = height * width
    height: int
    width: int

    @staticmethod
    def zeros(height, width, device="cuda"):
        return DeviceMatrix(torch.zeros(height * width, dtype=torch.int32, device=device), height, width)

    def ptr(self):
        return self.buf.data_ptr()


@dataclass
class Periphery:
    """Shared periphery histograms + bus ids (cuda/mod.rs:357-372)."""

    var_bus: int
    var_hist: torch.Tensor  # int32 [var_num_bins]
    tuple_bus: int
    tuple_hist: torch.Tensor  # int32 [sz0*sz1]
    tuple_sizes: tuple
    bitwise_bus: int
    bitwise_hist: torch.Tensor  # int32 [2 * 65536]: range counts | xor counts

    @staticmethod
    def fresh(device="cuda", var_bus=3, tuple_bus=7, bitwise_bus=6, max_bits=17, tuple_sizes=(256, 2048)):
        z = lambda n: torch.zeros(n, dtype=torch.int32, device=device)
        return Periphery(var_bus, z(1 << (max_bits + 1)), tuple_bus, z(tuple_sizes[0] * tuple_sizes[1]),
                         tuple_sizes, bitwise_bus, z(2 * 65536))

    def zero(self):
        """Clear the three histograms ON THE LIBRARY'S LAUNCH STREAM of the calling thread (powdr_gpu_get_stream): the kernels that
        accumulate into them are launched there, and a worker thread's stream (pw_prove_segments_multi: hipStreamNonBlocking) does not
        synchronise with torch's current stream — a zero_() issued on torch's stream could land after multiplicities were added."""
        s = abi.lib.powdr_gpu_get_stream()
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(int(s))) if s else contextlib.nullcontext()
        with ctx:
            for t in (self.var_hist, self.tuple_hist, self.bitwise_hist):
                t.zero_()


def apc_tracegen(output: DeviceMatrix, airs: list, subs: np.ndarray, num_calls: int):
    """airs: [(device int32 tensor col-major, width, height, row_block_size)]; subs int32 [n,4]."""
    dev = output.buf.device
    recs = (abi.OriginalAir * max(len(airs), 1))()
    for i, (t, w, h, b) in enumerate(airs):
        recs[i] = abi.OriginalAir(w, h, t.data_ptr(), b)
    d_airs = _dev(np.frombuffer(bytes(recs), dtype=np.uint8), dev)
    subs = np.ascontiguousarray(subs, dtype=np.int32).reshape(-1, 4)
    d_subs = _dev(subs, dev)
    rc = abi.lib._apc_tracegen(output.ptr(), output.height, d_airs.data_ptr(), d_subs.data_ptr(), len(subs), num_calls)
    abi.check(rc, "_apc_tracegen")
    return d_airs, d_subs  # keep alive until the stream has consumed them


def apc_apply_derived_expr(output: DeviceMatrix, num_calls: int, col_base, offs, lens, bytecode):
    dev = output.buf.device
    n = len(offs)
    specs = np.zeros(n, dtype=[("col_base", "<u8"), ("off", "<u4"), ("len", "<u4")])
    specs["col_base"], specs["off"], specs["len"] = col_base, offs, lens
    d_specs = _dev(specs, dev)
    d_bc = _dev(np.ascontiguousarray(bytecode, dtype=np.uint32), dev)
    rc = abi.lib._apc_apply_derived_expr(output.ptr(), output.height, num_calls, d_specs.data_ptr(), n, d_bc.data_ptr())
    abi.check(rc, "_apc_apply_derived_expr")
    return d_specs, d_bc


def apc_apply_bus(output: DeviceMatrix, num_calls: int, bytecode, interactions, spans, p: Periphery):
    dev = output.buf.device
    bytecode = np.ascontiguousarray(bytecode, dtype=np.uint32)
    interactions = np.ascontiguousarray(interactions, dtype=np.uint32).reshape(-1, 3)
    spans = np.ascontiguousarray(spans, dtype=np.uint32).reshape(-1, 2)
    d_bc, d_int, d_sp = _dev(bytecode, dev), _dev(interactions, dev), _dev(spans, dev)
    rc = abi.lib._apc_apply_bus(
        output.ptr(), num_calls, d_bc.data_ptr(), len(bytecode), d_int.data_ptr(), len(interactions),
        d_sp.data_ptr(), len(spans), p.var_bus, p.var_hist.data_ptr(), p.var_hist.numel(),
        p.tuple_bus, p.tuple_hist.data_ptr(), p.tuple_sizes[0], p.tuple_sizes[1],
        p.bitwise_bus, p.bitwise_hist.data_ptr())
    abi.check(rc, "_apc_apply_bus")
    return d_bc, d_int, d_sp


class PowdrCallMajorAir(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("cells_per_call", C.c_int32), ("reserved", C.c_int32)]


def apc_tracegen_callmajor(output: DeviceMatrix, airs: list, subs_cm: np.ndarray, num_calls: int):
    """Extension (layout half): airs = [(device int32 tensor [num_calls * cells_per_call], cells_per_call)],
    subs_cm int32 [n, 3] = {air, slot, apc_col}. out[apc_col * H + r] = air.buffer[r * cells_per_call + slot]."""
    abi.lib.powdr_apc_tracegen_callmajor.restype = C.c_int
    abi.lib.powdr_apc_tracegen_callmajor.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    recs = (PowdrCallMajorAir * max(len(airs), 1))()
    for i, (t, u) in enumerate(airs):
        recs[i] = PowdrCallMajorAir(t.data_ptr(), int(u), 0)
    s = np.ascontiguousarray(subs_cm, dtype=np.int32).reshape(-1, 3)
    rc = abi.lib.powdr_apc_tracegen_callmajor(output.ptr(), output.height, recs, len(airs), s.ctypes.data, len(s), num_calls)
    abi.check(rc, "powdr_apc_tracegen_callmajor")
