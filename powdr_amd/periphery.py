"""The receive side of the lookup buses: traces of the shared periphery chips from the histograms
`_apc_apply_bus` fills (include/powdr_gpu.h `powdr_periphery_*_trace`) and the bus interactions those
AIRs declare, in the table format of `pw_prover_create_logup` / `host.Apc.compile_bus(1)`.

The chips are external to the reference repository (openvm-circuit-primitives; instantiated in
openvm/src/powdr_extension/trace_generator/cuda/periphery.rs:33-85); the tuple <-> histogram index maps are
the reference's (openvm/cuda/src/apc_apply_bus.cu:74,89,104)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import abi

lib = abi.lib
PERIPHERY_SYMBOLS = ["powdr_periphery_var_range_trace", "powdr_periphery_tuple2_trace", "powdr_periphery_bitwise_trace"]
lib.powdr_periphery_var_range_trace.restype = C.c_int
lib.powdr_periphery_var_range_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
lib.powdr_periphery_tuple2_trace.restype = C.c_int
lib.powdr_periphery_tuple2_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
lib.powdr_periphery_bitwise_trace.restype = C.c_int
lib.powdr_periphery_bitwise_trace.argtypes = [C.c_void_p, C.c_void_p]

OP_PUSH_APC, OP_PUSH_CONST, OP_NEG = 0, 1, 5


def var_range_trace(hist: torch.Tensor) -> torch.Tensor:
    """[value, bits, mult] x len(hist) rows, column-major, Montgomery."""
    out = torch.empty(3 * hist.numel(), dtype=torch.int32, device=hist.device)
    abi.check(lib.powdr_periphery_var_range_trace(hist.data_ptr(), hist.numel(), out.data_ptr()), "powdr_periphery_var_range_trace")
    return out


def tuple2_trace(hist: torch.Tensor, sizes) -> torch.Tensor:
    """[v0, v1, mult] x sz0*sz1 rows."""
    out = torch.empty(3 * hist.numel(), dtype=torch.int32, device=hist.device)
    assert hist.numel() == sizes[0] * sizes[1]
    abi.check(lib.powdr_periphery_tuple2_trace(hist.data_ptr(), sizes[0], sizes[1], out.data_ptr()), "powdr_periphery_tuple2_trace")
    return out


def bitwise_trace(hist: torch.Tensor) -> torch.Tensor:
    """[x, y, x^y, mult_range, mult_xor] x 65 536 rows from the [range | xor] histogram."""
    assert hist.numel() == 2 * 65536
    out = torch.empty(5 * 65536, dtype=torch.int32, device=hist.device)
    abi.check(lib.powdr_periphery_bitwise_trace(hist.data_ptr(), out.data_ptr()), "powdr_periphery_bitwise_trace")
    return out


def _tables(bus: int, rows):
    """rows: list of (mult program, [arg programs]) -> (interactions[n,3], spans[m,2], bytecode)"""
    inter, spans, bc = [], [], []
    for mult, args in rows:
        inter.append((bus, len(args), len(spans)))
        for prog in [mult] + args:
            spans.append((len(bc), len(prog)))
            bc += prog
    return np.array(inter, np.uint32).reshape(-1, 3), np.array(spans, np.uint32).reshape(-1, 2), np.array(bc, np.uint32)


def _col(c):
    return [OP_PUSH_APC, c]


def _neg_col(c):
    return [OP_PUSH_APC, c, OP_NEG]


def var_range_interactions(bus: int = 3):
    """receive (value, bits) `mult` times"""
    return _tables(bus, [(_neg_col(2), [_col(0), _col(1)])])


def tuple2_interactions(bus: int = 7):
    return _tables(bus, [(_neg_col(2), [_col(0), _col(1)])])


def bitwise_interactions(bus: int = 6):
    """receive (x, y, 0, 0) `mult_range` times and (x, y, x^y, 1) `mult_xor` times"""
    return _tables(bus, [(_neg_col(3), [_col(0), _col(1), [OP_PUSH_CONST, 0], [OP_PUSH_CONST, 0]]),
                         (_neg_col(4), [_col(0), _col(1), _col(2), [OP_PUSH_CONST, 1]])])


def select_buses(interactions, buses):
    """Restrict an interaction table (interactions, spans, bytecode) to the given bus ids (spans and bytecode are
    kept whole; only the interaction rows are filtered)."""
    inter, spans, bc = interactions
    inter = np.asarray(inter, np.uint32).reshape(-1, 3)
    keep = np.isin(inter[:, 0], np.asarray(list(buses), np.uint32))
    return np.ascontiguousarray(inter[keep]), spans, bc
