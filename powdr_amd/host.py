"""ctypes binding of include/powdr_host.h: the C++ host mirror of the reference's GPU
trace-generation host path (APC loader, bytecode compilers, try_generate_witness)."""
from __future__ import annotations

import ctypes as C
import json

import numpy as np

from . import abi

lib = abi.lib
HOST_SYMBOLS = ["powdr_apc_from_json", "powdr_apc_free", "powdr_apc_width", "powdr_apc_poly_ids",
                "powdr_apc_num_constraints", "powdr_apc_num_bus_interactions", "powdr_apc_num_derived_columns",
                "powdr_apc_num_instructions", "powdr_apc_instruction_opcode", "powdr_apc_instruction_num_subs",
                "powdr_apc_compile_bus", "powdr_apc_compile_derived", "powdr_apc_compile_constraints",
                "powdr_apc_build_substitutions", "powdr_apc_generate_witness_gpu"]


class PowdrDeviceMatrix(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class PowdrPeriphery(C.Structure):
    _fields_ = [("var_range_bus_id", C.c_uint32), ("d_var_hist", C.c_void_p), ("var_num_bins", C.c_size_t),
                ("tuple2_bus_id", C.c_uint32), ("d_tuple2_hist", C.c_void_p), ("tuple2_sz0", C.c_uint32),
                ("tuple2_sz1", C.c_uint32), ("bitwise_bus_id", C.c_uint32), ("d_bitwise_hist", C.c_void_p)]


vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
lib.powdr_apc_from_json.restype = vp
lib.powdr_apc_from_json.argtypes = [C.c_char_p, sz, C.c_char_p, sz]
lib.powdr_apc_free.argtypes = [vp]
for name in ("powdr_apc_width", "powdr_apc_num_constraints", "powdr_apc_num_bus_interactions",
             "powdr_apc_num_derived_columns", "powdr_apc_num_instructions"):
    getattr(lib, name).restype = u32
    getattr(lib, name).argtypes = [vp]
lib.powdr_apc_poly_ids.restype = C.POINTER(C.c_uint64)
lib.powdr_apc_poly_ids.argtypes = [vp]
lib.powdr_apc_instruction_opcode.restype = u32
lib.powdr_apc_instruction_opcode.argtypes = [vp, u32]
lib.powdr_apc_instruction_num_subs.restype = u32
lib.powdr_apc_instruction_num_subs.argtypes = [vp, u32]
lib.powdr_apc_compile_bus.restype = sz
lib.powdr_apc_compile_bus.argtypes = [vp, sz, vp, vp, C.POINTER(sz), vp]
lib.powdr_apc_compile_derived.restype = sz
lib.powdr_apc_compile_derived.argtypes = [vp, sz, vp, vp]
lib.powdr_apc_compile_constraints.restype = sz
lib.powdr_apc_compile_constraints.argtypes = [vp, vp, vp]
lib.powdr_apc_build_substitutions.restype = sz
lib.powdr_apc_build_substitutions.argtypes = [vp, vp, vp, vp, vp, C.POINTER(sz)]
lib.powdr_apc_generate_witness_gpu.restype = C.c_int
lib.powdr_apc_generate_witness_gpu.argtypes = [vp, vp, vp, sz, sz, vp, vp]


class Apc:
    """An autoprecompile loaded from the reference's JSON wire format."""

    def __init__(self, doc):
        data = doc if isinstance(doc, (bytes, bytearray)) else json.dumps(doc).encode()
        err = C.create_string_buffer(512)
        self._h = lib.powdr_apc_from_json(data, len(data), err, 512)
        if not self._h:
            raise ValueError(err.value.decode())
        self.width = lib.powdr_apc_width(self._h)
        self.n_constraints = lib.powdr_apc_num_constraints(self._h)
        self.n_bus = lib.powdr_apc_num_bus_interactions(self._h)
        self.n_derived = lib.powdr_apc_num_derived_columns(self._h)
        self.n_instructions = lib.powdr_apc_num_instructions(self._h)

    def poly_ids(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib.powdr_apc_poly_ids(self._h), shape=(self.width,)).copy()

    def opcodes(self):
        return [lib.powdr_apc_instruction_opcode(self._h, i) for i in range(self.n_instructions)]

    def num_subs(self):
        return [lib.powdr_apc_instruction_num_subs(self._h, i) for i in range(self.n_instructions)]

    def compile_bus(self, height: int):
        n_spans = sz()
        n_bc = lib.powdr_apc_compile_bus(self._h, height, None, None, C.byref(n_spans), None)
        inter = np.zeros((self.n_bus, 3), np.uint32)
        spans = np.zeros((n_spans.value, 2), np.uint32)
        bc = np.zeros(n_bc, np.uint32)
        lib.powdr_apc_compile_bus(self._h, height, inter.ctypes.data, spans.ctypes.data, C.byref(n_spans), bc.ctypes.data)
        return inter, spans, bc

    def compile_derived(self, height: int):
        n_bc = lib.powdr_apc_compile_derived(self._h, height, None, None)
        specs = np.zeros(self.n_derived, dtype=[("col_base", "<u8"), ("off", "<u4"), ("len", "<u4")])
        bc = np.zeros(n_bc, np.uint32)
        lib.powdr_apc_compile_derived(self._h, height, specs.ctypes.data, bc.ctypes.data)
        return specs, bc

    def compile_constraints(self):
        n_bc = lib.powdr_apc_compile_constraints(self._h, None, None)
        spans = np.zeros((self.n_constraints, 2), np.uint32)
        bc = np.zeros(n_bc, np.uint32)
        lib.powdr_apc_compile_constraints(self._h, spans.ctypes.data, bc.ctypes.data)
        return bc, spans

    def build_substitutions(self, instr_air):
        ia = np.ascontiguousarray(instr_air, dtype=np.int32)
        n_airs = sz()
        n = lib.powdr_apc_build_substitutions(self._h, ia.ctypes.data, None, None, None, C.byref(n_airs))
        subs = np.zeros((n, 4), np.int32)
        ids = np.zeros(n_airs.value, np.int32)
        rbs = np.zeros(n_airs.value, np.int32)
        lib.powdr_apc_build_substitutions(self._h, ia.ctypes.data, subs.ctypes.data, ids.ctypes.data, rbs.ctypes.data, C.byref(n_airs))
        return subs, ids, rbs

    def generate_witness_gpu(self, instr_air, dummy, num_calls: int, d_output_ptr: int, periphery=None):
        """dummy: list of (device ptr, width, height) indexed by the AIR ids used in instr_air."""
        ia = np.ascontiguousarray(instr_air, dtype=np.int32)
        mats = (PowdrDeviceMatrix * max(len(dummy), 1))()
        for i, (ptr, w, h) in enumerate(dummy):
            mats[i] = PowdrDeviceMatrix(ptr, w, h)
        per = None
        if periphery is not None:
            p = periphery
            per = PowdrPeriphery(p.var_bus, p.var_hist.data_ptr(), p.var_hist.numel(), p.tuple_bus, p.tuple_hist.data_ptr(),
                                 p.tuple_sizes[0], p.tuple_sizes[1], p.bitwise_bus, p.bitwise_hist.data_ptr())
        rc = lib.powdr_apc_generate_witness_gpu(self._h, ia.ctypes.data, mats, len(dummy), num_calls, d_output_ptr,
                                                C.byref(per) if per is not None else None)
        abi.check(rc, "powdr_apc_generate_witness_gpu")

    def close(self):
        if self._h:
            lib.powdr_apc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
