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


lib.powdr_apc_from_json_at.restype = vp
lib.powdr_apc_from_json_at.argtypes = [C.c_char_p, sz, sz, C.c_char_p, sz]
lib.powdr_apc_from_cbor.restype = vp
lib.powdr_apc_from_cbor.argtypes = [C.c_char_p, sz, sz, C.c_char_p, sz]
lib.powdr_apc_count_in_json.restype = sz
lib.powdr_apc_count_in_json.argtypes = [C.c_char_p, sz]
lib.powdr_apc_count_in_cbor.restype = sz
lib.powdr_apc_count_in_cbor.argtypes = [C.c_char_p, sz]
lib.powdr_apc_bus_map_len.restype = sz
lib.powdr_apc_bus_map_len.argtypes = [vp]
lib.powdr_apc_bus_map_entry.restype = C.c_int
lib.powdr_apc_bus_map_entry.argtypes = [vp, sz, C.POINTER(C.c_uint64), C.POINTER(u32), C.POINTER(u32 * 2), C.c_char_p, sz]
lib.powdr_apc_periphery_from_bus_map.restype = C.c_int
lib.powdr_apc_periphery_from_bus_map.argtypes = [vp, C.POINTER(PowdrPeriphery)]

BUS_KINDS = ["ExecutionBridge", "Memory", "PcLookup", "VariableRangeChecker", "BitwiseLookup", "TupleRangeChecker", "Other"]


def count_apcs(data: bytes, fmt: str = "json") -> int:
    """Number of `Apc` maps inside a JSON / CBOR document (export file, CLI stage artifact)."""
    f = lib.powdr_apc_count_in_cbor if fmt == "cbor" else lib.powdr_apc_count_in_json
    return int(f(data, len(data)))


class PowdrAirStats(C.Structure):
    _fields_ = [("main_columns", C.c_uint64), ("constraints", C.c_uint64), ("bus_interactions", C.c_uint64)]


class PowdrApcCandidateInfo(C.Structure):
    _fields_ = [("execution_frequency", C.c_uint64), ("start_pc", C.c_uint64), ("n_blocks", u32), ("n_instructions", u32),
                ("before", PowdrAirStats), ("after", PowdrAirStats), ("width_before", C.c_uint64), ("value", C.c_uint64),
                ("cost_before", C.c_double), ("cost_after", C.c_double)]


lib.powdr_apc_candidates_from_json.restype = vp
lib.powdr_apc_candidates_from_json.argtypes = [C.c_char_p, sz, C.c_char_p, sz]
lib.powdr_apc_candidates_free.argtypes = [vp]
lib.powdr_apc_candidates_version.restype = C.c_uint64
lib.powdr_apc_candidates_version.argtypes = [vp]
lib.powdr_apc_candidates_count.restype = sz
lib.powdr_apc_candidates_count.argtypes = [vp]
lib.powdr_apc_candidates_num_labels.restype = sz
lib.powdr_apc_candidates_num_labels.argtypes = [vp]
lib.powdr_apc_candidates_get.restype = C.c_int
lib.powdr_apc_candidates_get.argtypes = [vp, sz, C.POINTER(PowdrApcCandidateInfo)]


def read_apc_candidates(data: bytes):
    """`apc_candidates.json` of cell PGO -> (version, [dict per candidate], number of labels)."""
    err = C.create_string_buffer(512)
    h = lib.powdr_apc_candidates_from_json(data, len(data), err, 512)
    if not h:
        raise ValueError(err.value.decode())
    try:
        out = []
        for i in range(lib.powdr_apc_candidates_count(h)):
            ci = PowdrApcCandidateInfo()
            assert lib.powdr_apc_candidates_get(h, i, C.byref(ci)) == 0
            st = lambda s: dict(main_columns=s.main_columns, constraints=s.constraints, bus_interactions=s.bus_interactions)
            out.append(dict(execution_frequency=ci.execution_frequency, start_pc=ci.start_pc, n_blocks=ci.n_blocks,
                            n_instructions=ci.n_instructions, before=st(ci.before), after=st(ci.after),
                            width_before=ci.width_before, value=ci.value, cost_before=ci.cost_before, cost_after=ci.cost_after))
        return int(lib.powdr_apc_candidates_version(h)), out, int(lib.powdr_apc_candidates_num_labels(h))
    finally:
        lib.powdr_apc_candidates_free(h)


class Apc:
    """An autoprecompile loaded from the reference's wire formats: the serde_json `Apc` / `ApcWithBusMap` export
    (a dict or JSON bytes), or — fmt="cbor" — a serde_cbor stage artifact of the CLI; `index` picks the Apc when the
    document holds several (the `select` artifact is a list of ApcWithStats)."""

    def __init__(self, doc, fmt: str = "json", index: int = 0):
        data = doc if isinstance(doc, (bytes, bytearray)) else json.dumps(doc).encode()
        err = C.create_string_buffer(512)
        if fmt == "cbor":
            self._h = lib.powdr_apc_from_cbor(bytes(data), len(data), index, err, 512)
        else:
            self._h = lib.powdr_apc_from_json_at(bytes(data), len(data), index, err, 512)
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

    def bus_map(self):
        """[(bus id, kind name, (sz0, sz1), variant name)] of an ApcWithBusMap export; [] if the document had none."""
        out = []
        for i in range(lib.powdr_apc_bus_map_len(self._h)):
            bid, kind, sizes, name = C.c_uint64(), u32(), (u32 * 2)(), C.create_string_buffer(64)
            assert lib.powdr_apc_bus_map_entry(self._h, i, C.byref(bid), C.byref(kind), C.byref(sizes), name, 64) == 0
            out.append((bid.value, BUS_KINDS[kind.value], (sizes[0], sizes[1]), name.value.decode()))
        return out

    def periphery_bus_ids(self):
        """(var-range bus, tuple bus, (sz0, sz1), bitwise bus) from the bus map; None for a bus the map does not name."""
        per = PowdrPeriphery(0xFFFFFFFF, None, 0, 0xFFFFFFFF, None, 0, 0, 0xFFFFFFFF, None)
        lib.powdr_apc_periphery_from_bus_map(self._h, C.byref(per))
        f = lambda v: None if v == 0xFFFFFFFF else v
        return f(per.var_range_bus_id), f(per.tuple2_bus_id), (per.tuple2_sz0, per.tuple2_sz1), f(per.bitwise_bus_id)

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

    def instruction_table(self):
        """powdr_apc_instruction_table: (ctypes array of PowdrOrigInstr for the instructions that keep a cell, words per call record)."""
        from .original_chips import PowdrOrigInstr

        lib.powdr_apc_instruction_table.restype = C.c_size_t
        lib.powdr_apc_instruction_table.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
        words = C.c_size_t()
        n = lib.powdr_apc_instruction_table(self._h, None, C.byref(words))
        if n == C.c_size_t(-1).value:
            raise ValueError("the block uses an opcode outside the thirteen RV32IM chips")
        out = (PowdrOrigInstr * max(n, 1))()
        lib.powdr_apc_instruction_table(self._h, out, None)
        return out, n, words.value

    def generate_witness_from_records(self, d_records_ptr: int, num_calls: int, d_output_ptr: int, periphery=None):
        """powdr_apc_generate_witness_from_records: the whole of a1-a3 from call records (no dummy traces)."""
        per = None
        if periphery is not None:
            p = periphery
            per = PowdrPeriphery(p.var_bus, p.var_hist.data_ptr(), p.var_hist.numel(), p.tuple_bus, p.tuple_hist.data_ptr(),
                                 p.tuple_sizes[0], p.tuple_sizes[1], p.bitwise_bus, p.bitwise_hist.data_ptr())
        lib.powdr_apc_generate_witness_from_records.restype = C.c_int
        lib.powdr_apc_generate_witness_from_records.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        rc = lib.powdr_apc_generate_witness_from_records(self._h, d_records_ptr, num_calls, d_output_ptr, C.byref(per) if per is not None else None)
        abi.check(rc, "powdr_apc_generate_witness_from_records")

    def close(self):
        if self._h:
            lib.powdr_apc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
