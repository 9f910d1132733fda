"""TEST INFRASTRUCTURE — Python side of the CPU oracle for APC trace generation.

Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

What it restates (paths relative to /root/reference):
  * the serde wire format of `Apc` / `SymbolicMachine` / `AlgebraicExpression`
      expression/src/lib.rs:209-246 (`[l,"+",r]`, `["-",e]`, numbers),
      autoprecompiles/src/expression.rs:51-80 (`"name@id"` references),
      constraint-solver/src/constraint_system.rs:97-137 (derived columns),
      autoprecompiles/src/lib.rs:177-195 (Apc, Substitution)
  * column order = ascending poly id of the unique references in constraints
    and bus interactions: autoprecompiles/src/powdr.rs:44-57,
    symbolic_machine.rs:129-133,201-209
  * a direct AST evaluator (expression/src/lib.rs:179-206) used to cross-check
    the bytecode evaluator of oracle/apc_oracle.c on small cases
  * the post-fix encoder and the Subst/OriginalAir table builder of the GPU
    host path: openvm/src/powdr_extension/trace_generator/cuda/mod.rs:49-177,
    272-328 (AIRs are numbered by first appearance instead of the reference's
    HashMap iteration order)
  * the per-instruction dummy-row addressing of the CPU path:
    autoprecompiles/src/trace_handler.rs:53-124
"""
from __future__ import annotations

import ctypes
import gzip
import json
import os
import subprocess
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

P = 0x78000001

OP_PUSH_APC, OP_PUSH_CONST, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_INV_OR_ZERO = range(7)

# ----------------------------------------------------------------------------------------
# wire format
# ----------------------------------------------------------------------------------------


def parse_expr(j):
    """expression/src/lib.rs:209-246 + autoprecompiles/src/expression.rs:51-80."""
    if isinstance(j, int):
        return ("num", j % P)
    if isinstance(j, str):
        pos = j.rfind("@")
        if pos < 0:
            raise ValueError(f"Invalid format for AlgebraicReference: {j}")
        return ("ref", j[:pos], int(j[pos + 1 :]))
    if isinstance(j, list):
        if len(j) == 3:
            return ("bin", j[1], parse_expr(j[0]), parse_expr(j[2]))
        if len(j) == 2:
            if j[0] != "-":
                raise ValueError(f"unknown unary operator {j[0]}")
            return ("neg", parse_expr(j[1]))
    raise ValueError(f"cannot parse expression {j!r}")


@dataclass
class BusInteraction:
    id: int
    mult: tuple
    args: list


@dataclass
class Derived:
    name: str
    poly_id: int
    kind: str  # "Constant" | "QuotientOrZero"
    constant: int = 0
    e1: tuple | None = None  # numerator
    e2: tuple | None = None  # denominator


@dataclass
class Apc:
    instructions: list  # [[opcode, a, b, c, d, e, f, g], ...] flattened over blocks
    constraints: list
    bus_interactions: list
    derived_columns: list
    subs: list  # per instruction: [(original_poly_index, apc_poly_id), ...]
    bus_map: dict = field(default_factory=dict)
    start_pc: int = 0  # pc of the first instruction of the first block

    def main_columns(self):
        """Ascending poly ids of all references in constraints and bus interactions."""
        ids = set()

        def walk(e):
            stack = [e]
            while stack:
                x = stack.pop()
                t = x[0]
                if t == "ref":
                    ids.add(x[2])
                elif t == "bin":
                    stack.append(x[2])
                    stack.append(x[3])
                elif t == "neg":
                    stack.append(x[1])

        for c in self.constraints:
            walk(c)
        for b in self.bus_interactions:
            walk(b.mult)
            for a in b.args:
                walk(a)
        return sorted(ids)

    def poly_id_to_index(self):
        return {pid: i for i, pid in enumerate(self.main_columns())}


def load_apc(d: dict) -> Apc:
    instrs = []
    for blk in d["block"]["blocks"]:
        instrs.extend(blk["instructions"])
    m = d["machine"]
    derived = []
    for var, method in m["derived_columns"]:
        pos = var.rfind("@")
        name, pid = var[:pos], int(var[pos + 1 :])
        if "Constant" in method:
            derived.append(Derived(name, pid, "Constant", constant=int(method["Constant"]) % P))
        else:
            e1, e2 = method["QuotientOrZero"]
            derived.append(Derived(name, pid, "QuotientOrZero", e1=parse_expr(e1), e2=parse_expr(e2)))
    return Apc(
        instructions=instrs,
        constraints=[parse_expr(c) for c in m["constraints"]],
        bus_interactions=[
            BusInteraction(int(b["id"]), parse_expr(b["mult"]), [parse_expr(a) for a in b["args"]])
            for b in m["bus_interactions"]
        ],
        derived_columns=derived,
        subs=[[(int(s["original_poly_index"]), int(s["apc_poly_id"])) for s in row] for row in d["subs"]],
        bus_map=d.get("bus_map", {}),
        start_pc=int(d["block"]["blocks"][0].get("start_pc", 0)) if d["block"]["blocks"] else 0,
    )


def load_apc_file(path) -> Apc:
    path = str(path)
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        return load_apc(json.load(f))


# ----------------------------------------------------------------------------------------
# direct AST evaluation (expression/src/lib.rs:179-206), pure Python: small cases only
# ----------------------------------------------------------------------------------------


def eval_ast(e, lookup):
    t = e[0]
    if t == "num":
        return e[1] % P
    if t == "ref":
        return lookup(e[2]) % P
    if t == "neg":
        return (-eval_ast(e[1], lookup)) % P
    l = eval_ast(e[2], lookup)
    r = eval_ast(e[3], lookup)
    if e[1] == "+":
        return (l + r) % P
    if e[1] == "-":
        return (l - r) % P
    if e[1] == "*":
        return (l * r) % P
    raise ValueError(e[1])


# ----------------------------------------------------------------------------------------
# post-fix encoder (cuda/mod.rs:49-177)
# ----------------------------------------------------------------------------------------


def emit_expr(bc: list, e, id_to_index: dict, apc_height: int):
    """cuda/mod.rs:49-81; iterative to survive the deep left-leaning sums of the fixtures."""
    out = []
    stack = [(e, False)]
    while stack:
        x, done = stack.pop()
        t = x[0]
        if t == "num":
            out += [OP_PUSH_CONST, x[1] % P]
        elif t == "ref":
            out += [OP_PUSH_APC, (id_to_index[x[2]] * apc_height) & 0xFFFFFFFF]
        elif t == "neg":
            if done:
                out.append(OP_NEG)
            else:
                stack.append((x, True))
                stack.append((x[1], False))
        else:
            if done:
                out.append({"+": OP_ADD, "-": OP_SUB, "*": OP_MUL}[x[1]])
            else:
                stack.append((x, True))
                stack.append((x[3], False))
                stack.append((x[2], False))
    bc.extend(out)


def compile_derived(apc: Apc, id_to_index: dict, apc_height: int):
    """cuda/mod.rs:100-141 -> (col_base[u64], span_off, span_len, bytecode)."""
    col_base, offs, lens, bc = [], [], [], []
    for d in apc.derived_columns:
        off = len(bc)
        if d.kind == "Constant":
            bc += [OP_PUSH_CONST, d.constant % P]
        else:
            emit_expr(bc, d.e2, id_to_index, apc_height)
            bc.append(OP_INV_OR_ZERO)
            emit_expr(bc, d.e1, id_to_index, apc_height)
            bc.append(OP_MUL)
        col_base.append(id_to_index[d.poly_id] * apc_height)
        offs.append(off)
        lens.append(len(bc) - off)
    return (
        np.array(col_base, dtype=np.uint64),
        np.array(offs, dtype=np.uint32),
        np.array(lens, dtype=np.uint32),
        np.array(bc, dtype=np.uint32),
    )


def compile_bus(apc: Apc, id_to_index: dict, apc_height: int):
    """cuda/mod.rs:143-177 -> (interactions[n,3], spans[m,2], bytecode)."""
    inter, spans, bc = [], [], []
    for b in apc.bus_interactions:
        off_idx = len(spans)
        for e in [b.mult] + list(b.args):
            off = len(bc)
            emit_expr(bc, e, id_to_index, apc_height)
            spans.append((off, len(bc) - off))
        inter.append((b.id, len(b.args), off_idx))
    return (
        np.array(inter, dtype=np.uint32).reshape(-1, 3),
        np.array(spans, dtype=np.uint32).reshape(-1, 2),
        np.array(bc, dtype=np.uint32),
    )


# ----------------------------------------------------------------------------------------
# instruction -> AIR (test harness stand-in for OriginalAirs::opcode_to_air, which the
# reference derives from the EXTERNAL VM config; opcode classes of OpenVM's RV32IM)
# ----------------------------------------------------------------------------------------

_OPCODE_CLASSES = [
    (0x200, 0x204, "BaseAlu"),
    (0x205, 0x207, "Shift"),
    (0x208, 0x209, "LessThan"),
    (0x210, 0x215, "LoadStore"),
    (0x216, 0x217, "LoadSignExtend"),
    (0x220, 0x221, "BranchEqual"),
    (0x225, 0x228, "BranchLessThan"),
    (0x230, 0x231, "JalLui"),
    (0x235, 0x235, "Jalr"),
    (0x240, 0x240, "Auipc"),
    (0x250, 0x250, "Mul"),
    (0x251, 0x253, "MulH"),
    (0x254, 0x257, "DivRem"),
]


def opcode_air(opcode: int) -> str:
    for lo, hi, name in _OPCODE_CLASSES:
        if lo <= opcode <= hi:
            return name
    return f"Opcode{opcode}"


@dataclass
class GpuTables:
    """What cuda/mod.rs:272-328 uploads: OriginalAir descriptors and Subst records."""

    air_names: list  # index = air_index
    row_block_size: list  # per air: instructions per call
    subs: np.ndarray  # [n,4] int32: air_index, col, row, apc_col


def build_gpu_tables(apc: Apc, id_to_index: dict, air_of=None) -> GpuTables:
    air_of = air_of or (lambda ins: opcode_air(ins[0]))
    groups: dict[str, list] = {}
    for ins, subs in zip(apc.instructions, apc.subs):
        if not subs:
            continue
        groups.setdefault(air_of(ins), []).append(subs)
    names, rbs, recs = [], [], []
    for air_index, (name, subs_by_row) in enumerate(groups.items()):
        names.append(name)
        rbs.append(len(subs_by_row))
        for row, subs in enumerate(subs_by_row):
            for orig, pid in subs:
                recs.append((air_index, orig, row, id_to_index[pid]))
    return GpuTables(names, rbs, np.array(recs, dtype=np.int32).reshape(-1, 4))


@dataclass
class CpuTables:
    """What trace_handler.rs:53-124 derives for the row-major CPU path."""

    air_names: list
    occurrences: np.ndarray  # per air
    instr_air: np.ndarray
    instr_table_offset: np.ndarray
    sub_begin: np.ndarray
    sub_pairs: np.ndarray  # [n,2] (dummy_trace_index, apc_index)


def build_cpu_tables(apc: Apc, id_to_index: dict, air_of=None) -> CpuTables:
    air_of = air_of or (lambda ins: opcode_air(ins[0]))
    names: list[str] = []
    instr_air, table_off, counts = [], [], {}
    sub_begin, pairs = [0], []
    for ins, subs in zip(apc.instructions, apc.subs):
        if not subs:
            continue
        name = air_of(ins)
        if name not in names:
            names.append(name)
        a = names.index(name)
        instr_air.append(a)
        table_off.append(counts.get(a, 0))
        counts[a] = counts.get(a, 0) + 1
        for orig, pid in subs:
            pairs.append((orig, id_to_index[pid]))
        sub_begin.append(len(pairs))
    occ = np.array([counts[a] for a in range(len(names))], dtype=np.int32)
    return CpuTables(
        names,
        occ,
        np.array(instr_air, dtype=np.int32),
        np.array(table_off, dtype=np.int32),
        np.array(sub_begin, dtype=np.uint32),
        np.array(pairs, dtype=np.uint32).reshape(-1, 2),
    )


def air_widths(apc: Apc, air_of=None) -> dict:
    """Width of each original AIR as far as the APC can tell: 1 + the largest
    substituted column index (pre-optimisation fixtures substitute every cell)."""
    air_of = air_of or (lambda ins: opcode_air(ins[0]))
    w: dict[str, int] = {}
    for ins, subs in zip(apc.instructions, apc.subs):
        if subs:
            n = air_of(ins)
            w[n] = max(w.get(n, 0), 1 + max(o for o, _ in subs))
    return w


# ----------------------------------------------------------------------------------------
# Montgomery <-> canonical (R = 2^32), for moving data across the C ABI in tests
# ----------------------------------------------------------------------------------------

_R = (1 << 32) % P
_RINV = pow(_R, P - 2, P)


def to_monty(a: np.ndarray) -> np.ndarray:
    return ((a.astype(np.uint64) * np.uint64(_R)) % np.uint64(P)).astype(np.uint32)


def from_monty(a: np.ndarray) -> np.ndarray:
    return ((a.astype(np.uint64) * np.uint64(_RINV)) % np.uint64(P)).astype(np.uint32)


# ----------------------------------------------------------------------------------------
# C oracle loader
# ----------------------------------------------------------------------------------------

_HERE = Path(__file__).resolve().parent
_LIB = None


# POWDR_ORACLE_SAN=1: the checker itself under AddressSanitizer + UBSan (gcc), in oracle/_build_san/ (tools/asan_cpu_suite.sh oracle)
_SAN = os.environ.get("POWDR_ORACLE_SAN", "") == "1"
_BUILD = "_build_san" if _SAN else "_build"
_SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"] if _SAN else []


def build_c_oracle(force=False) -> Path:
    out = _HERE / _BUILD / "liboracle.so"
    srcs = sorted(_HERE.glob("*.c")) + [f for f in sorted(_HERE.glob("*.cpp")) if f.name != "tuned_cpu.cpp"]  # (tuned_cpu.cpp: build_tuned_cpu)
    hdrs = sorted(_HERE.glob("*.h")) + sorted(_HERE.glob("*.hpp")) + sorted(_HERE.glob("*.inc"))
    if not force and out.exists() and all(out.stat().st_mtime >= s.stat().st_mtime for s in srcs + hdrs):
        return out
    out.parent.mkdir(exist_ok=True)
    objs = []
    for s in srcs:
        o = out.parent / (s.name + ".o")
        cc = ["gcc", "-std=c11"] if s.suffix == ".c" else ["g++", "-std=c++17"]
        subprocess.check_call(cc + ["-O1" if _SAN else "-O2", "-fPIC", "-fopenmp", "-c", str(s), "-o", str(o)] + _SAN_FLAGS)
        objs.append(str(o))
    subprocess.check_call(["g++", "-shared", "-fopenmp", "-o", str(out)] + _SAN_FLAGS + objs)
    return out


def build_tuned_cpu(force=False):
    """oracle/tuned_cpu.cpp (the TUNED CPU baseline of bench.py: Montgomery + AVX-512 LDE and Poseidon2 Merkle) as two shared
    objects, one compiled with the AVX-512 feature flags Zen 4+/Skylake-X+ share and a scalar Montgomery one; tuned_cpu() loads
    the AVX-512 build only on a CPU that reports avx512f/bw/dq/vl."""
    src = _HERE / "tuned_cpu.cpp"
    outs = {}
    for name, flags in (("avx512", ["-mavx512f", "-mavx512bw", "-mavx512dq", "-mavx512vl"]), ("scalar", [])):
        out = _HERE / _BUILD / f"libtuned_cpu_{name}.so"
        if force or not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
            out.parent.mkdir(exist_ok=True)
            subprocess.check_call(["g++", "-std=c++17", "-O1" if _SAN else "-O3", "-fPIC", "-fopenmp", "-shared", str(src), "-o", str(out)] + flags + _SAN_FLAGS)
        outs[name] = out
    return outs


_TUNED = None


def tuned_cpu():
    """ctypes handle of the tuned CPU baseline with the oracle's Poseidon2 constants installed."""
    global _TUNED
    if _TUNED is None:
        outs = build_tuned_cpu()
        flags = ""
        try:
            flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags"))
        except Exception:
            pass
        wide = all(f" {f}" in flags for f in ("avx512f", "avx512bw", "avx512dq", "avx512vl"))
        lib = ctypes.CDLL(str(outs["avx512" if wide else "scalar"]))
        from . import stark_model as sm

        e, i, d = sm.poseidon2_constants()
        lib.tc_set_poseidon2_constants(_p(np.ascontiguousarray(e.reshape(-1))), _p(i), _p(d))
        _TUNED = lib
    return _TUNED


def c_oracle():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(str(build_c_oracle()))
    return _LIB


def _p(a: np.ndarray, ty=ctypes.c_void_p):
    return a.ctypes.data_as(ty)


def c_eval_expr(bc: np.ndarray, trace: np.ndarray, r: int) -> int:
    lib = c_oracle()
    res = ctypes.c_uint32()
    bc = np.ascontiguousarray(bc, dtype=np.uint32)
    rc = lib.or_eval_expr(_p(bc), ctypes.c_uint32(len(bc)), _p(trace), ctypes.c_size_t(r), ctypes.byref(res))
    if rc:
        raise ValueError("malformed bytecode")
    return res.value


def c_apc_tracegen(H, width, airs_buf, airs_height, airs_rbs, subs, num_calls, out=None):
    """or_apc_tracegen; airs_buf: list of canonical column-major uint32 arrays."""
    lib = c_oracle()
    if out is None:
        out = np.zeros(H * width, dtype=np.uint32)
    ptrs = (ctypes.c_void_p * len(airs_buf))(*[a.ctypes.data for a in airs_buf])
    h = np.ascontiguousarray(airs_height, dtype=np.int32)
    b = np.ascontiguousarray(airs_rbs, dtype=np.int32)
    s = np.ascontiguousarray(subs, dtype=np.int32)
    lib.or_apc_tracegen(_p(out), ctypes.c_size_t(H), ptrs, _p(h), _p(b), _p(s), ctypes.c_size_t(len(s)), ctypes.c_int(num_calls))
    return out


def compact_call_major(airs_buf, airs_height, airs_rbs, subs, num_calls):
    """Reference-layout gather inputs -> the call-major compacted form of powdr_apc_tracegen_callmajor: per AIR the
    used (row, col) cells get slots in order of first appearance; buffer[r * U + slot] = dense[col * h + row + r * b].
    Returns (list of uint32 arrays [num_calls * U], cells_per_call, subs_cm int32 [n, 3])."""
    subs = np.ascontiguousarray(subs, dtype=np.int32).reshape(-1, 4)
    slots = [dict() for _ in airs_buf]
    subs_cm = np.zeros((len(subs), 3), np.int32)
    for i, (a, col, row, apc_col) in enumerate(subs):
        d = slots[a]
        subs_cm[i] = (a, d.setdefault((int(row), int(col)), len(d)), apc_col)
    bufs, cells = [], []
    r = np.arange(num_calls, dtype=np.int64)
    for a, d in enumerate(slots):
        U = len(d)
        buf = np.zeros(max(num_calls * U, 1), np.uint32)
        if U and num_calls:
            view = buf[:num_calls * U].reshape(num_calls, U)
            for (row, col), slot in d.items():
                view[:, slot] = airs_buf[a][col * int(airs_height[a]) + row + r * int(airs_rbs[a])]
        bufs.append(buf)
        cells.append(U)
    return bufs, np.array(cells, np.int32), subs_cm


def c_apc_tracegen_callmajor(H, width, bufs, cells, subs_cm, num_calls):
    lib = c_oracle()
    out = np.zeros(H * width, dtype=np.uint32)
    ptrs = (ctypes.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
    c = np.ascontiguousarray(cells, dtype=np.int32)
    s = np.ascontiguousarray(subs_cm, dtype=np.int32)
    lib.or_apc_tracegen_callmajor(_p(out), ctypes.c_size_t(H), ptrs, _p(c), _p(s), ctypes.c_size_t(len(s)), ctypes.c_int(num_calls))
    return out


def c_apc_apply_derived(out, H, num_calls, col_base, offs, lens, bc):
    lib = c_oracle()
    col_base = np.ascontiguousarray(col_base, dtype=np.uint64)
    offs = np.ascontiguousarray(offs, dtype=np.uint32)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    bc = np.ascontiguousarray(bc, dtype=np.uint32)
    rc = lib.or_apc_apply_derived(_p(out), ctypes.c_size_t(H), ctypes.c_int(num_calls), _p(col_base), _p(offs), _p(lens), ctypes.c_size_t(len(offs)), _p(bc))
    if rc:
        raise ValueError("malformed derived bytecode")
    return out


def c_apc_apply_bus(trace, num_calls, bc, inter, spans, var_bus, var_hist, tuple_bus, tuple_hist, sz0, sz1, bitwise_bus, bitwise_hist):
    lib = c_oracle()
    bc = np.ascontiguousarray(bc, dtype=np.uint32)
    inter = np.ascontiguousarray(inter, dtype=np.uint32)
    spans = np.ascontiguousarray(spans, dtype=np.uint32)
    rc = lib.or_apc_apply_bus(
        _p(trace), ctypes.c_int(num_calls), _p(bc), _p(inter), ctypes.c_size_t(len(inter)), _p(spans),
        ctypes.c_uint32(var_bus), _p(var_hist), ctypes.c_size_t(len(var_hist)),
        ctypes.c_uint32(tuple_bus), _p(tuple_hist), ctypes.c_uint32(sz0), ctypes.c_uint32(sz1),
        ctypes.c_uint32(bitwise_bus), _p(bitwise_hist),
    )
    if rc:
        raise ValueError("malformed bus bytecode")


def c_generate_witness(apc: Apc, cpu: CpuTables, id_to_index, dummy_rowmajor: list, dummy_width: list,
                       num_calls: int, periphery: dict):
    """or_generate_witness. periphery: dict(var_bus, var_hist, tuple_bus|None, tuple_hist, sz0, sz1,
    bitwise_bus|None, bitwise_hist). Returns row-major values [height, width]."""
    lib = c_oracle()
    width = len(id_to_index)
    height = 0 if num_calls == 0 else 1 << (num_calls - 1).bit_length()
    values = np.zeros(max(height * width, 1), dtype=np.uint32)
    # derived: bytecode with column-index operands (apc_height = 1)
    d_col, d_kind, d_const, d_spans, d_bc = [], [], [], [], []
    for d in apc.derived_columns:
        d_col.append(id_to_index[d.poly_id])
        if d.kind == "Constant":
            d_kind.append(0); d_const.append(d.constant); d_spans += [0, 0, 0, 0]
        else:
            d_kind.append(1); d_const.append(0)
            o1 = len(d_bc); emit_expr(d_bc, d.e1, id_to_index, 1); l1 = len(d_bc) - o1
            o2 = len(d_bc); emit_expr(d_bc, d.e2, id_to_index, 1); l2 = len(d_bc) - o2
            d_spans += [o1, l1, o2, l2]
    inter, spans, bc = compile_bus(apc, id_to_index, 1)
    u32 = lambda x: np.ascontiguousarray(np.array(x, dtype=np.uint32))
    d_col, d_kind, d_const, d_spans, d_bc = map(u32, (d_col, d_kind, d_const, d_spans, d_bc))
    ptrs = (ctypes.c_void_p * len(dummy_rowmajor))(*[a.ctypes.data for a in dummy_rowmajor])
    dw = np.ascontiguousarray(dummy_width, dtype=np.int32)
    ibus = np.ascontiguousarray(inter[:, 0]); inargs = np.ascontiguousarray(inter[:, 1]); ioff = np.ascontiguousarray(inter[:, 2])
    pz = periphery
    dummy_hist = np.zeros(1, dtype=np.uint32)
    rc = lib.or_generate_witness(
        _p(values), ctypes.c_size_t(height), ctypes.c_size_t(width), ctypes.c_size_t(num_calls),
        ptrs, _p(dw), ctypes.c_size_t(len(cpu.instr_air)), _p(cpu.instr_air), _p(cpu.instr_table_offset),
        _p(cpu.occurrences), _p(cpu.sub_begin), _p(np.ascontiguousarray(cpu.sub_pairs)),
        ctypes.c_size_t(len(d_col)), _p(d_col), _p(d_kind), _p(d_const), _p(d_spans), _p(d_bc),
        ctypes.c_size_t(len(inter)), _p(ibus), _p(inargs), _p(ioff), _p(np.ascontiguousarray(spans)), _p(bc),
        ctypes.c_uint32(pz["var_bus"]), _p(pz["var_hist"]), ctypes.c_size_t(len(pz["var_hist"])),
        ctypes.c_int(pz.get("tuple_bus") is not None), ctypes.c_uint32(pz.get("tuple_bus") or 0),
        _p(pz["tuple_hist"] if pz.get("tuple_bus") is not None else dummy_hist),
        ctypes.c_uint32(pz.get("sz0", 0)), ctypes.c_uint32(pz.get("sz1", 0)),
        ctypes.c_int(pz.get("bitwise_bus") is not None), ctypes.c_uint32(pz.get("bitwise_bus") or 0),
        _p(pz["bitwise_hist"] if pz.get("bitwise_bus") is not None else dummy_hist),
    )
    if rc:
        raise ValueError(f"or_generate_witness failed: {rc}")
    return values[: height * width].reshape(height, width)


# ---- periphery chips' traces from the histograms (include/powdr_gpu.h powdr_periphery_*_trace) --------------------
# The tuple <-> index maps are the reference's (openvm/cuda/src/apc_apply_bus.cu:74,89,104; cpu/periphery.rs:176-237)
# inverted; canonical values, column-major (one numpy row per column).

def var_range_trace(hist: np.ndarray) -> np.ndarray:
    i = np.arange(len(hist), dtype=np.uint64) + 1
    bits = np.floor(np.log2(i.astype(np.float64))).astype(np.uint64)
    bits = np.where((np.uint64(1) << bits) > i, bits - 1, bits)  # guard the float log at exact powers of two
    bits = np.where((np.uint64(2) << bits) <= i, bits + 1, bits)
    return np.stack([i - (np.uint64(1) << bits), bits, hist.astype(np.uint64) % P]).astype(np.uint32)


def tuple2_trace(hist: np.ndarray, sz0: int, sz1: int) -> np.ndarray:
    i = np.arange(sz0 * sz1, dtype=np.uint64)
    return np.stack([i // sz1, i % sz1, hist.astype(np.uint64) % P]).astype(np.uint32)


def bitwise_trace(hist: np.ndarray) -> np.ndarray:
    i = np.arange(65536, dtype=np.uint64)
    x, y = i >> 8, i & 255
    return np.stack([x, y, x ^ y, hist[:65536].astype(np.uint64) % P, hist[65536:].astype(np.uint64) % P]).astype(np.uint32)
