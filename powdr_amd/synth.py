"""Synthetic APC workloads in the reference's own wire format.

No guest program can be built without a Rust/RISC-V toolchain, so benchmarks and
parity tests run on seeded synthetic autoprecompiles whose SHAPES are pinned by
the reference's tests:

  C1  sha256-shaped      W = 1 204, H = 2^16,  377 constraints,   954 bus interactions
  C2  guest-keccak APC   W = 2 022, H = 2^20,  187 constraints, 1 734 bus interactions
      (openvm-riscv/src/lib.rs:1377-1458), gathered from 5 original AIRs
      (w, b) = (36,318) BaseAlu, (53,116) Shift, (41,241) LoadStore, (26,1), (18,1)
  C3  guest-ecrecover    W = 3 731, H = 2^22, 3 114 constraints, 2 314 bus interactions

The generator emits the JSON document `serde_json` would produce for
`Apc<BabyBear, Instr, _, _>` (autoprecompiles/src/lib.rs:185-195), so the product
host loader (csrc/host) and the oracle loader both parse exactly what they would
parse for a real exported APC. Everything is deterministic in `seed`.

Column kinds steer both the dummy-trace filler (so that lookups are in range)
and the constraint generator (so that every constraint vanishes on valid rows
and on the all-zero padding rows, as `is_valid`-guarded APC constraints do).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

P = 0x78000001

# opcode of the first instruction class member per original AIR (OpenVM RV32IM numbering)
AIR_OPCODES = {
    "BaseAlu": [0x200, 0x202, 0x203, 0x204],
    "Shift": [0x205, 0x206],
    "LoadStore": [0x210, 0x213],
    "BranchEqual": [0x220],
    "JalLui": [0x230],
    "LessThan": [0x208],
    "Mul": [0x250],
    "DivRem": [0x254],
}

BUS_EXEC, BUS_MEMORY, BUS_PC, BUS_VAR_RANGE, BUS_BITWISE, BUS_TUPLE = 0, 1, 2, 3, 6, 7
TUPLE_SIZES = (256, 2048)  # openvm-bus-interaction-handler/src/lib.rs:44
VAR_RANGE_MAX_BITS = 17  # bins = 2^(17+1)


@dataclass
class Shape:
    name: str
    width: int
    log_height: int
    n_constraints: int
    n_interactions: int
    airs: list  # [(air name, width, instructions per call)]
    n_quotient: int = 4
    config_id: int = 0
    honest_bitwise: bool = False  # bitwise-bus interactions are range pairs (x, y, 0, 0) only (segment_workload.py: a synthetic APC has no x ^ y column)


SHAPES = {
    "C1": Shape("sha256-shaped", 1204, 16, 377, 954,
                [("BaseAlu", 36, 160), ("Shift", 53, 60), ("LoadStore", 41, 96), ("LessThan", 37, 12)], config_id=1),
    "C2": Shape("guest-keccak", 2022, 20, 187, 1734,
                [("BaseAlu", 36, 318), ("Shift", 53, 116), ("LoadStore", 41, 241), ("BranchEqual", 26, 1), ("JalLui", 18, 1)], config_id=2),
    "C3": Shape("guest-ecrecover", 3731, 22, 3114, 2314,
                [("BaseAlu", 36, 420), ("Shift", 53, 64), ("LoadStore", 41, 310), ("Mul", 31, 96), ("LessThan", 37, 40), ("BranchEqual", 26, 1)], config_id=3),
    # C3 with dense gather sources (4 076 source cells per call instead of ~36 000) so that a 2^22-row run fits one
    # 288 GB GPU: same output shape, constraints and interactions as C3
    "C3p": Shape("guest-ecrecover (dense sources)", 3731, 22, 3114, 2314,
                 [("BaseAlu", 36, 30), ("Shift", 53, 20), ("LoadStore", 41, 30), ("Mul", 31, 10), ("LessThan", 37, 10), ("BranchEqual", 26, 1)], config_id=4),
    # small shapes for tests
    "T0": Shape("tiny", 24, 6, 9, 20, [("BaseAlu", 36, 3), ("Shift", 53, 2), ("JalLui", 18, 1)], n_quotient=2, config_id=100),
    "T1": Shape("small", 160, 10, 40, 120, [("BaseAlu", 36, 20), ("Shift", 53, 7), ("LoadStore", 41, 12), ("BranchEqual", 26, 1)], n_quotient=3, config_id=101),
}


@dataclass
class SynthApc:
    doc: dict  # the JSON document (wire format)
    shape: Shape
    poly_ids: list  # ascending; column index = position
    kinds: dict  # poly_id -> (kind, bound)   kind in valid|derived|bit|tri|byte|range|field
    instr_air: list  # per instruction: AIR name ('' if the instruction has no substitutions)
    airs: list  # [(name, width, effective row_block_size)] in order of first appearance
    source_of: dict  # poly_id -> (air name, block row, column) for substituted columns
    n_nodes_bus: int = 0
    n_nodes_constraints: int = 0


def _ref(pid):
    return f"c{pid}_0@{pid}"


def _count(e):
    if isinstance(e, list):
        return 1 + sum(_count(x) for x in e if not (isinstance(x, str) and x in "+-*"))
    return 1


def _stack_need(e):
    """Slots the reference's post-fix evaluator needs for `e` as compile_bus_to_gpu / emit_expr emit it (left operand first:
    cuda/mod.rs:49-98): its stack has 16 (expr_eval.cuh:22) and overflowing it is a device assert."""
    if not isinstance(e, list):
        return 1
    if len(e) == 2:  # unary minus
        return _stack_need(e[1])
    return max(_stack_need(e[0]), 1 + _stack_need(e[2]))


def generate(shape: Shape | str, seed: int = 0, verbosity: float = 1.0, density_note: str = "uniform") -> SynthApc:
    if isinstance(shape, str):
        shape = SHAPES[shape]
    rng = np.random.Generator(np.random.PCG64(0x504F574452 ^ shape.config_id ^ (seed << 20)))
    W = shape.width
    # sparse, ascending poly ids (the reference orders columns by id, powdr.rs:44-57)
    poly_ids = sorted(rng.choice(4 * W, size=W, replace=False).tolist())

    # ---- column kinds -------------------------------------------------------------------
    nq = shape.n_quotient
    order = rng.permutation(W).tolist()
    kinds: dict[int, tuple] = {}
    valid_pid = poly_ids[order[0]]
    kinds[valid_pid] = ("valid", 2)
    derived_pids = [poly_ids[i] for i in order[1 : 1 + nq]]
    for pid in derived_pids:
        kinds[pid] = ("derived", P)
    rest = [poly_ids[i] for i in order[1 + nq :]]
    n = len(rest)
    cuts = np.cumsum([max(2, int(n * f)) for f in (0.08, 0.02, 0.35, 0.25)]).tolist()
    for i, pid in enumerate(rest):
        if i < cuts[0]:
            kinds[pid] = ("bit", 2)
        elif i < cuts[1]:
            kinds[pid] = ("tri", 3)
        elif i < cuts[2]:
            kinds[pid] = ("byte", 256)
        elif i < cuts[3]:
            bits = int(rng.choice([11, 12, 13, 14, 17]))
            kinds[pid] = ("range", 1 << bits)
        else:
            kinds[pid] = ("field", P)
    by_kind: dict[str, list] = {}
    for pid, (k, bnd) in kinds.items():
        by_kind.setdefault(k, []).append(pid)
    for v in by_kind.values():
        v.sort()

    def pick(kind):
        lst = by_kind[kind]
        return lst[int(rng.integers(len(lst)))]

    def pad(e, nodes_target):
        """Inflate an expression the way un-optimised machines look (0 + x, x * 1)."""
        while _count(e) < nodes_target:
            r = rng.random()
            if r < 0.4:
                # `0 + e` holds the constant while e is evaluated: one stack slot per nesting level. Past 14 the expression would trip
                # the reference evaluator's own stack (16 slots); it is then padded the other way round — same random stream, so every
                # shape / seed that never came near the limit is unchanged
                e = [0, "+", e] if _stack_need(e) < 14 else [e, "*", 1]
            elif r < 0.7:
                e = [e, "*", 1]
            elif r < 0.85:
                e = [e, "-", 0]
            else:
                e = ["-", ["-", e]]
        return e

    # ---- derived columns ----------------------------------------------------------------
    derived_json = [[_ref(valid_pid), {"Constant": 1}]]
    derived_defs = {}
    non_derived = [p for p in rest]
    for k, pid in enumerate(derived_pids):
        a, b, c = (non_derived[int(i)] for i in rng.integers(len(non_derived), size=3))
        if k % 2 == 0:
            e1 = [[_ref(a), "*", _ref(b)], "+", _ref(c)]  # degree 2 numerator
            e2 = [_ref(pick("field")), "+", int(rng.integers(1, 1000))]
        else:
            e1 = [_ref(a), "+", [int(rng.integers(1, 1 << 20)), "*", _ref(b)]]
            e2 = 1  # plain expression column
            if k >= 1 and rng.random() < 0.5:
                e1 = [e1, "+", _ref(derived_pids[k - 1])]  # later derived columns may read earlier ones
        derived_json.append([_ref(pid), {"QuotientOrZero": [e1, e2]}])
        derived_defs[pid] = (e1, e2)

    # ---- substitutions: every non-derived column <- one cell of one original instruction -----
    cells_per_air = [(name, w, b) for name, w, b in shape.airs]
    total_cells = sum(w * b for _, w, b in cells_per_air)
    n_sub = len(non_derived)
    assert n_sub <= total_cells, "more APC columns than source cells"
    flat = rng.choice(total_cells, size=n_sub, replace=False)
    flat.sort()
    # instructions: b_k per AIR, shuffled program order
    instr_list = []
    for name, w, b in cells_per_air:
        for j in range(b):
            instr_list.append((name, j))
    perm = rng.permutation(len(instr_list)).tolist()
    instr_list = [instr_list[i] for i in perm]
    cell_owner = {}  # (air, j) -> [(col, pid)]
    base = 0
    bounds = []
    for name, w, b in cells_per_air:
        bounds.append((base, base + w * b, name, w))
        base += w * b
    sub_cols = rng.permutation(non_derived).tolist()
    for cell, pid in zip(flat.tolist(), sub_cols):
        for lo, hi, name, w in bounds:
            if lo <= cell < hi:
                j, col = divmod(cell - lo, w)
                cell_owner.setdefault((name, j), []).append((col, pid))
                break
    instructions, subs_json, instr_air = [], [], []
    eff_rows: dict[str, int] = {}
    air_order: list[str] = []
    source_of = {}
    for name, j in instr_list:
        ops = AIR_OPCODES[name]
        opcode = ops[int(rng.integers(len(ops)))]
        instructions.append([opcode] + [int(x) for x in rng.integers(0, 128, size=7)])
        owned = sorted(cell_owner.get((name, j), []))
        subs_json.append([{"original_poly_index": c, "apc_poly_id": p} for c, p in owned])
        if owned:
            if name not in air_order:
                air_order.append(name)
            row = eff_rows.get(name, 0)
            eff_rows[name] = row + 1
            instr_air.append(name)
            for c, p in owned:
                source_of[p] = (name, row, c)
        else:
            instr_air.append("")
    widths = {name: w for name, w, _ in cells_per_air}
    airs = [(name, widths[name], eff_rows[name]) for name in air_order]

    # ---- bus interactions ------------------------------------------------------------------
    n_int = shape.n_interactions
    n_exec = 2
    remaining = n_int - n_exec
    n_mem = int(remaining * 0.30)
    n_var = int(remaining * 0.45)
    n_bit = remaining - n_mem - n_var
    n_tuple = max(1, n_var // 20)
    n_var -= n_tuple
    plan = ["exec"] * n_exec + ["mem"] * n_mem + ["var"] * n_var + ["tuple"] * n_tuple + ["bitwise"] * n_bit
    plan = [plan[i] for i in rng.permutation(len(plan)).tolist()]
    mean_nodes = 6.0 * verbosity

    def mult_expr():
        r = rng.random()
        if r < 0.75:
            return _ref(valid_pid)
        if r < 0.92:
            return [_ref(valid_pid), "*", _ref(pick("bit"))]
        return [_ref(valid_pid), "*", int(rng.integers(2, 4))]

    def nodes():
        return max(1, int(rng.exponential(mean_nodes)))

    buses = []
    for kind in plan:
        if kind == "exec":
            buses.append({"id": BUS_EXEC, "mult": ["-", _ref(valid_pid)] if len(buses) % 2 else _ref(valid_pid),
                          "args": [pad(_ref(pick("field")), nodes()), pad(_ref(pick("field")), nodes())]})
        elif kind == "mem":
            args = [int(rng.integers(1, 3)), pad(_ref(pick("field")), nodes())]
            args += [_ref(pick("byte")) for _ in range(4)] + [pad(_ref(pick("field")), nodes())]
            buses.append({"id": BUS_MEMORY, "mult": mult_expr() if rng.random() < 0.5 else ["-", mult_expr()], "args": args})
        elif kind == "var":
            if rng.random() < 0.7:
                pid = pick("range")
                bits = kinds[pid][1].bit_length() - 1
                val = _ref(pid)
            else:
                val = [_ref(pick("byte")), "+", [256, "*", _ref(pick("byte"))]]
                bits = int(rng.choice([16, 17]))
            buses.append({"id": BUS_VAR_RANGE, "mult": mult_expr(), "args": [pad(val, nodes()), bits]})
        elif kind == "tuple":
            r11 = [p for p in by_kind["range"] if kinds[p][1] <= TUPLE_SIZES[1]]
            v1 = _ref(r11[int(rng.integers(len(r11)))]) if r11 else _ref(pick("byte"))
            buses.append({"id": BUS_TUPLE, "mult": mult_expr(), "args": [pad(_ref(pick("byte")), nodes()), v1]})
        else:
            sel = int(rng.integers(0, 2))
            x, y = _ref(pick("byte")), _ref(pick("byte"))
            if rng.random() < 0.3:
                x = [[255, "-", _ref(pick("byte"))], "*", 1]
            mult = mult_expr()
            args = [pad(x, nodes()), pad(y, nodes()), _ref(pick("byte")), sel]
            if shape.honest_bitwise:  # (drawn all the same: the random stream does not depend on the flag)
                args[2], args[3] = 0, 0
            buses.append({"id": BUS_BITWISE, "mult": mult, "args": args})

    # ---- constraints (vanish on valid rows and on all-zero rows) ---------------------------
    cons = [[_ref(valid_pid), "*", [_ref(valid_pid), "-", 1]]]

    def lin():
        a, b = pick("field"), pick("byte")
        return [[_ref(a), "+", [int(rng.integers(1, P)), "*", _ref(b)]], "+", int(rng.integers(0, P))]

    while len(cons) < shape.n_constraints:
        r = rng.random()
        if r < 0.35:
            b = _ref(pick("bit"))
            c = [b, "*", [b, "-", 1]]
            if rng.random() < 0.5:
                c = [c, "*", lin()]
        elif r < 0.45:
            t = _ref(pick("tri"))
            c = [t, "*", [[t, "-", 1], "*", [t, "-", 2]]]
        else:
            pid = derived_pids[int(rng.integers(len(derived_pids)))] if derived_pids else None
            if pid is None:
                continue
            e1, e2 = derived_defs[pid]
            body = [[_ref(pid), "*", e2], "-", e1]  # d*e2 - e1, degree <= 2
            c = [_ref(valid_pid), "*", body]
        cons.append(pad(c, int(_count(c) * (1 + rng.random() * verbosity))))

    doc = {
        "block": {"blocks": [{"start_pc": 0x200000, "instructions": instructions}]},
        "machine": {"constraints": cons, "bus_interactions": buses, "derived_columns": derived_json},
        "subs": subs_json,
        "optimistic_constraints": {"fetches_by_step": {}, "constraints_to_check_by_step": {}},
        "bus_map": {"bus_ids": {"0": "ExecutionBridge", "1": "Memory", "2": "PcLookup",
                                "3": {"Other": "VariableRangeChecker"}, "6": {"Other": "BitwiseLookup"},
                                "7": {"Other": {"TupleRangeChecker": list(TUPLE_SIZES)}}}},
    }
    # every column must be referenced by a constraint or a bus interaction to exist
    # (main_columns = unique references); reference the stragglers through memory sends.
    referenced = set()

    def walk(e):
        if isinstance(e, str):
            if "@" in e:
                referenced.add(int(e[e.rfind("@") + 1 :]))
        elif isinstance(e, list):
            for x in e:
                walk(x)

    for c in cons:
        walk(c)
    for b in buses:
        walk(b["mult"])
        for a in b["args"]:
            walk(a)
    missing = [p for p in poly_ids if p not in referenced]
    # fold the missing columns into the args of memory interactions (8 at a time)
    mem_idx = [i for i, b in enumerate(buses) if b["id"] == BUS_MEMORY]
    k = 0
    while missing:
        chunk, missing = missing[:8], missing[8:]
        e = _ref(chunk[0])
        for p in chunk[1:]:
            e = [e, "+", _ref(p)]
        b = buses[mem_idx[k % len(mem_idx)]]
        b["args"].append(e)
        k += 1

    s = SynthApc(doc, shape, poly_ids, kinds, instr_air, airs, source_of)
    s.n_nodes_bus = sum(_count(b["mult"]) + sum(_count(a) for a in b["args"]) for b in buses)
    s.n_nodes_constraints = sum(_count(c) for c in cons)
    return s


def next_pow2_or_zero(n: int) -> int:
    """openvm_circuit::utils::next_power_of_two_or_zero."""
    return 0 if n == 0 else 1 << (n - 1).bit_length()


def dummy_trace_dims(s: SynthApc, num_calls: int, pow2: bool = True):
    """[(air name, width, height, row_block_size)]; height like the original chips' traces:
    next_pow2(rows) (cuda/mod.rs:244-245 takes `common_main` of the dummy chip)."""
    out = []
    for name, w, b in s.airs:
        rows = b * num_calls
        h = next_pow2_or_zero(rows) if pow2 else (rows + 3) // 4 * 4
        out.append((name, w, max(h, 4), b))
    return out


_RNG_LIB = None


def _bounded_u32(rng: np.random.Generator, bound: int, n: int) -> np.ndarray:
    """rng.integers(0, bound, size=n, dtype=np.uint32), bit for bit, through libpowdr_synth_rng.so (synth_csrc/synth_rng.c: numpy's PCG64 +
    Lemire draws restated in C, ~10 x faster; the generator's state is handed over and written back). numpy itself when the library is
    missing or the bit generator is not PCG64."""
    global _RNG_LIB
    if n < (1 << 16):  # (a parallel region per call: not worth it for one cell's few thousand draws)
        return rng.integers(0, bound, size=n, dtype=np.uint32)
    if _RNG_LIB is None:
        import ctypes as C
        from pathlib import Path

        path = Path(__file__).resolve().parent / "lib" / "libpowdr_synth_rng.so"
        try:
            _RNG_LIB = C.CDLL(str(path))
            _RNG_LIB.synth_pcg64_bounded_u32.restype = C.c_int
            _RNG_LIB.synth_pcg64_bounded_u32.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        except OSError:
            _RNG_LIB = False
    st = rng.bit_generator.state
    if not _RNG_LIB or st.get("bit_generator") != "PCG64" or not (2 <= bound <= 0xFFFFFFFF):
        return rng.integers(0, bound, size=n, dtype=np.uint32)
    import ctypes as C

    class _S(C.Structure):
        _fields_ = [("state_hi", C.c_uint64), ("state_lo", C.c_uint64), ("inc_hi", C.c_uint64), ("inc_lo", C.c_uint64),
                    ("has_uint32", C.c_uint32), ("uinteger", C.c_uint32)]

    m64 = (1 << 64) - 1
    x, inc = st["state"]["state"], st["state"]["inc"]
    c = _S(x >> 64, x & m64, inc >> 64, inc & m64, int(st["has_uint32"]), int(st["uinteger"]))
    out = np.empty(n, dtype=np.uint32)
    if _RNG_LIB.synth_pcg64_bounded_u32(C.byref(c), bound, out.ctypes.data, n):
        return rng.integers(0, bound, size=n, dtype=np.uint32)
    st["state"]["state"] = (int(c.state_hi) << 64) | int(c.state_lo)
    st["has_uint32"], st["uinteger"] = int(c.has_uint32), int(c.uinteger)
    rng.bit_generator.state = st
    return out


def fill_dummy_traces_numpy(s: SynthApc, num_calls: int, seed: int = 0, pow2: bool = True):
    """Canonical column-major dummy traces (numpy uint32), cells that feed bounded column
    kinds drawn below their bound. Returns list of arrays in `s.airs` order."""
    rng = np.random.Generator(np.random.PCG64(seed + 977))
    dims = dummy_trace_dims(s, num_calls, pow2)
    bufs = []
    for name, w, h, b in dims:
        bufs.append(_bounded_u32(rng, P, w * h))
    idx = {name: i for i, (name, _, _, _) in enumerate(dims)}
    for pid, (name, row, col) in s.source_of.items():
        kind, bound = s.kinds[pid]
        if bound >= P:
            continue
        _, w, h, b = dims[idx[name]]
        view = bufs[idx[name]][col * h + row : col * h + row + b * num_calls : b]
        view[:] = _bounded_u32(rng, bound, len(view))
    return bufs, dims


# ---------------------------------------------------------------------------------------------------------------------
# Multi-AIR segments: shapes only — widths, heights, constraint and interaction counts. The proofs'
# cost does not depend on the values, so the bench proves random traces against programs of the right size and form.
# /root/reference/openvm-riscv/src/lib.rs:1114-1122: 19 non-powdr machines, main width 819, 643 constraints, 253 interactions.
# Thirteen of them are the RV32IM instruction AIRs whose ACTUAL constraints and bus interactions the reference snapshots in
# openvm-riscv/tests/openvm_constraints.txt (456 columns, 307 constraints, 227 interactions; parsed by powdr_amd/air_text.py into
# tests/golden/openvm_airs.npz): those are proven with their real programs. The other six (program, connector, memory boundary /
# Merkle, Poseidon2 periphery, range / bitwise lookups: EXTERNAL, no snapshot) carry random programs that make up the pinned totals.
REFERENCE_AIR_WIDTHS = {"BaseAlu": 36, "LessThan": 37, "Shift": 53, "BranchEqual": 26, "BranchLessThan": 32, "JalLui": 18, "Jalr": 28,
                        "LoadSignExtend": 36, "LoadStore": 41, "DivRem": 59, "MulH": 39, "Multiplication": 31, "Auipc": 20}
REFERENCE_AIR_COUNTS = {"BaseAlu": (22, 20), "LessThan": (28, 18), "Shift": (76, 24), "BranchEqual": (11, 11), "BranchLessThan": (25, 13),
                        "JalLui": (9, 10), "Jalr": (9, 16), "LoadSignExtend": (18, 18), "LoadStore": (25, 17), "DivRem": (64, 25), "MulH": (11, 24),
                        "Multiplication": (4, 19), "Auipc": (5, 12)}  # (constraints, interactions)
OTHER_SYSTEM_AIRS = [("connector", 2, 10), ("program", 4, 18), ("boundary", 21, 16), ("merkle", 32, 17), ("range_bitwise", 6, 18), ("poseidon2", 298, 14)]  # (name, width, log height)
REFERENCE_AIR_LOG_HEIGHTS = {"BaseAlu": 20, "LessThan": 16, "Shift": 19, "BranchEqual": 17, "BranchLessThan": 16, "JalLui": 14, "Jalr": 15,
                             "LoadSignExtend": 13, "LoadStore": 20, "DivRem": 12, "MulH": 10, "Multiplication": 15, "Auipc": 13}
SYSTEM_CONSTRAINTS, SYSTEM_INTERACTIONS = 643, 253
SYSTEM_AIR_WIDTHS = list(REFERENCE_AIR_WIDTHS.values()) + [w for _, w, _ in OTHER_SYSTEM_AIRS]
assert sum(SYSTEM_AIR_WIDTHS) == 819 and len(SYSTEM_AIR_WIDTHS) == 19
assert sum(c for c, _ in REFERENCE_AIR_COUNTS.values()) == 307 and sum(i for _, i in REFERENCE_AIR_COUNTS.values()) == 227
C4_APC_WIDTHS = [520, 440, 380, 330, 290, 260, 230, 210, 180, 160]  # 10 APC AIRs of a pairing-shaped segment, sum 3000


def _split(total, weights):
    """integers proportional to `weights` that add up to `total`"""
    w = np.asarray(weights, dtype=np.float64)
    x = np.floor(total * w / w.sum()).astype(int)
    for i in np.argsort(-(total * w / w.sum() - x))[: total - int(x.sum())]:
        x[i] += 1
    return [int(v) for v in x]


def segment_shape(kind: str, seed: int = 0, max_log_height: int = 20):
    """[(name, width, log_height, n_constraints, n_interactions)] of one segment.
    C4: 10 APC AIRs (sum W = 3 000) at 2^max_log_height rows + the 19 system AIRs (sum W = 819).
    C5: reth-shaped, ~57 APC AIRs with log-uniform heights 2^10..2^max and widths 30..4 000 (about 3 G cells) + the system AIRs.
    APC AIRs carry constraints / interactions at the keccak APC's densities (187 and 1 734 per 2 022 columns)."""
    shrink = 20 - max_log_height
    airs = []
    system = [(n, w, max(2, REFERENCE_AIR_LOG_HEIGHTS[n] - shrink), *REFERENCE_AIR_COUNTS[n]) for n, w in REFERENCE_AIR_WIDTHS.items()]
    ow = [w for _, w, _ in OTHER_SYSTEM_AIRS]
    oc, oi = _split(SYSTEM_CONSTRAINTS - 307, ow), _split(SYSTEM_INTERACTIONS - 227, ow)
    system += [(n, w, max(2, lh - shrink), oc[k], oi[k]) for k, (n, w, lh) in enumerate(OTHER_SYSTEM_AIRS)]
    apc = lambda name, w, lh: (name, w, lh, max(1, round(w * 187 / 2022)), max(1, round(w * 1734 / 2022)))
    if kind == "C4":
        airs = [apc(f"apc{k}", w, max_log_height) for k, w in enumerate(C4_APC_WIDTHS)]
    elif kind == "C5":
        rng = np.random.default_rng(1000 + seed)
        budget, total = 3.0e9 / (1 << (2 * shrink)) if shrink else 3.0e9, 0
        while len(airs) < 57:
            lh = int(rng.integers(10, 21)) - shrink
            w = int(np.exp(rng.uniform(np.log(30), np.log(4000))))
            if lh < 2 or total + (w << lh) > budget:
                if all((30 << max(2, h - shrink)) + total > budget for h in range(10, 21)):
                    break
                continue
            total += w << lh
            airs.append(apc(f"apc{len(airs)}", w, lh))
    else:
        raise ValueError(kind)
    return airs + system


_REFERENCE_AIRS = None


def reference_air_programs(name: str):
    """(cons_bc, cons_spans, (inter, ispans, ibc)) of one of the reference's RV32IM instruction AIRs (REFERENCE_AIR_WIDTHS) from the
    committed fixture tests/golden/openvm_airs.npz (made by tests/golden/make_openvm_airs.py from the reference's snapshot)."""
    global _REFERENCE_AIRS
    if _REFERENCE_AIRS is None:
        from pathlib import Path

        z = np.load(Path(__file__).resolve().parents[1] / "tests" / "golden" / "openvm_airs.npz")
        _REFERENCE_AIRS = {str(n): (z[f"a{k}_bc"], z[f"a{k}_spans"], (z[f"a{k}_inter"], z[f"a{k}_ispans"], z[f"a{k}_ibc"])) for k, n in enumerate(z["names"])}
    return _REFERENCE_AIRS[name]


def air_programs(name: str, width: int, n_constraints: int, n_interactions: int, seed: int):
    """Programs of one AIR of a segment shape: the real ones for the reference's instruction AIRs, random ones of the given size otherwise."""
    if name in REFERENCE_AIR_WIDTHS:
        bc, sp, it = reference_air_programs(name)
        assert width == REFERENCE_AIR_WIDTHS[name] and len(sp) == n_constraints and len(it[0]) == n_interactions
        return bc, sp, it
    return random_air_programs(width, n_constraints, n_interactions, seed)


def random_air_programs(width: int, n_constraints: int, n_interactions: int, seed: int):
    """Constraint programs (post-fix, column operands) of the forms an optimised APC has — a*b - c, a*(a - 1), a*b*c - d,
    a + k*b - c — and bus interactions with a column multiplicity and 2-4 degree-1 arguments (column, k - column,
    column + 256*column). Returns (cons_bc, cons_spans[n,2], (inter[n,3], ispans[m,2], ibc))."""
    rng = np.random.default_rng(seed)
    PA, PC, ADD, SUB, MUL = 0, 1, 2, 3, 4
    col = lambda: int(rng.integers(0, width))
    bc, spans = [], []
    for k in range(n_constraints):
        off, form = len(bc), k % 4
        if form == 0:
            bc += [PA, col(), PA, col(), MUL, PA, col(), SUB]
        elif form == 1:
            a = col()
            bc += [PA, a, PA, a, PC, 1, SUB, MUL]
        elif form == 2:
            bc += [PA, col(), PA, col(), MUL, PA, col(), MUL, PA, col(), SUB]
        else:
            bc += [PA, col(), PC, int(rng.integers(1, 1 << 16)), PA, col(), MUL, ADD, PA, col(), SUB]
        spans.append((off, len(bc) - off))
    ibc, ispans, inter = [], [], []
    for k in range(n_interactions):
        n_args = int(rng.integers(2, 5))
        inter.append((int(rng.choice([1, 3, 6, 7])), n_args, len(ispans)))
        off = len(ibc)
        ibc += [PA, col()]  # multiplicity: a column
        ispans.append((off, len(ibc) - off))
        for j in range(n_args):
            off, form = len(ibc), (k + j) % 3
            if form == 0:
                ibc += [PA, col()]
            elif form == 1:
                ibc += [PC, 255, PA, col(), SUB]
            else:
                ibc += [PA, col(), PC, 256, PA, col(), MUL, ADD]
            ispans.append((off, len(ibc) - off))
    return (np.array(bc, np.uint32), np.array(spans, np.uint32).reshape(-1, 2),
            (np.array(inter, np.uint32).reshape(-1, 3), np.array(ispans, np.uint32).reshape(-1, 2), np.array(ibc, np.uint32)))


def column_structured_substitutions(air_dims, n_sub: int, seed: int = 0, coverage: float = 0.65) -> np.ndarray:
    """Substitutions (air_index, col, row, apc_col) with the structure optimised APCs show in the reference's own snapshots:
    the surviving cells are FEW COLUMNS of an original AIR, each present in most of the block's instructions, not cells
    scattered over the whole block. Evidence: /root/reference/sp1-benchmarks/tests/apc_snapshots/complex/
    keccak_permutation.txt (13 693 -> 2 940 columns over 263 instructions, but only 73 distinct column names: the bitwise
    chip keeps b_low_bytes / c_low_bytes / result, 16 columns, in 100-155 of its 184 instructions) and
    /root/reference/openvm-riscv/tests/apc_snapshots/complex/aligned_memcpy.txt (the same 14 LoadStore columns —
    mem_ptr_limbs, prev_data, timestamps — survive in every LOADW / STOREW). `generate` scatters the cells uniformly over
    the block instead, which is the worst case for the gather: every 64-byte sector of every source column holds a used cell.
    air_dims: [(width, row_block_size)]; every AIR gets u_k ~ width-proportional used columns so that
    sum_k u_k * b_k * coverage ~ n_sub, rows are dropped at random down to exactly n_sub cells."""
    rng = np.random.default_rng(seed ^ 0xC01)
    dims = [(int(w), int(b)) for w, b in air_dims]
    total_w = sum(w for w, _ in dims)
    target = n_sub / coverage
    frac = target / sum(w * b for w, b in dims)
    cells = []
    for k, (w, b) in enumerate(dims):
        u = min(w, max(1, int(round(frac * w + 0.5))))
        for c in rng.choice(w, size=u, replace=False):
            for r in range(b):
                cells.append((k, int(c), r))
    while len(cells) < n_sub:  # tiny shapes: top up with unused cells
        k = int(rng.integers(len(dims)))
        cand = (k, int(rng.integers(dims[k][0])), int(rng.integers(dims[k][1])))
        if cand not in cells:
            cells.append(cand)
    keep = rng.choice(len(cells), size=n_sub, replace=False)
    keep.sort()
    out = np.zeros((n_sub, 4), np.int32)
    apc_cols = rng.permutation(n_sub)
    for i, j in enumerate(keep.tolist()):
        k, c, r = cells[j]
        out[i] = (k, c, r, int(apc_cols[i]))
    assert total_w > 0
    return out
