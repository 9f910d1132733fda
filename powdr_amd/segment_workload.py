"""An HONEST multi-AIR segment for the C4 / C5 benchmark legs (BASELINE configs[3] / configs[4]).

The reference proves, per segment, the traces of every chip that took part in it: the autoprecompile chips, the original RV32IM
instruction chips and the shared periphery chips that receive their lookups (`engine.prove(pk, ProvingContext{per_trace})`,
/root/reference/openvm/src/trace_generation.rs:97-139; 19 non-powdr AIRs, /root/reference/openvm-riscv/src/lib.rs:1114-1122).
This module builds such a segment out of the library's own trace generators, so that what the bench times is a statement a
verifier accepts:

  * APC AIRs          synthetic autoprecompiles (powdr_amd/synth.py: constraints at the keccak APC's density, bus interactions
                      exec / memory / var-range / tuple / bitwise) whose traces are GENERATED — gather from the original chips'
                      dummy traces, derived columns, bus replay into the shared periphery histograms
                      (powdr_apc_generate_witness_gpu = try_generate_witness, cuda/mod.rs:201-401); every constraint holds on them;
  * instruction AIRs  the thirteen RV32IM chips of the reference's snapshot with their REAL constraints and interactions
                      (tests/golden/openvm_airs.npz), traces expanded from call records (powdr_original_airs_expand), their range /
                      bitwise / tuple lookups replayed into the same histograms (_apc_apply_bus on the chip's trace);
  * periphery AIRs    variable range checker, range tuple checker, bitwise lookup: traces from the histograms
                      (powdr_periphery_*_trace), the receive side of buses 3 / 7 / 6.

ONE proof per segment (pw_prove_segment, LogUp inside), checked by pw_verify_segment. The lookup buses have both sides in the
segment, so their LogUp sums cancel — `balance_witness()` proves the same traces with every AIR restricted to buses 3 / 6 / 7 and
verifies with check_balance. The memory, execution-bridge and program buses are proven on the send side only: their receivers
(offline memory checking, connector, program chip; with Merkle / Poseidon2 the remaining 5 of the 19 system AIRs, 357 of the 819
columns) are external chips that nothing in this repository generates — they are left out rather than faked.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import abi, host, original_chips as oc, periphery, prover, synth, tracegen as tg

P = 0x78000001
LOOKUP_BUSES = (synth.BUS_VAR_RANGE, synth.BUS_BITWISE, synth.BUS_TUPLE)


def build_apc_workload(shape, log_h: int, exact_heights: bool, seed: int, calls_fraction: float = 1.0, out: torch.Tensor | None = None,
                       per: tg.Periphery | None = None, data_seed: int | None = None):
    """One APC AIR's trace-generation inputs on the GPU: the APC (host mirror), the original chips' dummy traces (random, cells that
    feed bounded column kinds — bytes, range-checked limbs, flags — drawn below their bound so that every lookup is in range), the
    output matrix and the periphery histograms. `shape`: a synth.Shape or the name of one. `seed` fixes the AIR (its substitutions,
    constraints, interactions); `data_seed` (None: the process's global generator for the uniform cells, seed + 1 for the bounded ones,
    as before round 6) the VALUES — the ranks of a run prove the same AIR on different segments of an execution."""
    s = synth.generate(shape, seed=seed)
    apc = host.Apc(s.doc)
    H = 1 << log_h
    calls = max(1, int(H * calls_fraction))
    order, instr_air = [], []  # AIR ids by first appearance among instructions with substitutions
    for n in s.instr_air:
        if n and n not in order:
            order.append(n)
        instr_air.append(order.index(n) if n else -1)
    dims = {n: (w, b) for n, w, b in s.airs}
    dummy, tensors, src_bytes = [], {}, 0
    for n in order:
        w, b = dims[n]
        rows = b * calls
        h = max(4, (rows + 3) // 4 * 4 if exact_heights else synth.next_pow2_or_zero(rows))
        t = torch.empty(w * h, dtype=torch.int32, device="cuda")
        if data_seed is None:
            t.random_(0, P)
        else:
            t.random_(0, P, generator=torch.Generator(device="cuda").manual_seed(1000003 * data_seed + len(dummy)))
        tensors[n] = (t, w, h, b)
        dummy.append((t.data_ptr(), w, h))
        src_bytes += t.numel() * 4
    g = torch.Generator(device="cuda").manual_seed(seed + 1 if data_seed is None else 7919 * data_seed + 1)
    for pid, (name, row, col) in s.source_of.items():
        kind, bound = s.kinds[pid]
        if bound >= P:
            continue
        t, w, h, b = tensors[name]
        v = torch.randint(0, bound, (calls,), dtype=torch.int64, device="cuda", generator=g)
        t[col * h + row: col * h + row + b * calls: b] = ((v << 32) % P).to(torch.int32)  # Montgomery form
    if out is None:
        out = torch.empty(apc.width * H, dtype=torch.int32, device="cuda")
    if per is None:
        per = tg.Periphery.fresh()
    cons_bc, cons_spans = apc.compile_constraints()
    return dict(synth=s, apc=apc, instr_air=instr_air, air_names=order, dummy=dummy, tensors=tensors, out=out, per=per,
                calls=calls, log_h=log_h, H=H, W=apc.width, cons=(cons_bc, cons_spans), src_bytes=src_bytes)


def _synth_lib():
    """libpowdr_synth.so (powdr_amd/synth_csrc/synth_fill.hip): the benches' input generator, a library of its own."""
    global _SYNTH
    if _SYNTH is None:
        from . import build as b

        if not b.SYNTH_LIB.exists():
            raise RuntimeError(f"{b.SYNTH_LIB} is missing: run `python -m powdr_amd.build`")
        _SYNTH = C.CDLL(str(b.SYNTH_LIB))
        _SYNTH.powdr_synth_fill_sources.restype = C.c_int
        _SYNTH.powdr_synth_fill_sources.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    return _SYNTH


_SYNTH = None


def source_bounds(wl) -> dict:
    """Per source matrix of an APC workload: the device table bounds[col * b + row] of the cells that feed bounded APC columns
    (0 = any field element) — what powdr_synth_fill_sources needs to refill the matrix from a seed."""
    s = wl["synth"]
    tabs = {n: np.zeros(w * b, np.uint32) for n, (t, w, h, b) in wl["tensors"].items()}
    for pid, (name, row, col) in s.source_of.items():
        kind, bound = s.kinds[pid]
        if bound < P:
            t, w, h, b = wl["tensors"][name]
            cur = tabs[name][col * b + row]
            tabs[name][col * b + row] = bound if cur == 0 else min(int(cur), bound)
    return {n: torch.from_numpy(a.view(np.int32)).cuda() for n, a in tabs.items()}


def refill_sources(wl, bounds: dict, seed: int) -> None:
    """The original chips' dummy traces of this APC, regenerated on the device from `seed` (one write-only launch per matrix, on the
    library's launch stream): uniform field elements, cells that feed bounded columns drawn below their bound."""
    lib, st = _synth_lib(), abi.lib.powdr_gpu_get_stream()
    for k, (n, (t, w, h, b)) in enumerate(wl["tensors"].items()):
        abi.check(lib.powdr_synth_fill_sources(t.data_ptr(), w, h, b, bounds[n].data_ptr(), (seed * 1000003 + k * 7919 + 1) & ((1 << 64) - 1), st),
                  "powdr_synth_fill_sources")


def apc_air_shape(name: str, width: int, log_h: int, config_id: int) -> synth.Shape:
    """A synthetic APC of `width` columns at the keccak APC's densities (187 constraints / 1 734 interactions per 2 022 columns) with
    DENSE gather sources (about 1.3 source cells per APC cell instead of keccak's 13.6: ten such AIRs must fit beside the prover) and
    honest bitwise lookups (range pairs (x, y, 0, 0): a synthetic APC has no column that holds x ^ y)."""
    nc, ni = max(2, round(width * 187 / 2022)), max(8, round(width * 1734 / 2022))
    cells = int(width * 1.3) + 60
    airs = [("BaseAlu", 36, max(1, round(cells * 0.40 / 36))), ("Shift", 53, max(1, round(cells * 0.25 / 53))),
            ("LoadStore", 41, max(1, round(cells * 0.35 / 41)))]
    return synth.Shape(name, width, log_h, nc, ni, airs, n_quotient=min(4, max(1, width // 40)), config_id=config_id, honest_bitwise=True)


def instruction_block(counts: dict, seed: int):
    """A block with `counts[kind]` instructions of every chip kind (opcodes drawn from the chip's range, operands in the shape the
    program ROM holds them: original_chips.sanitise_instructions), loads / stores word-aligned."""
    rng = np.random.default_rng(seed)
    ins = []
    for lo, hi, k in oc.OPCODE_RANGES:
        for _ in range(counts.get(k, 0)):
            op = int(rng.integers(lo, hi + 1))
            if k == oc.LOAD_STORE:
                op = int(rng.choice([528, 531]))  # LOADW / STOREW: any word-aligned pointer is legal
            ins.append([op] + [int(x) for x in rng.integers(0, 1 << 12, size=7)])
    ins = [ins[i] for i in rng.permutation(len(ins))]
    out = oc.sanitise_instructions(ins)
    for i in out:
        k = oc.kind_of_opcode(i[0])
        if k in (oc.LOAD_STORE, oc.LOAD_SIGN_EXTEND, oc.JALR):
            i[3] &= 0x7FFC  # a positive, word-aligned 16-bit immediate
            i[7] = 0
        if k == oc.LOAD_SIGN_EXTEND:
            i[0] = 535  # LOADH: half-word aligned
    return out


def plausible_records(table: oc.InstructionTable, num_calls: int, seed: int):
    """Call records whose rows pass every lookup of their chips: random operand words (their limbs are bytes by construction),
    base registers of loads / stores / jumps that give small word-aligned targets, timestamps that increase."""
    rec = oc.random_records_device(table, num_calls, seed=seed).view(table.words_per_call, num_calls)
    g = torch.Generator(device="cuda").manual_seed(seed + 7)
    for e in table.entries:
        if e.kind in (oc.LOAD_STORE, oc.LOAD_SIGN_EXTEND, oc.JALR):
            rec[e.rec_off] = torch.randint(0, 1 << 22, (num_calls,), dtype=torch.int32, device="cuda", generator=g) << 2  # rs1: < 2^24, 4-aligned
    return rec.reshape(-1)


class RecordStager:
    """plausible_records without the per-instruction launches (a dozen launches for the whole block): the index tensors are built
    once, fill() draws the records of one segment into an existing buffer on the library's launch stream."""

    def __init__(self, table: oc.InstructionTable, num_calls: int):
        self.words, self.calls = table.words_per_call, num_calls
        rows, delta, ptr_rows = [], [], []
        for e in table.entries:
            n_prev = oc.N_PREV_TS[e.kind]
            first = e.rec_off + oc.RECORD_WORDS[e.kind] - n_prev
            rows += list(range(first, first + n_prev))
            delta += [e.ts_delta] * n_prev
            if e.kind in (oc.LOAD_STORE, oc.LOAD_SIGN_EXTEND, oc.JALR):
                ptr_rows.append(e.rec_off)
        dev = lambda a, dt: torch.tensor(a, dtype=dt, device="cuda")
        self.rows, self.delta, self.ptr_rows = dev(rows, torch.int64), dev(delta, torch.int32), dev(ptr_rows, torch.int64)

    def fill(self, rec_flat: torch.Tensor, seed: int, calls: int | None = None) -> None:
        """calls: this segment's number of calls (<= the number the buffer was sized for); the records are word-major, so a
        segment with fewer calls occupies the first words * calls words of the buffer."""
        s = abi.lib.powdr_gpu_get_stream()
        import contextlib

        calls = self.calls if calls is None else calls
        assert 0 < calls <= self.calls
        with (torch.cuda.stream(torch.cuda.ExternalStream(int(s))) if s else contextlib.nullcontext()):
            g = torch.Generator(device="cuda").manual_seed(seed)
            rec = rec_flat[:self.words * calls].view(self.words, calls)
            rec.random_(-(1 << 31), (1 << 31) - 1, generator=g)
            base = torch.randint(1 << 10, 1 << 26, (calls,), dtype=torch.int32, device="cuda", generator=g)
            rec[0] = base
            if len(self.rows):
                gap = torch.randint(1, 1 << 20, (len(self.rows), calls), dtype=torch.int32, device="cuda", generator=g)
                rec[self.rows] = torch.clamp(base[None, :] + self.delta[:, None] - gap, min=0)
            if len(self.ptr_rows):  # rs1 of loads / stores / jumps: < 2^24, 4-aligned
                rec[self.ptr_rows] = torch.randint(0, 1 << 22, (len(self.ptr_rows), calls), dtype=torch.int32, device="cuda", generator=g) << 2


def _offset_operands(ibc: np.ndarray, ispans: np.ndarray, height: int) -> np.ndarray:
    """post-fix programs with COLUMN operands -> the reference ABI's element offsets col * H (cuda/mod.rs:61-63)"""
    out = np.array(ibc, dtype=np.uint32, copy=True)
    for off, ln in np.asarray(ispans).reshape(-1, 2).tolist():
        ip = off
        while ip < off + ln:
            op = int(out[ip])
            if op == 0:
                out[ip + 1] = int(out[ip + 1]) * height
                ip += 2
            elif op == 1:
                ip += 2
            else:
                ip += 1
    return out


class _BusReplay:
    """_apc_apply_bus on an arbitrary trace with an interaction table kept on the device (uploaded once)."""

    def __init__(self, interactions, height: int):
        inter, ispans, ibc = interactions
        inter = np.ascontiguousarray(inter, np.uint32).reshape(-1, 3)
        ispans = np.ascontiguousarray(ispans, np.uint32).reshape(-1, 2)
        self.n_bc, self.n_int, self.n_sp = len(ibc), len(inter), len(ispans)
        bc = _offset_operands(np.ascontiguousarray(ibc, np.uint32), ispans, height)
        self.d = [torch.from_numpy(a.view(np.int32).reshape(-1).copy()).cuda() for a in (bc, inter, ispans)]
        self.height = height

    def __call__(self, d_trace_ptr: int, p: tg.Periphery):
        rc = abi.lib._apc_apply_bus(d_trace_ptr, self.height, self.d[0].data_ptr(), self.n_bc, self.d[1].data_ptr(), self.n_int,
                                    self.d[2].data_ptr(), self.n_sp, p.var_bus, p.var_hist.data_ptr(), p.var_hist.numel(),
                                    p.tuple_bus, p.tuple_hist.data_ptr(), p.tuple_sizes[0], p.tuple_sizes[1],
                                    p.bitwise_bus, p.bitwise_hist.data_ptr())
        abi.check(rc, "_apc_apply_bus")


def draw_segment_shape(seed: int, u: int, n_segments: int, apc_max_calls, instr_max_calls: int) -> dict:
    """HonestSegment.draw_shape as a pure function of the caps (testable without a GPU): per APC chip its number of calls, for the
    instruction chips the number of block executions, of segment u of n_segments."""
    rng = np.random.default_rng([1 + seed, 7919 + u])
    n = len(apc_max_calls) + 1  # + the instruction block
    if u == 0:
        f = np.ones(n)
    elif u == n_segments - 1:
        f = 2.0 ** -rng.uniform(3.0, 5.0, size=n)
    else:
        f = 2.0 ** -rng.uniform(0.0, 2.0, size=n)
        f[int(rng.integers(0, n))] = 1.0
    # (at least 3 calls: the library sizes an APC trace as next_pow2(calls) rows and the provers want >= 4 rows, the shapes' minimum)
    apc_calls = [max(min(3, cap), int(cap * f[k])) for k, cap in enumerate(apc_max_calls)]
    return dict(segment=u, apc_calls=apc_calls, instr_calls=max(1, int(instr_max_calls * f[-1])))


class HonestSegment:
    """kind "C4": 10 APC AIRs (widths synth.C4_APC_WIDTHS, 2^max_log_height rows) + 13 instruction AIRs + 3 periphery AIRs.
    kind "C5": reth-shaped — the APC AIRs of synth.segment_shape("C5") (log-uniform heights and widths) + the same system AIRs."""

    def __init__(self, kind: str, max_log_height: int = 20, seed: int = 0, queries: int = 100, pow_bits: int = 16, logup: bool = True,
                 max_apc_airs: int | None = None, specialise_all: bool = False):
        self.kind, self.logup, self.queries, self.pow_bits = kind, logup, queries, pow_bits
        shrink = 20 - max_log_height
        self.per = tg.Periphery.fresh()
        self.airs = []  # dict(name, role, width, log_h, cons, inter, trace(tensor), prover)
        # ---- APC AIRs
        apc_shapes = [s for s in synth.segment_shape(kind, seed=seed, max_log_height=max_log_height) if s[0].startswith("apc")]
        if max_apc_airs is not None:
            apc_shapes = apc_shapes[:max_apc_airs]
        self.apcs = []
        for k, (name, w, lh, _, _) in enumerate(apc_shapes):
            wl = build_apc_workload(apc_air_shape(name, w, lh, config_id=4000 + k), lh, True, seed=seed * 131 + k, per=self.per)
            self.apcs.append(wl)
            self.airs.append(dict(name=name, role="apc", width=wl["W"], log_h=lh, cons=wl["cons"], inter=wl["apc"].compile_bus(1), trace=wl["out"]))
        # ---- instruction AIRs: one block whose per-kind instruction counts give the reference heights at 2^10 calls
        self.calls = max(4, 1 << max(0, 10 - shrink))
        counts = {}
        for n, lh in synth.REFERENCE_AIR_LOG_HEIGHTS.items():
            k = oc.KIND_NAMES.index(n)
            counts[k] = max(1, (1 << max(2, lh - shrink)) // self.calls)
        self.block = instruction_block(counts, seed + 99)
        self.table = oc.InstructionTable(self.block, [True] * len(self.block), 0x200000)
        self.records = plausible_records(self.table, self.calls, seed + 5)
        heights = oc.dummy_trace_heights(self.table, self.calls)
        self.instr_bufs, self.replays = [None] * oc.N_KINDS, []
        for k, n in enumerate(oc.KIND_NAMES):
            if not heights[k]:
                continue
            t = torch.zeros(oc.WIDTHS[k] * heights[k], dtype=torch.int32, device="cuda")
            self.instr_bufs[k] = (t.data_ptr(), heights[k])
            bc, sp, it = synth.reference_air_programs(n)
            self.airs.append(dict(name=n, role="instruction", width=oc.WIDTHS[k], log_h=heights[k].bit_length() - 1, cons=(bc, sp), inter=it, trace=t))
            self.replays.append((_BusReplay(it, heights[k]), t))
        # ---- periphery AIRs (no constraints: lookup tables with multiplicity columns; the chips are external, the layouts ours)
        empty = (np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32))
        self.per_traces = {}
        for name, w, rows, it in (("var_range", 3, self.per.var_hist.numel(), periphery.var_range_interactions(self.per.var_bus)),
                                  ("tuple2", 3, self.per.tuple_hist.numel(), periphery.tuple2_interactions(self.per.tuple_bus)),
                                  ("bitwise", 5, 65536, periphery.bitwise_interactions(self.per.bitwise_bus))):
            t = torch.zeros(w * rows, dtype=torch.int32, device="cuda")
            self.per_traces[name] = t
            self.airs.append(dict(name=name, role="periphery", width=w, log_h=rows.bit_length() - 1, cons=empty, inter=it, trace=t))
        for a in self.airs:
            a["prover"] = prover.Prover(a["width"], a["cons"][0], a["cons"][1], num_queries=queries, pow_bits=pow_bits,
                                        interactions=a["inter"] if logup else None)
        # the AIR set of an execution is fixed at key generation and proven in every segment: its specialised kernels are compiled once,
        # for EVERY AIR (pw_provers_specialise), not only for those whose first trace happens to be tall (the library's own rule when it
        # meets a prover for the first time inside a proof)
        self.specialised = prover.specialise_all([a["prover"] for a in self.airs]) if specialise_all else None
        self.source_bytes = sum(wl["src_bytes"] for wl in self.apcs) + self.records.numel() * 4
        self._bounds, self._stager, self.data_seed = None, None, None
        # ---- the shape the buffers were sized for (every AIR at its cap); a segment's OWN heights: set_shape()
        self.seed = seed
        self.max_calls = self.calls
        for wl in self.apcs:
            wl["max_calls"], wl["max_tensors"] = wl["calls"], dict(wl["tensors"])
        self._instr = [(k, a) for a in self.airs if a["role"] == "instruction" for k in [oc.KIND_NAMES.index(a["name"])]]
        self._replay_cache = {(k, self.instr_bufs[k][1]): r for (k, _), (r, _) in zip(self._instr, self.replays)}
        self.shape = None  # None: the maximal shape
        self._refresh()

    def _refresh(self):
        self.cells = sum(a["width"] << a["log_h"] for a in self.airs)
        self.cells_by_role = {r: sum(a["width"] << a["log_h"] for a in self.airs if a["role"] == r) for r in ("apc", "instruction", "periphery")}
        self.seg = [(a["prover"], a["trace"].data_ptr(), a["log_h"]) for a in self.airs]

    # ---- segments that differ in SHAPE ---------------------------------------------------------------------
    def draw_shape(self, u: int, n_segments: int = 8) -> dict:
        """The trace heights of segment u as a metered execution would report them
        (/root/reference/openvm/src/trace_generation.rs:113-131: every segment carries its own `trace_heights`; a segment is cut
        when ONE chip reaches its height limit, /root/reference/openvm-riscv/src/lib.rs:270-286 — the others are wherever they are).
        Per APC chip the number of calls, for the instruction chips the number of block executions:
          u == 0              every chip at its cap (the shape the buffers are sized for; under a device budget the segment
                              whose tall AIR crosses the streaming threshold)
          u == n_segments-1   the execution's tail: every chip at <= 1/8 of its cap
          otherwise           one chip (drawn) at its cap, the others log-uniform over the two octaves below theirs."""
        return draw_segment_shape(self.seed, u, n_segments, [wl["max_calls"] for wl in self.apcs], self.max_calls)

    def set_shape(self, shape: dict | None) -> None:
        """Re-shape the resident segment: every APC AIR gets next_pow2(calls) rows (cuda/mod.rs:266: `next_power_of_two_or_zero`), the
        gather sources behind it row_block_size * calls rows, every instruction AIR next_pow2(its rows of the block * executions).
        The buffers stay (they hold the maximal shape); matrices are column-major, so a shorter one occupies a prefix. The inputs
        must be staged afterwards (stage_inputs): the sources' layout depends on their height."""
        if shape is None:
            shape = dict(apc_calls=[wl["max_calls"] for wl in self.apcs], instr_calls=self.max_calls)
        stream = abi.lib.powdr_gpu_get_stream()
        import contextlib

        apc_airs = [a for a in self.airs if a["role"] == "apc"]
        for wl, a, calls in zip(self.apcs, apc_airs, shape["apc_calls"]):
            assert 0 < calls <= wl["max_calls"] and (calls >= 3 or calls == wl["max_calls"])
            lh = max(2, (calls - 1).bit_length())
            wl["calls"], wl["log_h"], wl["H"] = calls, lh, 1 << lh
            a["log_h"] = lh
            wl["tensors"], wl["dummy"] = {}, []
            for n in wl["air_names"]:
                t, w, h_max, b = wl["max_tensors"][n]
                h = max(4, (b * calls + 3) // 4 * 4)
                assert h <= h_max
                wl["tensors"][n] = (t, w, h, b)
                wl["dummy"].append((t.data_ptr(), w, h))
        calls = shape["instr_calls"]
        assert 0 < calls <= self.max_calls
        changed = calls != self.calls
        self.calls = calls
        heights = oc.dummy_trace_heights(self.table, calls)
        self.replays = []
        with (torch.cuda.stream(torch.cuda.ExternalStream(int(stream))) if stream else contextlib.nullcontext()):
            for k, a in self._instr:
                h = heights[k]
                t = a["trace"]
                assert oc.WIDTHS[k] * h <= t.numel()
                if changed:
                    t[:oc.WIDTHS[k] * h].zero_()  # padding rows are only ever written here (the expanders write the rows of real calls)
                self.instr_bufs[k] = (t.data_ptr(), h)
                a["log_h"] = h.bit_length() - 1
                if (k, h) not in self._replay_cache:
                    self._replay_cache[(k, h)] = _BusReplay(a["inter"], h)
                self.replays.append((self._replay_cache[(k, h)], t))
        self.shape = shape
        self._refresh()

    def heights(self) -> list:
        return [a["log_h"] for a in self.airs]

    def shape_heights(self, shape: dict | None) -> list:
        """log2 heights of every AIR (in self.airs order) a segment of this shape has — without re-shaping anything"""
        if shape is None:
            shape = dict(apc_calls=[wl["max_calls"] for wl in self.apcs], instr_calls=self.max_calls)
        apc = iter(shape["apc_calls"])
        ih = oc.dummy_trace_heights(self.table, shape["instr_calls"])
        out = []
        for a in self.airs:
            if a["role"] == "apc":
                out.append(max(2, (next(apc) - 1).bit_length()))
            elif a["role"] == "instruction":
                out.append(ih[oc.KIND_NAMES.index(a["name"])].bit_length() - 1)
            else:
                out.append(a["log_h"])
        return out

    def shape_cells(self, shape: dict | None) -> int:
        """main-trace cells (rows x columns over all AIRs) of a segment of this shape: what the placement balances"""
        return sum(a["width"] << lh for a, lh in zip(self.airs, self.shape_heights(shape)))

    # ---- the inputs of ANOTHER segment of the same execution (same chips, other rows) ---------------------------------------
    def stage_inputs(self, data_seed: int, shape: dict | None = None):
        """Replace this segment's inputs — the original chips' dummy traces behind every APC AIR and the call records of the
        instruction AIRs — by those of segment `data_seed`, generated on the device in the buffers the resident segment already has
        (the reference cuts ONE execution into segments: /root/reference/openvm-riscv/src/lib.rs:585-592 — the same AIRs, different
        rows; nothing about the AIRs, their programs or the provers changes). Runs on the library's launch stream, in order with the
        trace generation that follows."""
        if self._bounds is None:
            self._bounds = [source_bounds(wl) for wl in self.apcs]
            self._stager = RecordStager(self.table, self.max_calls)
        if shape is not None or self.shape is not None:
            self.set_shape(shape)  # this segment's own heights (shape None: back to the caps)
        for k, wl in enumerate(self.apcs):
            refill_sources(wl, self._bounds[k], data_seed * 4099 + k)
        self._stager.fill(self.records, data_seed * 31 + 5, self.calls)
        self.data_seed = data_seed

    # ---- the timed pieces -----------------------------------------------------------------------------------------------
    def generate_traces(self):
        """Trace generation of the whole segment: APC AIRs (gather + derived columns + bus replay), instruction AIRs (record
        expansion + replay of their lookups), periphery AIRs from the histograms every other AIR filled."""
        p = self.per
        p.zero()  # on the library's launch stream: this runs on worker threads' streams too (pw_prove_segments_multi)
        for wl in self.apcs:
            wl["apc"].generate_witness_gpu(wl["instr_air"], wl["dummy"], wl["calls"], wl["out"].data_ptr(), p)
        oc.expand(self.records.data_ptr(), self.calls, self.table, self.instr_bufs)
        for replay, t in self.replays:
            replay(t.data_ptr(), p)
        abi.check(abi.lib.powdr_periphery_var_range_trace(p.var_hist.data_ptr(), p.var_hist.numel(), self.per_traces["var_range"].data_ptr()), "var_range_trace")
        abi.check(abi.lib.powdr_periphery_tuple2_trace(p.tuple_hist.data_ptr(), p.tuple_sizes[0], p.tuple_sizes[1], self.per_traces["tuple2"].data_ptr()), "tuple2_trace")
        abi.check(abi.lib.powdr_periphery_bitwise_trace(p.bitwise_hist.data_ptr(), self.per_traces["bitwise"].data_ptr()), "bitwise_trace")

    def prove(self, copy: bool = False, hand_over=None):
        """hand_over: pw_prove_segment_consuming — every chip moves its trace into the engine (cuda/mod.rs:415-419); the next
        generate_traces() rewrites them anyway."""
        if hand_over is True:  # (the instruction AIRs' padding rows are written once per shape, not per generation: those traces stay ours)
            hand_over = [a["role"] != "instruction" for a in self.airs]
        return prover.prove_segment(self.seg, logup=self.logup, copy=copy, hand_over=hand_over)

    # ---- checks (outside the timed region) ------------------------------------------------------------------------------
    def descriptions(self, buses=None):
        return [(a["width"], a["log_h"], a["cons"][0], a["cons"][1], a["inter"] if buses is None else periphery.select_buses(a["inter"], buses)) for a in self.airs]

    def verify(self, proof) -> int:
        """pw_verify_segment on the whole statement (every constraint identity, every LogUp column, openings, FRI, queries)."""
        return int(prover.verify_segment(self.descriptions(), proof, self.queries, self.pow_bits, self.logup)[0])

    def check_constraints(self) -> int:
        """the device's mock prover on the current traces: violated (row, constraint) pairs over all AIRs"""
        return sum(a["prover"].check_constraints(a["trace"].data_ptr(), a["log_h"])[0] for a in self.airs if len(a["cons"][1]))

    def balance_witness(self):
        """The lookup buses (3 var-range, 6 bitwise, 7 tuple) have their senders AND receivers in this segment: the same traces,
        every AIR restricted to those buses, proven and verified with check_balance — the senders' LogUp sums and the periphery AIRs'
        cancel exactly. Returns (verify code, total sum words)."""
        provers = [prover.Prover(a["width"], a["cons"][0], a["cons"][1], num_queries=min(self.queries, 8), pow_bits=0,
                                 interactions=periphery.select_buses(a["inter"], LOOKUP_BUSES)) for a in self.airs]
        seg = [(pr, a["trace"].data_ptr(), a["log_h"]) for pr, a in zip(provers, self.airs)]
        proof = prover.prove_segment(seg, logup=True, copy=True)
        rc, total = prover.verify_segment(self.descriptions(LOOKUP_BUSES), proof, min(self.queries, 8), 0, True, check_balance=True)[:2]
        for pr in provers:
            pr.close()
        return int(rc), total

    def device_bytes(self) -> int:
        return sum(a["prover"].device_bytes() for a in self.airs if a.get("prover") is not None)

    def release_provers(self):
        """give the provers' device buffers back (the traces and the trace generators stay): room for the balance witness"""
        for a in self.airs:
            if a.get("prover") is not None:
                a["prover"].close()
                a["prover"] = None
        torch.cuda.empty_cache()

    def close(self):
        self.release_provers()
        for wl in self.apcs:
            wl["apc"].close()
