"""SURVEY.md §8 row (f)-1, producer half: the thirteen original RV32IM chips an autoprecompile is built from as record -> row
expanders (powdr_amd/csrc/original_chips.hip), and the APC gather fused into them (powdr_apc_tracegen_records).

The chips are EXTERNAL to the reference checkout; their columns, constraints and bus interactions are not (openvm-riscv/tests/
openvm_constraints.txt, parsed into tests/golden/openvm_airs.npz). CPU: the numpy restatement (oracle/original_chips.py)
satisfies every one of those constraints and every row's bus interactions are legal ones (range checks in range, PC lookup = the
instruction, memory bus = a word-level RV32IM model) — for all 36 opcodes, on edge records, and for the reference's real keccak
block, whose own pre-optimisation APC constraints (28 627) then hold on the gathered APC trace; a mutation sweep shows the two
checks pin every cell the reference's proof pins. GPU: device expansion == restatement, the device's mock prover finds no
violation, fused == expand + _apc_tracegen == oracle."""
from pathlib import Path

import numpy as np
import pytest

from oracle import apc_model as om
from oracle import original_chips as oc
from powdr_amd import synth

GOLDEN = Path(__file__).parent / "golden"
P = om.P


def all_opcode_block(seed=5, copies=3):
    rng = np.random.default_rng(seed)
    ins = []
    for op in oc.ALL_OPCODES * copies:
        ins.append([op] + [int(x) for x in rng.integers(0, 1 << 20, size=7)])
    ins = oc.sanitise_instructions(ins)
    # a load into x0 (needs_write = 0), a JAL that does not write, a store with a negative offset, a large shift immediate, negative
    # immediates for SUB / SLT / SLTU, a backward branch, a JALR and a LOADB that do not write
    ins += [[528, 0, 8, 0xFFF0, 1, 2, 0, 1], [560, 0, 0, 16, 1, 0, 0, 0], [531, 12, 8, 0x8000, 1, 2, 1, 1], [519, 4, 8, 31, 1, 0, 0, 0],
            [518, 4, 8, 0, 1, 0, 0, 0], [513, 4, 8, 0xFFFFFF, 1, 0, 0, 0], [520, 4, 8, 0xFFFF80, 1, 0, 0, 0], [521, 4, 8, 0xFFFF80, 1, 0, 0, 0],
            [549, 4, 8, P - 64, 1, 1, 0, 0], [565, 0, 8, 0xFFFC, 1, 0, 0, 1], [534, 0, 8, 3, 1, 2, 0, 0]]
    return ins


def keccak_block():
    z = np.load(GOLDEN / "keccak_apc_pre_opt.apc.npz")
    return z, z["instructions"].tolist(), int(z["start_pc"][0])


def edge_records(table, wpc, calls, seed):
    rec = oc.random_records(table, wpc, calls, seed=seed)
    for ins in table:  # data words at their extremes in the first calls (timestamps stay consistent)
        o, k = int(ins["rec_off"]), int(ins["kind"])
        n_data = oc.RECORD_WORDS[k] - oc.N_PREV_TS[k]
        for w in range(n_data):
            rec[o + w, 0], rec[o + w, 1], rec[o + w, 2] = 0, 0xFFFFFFFF, 0x80000000
        if n_data >= 2 and calls >= 8:  # the division corner cases (and whatever they mean to the other two-operand chips)
            for call, (x, y) in enumerate([(0x80000000, 0xFFFFFFFF), (7, 0), (0xFFFFFFF9, 2), (7, 0xFFFFFFFE), (0xFFFFFFFF, 5)], start=3):
                rec[o, call], rec[o + 1, call] = x, y
    return rec


@pytest.mark.parametrize("block", ["all_opcodes", "keccak"])
def test_restatement_satisfies_the_reference_constraints_and_lookups(block):
    if block == "keccak":
        z, ins, start_pc = keccak_block()
    else:
        ins, start_pc = all_opcode_block(), 0x200000
    table, idx, rbs, wpc = oc.build_instruction_table(ins, [True] * len(ins), start_pc)
    calls = 40
    rec = edge_records(table, wpc, calls, seed=1)
    traces = oc.expand_dummy_traces(table, rec, rbs)
    assert sorted(traces) == [k for k in range(oc.N_KINDS) if rbs[k]] and (block == "keccak" or len(traces) == 13)
    for k, t in traces.items():
        name = oc.KIND_NAMES[k]
        bc, sp, _ = synth.reference_air_programs(name)
        assert t.shape[0] == oc.WIDTHS[k] == synth.REFERENCE_AIR_WIDTHS[name]
        bad, first = oc.check_constraints(bc, sp, [t[c] for c in range(t.shape[0])])  # padding rows (all zero) included
        assert bad == 0, (name, first)
        assert int(np.count_nonzero(t[:, : rbs[k] * calls])) > 0
    # every bus interaction of every row: range checks in range, bitwise rows, PC lookup = the instruction (for the keccak block: the
    # reference's REAL instructions), execution bridge and memory bus = the word-level RV32IM model
    for ins_row in table:
        fails = oc.check_interactions(synth.reference_air_programs(oc.KIND_NAMES[int(ins_row["kind"])])[2], ins_row, rec, oc.expand_rows(ins_row, rec, rec[0]))
        assert not fails, fails[:3]


def test_constraints_and_lookups_pin_every_cell_the_proof_pins():
    """Mutation sweep: +1 on any single column of any opcode's rows is caught by the algebraic constraints or by the interaction
    checks — except for the columns the reference's AIRs themselves leave free (auxiliary columns of a disabled memory access,
    r_inv of an unsigned division)."""
    ins = all_opcode_block(copies=1)
    table, idx, rbs, wpc = oc.build_instruction_table(ins, [True] * len(ins), 0x200000)
    rec = edge_records(table, wpc, 24, seed=4)
    z = np.load(GOLDEN / "openvm_airs.npz")
    names = [str(x) for x in z["names"]]
    free_ok = ("prev_timestamp", "timestamp_lt_aux", "prev_data", "r_inv")
    seen = set()
    for t in table:
        key = (int(t["opcode"]), int(t["e"]), int(t["f"]))
        if key in seen:
            continue
        seen.add(key)
        name = oc.KIND_NAMES[int(t["kind"])]
        bc, sp, inter = synth.reference_air_programs(name)
        colnames = [str(x) for x in z[f"a{names.index(name)}_columns"]]
        rows = [np.asarray(r).astype(np.int64) % P for r in oc.expand_rows(t, rec, rec[0])]
        for c in range(len(rows)):
            mut = list(rows)
            mut[c] = (rows[c] + 1) % P
            if oc.check_constraints(bc, sp, mut)[0] or oc.check_interactions(inter, t, rec, mut):
                continue
            assert any(x in colnames[c] for x in free_ok), (name, key, colnames[c])
            disabled = (int(t["e"]) == 0 and "reads_aux__1" in colnames[c]) or int(t["f"]) == 0 or (name == "DivRem" and key[0] in (597, 599))
            assert disabled, (name, key, colnames[c])
    assert {key[0] for key in seen} == set(oc.ALL_OPCODES) and len(oc.ALL_OPCODES) == 36


def test_keccak_block_apc_constraints_hold_on_the_gathered_trace():
    """Records -> dummy traces -> the reference's gather (oracle) -> the APC trace of the reference's OWN pre-optimisation keccak
    APC: all of its 28 627 constraints vanish on every row that holds a call (they pin pcs, operands and flags of the 677
    instructions too — which is also why the all-zero padding rows of this UNOPTIMISED machine do not satisfy them: the is_valid
    guards come with the optimiser, autoprecompiles/src/lib.rs:470-524)."""
    z, ins, start_pc = keccak_block()
    table, idx, rbs, wpc = oc.build_instruction_table(ins, [True] * len(ins), start_pc)
    assert len(table) == 677 and wpc * 4 < 17000  # < 17 KB of records per call (the dummy traces: 110 KB)
    calls, H = 12, 16
    traces = oc.expand_dummy_traces(table, oc.random_records(table, wpc, calls, seed=3), rbs)
    kind_of = {n: i for i, n in enumerate(oc.KIND_NAMES)}
    names = [str(n) for n in z["air_names"]]
    assert [rbs[kind_of[n]] for n in names] == z["row_block_size"].tolist()
    W = len(z["poly_ids"])
    trace = om.c_apc_tracegen(H, W, [traces[kind_of[n]].reshape(-1) for n in names], [traces[kind_of[n]].shape[1] for n in names],
                              z["row_block_size"], z["subs"], calls).reshape(W, H)
    bad, first = oc.check_constraints(z["cons_bc"], z["cons_spans"], [trace[c, :calls] for c in range(W)])
    assert len(z["cons_spans"]) == 28627 and bad == 0, first
    assert (trace[:, calls:] == 0).all()


def test_product_host_table_equals_the_restatement():
    from powdr_amd import original_chips as pc

    for ins, start_pc in [(all_opcode_block(), 0x200000), keccak_block()[1:]]:
        has = [i % 7 != 3 for i in range(len(ins))]
        t = pc.InstructionTable(ins, has, start_pc)
        table, idx, rbs, wpc = oc.build_instruction_table(ins, has, start_pc)
        assert len(t) == len(table) and t.row_block_size == rbs and t.words_per_call == wpc
        for e, o in zip(t.entries, table):
            assert [getattr(e, n) for n, _ in pc.PowdrOrigInstr._fields_] == [int(o[n]) for n in o.dtype.names]
    assert pc.sanitise_instructions(all_opcode_block()[:56]) == oc.sanitise_instructions(all_opcode_block()[:56])


def test_library_expanders_on_the_host_equal_the_restatement():
    """The product's OWN expander code (csrc/original_chips.hip: host-device templates, here run on the host through
    powdr_original_row_expand_host) against the numpy restatement: every cell of every opcode's row, on edge records and on random
    ones — the check the GPU tests repeat on the device, available to a CPU-only run."""
    from powdr_amd import original_chips as pc

    ins = all_opcode_block()
    t = pc.InstructionTable(ins, [True] * len(ins), 0x200000)
    table, idx, rbs, wpc = oc.build_instruction_table(ins, [True] * len(ins), 0x200000)
    calls = 24
    rec = edge_records(table, wpc, calls, seed=8)
    seen = set()
    for entry, row in zip(t.entries, table):
        want = np.stack([np.asarray(v) % P for v in oc.expand_rows(row, rec, rec[0])]).astype(np.uint32)  # [width, calls]
        o, n = int(row["rec_off"]), oc.RECORD_WORDS[int(row["kind"])]
        for r in range(calls):
            got = pc.expand_row_host(entry, rec[o:o + n, r], int(rec[0, r]) + int(row["ts_delta"]))
            assert (got == want[:, r]).all(), (oc.KIND_NAMES[int(row["kind"])], int(row["opcode"]), r, np.nonzero(got != want[:, r])[0][:4])
        seen.add(int(row["opcode"]))
    assert seen == set(oc.ALL_OPCODES)
    bad = pc.PowdrOrigInstr(0, 9999, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1)
    with pytest.raises(ValueError):
        pc.expand_row_host(bad, [0] * 6, 0)


@pytest.mark.parametrize("shape", ["T1", "C2"])
def test_typed_substitutions_feed_bounded_columns_from_bounded_cells(shape):
    """original_chips.typed_substitutions: the synthetic APC keeps the same columns per instruction, every instruction's cells stay
    distinct, and on FRESH records the restatement's rows (oracle/original_chips.py expand_rows) hold a value below the bound in
    every cell that feeds a bounded column (bit, tri, byte, k-bit range) — what makes the flow from records look up in range."""
    from powdr_amd import original_chips as pc

    s = synth.generate(shape, seed=0)
    blk = s.doc["block"]["blocks"][0]
    ins = oc.sanitise_instructions(blk["instructions"])
    typed, untyped = pc.typed_substitutions(ins, s.doc["subs"], s.kinds, int(blk["start_pc"]), seed=1)
    assert untyped == 0
    again, _ = pc.typed_substitutions(ins, s.doc["subs"], s.kinds, int(blk["start_pc"]), seed=1)
    assert again == typed
    moved = 0
    for before, after in zip(s.doc["subs"], typed):
        assert sorted(x["apc_poly_id"] for x in before) == sorted(x["apc_poly_id"] for x in after)
        cols = [x["original_poly_index"] for x in after]
        assert len(set(cols)) == len(cols) and cols == sorted(cols)
        was = {x["apc_poly_id"]: x["original_poly_index"] for x in before}
        moved += sum(was[x["apc_poly_id"]] != x["original_poly_index"] for x in after)
        for x in after:  # an unbounded column keeps its cell unless a bounded one needed it
            if s.kinds[x["apc_poly_id"]][1] >= P and was[x["apc_poly_id"]] != x["original_poly_index"]:
                assert was[x["apc_poly_id"]] in cols
    assert moved > 0
    has = [len(x) > 0 for x in typed]
    table, index, rbs, wpc = oc.build_instruction_table(ins, has, int(blk["start_pc"]))
    calls = 24
    rec = oc.random_records(table, wpc, calls, seed=77)
    checked = 0
    for row, i in zip(table, index):
        cells = oc.expand_rows(row, rec, rec[0])
        for x in typed[int(i)]:
            bound = s.kinds[x["apc_poly_id"]][1]
            if bound < P:
                v = np.broadcast_to(np.asarray(cells[x["original_poly_index"]]) % P, (calls,))
                assert int(v.max()) < bound, (oc.KIND_NAMES[int(row["kind"])], x, int(v.max()), bound)
                checked += 1
    assert checked == sum(1 for r in typed for x in r if s.kinds[x["apc_poly_id"]][1] < P) > 0


def test_library_instruction_table_equals_the_restatement():
    """powdr_apc_instruction_table (C++ host library, from the APC's own block and substitutions) == the Python mirror == the
    restatement, for the synthetic C2 APC (instructions without a surviving cell have no entry but still advance the timestamp)."""
    from powdr_amd import host, original_chips as pc

    s = synth.generate("C2", seed=0)
    doc = s.doc
    blk = doc["block"]["blocks"][0]
    has = [len(x) > 0 for x in doc["subs"]]
    h_apc = host.Apc(doc)
    arr, n, words = h_apc.instruction_table()
    table, idx, rbs, wpc = oc.build_instruction_table(blk["instructions"], has, int(blk["start_pc"]))
    assert n == len(table) == sum(has) and words == wpc
    for e, o in zip(arr, table):
        assert [getattr(e, name) for name, _ in pc.PowdrOrigInstr._fields_] == [int(o[name]) for name in o.dtype.names]
    h_apc.close()


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, original_chips as pc, prover, tracegen

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (run with -m gpu on the GPU box)")
    return torch, abi, pc, prover, tracegen


def _device_traces(gpu, t, rec, calls):
    torch, abi, pc, prover, tg = gpu
    d_rec = torch.from_numpy(rec.view(np.int32).reshape(-1).copy()).cuda()
    heights = pc.dummy_trace_heights(t, calls)
    bufs = [torch.zeros(pc.WIDTHS[k] * heights[k], dtype=torch.int32, device="cuda") if heights[k] else None for k in range(pc.N_KINDS)]
    pc.expand(d_rec.data_ptr(), calls, t, [(b.data_ptr(), heights[k]) if b is not None else None for k, b in enumerate(bufs)])
    torch.cuda.synchronize()
    return d_rec, bufs, heights


@pytest.mark.gpu
@pytest.mark.parametrize("block,calls", [("all_opcodes", 300), ("keccak", 70)])
def test_device_expansion_matches_the_restatement_and_the_constraints(gpu, block, calls):
    torch, abi, pc, prover, tg = gpu
    if block == "keccak":
        z, ins, start_pc = keccak_block()
    else:
        ins, start_pc = all_opcode_block(), 0x200000
    t = pc.InstructionTable(ins, [True] * len(ins), start_pc)
    table, idx, rbs, wpc = oc.build_instruction_table(ins, [True] * len(ins), start_pc)
    rec = edge_records(table, wpc, calls, seed=2)
    want = oc.expand_dummy_traces(table, rec, rbs)
    d_rec, bufs, heights = _device_traces(gpu, t, rec, calls)
    assert sorted(want) == [k for k in range(pc.N_KINDS) if heights[k]]
    for k in sorted(want):
        name = oc.KIND_NAMES[k]
        got = om.from_monty(bufs[k].cpu().numpy().view(np.uint32)).reshape(pc.WIDTHS[k], heights[k])
        assert heights[k] == want[k].shape[1] and (got == want[k]).all(), name
        # the device's mock prover (the reference's prove_mock / debug_proving_ctx) on the chip's own constraints: no violation
        bc, sp, _ = synth.reference_air_programs(name)
        pr = prover.Prover(pc.WIDTHS[k], bc, sp, num_queries=1)
        assert pr.check_constraints(bufs[k].data_ptr(), heights[k].bit_length() - 1) == (0, None, None), name
        # one wrong cell is found: an opcode / validity flag of the first row, off by one (a result limb would not do — XOR / OR / AND
        # results are only constrained through the bitwise lookup bus, not algebraically)
        flag_col = {"BaseAlu": 31, "Shift": 31, "LoadStore": 27, "BranchEqual": 20, "JalLui": 16, "LessThan": 28, "BranchLessThan": 20, "Jalr": 23,
                    "LoadSignExtend": 23, "DivRem": 55, "MulH": 36, "Multiplication": 30, "Auipc": 10}[name]
        bufs[k][flag_col * heights[k]] += int(om.to_monty(np.array([1], np.uint32))[0])
        n_bad, row, _ = pr.check_constraints(bufs[k].data_ptr(), heights[k].bit_length() - 1)
        assert n_bad >= 1 and row == 0, name
        pr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("calls", [1, 37, 128, 1000])
def test_fused_records_gather_equals_expand_then_gather(gpu, calls):
    """powdr_apc_tracegen_records == powdr_original_airs_expand + _apc_tracegen == the oracle's gather of the restated traces,
    on the synthetic C2 APC's own substitutions (2 017 scattered cells of 27 521; instructions without a surviving cell have no
    row and no record) — padding rows zero, untouched columns untouched, duplicate destinations: the last wins."""
    torch, abi, pc, prover, tg = gpu
    s = synth.generate("C2", seed=0)
    apc = om.load_apc(s.doc)
    idx = apc.poly_id_to_index()
    gt = om.build_gpu_tables(apc, idx)
    ins = oc.sanitise_instructions(apc.instructions)
    has = [len(x) > 0 for x in apc.subs]
    t = pc.InstructionTable(ins, has, 0x200000)
    table, _, rbs, wpc = oc.build_instruction_table(ins, has, 0x200000)
    kinds = [oc.KIND_NAMES.index(n) for n in gt.air_names]
    assert [rbs[k] for k in kinds] == gt.row_block_size
    rec = oc.random_records(table, wpc, calls, seed=calls)
    H, W = synth.next_pow2_or_zero(calls), len(idx)
    H = max(H, 2)
    traces = oc.expand_dummy_traces(table, rec, rbs)
    want = om.c_apc_tracegen(H, W, [traces[k].reshape(-1) for k in kinds], [traces[k].shape[1] for k in kinds], gt.row_block_size, gt.subs, calls)
    d_rec, bufs, heights = _device_traces(gpu, t, rec, calls)
    # (1) the reference flow on the device: expanded dummy traces -> _apc_tracegen
    out1 = tg.DeviceMatrix.zeros(H, W)
    keep = tg.apc_tracegen(out1, [(bufs[k], pc.WIDTHS[k], heights[k], rbs[k]) for k in kinds], gt.subs, calls)
    # (2) fused
    rsubs, n = t.record_substitutions(gt.subs, kinds)
    out2 = tg.DeviceMatrix(torch.full((H * W,), 0x55, dtype=torch.int32, device="cuda"), H, W)
    pc.tracegen_records(out2.ptr(), H, d_rec.data_ptr(), calls, t, rsubs, n)
    torch.cuda.synchronize()
    got1 = om.from_monty(out1.buf.cpu().numpy().view(np.uint32))
    assert (got1 == want).all()
    m2 = out2.buf.view(W, H)
    untouched = sorted(set(range(W)) - set(gt.subs[:, 3].tolist()))
    assert untouched and all(int((m2[c] != 0x55).sum()) == 0 for c in untouched)
    m2[untouched] = 0
    assert (om.from_monty(out2.buf.cpu().numpy().view(np.uint32)) == want).all()
    del keep


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["keccak_apc_pre_opt", "ecrecover_apc_pre_opt"])
def test_fused_real_blocks_satisfy_the_apc_constraints_on_the_device(gpu, fixture):
    """The reference's real blocks end to end on the device: records -> powdr_apc_tracegen_records -> the pre-optimisation APC trace
    (keccak: 27 521 columns; ecrecover: 750 instructions incl. MUL / MULHU / SLTU / AUIPC / JALR, on the records of an execution);
    pw_prover_check_constraints with the APC's own constraints (28 627 / 23 629) finds no violation, and the trace equals the
    restatement's gather word for word."""
    from oracle import rv32_vm

    torch, abi, pc, prover, tg = gpu
    z = np.load(GOLDEN / f"{fixture}.apc.npz")
    ins, start_pc = z["instructions"].tolist(), int(z["start_pc"][0])
    t = pc.InstructionTable(ins, [True] * len(ins), start_pc)
    table, _, rbs, wpc = oc.build_instruction_table(ins, [True] * len(ins), start_pc)
    W = len(z["poly_ids"])
    if fixture.startswith("keccak"):
        calls = H = 256  # no padding rows: the unoptimised machine's constraints pin pcs and operands on every row
        rec = oc.random_records(table, wpc, calls, seed=9)
    else:
        calls = H = 8
        rec, _ = rv32_vm.execute_block(table, [start_pc + 4 * i for i in range(len(ins))], wpc, calls, seed=5)
    d_rec = torch.from_numpy(rec.view(np.int32).reshape(-1).copy()).cuda()
    kinds = [oc.KIND_NAMES.index({"Mul": "Multiplication"}.get(str(n), str(n))) for n in z["air_names"]]
    rsubs, n = t.record_substitutions(z["subs"], kinds)
    out = tg.DeviceMatrix.zeros(H, W)
    pc.tracegen_records(out.ptr(), H, d_rec.data_ptr(), calls, t, rsubs, n)
    torch.cuda.synchronize()
    pr = prover.Prover(W, z["cons_bc"], z["cons_spans"], num_queries=1)
    assert pr.check_constraints(out.ptr(), H.bit_length() - 1) == (0, None, None)
    pr.close()
    traces = oc.expand_dummy_traces(table, rec, rbs)
    want = om.c_apc_tracegen(H, W, [traces[k].reshape(-1) for k in kinds], [traces[k].shape[1] for k in kinds], z["row_block_size"].tolist(), z["subs"], calls)
    assert (om.from_monty(out.buf.cpu().numpy().view(np.uint32)) == want).all()
    with pytest.raises(abi.HipError):  # a column beyond the chip's width
        bad = (pc.PowdrRecordSubst * 1)(pc.PowdrRecordSubst(0, 60, 0))
        pc.tracegen_records(out.ptr(), H, d_rec.data_ptr(), calls, t, bad, 1)
    with pytest.raises(abi.HipError):  # one source cell to two APC columns
        twice = (pc.PowdrRecordSubst * 2)(pc.PowdrRecordSubst(0, 1, 0), pc.PowdrRecordSubst(0, 1, 1))
        pc.tracegen_records(out.ptr(), H, d_rec.data_ptr(), calls, t, twice, 2)


@pytest.mark.gpu
def test_one_segment_proof_of_all_thirteen_chips_from_records_verifies(gpu):
    """Records -> the thirteen instruction AIRs on the device -> ONE segment proof (pw_prove_segment) with the reference's REAL
    constraints and bus interactions of every AIR (LogUp phase included) -> accepted by the product's and the oracle's verifier;
    with one flag cell of one AIR off by one the constraint identity of exactly that AIR fails."""
    from oracle import stark_model as sm

    torch, abi, pc, prover, tg = gpu
    ins = all_opcode_block()
    t = pc.InstructionTable(ins, [True] * len(ins), 0x200000)
    table, _, rbs, wpc = oc.build_instruction_table(ins, [True] * len(ins), 0x200000)
    calls = 100
    rec = edge_records(table, wpc, calls, seed=11)
    d_rec, bufs, heights = _device_traces(gpu, t, rec, calls)
    nq = 6
    provers, airs, descs = [], [], []
    for k, name in enumerate(oc.KIND_NAMES):
        bc, sp, it = synth.reference_air_programs(name)
        lh = heights[k].bit_length() - 1
        provers.append(prover.Prover(pc.WIDTHS[k], bc, sp, interactions=it, num_queries=nq))
        airs.append((provers[-1], bufs[k].data_ptr(), lh))
        descs.append((pc.WIDTHS[k], lh, bc, sp, it))
    proof = prover.prove_segment(airs, logup=True)
    assert prover.verify_segment(descs, proof, nq, 0, True)[0] == 0
    oracle_airs = [(None, pc.WIDTHS[k], descs[k][1], descs[k][2], descs[k][3], descs[k][4]) for k in range(pc.N_KINDS)]  # traces are not needed to verify
    assert sm.verify_segment(proof, oracle_airs, nq, 0, True)[0] == 0
    victim = pc.DIV_REM
    bufs[victim][55 * heights[victim] + 1] += int(om.to_monty(np.array([1], np.uint32))[0])  # opcode_div_flag of the second row
    bad = prover.prove_segment(airs, logup=True)
    assert prover.verify_segment(descs, bad, nq, 0, True)[0] == ((victim + 1) << 8) | 2
    for pr in provers:
        pr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("calls,typed", [(100, False), (1024, False), (1024, True), (20000, True)])
def test_generate_witness_from_records_equals_the_reference_flow(gpu, calls, typed):
    """powdr_apc_generate_witness_from_records (a1-a3 from call records, one host call) == powdr_original_airs_expand on the same
    records followed by powdr_apc_generate_witness_gpu (the reference flow: dummy traces -> gather -> derived -> bus): the same trace
    in every column and the same three periphery histograms. typed: with the substitutions of the bench's timed step from records
    (original_chips.typed_substitutions) — every bounded column of the generated trace then stays below its bound; 20 000 calls take
    the binned histogram path."""
    from powdr_amd import host

    torch, abi, pc, prover, tg = gpu
    s = synth.generate("C2", seed=0)
    doc = dict(s.doc)
    blk = dict(doc["block"]["blocks"][0])
    blk["instructions"] = oc.sanitise_instructions(blk["instructions"])
    doc["block"] = dict(doc["block"], blocks=[blk])
    if typed:
        doc["subs"], untyped = pc.typed_substitutions(blk["instructions"], doc["subs"], s.kinds, int(blk["start_pc"]))
        assert untyped == 0
    h_apc = host.Apc(doc)
    has = [len(x) > 0 for x in doc["subs"]]
    t = pc.InstructionTable(blk["instructions"], has, int(blk["start_pc"]))
    arr, n, words = h_apc.instruction_table()
    assert n == len(t) and words == t.words_per_call
    rec = pc.random_records_device(t, calls, seed=calls)
    H, W = synth.next_pow2_or_zero(calls), h_apc.width
    # the reference flow: full dummy traces, then the three-stage witness generation
    apc = om.load_apc(doc)
    gt = om.build_gpu_tables(apc, apc.poly_id_to_index())
    names = list(dict.fromkeys(om.opcode_air(int(i[0])) for i, h_ in zip(blk["instructions"], has) if h_))
    kind = lambda name: oc.KIND_NAMES.index({"Mul": "Multiplication"}.get(name, name))
    heights = pc.dummy_trace_heights(t, calls)
    bufs = [torch.zeros(pc.WIDTHS[k] * heights[k], dtype=torch.int32, device="cuda") if heights[k] else None for k in range(pc.N_KINDS)]
    pc.expand(rec.data_ptr(), calls, t, [(b.data_ptr(), heights[k]) if b is not None else None for k, b in enumerate(bufs)])
    instr_air = [names.index(om.opcode_air(int(i[0]))) for i in blk["instructions"]]
    dummy = [(bufs[kind(nm)].data_ptr(), pc.WIDTHS[kind(nm)], heights[kind(nm)]) for nm in names]
    out1, out2 = tg.DeviceMatrix.zeros(H, W), tg.DeviceMatrix(torch.full((H * W,), 0x55, dtype=torch.int32, device="cuda"), H, W)
    per1, per2 = tg.Periphery.fresh(), tg.Periphery.fresh()
    h_apc.generate_witness_gpu(instr_air, dummy, calls, out1.ptr(), per1)
    h_apc.generate_witness_from_records(rec.data_ptr(), calls, out2.ptr(), per2)
    torch.cuda.synchronize()
    assert torch.equal(out1.buf, out2.buf)
    for a, b in ((per1.var_hist, per2.var_hist), (per1.tuple_hist, per2.tuple_hist), (per1.bitwise_hist, per2.bitwise_hist)):
        assert torch.equal(a, b)
    assert int(per2.var_hist.sum()) > 0
    if typed:
        index = apc.poly_id_to_index()
        bounded = [(index[pid], b) for pid, (_, b) in s.kinds.items() if b < P and pid in s.source_of]
        view = out2.buf.view(W, H)[:, :calls]
        for c, b in bounded[:: max(1, len(bounded) // 200)]:
            assert int(om.from_monty(view[c].cpu().numpy().astype(np.uint32)).max()) < b, (c, b)
    h_apc.close()
