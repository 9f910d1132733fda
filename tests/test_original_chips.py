"""SURVEY.md §8 row (f)-1, producer half: the five original RV32IM chips of a keccak autoprecompile as record -> row expanders
(powdr_amd/csrc/original_chips.hip), and the APC gather fused into them (powdr_apc_tracegen_records).

The chips are EXTERNAL to the reference checkout; their columns and constraints are not (openvm-riscv/tests/
openvm_constraints.txt, parsed into tests/golden/openvm_airs.npz). CPU: the numpy restatement (oracle/original_chips.py)
satisfies every one of those constraints — for all opcodes, and for the reference's real keccak block, whose own pre-optimisation
APC constraints (28 627) then hold on the gathered APC trace. GPU: device expansion == restatement, the device's mock prover
finds no violation, fused == expand + _apc_tracegen == oracle."""
from pathlib import Path

import numpy as np
import pytest

from oracle import apc_model as om
from oracle import original_chips as oc
from powdr_amd import synth

GOLDEN = Path(__file__).parent / "golden"
P = om.P


def all_opcode_block(seed=5, copies=4):
    rng = np.random.default_rng(seed)
    ins = []
    for op in [512, 513, 514, 515, 516, 517, 518, 519, 528, 531, 544, 545, 560, 561] * copies:
        ins.append([op] + [int(x) for x in rng.integers(0, 1 << 20, size=7)])
    ins = oc.sanitise_instructions(ins)
    # a load into x0 (needs_write = 0), a JAL that does not write, a store with a negative offset, a large shift immediate
    ins += [[528, 0, 8, 0xFFF0, 1, 2, 0, 1], [560, 0, 0, 16, 1, 0, 0, 0], [531, 12, 8, 0x8000, 1, 2, 1, 1], [519, 4, 8, 31, 1, 0, 0, 0],
            [518, 4, 8, 0, 1, 0, 0, 0], [513, 4, 8, 0xFFFFFF, 1, 0, 0, 0]]
    return ins


def keccak_block():
    z = np.load(GOLDEN / "keccak_apc_pre_opt.apc.npz")
    return z, z["instructions"].tolist(), int(z["start_pc"][0])


def edge_records(table, wpc, calls, seed):
    rec = oc.random_records(table, wpc, calls, seed=seed)
    for ins in table:  # data words at their extremes in the first calls (timestamps stay consistent)
        o, k = int(ins["rec_off"]), int(ins["kind"])
        n_data = oc.RECORD_WORDS[k] - {0: 3, 1: 3, 2: 3, 3: 2, 4: 1}[k]
        for w in range(n_data):
            rec[o + w, 0] = 0
            rec[o + w, 1] = 0xFFFFFFFF if k != oc.KIND_LOAD_STORE or w else 0x0FFFFFFC
            rec[o + w, 2] = 0x80000000 if k != oc.KIND_LOAD_STORE or w else 0x00000004
    return rec


@pytest.mark.parametrize("block", ["all_opcodes", "keccak"])
def test_restatement_satisfies_the_reference_constraints(block):
    if block == "keccak":
        z, ins, start_pc = keccak_block()
    else:
        ins, start_pc = all_opcode_block(), 0x200000
    table, idx, rbs, wpc = oc.build_instruction_table(ins, [True] * len(ins), start_pc)
    calls = 40
    traces = oc.expand_dummy_traces(table, edge_records(table, wpc, calls, seed=1), rbs)
    for k, name in enumerate(oc.KIND_NAMES):
        bc, sp, _ = synth.reference_air_programs(name)
        t = traces[k]
        assert t.shape[0] == oc.WIDTHS[k] == synth.REFERENCE_AIR_WIDTHS[name]
        bad, first = oc.check_constraints(bc, sp, [t[c] for c in range(t.shape[0])])  # padding rows (all zero) included
        assert bad == 0, (name, first)
        assert int(np.count_nonzero(t[:, : rbs[k] * calls])) > 0


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


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, original_chips as pc, prover, tracegen

    assert torch.cuda.is_available()
    return torch, abi, pc, prover, tracegen


def _device_traces(gpu, t, rec, calls):
    torch, abi, pc, prover, tg = gpu
    d_rec = torch.from_numpy(rec.view(np.int32).reshape(-1).copy()).cuda()
    heights = pc.dummy_trace_heights(t, calls)
    bufs = [torch.zeros(pc.WIDTHS[k] * heights[k], dtype=torch.int32, device="cuda") if heights[k] else None for k in range(5)]
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
    for k, name in enumerate(oc.KIND_NAMES):
        got = om.from_monty(bufs[k].cpu().numpy().view(np.uint32)).reshape(pc.WIDTHS[k], heights[k])
        assert heights[k] == want[k].shape[1] and (got == want[k]).all(), name
        # the device's mock prover (the reference's prove_mock / debug_proving_ctx) on the chip's own constraints: no violation
        bc, sp, _ = synth.reference_air_programs(name)
        pr = prover.Prover(pc.WIDTHS[k], bc, sp, num_queries=1)
        assert pr.check_constraints(bufs[k].data_ptr(), heights[k].bit_length() - 1) == (0, None, None), name
        # one wrong cell is found: an opcode / validity flag of the first row, off by one (a result limb would not do — XOR / OR / AND
        # results are only constrained through the bitwise lookup bus, not algebraically)
        flag_col = {"BaseAlu": 31, "Shift": 31, "LoadStore": 27, "BranchEqual": 20, "JalLui": 16}[name]
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
def test_fused_keccak_block_satisfies_the_apc_constraints_on_the_device(gpu):
    """The reference's real keccak block end to end on the device: records -> powdr_apc_tracegen_records -> the 27 521-column
    pre-optimisation APC trace; pw_prover_check_constraints with the APC's own 28 627 constraints finds no violation."""
    torch, abi, pc, prover, tg = gpu
    z, ins, start_pc = keccak_block()
    t = pc.InstructionTable(ins, [True] * len(ins), start_pc)
    table, _, rbs, wpc = oc.build_instruction_table(ins, [True] * len(ins), start_pc)
    calls, H, W = 256, 256, len(z["poly_ids"])  # no padding rows: the unoptimised machine's constraints pin pcs and operands on every row
    rec = oc.random_records(table, wpc, calls, seed=9)
    d_rec = torch.from_numpy(rec.view(np.int32).reshape(-1).copy()).cuda()
    kinds = [oc.KIND_NAMES.index(str(n)) for n in z["air_names"]]
    rsubs, n = t.record_substitutions(z["subs"], kinds)
    out = tg.DeviceMatrix.zeros(H, W)
    pc.tracegen_records(out.ptr(), H, d_rec.data_ptr(), calls, t, rsubs, n)
    torch.cuda.synchronize()
    pr = prover.Prover(W, z["cons_bc"], z["cons_spans"], num_queries=1)
    assert pr.check_constraints(out.ptr(), 8) == (0, None, None)
    pr.close()
    with pytest.raises(abi.HipError):  # a column beyond the chip's width
        bad = (pc.PowdrRecordSubst * 1)(pc.PowdrRecordSubst(0, 60, 0))
        pc.tracegen_records(out.ptr(), H, d_rec.data_ptr(), calls, t, bad, 1)
