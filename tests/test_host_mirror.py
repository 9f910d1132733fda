"""The C++ host mirror (include/powdr_host.h) against the oracle's independent Python
restatement of the same reference code: identical column order, bytecode, spans and
Subst tables — and against the committed golden hashes of the reference's own fixtures."""
import gzip
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

GOLDEN = Path(__file__).parent / "golden"


def air_ids(apc_doc_instructions, names):
    """instr_air for the C++ host: ids by first appearance of the AIR name."""
    order = []
    out = []
    for n in names:
        if n not in order:
            order.append(n)
        out.append(order.index(n))
    return out, order


def check_against_oracle(doc, instr_names, heights=(1, 8, 1 << 20)):
    from powdr_amd import host

    h_apc = host.Apc(doc)
    apc = om.load_apc(json.loads(doc) if isinstance(doc, (bytes, bytearray)) else doc)
    idx = apc.poly_id_to_index()
    assert h_apc.width == len(idx)
    assert (h_apc.poly_ids() == np.array(sorted(idx), dtype=np.uint64)).all()
    assert h_apc.n_constraints == len(apc.constraints) and h_apc.n_bus == len(apc.bus_interactions)
    assert h_apc.opcodes() == [i[0] for i in apc.instructions]
    assert h_apc.num_subs() == [len(s) for s in apc.subs]
    for H in heights:
        inter, spans, bc = h_apc.compile_bus(H)
        o_inter, o_spans, o_bc = om.compile_bus(apc, idx, H)
        assert (inter == o_inter).all() and (spans == o_spans).all() and (bc == o_bc).all()
        specs, dbc = h_apc.compile_derived(H)
        cb, offs, lens, o_dbc = om.compile_derived(apc, idx, H)
        assert (specs["col_base"] == cb).all() and (specs["off"] == offs).all() and (specs["len"] == lens).all()
        assert (dbc == o_dbc).all()
    cbc, cspans = h_apc.compile_constraints()
    o_cbc, o_cspans = sm.compile_constraints(apc, idx)
    assert (cbc == o_cbc).all() and (cspans == o_cspans).all()
    ids, order = air_ids(None, instr_names)
    subs, air_id_out, rbs = h_apc.build_substitutions(ids)
    by_obj = {id(ins): n for ins, n in zip(apc.instructions, instr_names)}
    gt = om.build_gpu_tables(apc, idx, air_of=lambda ins: by_obj[id(ins)])
    assert (subs == gt.subs).all()
    assert [order[i] for i in air_id_out] == gt.air_names and rbs.tolist() == gt.row_block_size
    return h_apc, apc, bc


@pytest.mark.parametrize("shape", ["T0", "T1", "C1"])
def test_cpp_host_matches_oracle_on_synthetic_apcs(shape):
    s = synth.generate(shape, seed=2)
    names = [n if n else om.opcode_air(ins[0]) for n, ins in zip(s.instr_air, s.doc["block"]["blocks"][0]["instructions"])]
    check_against_oracle(s.doc, names)


def test_cpp_host_rejects_malformed_documents():
    from powdr_amd import host

    s = synth.generate("T0", seed=2)
    for bad in (b"{", b'{"block":{}}', json.dumps({**s.doc, "subs": s.doc["subs"][:-1]}).encode()):
        with pytest.raises(ValueError):
            host.Apc(bad)
    doc = json.loads(json.dumps(s.doc))
    doc["machine"]["constraints"][0] = ["nope", "*", 3]
    with pytest.raises(ValueError, match="AlgebraicReference"):
        host.Apc(doc)


@pytest.mark.parametrize("name", ["single_div_nondet", "keccak_apc_pre_opt", "wasm_register_reuse"])
def test_cpp_host_on_reference_fixtures(name, reference_dir):
    """The reference's own APC fixtures: same counts as its tests pin
    (autoprecompiles/tests/optimizer.rs:66-84) and the same bytecode as the golden hash."""
    raw = gzip.open(reference_dir / "autoprecompiles/tests" / f"{name}.json.gz").read()
    doc = json.loads(raw)
    instrs = [i for b in doc["block"]["blocks"] for i in b["instructions"]]
    names = [om.opcode_air(i[0]) for i in instrs]
    h_apc, apc, _ = check_against_oracle(raw, names, heights=(1,))
    summ = json.loads((GOLDEN / "apc_fixtures_summary.json").read_text())[name]
    assert h_apc.width == summ["main_columns"] and h_apc.n_bus == summ["bus_interactions"]
    assert h_apc.n_constraints == summ["constraints"]
    inter, spans, bc = h_apc.compile_bus(1)
    assert hashlib.sha256(bc.tobytes()).hexdigest() == summ["bus_bytecode_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,calls", [("T0", 9), ("T1", 777), ("C1", 1500)])
def test_generate_witness_gpu_matches_oracle(shape, calls):
    """powdr_apc_generate_witness_gpu == try_generate_witness: JSON in, trace + histograms out."""
    import torch
    from powdr_amd import host, tracegen as tg
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate(shape, seed=13)
    apc, idx, want, hist, (bufs, dims, gt, order) = run_oracle_gpu_convention(s, calls, seed=13)
    W, H = want.shape
    names = [n if n else om.opcode_air(ins[0]) for n, ins in zip(s.instr_air, s.doc["block"]["blocks"][0]["instructions"])]
    ids, air_order = air_ids(None, names)
    name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
    dev = []
    dummy = []
    for n in air_order:
        if n in name_to:
            _, w, h, _ = dims[name_to[n]]
            t = torch.from_numpy(om.to_monty(bufs[name_to[n]]).view(np.int32)).cuda()
            dev.append(t)
            dummy.append((t.data_ptr(), w, h))
        else:
            dummy.append((0, 0, 0))
    h_apc = host.Apc(s.doc)
    out = torch.full((H * W,), 7, dtype=torch.int32, device="cuda")  # zero-filled by the callee
    per = tg.Periphery.fresh()
    for _ in range(2):  # second call exercises the cached tables
        for t in (per.var_hist, per.tuple_hist, per.bitwise_hist):
            t.zero_()
        h_apc.generate_witness_gpu(ids, dummy, calls, out.data_ptr(), per)
        torch.cuda.synchronize()
        assert (om.from_monty(out.cpu().numpy().view(np.uint32)).reshape(W, H) == want).all()
        assert (per.var_hist.cpu().numpy().view(np.uint32) == hist["var"]).all()
        assert (per.tuple_hist.cpu().numpy().view(np.uint32) == hist["tuple"]).all()
        assert (per.bitwise_hist.cpu().numpy().view(np.uint32) == hist["bitwise"]).all()


def test_xbc_compiler_preserves_values():
    """The plan-time compiler (post-fix -> accumulator code with folding and reordering) against the
    oracle's evaluator, on random expressions and on the identities it folds."""
    import ctypes as C

    from powdr_amd import abi
    from tests.test_oracle_apc import _random_expr

    lib = abi.lib
    lib.powdr_xbc_eval_host.restype = C.c_int
    rng = np.random.default_rng(5)
    W, H = 10, 4
    ids = list(range(W))
    idx = {p: p for p in ids}
    trace = rng.integers(0, om.P, size=W * H, dtype=np.uint32)
    trace[0:H] = 0  # a zero column: exercises x*0, 0-x, inv_or_zero(0)
    tm = om.to_monty(trace)

    def check(bc, r):
        bc = np.array(bc, dtype=np.uint32)
        try:
            want = om.c_eval_expr(bc, trace, r)
        except ValueError:
            want = None
        res, n = C.c_uint32(), C.c_uint32()
        rc = lib.powdr_xbc_eval_host(bc.ctypes.data_as(C.c_void_p), C.c_uint32(len(bc)), tm.ctypes.data_as(C.c_void_p),
                                     C.c_size_t(r), C.byref(res), C.byref(n))
        if want is None:
            assert rc != 0
            return None
        assert rc == 0
        got = int(om.from_monty(np.array([res.value], dtype=np.uint32))[0])
        assert got == want, (bc.tolist(), r)
        return n.value

    for depth in (2, 4, 6, 7):
        for _ in range(150):
            e = _random_expr(rng, ids, depth)
            bc = []
            om.emit_expr(bc, e, idx, H)
            check(bc, int(rng.integers(H)))
    A, Cn = om.OP_PUSH_APC, om.OP_PUSH_CONST
    x, y = [A, 1 * H], [A, 2 * H]
    cases = [
        [Cn, 0] + x + [om.OP_ADD], x + [Cn, 0, om.OP_SUB], [Cn, 0] + x + [om.OP_SUB], x + [Cn, 1, om.OP_MUL],
        x + [Cn, 0, om.OP_MUL], [Cn, 5, Cn, 7, om.OP_MUL, Cn, 3, om.OP_SUB], x + [om.OP_NEG, om.OP_NEG],
        [Cn, 9] + x + y + [om.OP_ADD, om.OP_SUB], x + y + [om.OP_MUL] + x + y + [om.OP_SUB, om.OP_SUB],
        [A, 0, om.OP_INV_OR_ZERO] + x + [om.OP_MUL], x + [om.OP_INV_OR_ZERO] + x + [om.OP_MUL], [Cn, 4, om.OP_INV_OR_ZERO],
    ]
    for bc in cases:
        for r in range(H):
            check(bc, r)
    assert check(x + [Cn, 1, om.OP_MUL, Cn, 0, om.OP_ADD], 0) == 1  # folds to a single load
    # malformed programs are rejected (the kernels then keep the post-fix interpreter)
    for bad in ([om.OP_ADD], x + y, [Cn], [7, 0], x + [om.OP_MUL]):
        assert check(bad, 0) is None


def test_small_form_analysis_preserves_values():
    """The fast bus kernel evaluates multiplicities and arguments that are bilinear in at most two columns from a
    closed form (csrc/small_form.hpp). Whatever the analysis accepts must evaluate like the expression; shapes it
    must accept (the ones bus interactions have) and shapes it must refuse are pinned."""
    import ctypes as C

    from powdr_amd import abi
    from tests.test_oracle_apc import _random_expr

    lib = abi.lib
    lib.powdr_small_form_eval_host.restype = C.c_int
    rng = np.random.default_rng(11)
    W, H = 6, 4
    ids = list(range(W))
    idx = {p: p for p in ids}
    trace = rng.integers(0, om.P, size=W * H, dtype=np.uint32)
    tm = om.to_monty(trace)

    def run(bc, r):
        bc = np.array(bc, dtype=np.uint32)
        res, fl = C.c_uint32(), C.c_uint32()
        rc = lib.powdr_small_form_eval_host(bc.ctypes.data_as(C.c_void_p), C.c_uint32(len(bc)), tm.ctypes.data_as(C.c_void_p),
                                            C.c_size_t(r), C.byref(res), C.byref(fl))
        if rc != 0:
            return None, None
        want = om.c_eval_expr(bc, trace, r)
        got = int(om.from_monty(np.array([res.value], dtype=np.uint32))[0])
        assert got == want, (bc.tolist(), r)
        return got, fl.value

    accepted = 0
    for depth in (1, 2, 3, 4, 5):
        for _ in range(300):
            e = _random_expr(rng, ids[:3], depth)  # few columns: many expressions stay within two
            bc = []
            om.emit_expr(bc, e, idx, H)
            if om.OP_INV_OR_ZERO in bc[::1] and run(bc, 0)[0] is None:
                continue
            for r in range(H):
                accepted += run(bc, r)[0] is not None
    assert accepted > 300
    A, Cn, ADD, SUB, MUL, NEG = om.OP_PUSH_APC, om.OP_PUSH_CONST, om.OP_ADD, om.OP_SUB, om.OP_MUL, om.OP_NEG
    x, y, z = [A, 1 * H], [A, 2 * H], [A, 3 * H]
    must_accept = {
        "column": (x, 8 | 1), "constant": ([Cn, 17], 16), "255 - byte": ([Cn, 255] + x + [SUB], 1),
        "byte + 256 byte": (x + [Cn, 256] + y + [MUL, ADD], 1 | 2), "valid * flag": (x + y + [MUL], 1 | 2 | 4),
        "valid * 3": (x + [Cn, 3, MUL], 1), "x - x + y": (x + x + [SUB] + y + [ADD], 8 | 1),
        "(x + 1)(y + 2)": (x + [Cn, 1, ADD] + y + [Cn, 2, ADD, MUL], 1 | 2 | 4), "-(x) * 0 + 5": (x + [NEG, Cn, 0, MUL, Cn, 5, ADD], 16),
    }
    for name, (bc, flags) in must_accept.items():
        for r in range(H):
            v, fl = run(bc, r)
            assert v is not None and fl == flags, (name, fl)
    must_refuse = {"square": x + x + [MUL], "three columns": x + y + [ADD] + z + [ADD], "x y x": x + y + [MUL] + x + [MUL],
                   "inverse": x + [om.OP_INV_OR_ZERO], "malformed": x + [ADD]}
    for name, bc in must_refuse.items():
        assert run(bc, 0)[0] is None, name
