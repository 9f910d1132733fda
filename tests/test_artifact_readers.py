"""APC artifact readers of the host library: the reference's three on-disk formats, read without the
Rust toolchain — `ApcWithBusMap` JSON exports (autoprecompiles/src/export.rs:77-93,271-276), the CLI's serde_cbor stage
artifacts (cli-openvm-riscv/src/main.rs:380-407) and `apc_candidates.json` v4 (autoprecompiles/src/pgo/cell/mod.rs:34-97).
CPU only. The reference checkout has no .cbor fixture (the blobs are missing), so the CBOR documents are
produced here from the reference's own JSON fixtures with a minimal RFC 8949 encoder that writes what serde_cbor writes
(maps with text keys, definite lengths, shortest integer forms)."""
import gzip
import hashlib
import json
import struct

import numpy as np
import pytest

from powdr_amd import host


def cbor(v) -> bytes:
    def head(major, n):
        if n < 24:
            return bytes([major << 5 | n])
        for info, fmt, lim in ((24, ">B", 1 << 8), (25, ">H", 1 << 16), (26, ">I", 1 << 32), (27, ">Q", 1 << 64)):
            if n < lim:
                return bytes([major << 5 | info]) + struct.pack(fmt, n)
        raise ValueError(n)

    if v is None:
        return b"\xf6"
    if v is True:
        return b"\xf5"
    if v is False:
        return b"\xf4"
    if isinstance(v, int):
        return head(0, v) if v >= 0 else head(1, -1 - v)
    if isinstance(v, float):
        return b"\xfb" + struct.pack(">d", v)
    if isinstance(v, str):
        b = v.encode()
        return head(3, len(b)) + b
    if isinstance(v, (list, tuple)):
        return head(4, len(v)) + b"".join(cbor(x) for x in v)
    if isinstance(v, dict):
        return head(5, len(v)) + b"".join(cbor(k) + cbor(x) for k, x in v.items())
    raise TypeError(type(v))


def fingerprint(apc: host.Apc):
    """Everything the prover and the trace generator consume, hashed."""
    h = hashlib.sha256()
    h.update(apc.poly_ids().tobytes())
    for a in apc.compile_bus(64) + apc.compile_derived(64) + apc.compile_constraints():
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(np.array(apc.opcodes(), np.uint32).tobytes())
    h.update(np.array(apc.num_subs(), np.uint32).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def fixture_doc(reference_dir):
    return json.load(gzip.open(reference_dir / "autoprecompiles" / "tests" / "single_div_nondet.json.gz"))


def test_apc_with_bus_map_export(reference_dir):
    """The reference's tests/*.json.gz fixtures ARE ApcWithBusMap exports: the machine and the bus map are read."""
    for name, n_cols in (("single_div_nondet", None), ("apc_reth_op_bug", None)):
        raw = gzip.open(reference_dir / "autoprecompiles" / "tests" / f"{name}.json.gz").read()
        doc = json.loads(raw)
        apc = host.Apc(raw)
        assert host.count_apcs(raw) == 1
        bm = apc.bus_map()
        want = doc["bus_map"]["bus_ids"]
        assert len(bm) == len(want) and [b[0] for b in bm] == [int(k) for k in want]
        kinds = {b[0]: b[1] for b in bm}
        assert kinds[0] == "ExecutionBridge" and kinds[1] == "Memory" and kinds[2] == "PcLookup"
        var, tup, sizes, bw = apc.periphery_bus_ids()
        # ids are per document (the reth-op fixture has the tuple checker on bus 8, not the default 7 of
        # openvm-bus-interaction-handler/src/bus_map.rs:9-14): compare with the document itself
        other = {int(k): v["Other"] for k, v in want.items() if isinstance(v, dict)}
        exp_var = [k for k, v in other.items() if v == "VariableRangeChecker"]
        exp_bw = [k for k, v in other.items() if v == "BitwiseLookup"]
        exp_tup = [(k, tuple(v["TupleRangeChecker"])) for k, v in other.items() if isinstance(v, dict)]
        assert [var] == exp_var and [bw] == exp_bw and [(tup, sizes)] == exp_tup
        assert sizes == (256, 2048)  # openvm-bus-interaction-handler/src/lib.rs:44
        apc.close()


def test_cbor_artifacts_give_the_same_apc(fixture_doc):
    """serde_cbor artifacts: a bare Apc, the `select` stage's Vec<ApcWithStats>, and a `setup`-like document that
    nests the APCs deep inside the VM configuration — every route yields the APC the JSON export yields."""
    ref = host.Apc(fixture_doc)
    want = fingerprint(ref)
    # 1. the same document as CBOR
    a = host.Apc(cbor(fixture_doc), fmt="cbor")
    assert fingerprint(a) == want and a.bus_map() == ref.bus_map()
    # 2. Vec<ApcWithStats{apc, stats, evaluation_result}> (adapter.rs:22-27); instructions serialised as structs
    apc_only = {k: v for k, v in fixture_doc.items() if k != "bus_map"}
    as_struct = json.loads(json.dumps(apc_only))
    for blk in as_struct["block"]["blocks"]:
        blk["instructions"] = [dict(zip(["opcode", "a", "b", "c", "d", "e", "f", "g"], ins)) for ins in blk["instructions"]]
    stats = {"widths": {"before": {"preprocessed": 0, "main": 59}, "after": {"preprocessed": 0, "main": 30}}}
    ev = {"before": {"main_columns": 59, "constraints": 40, "bus_interactions": 25}, "after": {"main_columns": 30, "constraints": 14, "bus_interactions": 16}}
    select = [{"apc": apc_only, "stats": stats, "evaluation_result": ev}, {"apc": as_struct, "stats": stats, "evaluation_result": ev}]
    blob = cbor(select)
    assert host.count_apcs(blob, "cbor") == 2
    for i in range(2):
        b = host.Apc(blob, fmt="cbor", index=i)
        assert fingerprint(b) == want and b.bus_map() == []
    with pytest.raises(ValueError, match="out of range"):
        host.Apc(blob, fmt="cbor", index=2)
    # 3. nested (setup artifact shape: CompiledProgram{exe, vm_config{original, powdr{precompiles[..{apc}..]}}})
    setup = {"exe": {"program": [1, 2, 3], "pc_start": 0}, "vm_config": {"original": {"x": None, "y": 1.5, "z": -7},
             "powdr": {"precompiles": [{"name": "apc0", "opcode": 8192, "apc": apc_only, "apc_stats": stats}]}}}
    assert host.count_apcs(cbor(setup), "cbor") == 1
    assert fingerprint(host.Apc(cbor(setup), fmt="cbor")) == want
    # the JSON reader walks nested documents too
    assert host.count_apcs(json.dumps(setup).encode()) == 1
    assert fingerprint(host.Apc(json.dumps(setup).encode())) == want


def test_cbor_encodings_and_malformed_input(fixture_doc):
    doc = cbor(fixture_doc)
    # indefinite-length containers and tags decode to the same document
    indef = b"\xd9\xd9\xf7" + b"\xbf" + b"".join(cbor(k) + (b"\x9f" + b"".join(cbor(x) for x in v) + b"\xff" if isinstance(v, list) else cbor(v))
                                                 for k, v in fixture_doc.items()) + b"\xff"
    assert fingerprint(host.Apc(indef, fmt="cbor")) == fingerprint(host.Apc(doc, fmt="cbor"))
    for bad in (doc[: len(doc) // 2], b"\xa1", b"\x9b\xff\xff\xff\xff\xff\xff\xff\xff", b""):
        with pytest.raises(ValueError):
            host.Apc(bad, fmt="cbor")
        assert host.count_apcs(bad, "cbor") == 0
    with pytest.raises(ValueError, match="no Apc"):
        host.Apc(cbor({"a": [1, 2, {"b": "c"}]}), fmt="cbor")


def test_keccak_fixture_through_cbor(reference_dir):
    """The large fixture (27 521 columns, 13 262 interactions, 28 627 constraints; deep left-leaning sums) through the
    CBOR route: same counts as the reference pins (autoprecompiles/tests/optimizer.rs:66-84)."""
    doc = json.load(gzip.open(reference_dir / "autoprecompiles" / "tests" / "keccak_apc_pre_opt.json.gz"))
    a = host.Apc(cbor(doc), fmt="cbor")
    assert (a.width, a.n_bus, a.n_constraints) == (27521, 13262, 28627)
    b = host.Apc(doc)
    assert (a.poly_ids() == b.poly_ids()).all()
    ca, cb = a.compile_constraints(), b.compile_constraints()
    assert (ca[0] == cb[0]).all() and (ca[1] == cb[1]).all()


def test_apc_candidates_json_v4():
    """apc_candidates.json as cell PGO writes it (JsonExport{version: 4, apcs, labels}, pgo/cell/mod.rs:83-97;
    ApcCandidateJsonExport :34-52; EvaluationResult / AirStats evaluation.rs:12-21,50-59)."""
    doc = {"version": 4, "labels": {"2099200": ["main"], "2099300": ["memcpy", "loop0"]}, "apcs": [
        {"execution_frequency": 1200, "original_blocks": [{"start_pc": 2099200, "instructions": ["ADD rd_ptr = 8 ...", "XOR ...", "LOADW ..."]},
                                                           {"start_pc": 2099300, "instructions": ["STOREW ..."]}],
         "stats": {"before": {"main_columns": 166, "constraints": 120, "bus_interactions": 77},
                   "after": {"main_columns": 38, "constraints": 11, "bus_interactions": 20}},
         "width_before": 166, "value": 153600, "cost_before": 166.0, "cost_after": 38.5},
        {"execution_frequency": 7, "original_blocks": [{"start_pc": 2100000, "instructions": ["BEQ ..."]}],
         "stats": {"before": {"main_columns": 26, "constraints": 11, "bus_interactions": 8},
                   "after": {"main_columns": 9, "constraints": 3, "bus_interactions": 5}},
         "width_before": 26, "value": 119, "cost_before": 2.6e1, "cost_after": 9}]}
    version, apcs, n_labels = host.read_apc_candidates(json.dumps(doc).encode())
    assert version == 4 and n_labels == 2 and len(apcs) == 2
    a, b = apcs
    assert a["execution_frequency"] == 1200 and a["start_pc"] == 2099200 and a["n_blocks"] == 2 and a["n_instructions"] == 4
    assert a["before"] == doc["apcs"][0]["stats"]["before"] and a["after"] == doc["apcs"][0]["stats"]["after"]
    assert a["width_before"] == 166 and a["value"] == 153600 and a["cost_before"] == 166.0 and a["cost_after"] == 38.5
    assert b["n_blocks"] == 1 and b["cost_before"] == 26.0 and b["cost_after"] == 9.0
    # version 3 layout: a single original_block
    v3 = {"version": 3, "labels": {}, "apcs": [{**doc["apcs"][1], "original_block": doc["apcs"][1]["original_blocks"][0]}]}
    del v3["apcs"][0]["original_blocks"]
    version, apcs, _ = host.read_apc_candidates(json.dumps(v3).encode())
    assert version == 3 and apcs[0]["n_blocks"] == 1 and apcs[0]["n_instructions"] == 1
    with pytest.raises(ValueError):
        host.read_apc_candidates(b'{"version": 4}')


def test_mutated_apc_documents_are_rejected_or_read_never_crash():
    """Structural fuzz of the host library's document reader (JSON -> DOM -> PowdrApc): keys dropped, values replaced by the wrong
    type, expression nodes malformed, substitutions pointing nowhere. Every mutant is either rejected with an error message or read
    into a handle whose table compilers run — the C++ side never lets an exception or a bad index through the C ABI."""
    import copy
    import random

    from powdr_amd import synth

    doc = synth.generate("C1", seed=3).doc
    rng = random.Random(7)
    junk = [None, 1, "x", [], {}, -5, 2 ** 40, "a@b", "noat", [1, "^", 2], ["-"], [1, 2, 3, 4], "is_valid@999999"]

    def mutate(x):
        if isinstance(x, dict) and x:
            k = rng.choice(list(x.keys()))
            r = rng.random()
            if r < 0.15:
                x.pop(k)
            elif r < 0.3:
                x[k] = rng.choice(junk)
            else:
                mutate(x[k])
        elif isinstance(x, list) and x:
            i = rng.randrange(len(x))
            r = rng.random()
            if r < 0.15:
                x.pop(i)
            elif r < 0.3:
                x[i] = rng.choice(junk)
            else:
                mutate(x[i])

    read = rejected = 0
    for _ in range(250):
        d = copy.deepcopy(doc)
        for _ in range(rng.randrange(1, 4)):
            mutate(d)
        try:
            h = host.Apc(d)
        except (ValueError, RuntimeError, TypeError, KeyError, OverflowError):
            rejected += 1
            continue
        for call in (lambda: h.compile_bus(1), h.compile_constraints, lambda: h.compile_derived(1), h.instruction_table):
            try:
                call()
            except (ValueError, RuntimeError):
                pass
        h.close()
        read += 1
    assert read + rejected == 250 and rejected > 60 and read > 5


def test_byte_level_fuzz_and_adversarial_documents():
    """Round 6: the readers face files from another machine. Byte-level mutants of a JSON and a CBOR artifact (flips, deletions,
    insertions, truncations) are rejected with a message or read — never a crash; and the classic attacks on a recursive-descent
    reader are error messages: 200 000 opening brackets (the JSON reader had no depth bound: a stack overflow until this round; the
    CBOR reader had one), length prefixes far beyond the input, an integer that does not fit 64 bits (it used to wrap silently)."""
    import json
    import random

    from powdr_amd import synth

    doc = synth.generate("T1", seed=3).doc
    rng = random.Random(11)

    def mutants(b, n):
        for _ in range(n):
            x = bytearray(b)
            for _ in range(rng.randrange(1, 6)):
                r, i = rng.random(), rng.randrange(len(x))
                if r < 0.4:
                    x[i] = rng.randrange(256)
                elif r < 0.6:
                    del x[i:i + rng.randrange(1, 20)]
                elif r < 0.8:
                    x[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 8)))
                else:
                    x = x[:i]
                if not x:
                    x = bytearray(b"\x00")
            yield bytes(x)

    for fmt, base in (("json", json.dumps(doc).encode()), ("cbor", cbor(doc))):
        read = rejected = 0
        for m in mutants(base, 300):
            try:
                h = host.Apc(m, fmt=fmt)
            except (ValueError, RuntimeError, TypeError, KeyError, OverflowError, UnicodeDecodeError):
                rejected += 1
                host.count_apcs(m, fmt)
                continue
            for call in (lambda: h.compile_bus(1), h.compile_constraints, lambda: h.compile_derived(1)):
                try:
                    call()
                except (ValueError, RuntimeError):
                    pass
            h.close()
            read += 1
        assert read + rejected == 300 and rejected > 200
    for payload, fmt, what in ((b"[" * 200000, "json", "nesting too deep"), (b'{"a":' * 100000, "json", "nesting too deep"),
                               (b"\x81" * 200000, "cbor", "nesting too deep"), (b"\xc1" * 200000, "cbor", "unexpected end"),
                               (b"\x9b\x7f\xff\xff\xff\xff\xff\xff\xff", "cbor", "longer than the input"),
                               (b"\x5b\x7f\xff\xff\xff\xff\xff\xff\xff\x00", "cbor", "past the end"),
                               (b"\xbb\x00\x00\x00\x10\x00\x00\x00\x00", "cbor", "longer than the input"),
                               (b"[" + b"9" * 100000 + b"]", "json", "beyond 64 bits"), (b'{"block": 18446744073709551617}', "json", "beyond 64 bits")):
        with pytest.raises(ValueError, match=what):
            host.Apc(payload, fmt=fmt)
        assert host.count_apcs(payload, fmt) == 0
    # 2^64 - 1 still reads as a number (no Apc in the document: a different error), a float statistic of any size is fine
    with pytest.raises(ValueError, match="no Apc"):
        host.Apc(b'{"x": 18446744073709551615, "y": 123456789012345678901234567890.5e3}', fmt="json")
