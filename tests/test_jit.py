"""Run-time specialised expression kernels (csrc/jit.hpp, jit_codegen.hpp, prover_jit.hip): the AIR's
constraint / interaction programs emitted as straight-line HIP, compiled with hiprtc and run instead of the interpreter
kernels. Exact field arithmetic => the proof words must not depend on the path: every test proves with POWDR_JIT=1 and
POWDR_JIT=0 and compares both with the CPU oracle. CPU part: code generation + hiprtc cross-compilation need no GPU."""
from pathlib import Path

import numpy as np
import pytest

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

P = om.P
PA, PC, ADD, SUB, MUL, NEG = om.OP_PUSH_APC, om.OP_PUSH_CONST, om.OP_ADD, om.OP_SUB, om.OP_MUL, om.OP_NEG


def _tables(shape, seed=0):
    s = synth.generate(shape, seed=seed)
    apc = om.load_apc(s.doc)
    idx = apc.poly_id_to_index()
    return len(idx), sm.compile_constraints(apc, idx), sm.compile_interactions(apc, idx)


def hand_made_air(W=9):
    """Interaction shapes the synthetic APCs do not have: no arguments at all, constant-only arguments (degree-0 denominators:
    more than two members per LogUp group), a multiplicity of degree 2, a group of one, nested expressions that need the stack."""
    bc, spans = [], []

    def span(words, sink, spans_):
        spans_.append((len(sink), len(words)))
        sink.extend(words)

    span([PA, 0, PA, 1, MUL, PA, 2, SUB], bc, spans)                       # a*b - c
    span([PA, 3, PA, 3, PC, 1, SUB, MUL], bc, spans)                       # d*(d-1)
    span([PA, 0, PA, 1, ADD, PA, 2, PA, 3, ADD, MUL, PA, 4, NEG, ADD], bc, spans)  # (a+b)*(c+d) - e: both operands compound
    ibc, ispans, inter = [], [], []

    def interaction(bus, mult, args):
        inter.append((bus, len(args), len(ispans)))
        span(mult, ibc, ispans)
        for a in args:
            span(a, ibc, ispans)

    interaction(5, [PA, 3], [])                                             # no arguments
    interaction(5, [PC, 1], [[PC, 7], [PC, 9]])                             # constants only: joins the group of the previous ones
    interaction(5, [PA, 4], [[PC, 3]])
    interaction(5, [PA, 5], [[PC, 11], [PC, 12], [PC, 13]])
    interaction(3, [PA, 3, PA, 4, MUL], [[PA, 5], [PC, 12]])                # degree-2 multiplicity
    interaction(6, [PA, 3], [[PA, 0, PA, 1, ADD, PA, 2, PA, 6, SUB, MUL], [PA, 7], [PA, 8], [PC, 1]])  # degree-2 argument: a group of its own
    interaction(1, [PC, 2, NEG], [[PA, 1], [PA, 2], [PA, 3], [PA, 4], [PA, 5], [PA, 6], [PA, 7]])      # seven arguments
    return (W, (np.array(bc, np.uint32), np.array(spans, np.uint32).reshape(-1, 2)),
            (np.array(inter, np.uint32).reshape(-1, 3), np.array(ispans, np.uint32).reshape(-1, 2), np.array(ibc, np.uint32)))


@pytest.mark.parametrize("shape", ["T0", "T1", "hand"])
@pytest.mark.parametrize("logup", [False, True])
def test_specialised_kernels_compile_without_a_gpu(shape, logup):
    from powdr_amd import prover

    W, (bc, spans), it = hand_made_air() if shape == "hand" else _tables(shape)
    r = prover.jit_compile_check(W, bc, spans, it if logup else None)
    assert r["rc"] == 0, r["error"]
    assert r["kernels"] >= 1 and r["chunks"] >= r["kernels"] and r["code_bytes"] > 4000
    if logup and shape == "hand":
        assert len(prover.logup_group_starts(it)) - 1 < len(it[0]) - 2  # the constant-only interactions share a group


def test_code_objects_are_kept_on_disk_across_provers(tmp_path, monkeypatch):
    """The on-disk cache of compiled units ($POWDR_JIT_CACHE_DIR): the first compilation of an AIR's kernels writes one entry per
    translation unit, the same AIR again (its programs released in between: the in-process cache only holds live programs) loads
    them instead of compiling; a truncated or foreign entry is ignored and replaced; POWDR_JIT_CACHE=0 switches the cache off."""
    from powdr_amd import prover

    monkeypatch.setenv("POWDR_JIT_CACHE_DIR", str(tmp_path / "cache" / "nested"))
    W, (bc, spans), it = hand_made_air()
    s0 = prover.jit_cache_stats()
    r1 = prover.jit_compile_check(W, bc, spans, it)
    s1 = prover.jit_cache_stats()
    assert r1["rc"] == 0 and s1["compiled"] - s0["compiled"] == r1["kernels"] and s1["from_disk"] == s0["from_disk"]
    entries = sorted((tmp_path / "cache" / "nested").glob("*.pwjc"))
    assert len(entries) == r1["kernels"] and not list((tmp_path / "cache" / "nested").glob("*.tmp*"))
    r2 = prover.jit_compile_check(W, bc, spans, it)
    s2 = prover.jit_cache_stats()
    assert r2 == r1 and s2["compiled"] == s1["compiled"] and s2["from_disk"] - s1["from_disk"] == r1["kernels"]
    # a damaged entry: ignored, compiled again, rewritten whole
    whole = entries[0].read_bytes()
    entries[0].write_bytes(whole[: len(whole) // 2])
    r3 = prover.jit_compile_check(W, bc, spans, it)
    s3 = prover.jit_cache_stats()
    assert r3 == r1 and s3["compiled"] - s2["compiled"] == 1 and s3["from_disk"] - s2["from_disk"] == r1["kernels"] - 1
    assert entries[0].read_bytes()[:8] == b"PWJC0001" and len(entries[0].read_bytes()) > len(whole) // 2  # (code objects are not byte-reproducible)
    # an entry whose stored source is another unit's (a file-name collision): not used
    entries[0].write_bytes(entries[-1].read_bytes() if len(entries) > 1 else whole[:24] + b"x" * (len(whole) - 24))
    r4 = prover.jit_compile_check(W, bc, spans, it)
    assert r4 == r1 and prover.jit_cache_stats()["compiled"] - s3["compiled"] == 1
    monkeypatch.setenv("POWDR_JIT_CACHE", "0")
    s4 = prover.jit_cache_stats()
    r5 = prover.jit_compile_check(W, bc, spans, it)
    s5 = prover.jit_cache_stats()
    assert r5 == r1 and s5["compiled"] - s4["compiled"] == r1["kernels"] and s5["from_disk"] == s4["from_disk"]


def test_ranks_that_share_a_cold_cache_compile_every_unit_once(tmp_path):
    """N ranks of one node specialise the same AIRs against ONE on-disk cache. The compile phase runs
    under the cache directory's lock (flock), so the first process compiles and the others load: over 4 concurrent processes with a
    cold cache every translation unit is compiled exactly once, and a symlinked cache directory is refused with a diagnostic."""
    import json
    import os
    import subprocess
    import sys

    root = Path(__file__).resolve().parents[1]
    code = ("import json, sys; sys.path.insert(0, %r)\n"
            "from tests.test_jit import hand_made_air, _tables\n"
            "from powdr_amd import prover\n"
            "out = []\n"
            "for shape in ('hand', 'T1'):\n"
            "    W, (bc, spans), it = hand_made_air() if shape == 'hand' else _tables(shape)\n"
            "    r = prover.jit_compile_check(W, bc, spans, it)\n"
            "    out.append(r)\n"
            "print(json.dumps(dict(results=out, stats=prover.jit_cache_stats())))\n") % str(root)
    env = dict(os.environ, POWDR_JIT_CACHE_DIR=str(tmp_path / "shared"), POWDR_JIT_THREADS="2")
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=root) for _ in range(4)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [e[-800:] for _, e in outs]
    recs = [json.loads(o.strip().splitlines()[-1]) for o, _ in outs]
    units = sum(r["kernels"] for r in recs[0]["results"])
    assert all(r["rc"] == 0 for rec in recs for r in rec["results"]) and all(rec["results"] == recs[0]["results"] for rec in recs)
    assert sum(rec["stats"]["compiled"] for rec in recs) == units, [rec["stats"] for rec in recs]
    assert sum(rec["stats"]["from_disk"] for rec in recs) == 3 * units
    assert len(list((tmp_path / "shared").glob("*.pwjc"))) == units and not list((tmp_path / "shared").glob("*.tmp*"))
    # a cache directory reached through a symbolic link is not trusted — and says so
    (tmp_path / "link").symlink_to(tmp_path / "shared")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, POWDR_JIT_CACHE_DIR=str(tmp_path / "link")), cwd=root,
                         timeout=600)
    assert out.returncode == 0 and "is a symbolic link" in out.stderr and out.stderr.count("powdr jit:") == 1
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["stats"]["compiled"] == units and rec["stats"]["from_disk"] == 0


def test_jit_is_off_with_POWDR_JIT_0(monkeypatch):
    from powdr_amd import prover

    monkeypatch.setenv("POWDR_JIT", "0")
    W, (bc, spans), it = _tables("T0")
    r = prover.jit_compile_check(W, bc, spans, it)
    assert r["rc"] == 1 and r["kernels"] == 0


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def gpu():
    import torch
    from powdr_amd import abi, prover

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (run with -m gpu on the GPU box)")
    return torch, abi, prover


def to_dev(torch, a):
    return torch.from_numpy(om.to_monty(np.ascontiguousarray(a, dtype=np.uint32)).view(np.int32)).cuda()


def _prove_both_paths(gpu, monkeypatch, flat, W, log_h, bc, spans, it, nq=5, pow_bits=3):
    torch, abi, prover = gpu
    d_t = to_dev(torch, flat)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("POWDR_JIT", mode)
        pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pow_bits, interactions=it)
        abi.call_stats(reset=True)
        out[mode] = pr.prove(d_t.data_ptr(), log_h)
        st = abi.call_stats()
        info = pr.specialised()
        if mode == "1":
            assert info["state"] == 1 and info["kernels"] >= 1, info
            assert st["jit_launches"] >= 1 and st["interpreter_launches"] == 0, st
            assert (pr.prove(d_t.data_ptr(), log_h) == out[mode]).all()  # second proof: the same kernels again
        else:
            assert info["state"] <= 0 and st["jit_launches"] == 0 and st["interpreter_launches"] >= 1, (info, st)
        pr.close()
    assert len(out["1"]) == len(out["0"]) and (out["1"] == out["0"]).all(), f"first differing word {int(np.argmax(out['1'] != out['0']))}"
    return out["1"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,calls", [("T0", 7), ("T0", 64), ("T1", 1000), ("T1", 4000), ("C1", 700)])
@pytest.mark.parametrize("logup", [False, True])
def test_proof_bytes_do_not_depend_on_the_path(gpu, monkeypatch, shape, calls, logup):
    """Specialised kernels == interpreter == oracle, constraints-only and with the LogUp phase, with zero-padding rows."""
    from tests.test_oracle_apc import run_oracle_gpu_convention

    s = synth.generate(shape, seed=5)
    apc, idx, trace, _, _ = run_oracle_gpu_convention(s, calls, seed=5)
    W, H = trace.shape
    log_h = H.bit_length() - 1
    flat = np.ascontiguousarray(trace).reshape(-1)
    bc, spans = sm.compile_constraints(apc, idx)
    it = sm.compile_interactions(apc, idx) if logup else None
    got = _prove_both_paths(gpu, monkeypatch, flat, W, log_h, bc, spans, it)
    want = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=5, pow_bits=3) if logup else sm.prove(flat, W, log_h, bc, spans, num_queries=5, pow_bits=3)
    assert len(got) == len(want) and (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("log_h", [1, 5, 11])
def test_hand_made_interaction_shapes(gpu, monkeypatch, log_h):
    """Groups of one, of four (constant-only denominators), interactions without arguments, degree-2 multiplicities and
    arguments, stack-using expressions — on random traces (byte parity does not need satisfied constraints)."""
    W, (bc, spans), it = hand_made_air()
    rng = np.random.default_rng(log_h)
    flat = rng.integers(0, P, W << log_h, dtype=np.uint32)
    flat[3 << log_h: (3 << log_h) + max(1, (1 << log_h) // 3)] = 0  # rows with zero multiplicities (no inversion there)
    got = _prove_both_paths(gpu, monkeypatch, flat, W, log_h, bc, spans, it, nq=4, pow_bits=0)
    assert (got == sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=4, pow_bits=0)).all()
    got = _prove_both_paths(gpu, monkeypatch, flat, W, log_h, bc, spans, None, nq=4, pow_bits=0)
    assert (got == sm.prove(flat, W, log_h, bc, spans, num_queries=4, pow_bits=0)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("logup", [False, True])
def test_segment_proof_with_specialised_kernels(gpu, monkeypatch, logup):
    """pw_prove_segment: all AIRs' kernels compiled in one batch; mixed heights; an AIR without interactions inside a LogUp
    segment; the same words as the interpreter path and the oracle."""
    torch, abi, prover = gpu
    from tests.test_segment_proof import descs_of, synthetic_airs

    airs = synthetic_airs([("T0", 30), ("T1", 900), ("T0", 5), ("T1", 64)], seed0=21)
    W, (bc, spans), it = hand_made_air()
    rng = np.random.default_rng(3)
    airs.append((rng.integers(0, P, W << 6, dtype=np.uint32), W, 6, bc, spans, it))
    no_inter = (np.zeros((0, 3), np.uint32), np.zeros((0, 2), np.uint32), np.zeros(0, np.uint32))
    airs.append((rng.integers(0, P, 5 << 4, dtype=np.uint32), 5, 4, np.array([PA, 0, PA, 1, MUL], np.uint32), np.array([[0, 5]], np.uint32), no_inter))
    want = sm.prove_segment(airs, num_queries=4, pow_bits=2, logup=logup)
    traces = [to_dev(torch, a[0]) for a in airs]
    for mode in ("1", "0"):
        monkeypatch.setenv("POWDR_JIT", mode)
        provers = [prover.Prover(a[1], a[3], a[4], num_queries=4, pow_bits=2, interactions=a[5] if logup else None) for a in airs]
        abi.call_stats(reset=True)
        got = prover.prove_segment([(pr, t.data_ptr(), a[2]) for pr, t, a in zip(provers, traces, airs)], logup=logup)
        st = abi.call_stats()
        assert (st["jit_launches"] > 0) == (mode == "1") and (st["interpreter_launches"] > 0) == (mode == "0"), st
        assert len(got) == len(want) and (got == want).all(), (mode, int(np.argmax(got != want)))
        for pr in provers:
            pr.close()


@pytest.mark.gpu
def test_default_policy_specialises_tall_traces_only(gpu, monkeypatch):
    """Unset POWDR_JIT: a 2^10-row proof keeps the interpreter (compiling costs seconds), the same prover is specialised once
    it meets a trace of 2^POWDR_JIT_MIN_LOG_HEIGHT rows; pw_prover_specialise does it at set-up time."""
    torch, abi, prover = gpu
    monkeypatch.delenv("POWDR_JIT", raising=False)
    monkeypatch.setenv("POWDR_JIT_MIN_LOG_HEIGHT", "12")
    W, (bc, spans), it = _tables("T0")
    rng = np.random.default_rng(0)
    pr = prover.Prover(W, bc, spans, num_queries=3, interactions=it)
    small = to_dev(torch, rng.integers(0, P, W << 10, dtype=np.uint32))
    tall = to_dev(torch, rng.integers(0, P, W << 12, dtype=np.uint32))
    pr.prove(small.data_ptr(), 10)
    assert pr.specialised()["state"] == 0
    p_tall = pr.prove(tall.data_ptr(), 12)
    assert pr.specialised()["state"] == 1
    flat = om.from_monty(tall.cpu().numpy().view(np.uint32))
    assert (p_tall == sm.prove_logup(flat, W, 12, bc, spans, *it, num_queries=3, pow_bits=0)).all()
    pr.close()
    pr2 = prover.Prover(W, bc, spans, num_queries=3)
    assert pr2.specialise() and pr2.specialised()["state"] == 1
    pr2.close()
    # pw_provers_specialise: a set of AIRs compiled in one batch at set-up, whatever their heights (round 6: the segment legs do this; a
    # short trace under the interpreter pays the AIR's whole program per lane) — the short proofs then run specialised kernels, same words
    W1, (bc1, sp1), it1 = _tables("T1")
    ps = [prover.Prover(W, bc, spans, num_queries=3, interactions=it), prover.Prover(W1, bc1, sp1, num_queries=3, interactions=it1),
          prover.Prover(W, bc, spans, num_queries=3)]
    assert all(p_.specialised()["state"] == 0 for p_ in ps)
    assert prover.specialise_all(ps) == 3 and all(p_.specialised()["state"] == 1 for p_ in ps)
    abi.call_stats(reset=True)
    got = ps[0].prove(small.data_ptr(), 10)
    assert abi.call_stats()["jit_launches"] > 0 and abi.call_stats()["interpreter_launches"] == 0
    flat = om.from_monty(small.cpu().numpy().view(np.uint32))
    assert (got == sm.prove_logup(flat, W, 10, bc, spans, *it, num_queries=3, pow_bits=0)).all()
    assert prover.specialise_all(ps) == 3  # (nothing left to do)
    for p_ in ps:
        p_.close()


# ---- the generated code executed on the HOST ----------------------------------------------------------------------------------
_SHIM = """
#include <cstddef>
#include <cstdint>
#include <cstring>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(x)
static struct { unsigned x, y, z; } blockIdx, threadIdx;
"""
_W4 = 11  # F_p[X] / (X^4 - 11)


def _ext_mul(a, b):
    c = [0] * 7
    for i in range(4):
        for j in range(4):
            c[i + j] += a[i] * b[j]
    return [(c[k] + _W4 * (c[k + 4] if k < 3 else 0)) % P for k in range(4)]


def _ext_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def _ext_scale(a, s):
    return [x * s % P for x in a]


def _ext_inv(a):
    r, base, e = [1, 0, 0, 0], list(a), P ** 4 - 2
    while e:
        if e & 1:
            r = _ext_mul(r, base)
        base = _ext_mul(base, base)
        e >>= 1
    return r


def _eval_postfix(code, row):
    st, i = [], 0
    while i < len(code):
        op = int(code[i])
        if op in (PA, PC):
            st.append(int(row[int(code[i + 1])]) if op == PA else int(code[i + 1]) % P)
            i += 2
        elif op == NEG:
            st.append(-st.pop() % P)
            i += 1
        else:
            y, x = st.pop(), st.pop()
            st.append((x + y) % P if op == ADD else (x - y) % P if op == SUB else x * y % P)
            i += 1
    return st[0]


def _run_units_on_the_host(tmp_path, units, which, total_chunks, arrays, n_rows):
    """Every unit: shim + generated source + a driver that walks (chunk, row) like the grid would; built with the host compiler of the
    ROCm LLVM (plain C++: the embedded headers have host paths for everything), loaded with ctypes."""
    import ctypes as C
    import shutil
    import subprocess

    cxx = "/opt/rocm/lib/llvm/bin/clang++" if Path("/opt/rocm/lib/llvm/bin/clang++").exists() else shutil.which("g++")
    csrc = Path(__file__).resolve().parents[1] / "powdr_amd" / "csrc"
    for k, u in enumerate(units):
        if which == 0:
            driver = f"""
extern "C" void run(const uint32_t* T, const uint32_t* Pm, size_t N, const bb::Ext* apow, const uint32_t* al4, const bb::Ext* blpow, uint32_t* part, uint32_t* unused) {{
    bb::Ext al; memcpy(&al, al4, 16);
    for (unsigned y = 0; y < {u['n_chunks']}u; ++y) for (size_t j = 0; j < N; ++j) {{
        blockIdx.x = (unsigned)(j / 256); blockIdx.y = y; threadIdx.x = (unsigned)(j % 256);
        {u['kernel']}(T, Pm, N, apow, al, blpow, part);
    }}
}}"""
        else:
            driver = f"""
extern "C" void run(const uint32_t* T, const uint32_t* Pm, size_t N, const bb::Ext* apow, const uint32_t* al4, const bb::Ext* blpow, uint32_t* perm, uint32_t* rowsum) {{
    bb::Ext al; memcpy(&al, al4, 16);
    for (unsigned y = 0; y < {u['n_chunks']}u; ++y) for (size_t j = 0; j < N; ++j) {{
        blockIdx.x = (unsigned)(j / 256); blockIdx.y = y; threadIdx.x = (unsigned)(j % 256);
        {u['kernel']}(T, N, al, blpow, perm, rowsum);
    }}
}}"""
        src = tmp_path / f"unit_{which}_{total_chunks}_{k}.cpp"
        src.write_text(_SHIM + u["source"] + driver)
        so = src.with_suffix(".so")
        subprocess.run([cxx, "-x", "c++", "-std=c++17", "-O1", "-shared", "-fPIC", f"-I{csrc}", str(src), "-o", str(so)], check=True, capture_output=True)
        fn = C.CDLL(str(so)).run
        fn.restype = None
        fn.argtypes = [C.c_void_p] * 2 + [C.c_size_t] + [C.c_void_p] * 5
        fn(*[a.ctypes.data if isinstance(a, np.ndarray) else a for a in (arrays["T"], arrays["Pm"], n_rows, arrays["apow"], arrays["al"], arrays["blpow"], arrays["out0"],
                                                                         arrays["out1"])])


def _golden_machine_air(name):
    """(W, (bc, spans), interactions) of one of the reference's golden APC machines (tests/golden/apc_snapshots.json.gz)"""
    import gzip
    import json

    from powdr_amd import air_text

    snap = json.loads(gzip.open(Path(__file__).parent / "golden" / "apc_snapshots.json.gz").read())[name]
    air = air_text.TextAir("apc", snap["columns"], snap["constraints"], [(b, m, a) for b, m, a in snap["interactions"]])
    bc, spans, it = air.tables()
    return len(snap["columns"]), (bc, spans), it


@pytest.mark.parametrize("air,chunk_cost", [("hand", 200), ("hand", 100000), ("complex/memcpy_block", 400), ("single_instructions/single_div", 100000)])
def test_generated_code_runs_on_the_host_and_means_what_it_should(tmp_path, air, chunk_cost):
    """The code generator's output EXECUTED, without a GPU: the generated translation units are plain C++ over the embedded headers,
    so a shim (`__global__` = nothing, blockIdx / threadIdx = globals) builds them for the host. On random data, for a chunking into
    several chunks and units and for one big chunk:
      * the LogUp permutation columns are q_g = sum_{i in g} m_i / (alpha + bus_i + sum_j beta^(j+1) a_ij), the per-chunk row sums add
        up to sum_g q_g,
      * the quotient numerator's chunk parts add up to sum_c alpha^c C_c + sum_g alpha^(nc+g) (q_g prod d_i - sum_i m_i prod_{l != i} d_l)
    in F_p^4, computed here with plain modular arithmetic. (The same kernels on the GPU: the tests below.)"""
    from powdr_amd import prover

    W, (bc, spans), it = hand_made_air() if air == "hand" else _golden_machine_air(air)
    inter, ispans, ibc = it
    starts = prover.logup_group_starts(it)
    n_groups, n_cons, N = len(starts) - 1, len(spans), 24 if air == "hand" else 6
    rng = np.random.default_rng(chunk_cost)
    canon = {"T": rng.integers(0, P, (W, N)), "Pm": rng.integers(0, P, (4 * n_groups + 4, N)), "apow": rng.integers(0, P, (n_cons + n_groups + 2, 4)),
             "al": rng.integers(0, P, 4), "blpow": rng.integers(0, P, (9, 4))}
    if air == "hand":
        canon["T"][3, :5] = 0  # rows whose multiplicities vanish: the zero-multiplicity guard
        canon["T"][4, :3] = 0
        canon["T"][5, :3] = 0
    else:
        canon["T"][:, 0] = 0   # an all-zero row (a padding row of a real trace): every multiplicity vanishes
    monty = {k: om.to_monty(np.ascontiguousarray(v, np.uint32)) for k, v in canon.items()}
    T = canon["T"]
    # ---- expected values, canonical -----------------------------------------------------------------------------------------
    al, bl = [int(x) for x in canon["al"]], [[int(x) for x in b] for b in canon["blpow"]]
    q_want = np.zeros((n_groups, N, 4), np.int64)
    quot_want = np.zeros((N, 4), np.int64)
    for r in range(N):
        row = T[:, r]
        acc = [0, 0, 0, 0]
        for c, (off, ln) in enumerate(spans.tolist()):
            acc = _ext_add(acc, _ext_scale([int(x) for x in canon["apow"][c]], _eval_postfix(bc[off:off + ln], row)))
        for g in range(n_groups):
            ms, ds = [], []
            for i in range(int(starts[g]), int(starts[g + 1])):
                bus, n_args, s0 = (int(x) for x in inter[i])
                ev = lambda s: _eval_postfix(ibc[int(ispans[s][0]):int(ispans[s][0]) + int(ispans[s][1])], row)
                ms.append(ev(s0))
                d = _ext_add(al, [bus % P, 0, 0, 0])
                for j in range(n_args):
                    d = _ext_add(d, _ext_scale(bl[j + 1], ev(s0 + 1 + j)))  # blpow[k] = beta^k: argument j meets beta^(j+1)
                ds.append(d)
            q = [0, 0, 0, 0]
            for m, d in zip(ms, ds):
                q = _ext_add(q, _ext_scale(_ext_inv(d), m))
            q_want[g, r] = q
            pq = [int(canon["Pm"][4 * g + k, r]) for k in range(4)]  # the committed (here: random) q_g the quotient reads
            prod_all = [1, 0, 0, 0]
            for d in ds:
                prod_all = _ext_mul(prod_all, d)
            term = _ext_mul(pq, prod_all)
            for i, m in enumerate(ms):
                rest = [1, 0, 0, 0]
                for l, d in enumerate(ds):
                    if l != i:
                        rest = _ext_mul(rest, d)
                term = _ext_add(term, _ext_scale(rest, -m % P))
            acc = _ext_add(acc, _ext_mul([int(x) for x in canon["apow"][n_cons + g]], term))
        quot_want[r] = acc
    # ---- the generated code ---------------------------------------------------------------------------------------------------
    for which in (1, 0):
        units, total = prover.jit_generated_sources(W, bc, spans, it, which, chunk_cost, 2)
        assert units and total >= (2 if chunk_cost <= 400 else 1) and sum(u["n_chunks"] for u in units) == total
        out0 = np.zeros((max(total, n_groups) * 4 + 4, N), np.uint32) if which == 0 else np.zeros((4 * n_groups + 4, N), np.uint32)
        out1 = np.zeros((total * 4, N), np.uint32)
        _run_units_on_the_host(tmp_path, units, which, total, dict(monty, out0=out0, out1=out1), N)
        if which == 1:
            got_q = om.from_monty(out0[:4 * n_groups]).astype(np.int64).reshape(n_groups, 4, N).transpose(0, 2, 1)
            assert (got_q == q_want).all()
            rowsum = om.from_monty(out1).astype(np.int64).reshape(total, 4, N).sum(axis=0) % P
            assert (rowsum.T == q_want.sum(axis=0) % P).all()
        else:
            parts = om.from_monty(out0[:total * 4]).astype(np.int64).reshape(total, 4, N).sum(axis=0) % P
            assert (parts.T == quot_want).all()


def test_malformed_programs_get_no_prover():
    """Round 6: a constraint or interaction program that names a column the trace does not have, runs past its bytecode, uses an unknown
    opcode or leaves the stack unbalanced never reaches the device (pw_prover_create* return NULL; here through the host-only creation
    path of pw_jit_generated_source: no units). Before, a malformed CONSTRAINT program fell back to the post-fix interpreter as it was."""
    from powdr_amd import prover as _prover

    prover_mod = lambda: _prover
    W, (bc, spans), it = _tables("T1")
    assert len(prover_mod().jit_generated_sources(W, bc, spans, it, which=0)[0]) >= 1

    def first_column_operand(code):
        ip = 0
        while code[ip] != 0:
            ip += 2 if code[ip] == 1 else 1
        return ip + 1

    gen = lambda b, s, i, which=0: len(prover_mod().jit_generated_sources(W, b, s, i, which=which)[0])
    b = np.array(bc, dtype=np.uint32, copy=True)
    b[first_column_operand(b)] = W
    assert gen(b, spans, it) == 0
    s2 = np.array(spans, dtype=np.uint32, copy=True).reshape(-1, 2)
    s2[0, 1] = 10 ** 6
    assert gen(bc, s2, it) == 0
    b = np.array(bc, dtype=np.uint32, copy=True)
    b[0] = 77
    assert gen(b, spans, it) == 0
    b = np.array(bc, dtype=np.uint32, copy=True)
    sp = np.asarray(spans).reshape(-1, 2)
    b[sp[0, 0] + sp[0, 1] - 1] = 1  # the last operator replaced by a PUSH_CONST without an operand inside the span
    assert gen(b, spans, it) == 0
    i2 = np.array(it[2], dtype=np.uint32, copy=True)
    i2[first_column_operand(i2)] = W + 5
    assert gen(bc, spans, (it[0], it[1], i2), which=1) == 0
    i1 = np.array(it[1], dtype=np.uint32, copy=True).reshape(-1, 2)
    i1[0, 0] = len(it[2]) + 3
    assert gen(bc, spans, (it[0], i1, it[2]), which=1) == 0
