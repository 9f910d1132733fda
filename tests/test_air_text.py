"""powdr_amd/air_text.py reads the reference's textual machines (openvm-riscv/tests/openvm_constraints.txt, apc_snapshots/**): every
result pin of trace generation goes through it, so the parser itself is pinned — random expression trees rendered the way
`SymbolicMachine`'s Display writes them (binary + - *, unary minus, parentheses only where precedence needs them or around every
operand) evaluate to the same field element through the parsed post-fix code as directly."""
import random

import numpy as np
import pytest

from oracle import original_chips as oc
from powdr_amd import air_text

P = air_text.P
COLS = ["a__0_0", "b__1_12", "is_valid", "from_state__timestamp_0", "reads_aux__0__base__timestamp_lt_aux__lower_decomp__0_3", "x"]


def tree(rng, depth):
    if depth == 0 or rng.random() < 0.25:
        return ("col", rng.randrange(len(COLS))) if rng.random() < 0.6 else ("num", rng.choice([0, 1, 2, 255, 256, 65536, 2013265920, 1006632961, 7864320]))
    k = rng.choice(["+", "-", "*", "*", "neg"])
    if k == "neg":
        return ("neg", tree(rng, depth - 1))
    return (k, tree(rng, depth - 1), tree(rng, depth - 1))


def value(t, vals):
    if t[0] == "col":
        return vals[t[1]]
    if t[0] == "num":
        return t[1] % P
    if t[0] == "neg":
        return -value(t[1], vals) % P
    x, y = value(t[1], vals), value(t[2], vals)
    return (x + y) % P if t[0] == "+" else (x - y) % P if t[0] == "-" else x * y % P


PREC = {"+": 1, "-": 1, "*": 2, "neg": 3, "col": 4, "num": 4}


def render(t, rng, minimal):
    """minimal: parentheses only where left-associative precedence needs them; else around every compound operand (both styles occur)."""
    if t[0] == "col":
        return COLS[t[1]]
    if t[0] == "num":
        return str(t[1])
    if t[0] == "neg":
        inner = render(t[1], rng, minimal)
        return "-" + (inner if PREC[t[1][0]] >= 3 and minimal else f"({inner})")
    a, b = render(t[1], rng, minimal), render(t[2], rng, minimal)
    if not minimal or PREC[t[1][0]] < PREC[t[0]]:
        a = f"({a})" if t[1][0] not in ("col", "num") else a
    if not minimal or PREC[t[2][0]] <= PREC[t[0]]:  # right operand of a left-associative operator
        b = f"({b})" if t[2][0] not in ("col", "num") else b
    sp = rng.choice([" ", " ", ""])
    return f"{a}{sp}{t[0]}{sp}{b}"


@pytest.mark.parametrize("seed", range(6))
def test_parsed_expressions_evaluate_like_their_trees(seed):
    rng = random.Random(seed)
    col_index = {c: i for i, c in enumerate(COLS)}
    vals = [rng.randrange(P) for _ in COLS]
    cols = [np.array([v], np.int64) for v in vals]
    for _ in range(300):
        t = tree(rng, rng.randrange(1, 6))
        text = render(t, rng, minimal=rng.random() < 0.5)
        code = air_text.compile_expr(text, col_index)
        got = int(np.broadcast_to(oc.eval_postfix(np.array(code, np.uint32), cols), (1,))[0])
        assert got == value(t, vals), text


def test_malformed_text_is_rejected():
    col_index = {c: i for i, c in enumerate(COLS)}
    for bad in ["a__0_0 +", "(x", "x)", "x y", "unknown_column + 1", "", "* x", "x ** 2"]:
        with pytest.raises(ValueError):
            air_text.compile_expr(bad, col_index)
    with pytest.raises(ValueError):
        air_text.parse_airs("# A\nSymbolic machine using 1 unique main columns:\n  x\n\n// Algebraic constraints:\nx * x\n")  # no "= 0"
    with pytest.raises(KeyError):
        air_text.parse_instruction("FROB 1 2 3 1 0")
    assert air_text.parse_instruction("BLTU 44 48 -44 1 1") == [550, 44, 48, P - 44, 1, 1, 0, 0]
    assert air_text.parse_instruction("LOADW rd_rs2_ptr = 60, rs1_ptr = 56, imm = 0, mem_as = 2, needs_write = 1, imm_sign = 0") == [528, 60, 56, 0, 1, 2, 1, 0]


def test_the_air_fixture_is_the_reference_snapshot(reference_dir):
    """tests/golden/openvm_airs.npz == a fresh parse of the reference's openvm-riscv/tests/openvm_constraints.txt (the file its own
    `machine_extraction` test compares against): 13 AIRs, and per AIR the header's own column count."""
    import re
    from pathlib import Path

    text = (reference_dir / "openvm-riscv" / "tests" / "openvm_constraints.txt").read_text()
    airs = air_text.parse_airs(text)
    z = np.load(Path(__file__).parent / "golden" / "openvm_airs.npz")
    assert len(airs) == 13 == len(z["names"])
    declared = [int(x) for x in re.findall(r"Symbolic machine using (\d+) unique main columns", text)]
    assert [a.width for a in airs] == declared == z["widths"].tolist()
    assert sum(len(a.constraints) for a in airs) == 307 and sum(len(a.interactions) for a in airs) == 227
    for k, a in enumerate(airs):
        bc, spans, (inter, ispans, ibc) = a.tables()
        assert a.columns == [str(c) for c in z[f"a{k}_columns"]] and str(z["full_names"][k]) == a.name
        assert (bc == z[f"a{k}_bc"]).all() and (spans == z[f"a{k}_spans"]).all() and (inter == z[f"a{k}_inter"]).all() and (ibc == z[f"a{k}_ibc"]).all()
