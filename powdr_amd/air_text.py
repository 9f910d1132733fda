"""Parser for the reference's textual AIR snapshots (`# <AIR name>` / `Symbolic machine using N unique main columns:` /
`// Bus k (NAME):` lines `mult=<expr>, args=[<expr>, ...]` / `// Algebraic constraints:` lines `<expr> = 0`), the format of
/root/reference/openvm-riscv/tests/openvm_constraints.txt (written by `SymbolicMachine`'s Display, autoprecompiles/src/
symbolic_machine.rs) — the original RV32IM instruction AIRs an autoprecompile is built from.

Output: the tables pw_prover_create / pw_prover_create_logup take (post-fix bytecode with COLUMN-INDEX operands), so that the
real constraints of the original chips can be proven and mock-checked (pw_prover_check_constraints) by this library."""
from __future__ import annotations

import re
from dataclasses import dataclass, field

import numpy as np

P = 0x78000001
OP_PUSH_COL, OP_PUSH_CONST, OP_ADD, OP_SUB, OP_MUL, OP_NEG = 0, 1, 2, 3, 4, 5

_TOKEN = re.compile(r"\s*(?:(\d+)|([A-Za-z_][A-Za-z_0-9]*)|(.))")


def _tokens(text):
    out = []
    for num, ident, sym in _TOKEN.findall(text):
        if num:
            out.append(("num", int(num) % P))
        elif ident:
            out.append(("id", ident))
        elif sym.strip():
            out.append(("sym", sym))
    return out


class _Parser:
    """expr := term (('+' | '-') term)* ; term := unary ('*' unary)* ; unary := '-' unary | atom ; atom := num | id | '(' expr ')'
    emitting post-fix code as it goes (left-associative, like the Display it reads)."""

    def __init__(self, toks, col_index):
        self.t, self.i, self.col, self.code = toks, 0, col_index, []

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def eat(self, kind=None, val=None):
        k, v = self.peek()
        if k is None or (kind and k != kind) or (val is not None and v != val):
            raise ValueError(f"unexpected token {self.peek()} at {self.i}")
        self.i += 1
        return v

    def expr(self):
        self.term()
        while self.peek() in (("sym", "+"), ("sym", "-")):
            op = self.eat()
            self.term()
            self.code.append(OP_ADD if op == "+" else OP_SUB)

    def term(self):
        self.unary()
        while self.peek() == ("sym", "*"):
            self.eat()
            self.unary()
            self.code.append(OP_MUL)

    def unary(self):
        if self.peek() == ("sym", "-"):
            self.eat()
            self.unary()
            self.code.append(OP_NEG)
        else:
            self.atom()

    def atom(self):
        k, v = self.peek()
        if k == "num":
            self.eat()
            self.code += [OP_PUSH_CONST, v]
        elif k == "id":
            self.eat()
            if v not in self.col:
                raise ValueError(f"unknown column {v}")
            self.code += [OP_PUSH_COL, self.col[v]]
        elif (k, v) == ("sym", "("):
            self.eat()
            self.expr()
            self.eat("sym", ")")
        else:
            raise ValueError(f"unexpected token {self.peek()}")


def compile_expr(text: str, col_index: dict) -> list:
    p = _Parser(_tokens(text), col_index)
    p.expr()
    if p.i != len(p.t):
        raise ValueError(f"trailing tokens in {text!r}")
    return p.code


def _split_args(s: str):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
            continue
        depth += ch == "("
        depth -= ch == ")"
        cur += ch
    if cur.strip():
        out.append(cur)
    return out


@dataclass
class TextAir:
    name: str
    columns: list
    constraints: list = field(default_factory=list)   # expression texts
    interactions: list = field(default_factory=list)  # (bus id, mult text, [arg texts])

    @property
    def width(self):
        return len(self.columns)

    def tables(self):
        """(cons_bytecode, cons_spans[n, 2], (interactions[n, 3], inter_spans[m, 2], inter_bytecode)) — pw_prover_create_logup's arguments."""
        col = {n: i for i, n in enumerate(self.columns)}
        bc, spans = [], []
        for c in self.constraints:
            code = compile_expr(c, col)
            spans.append((len(bc), len(code)))
            bc += code
        ibc, ispans, inter = [], [], []
        for bus, mult, args in self.interactions:
            inter.append((bus, len(args), len(ispans)))
            for e in [mult] + args:
                code = compile_expr(e, col)
                ispans.append((len(ibc), len(code)))
                ibc += code
        return (np.array(bc, np.uint32), np.array(spans, np.uint32).reshape(-1, 2),
                (np.array(inter, np.uint32).reshape(-1, 3), np.array(ispans, np.uint32).reshape(-1, 2), np.array(ibc, np.uint32)))


def parse_airs(text: str) -> list:
    airs, cur, mode, bus = [], None, None, None
    for raw in text.splitlines():
        line = raw.strip()
        if raw.startswith("# "):
            cur = TextAir(raw[2:].strip(), [])
            airs.append(cur)
            mode = None
        elif line.startswith("Symbolic machine using"):
            mode = "cols"
        elif line.startswith("// Bus"):
            bus = int(re.match(r"// Bus (\d+)", line).group(1))
            mode = "bus"
        elif line.startswith("// Algebraic constraints"):
            mode = "cons"
        elif not line:
            if mode == "cols":
                mode = None
        elif mode == "cols":
            cur.columns.append(line)
        elif mode == "bus":
            m = re.match(r"mult=(.*), args=\[(.*)\]$", line)
            if not m:
                raise ValueError(f"cannot parse interaction {line!r}")
            cur.interactions.append((bus, m.group(1), _split_args(m.group(2))))
        elif mode == "cons":
            if not line.endswith("= 0"):
                raise ValueError(f"cannot parse constraint {line!r}")
            cur.constraints.append(line[: -len("= 0")].strip())
    return airs


# ---- APC snapshots: `Instructions:` / `APC advantage:` / `Symbolic machine ...` (openvm/src/test_utils.rs:63-66 writes them, the
# reference's apc_builder_* tests compare against openvm-riscv/tests/apc_snapshots/**) -----------------------------------------
MNEMONICS = {"ADD": 512, "SUB": 513, "XOR": 514, "OR": 515, "AND": 516, "SLL": 517, "SRL": 518, "SRA": 519, "SLT": 520, "SLTU": 521,
             "LOADW": 528, "LOADBU": 529, "LOADHU": 530, "STOREW": 531, "STOREH": 532, "STOREB": 533, "LOADB": 534, "LOADH": 535,
             "BEQ": 544, "BNE": 545, "BLT": 549, "BLTU": 550, "BGE": 551, "BGEU": 552, "JAL": 560, "LUI": 561, "JALR": 565, "AUIPC": 576,
             "MUL": 592, "MULH": 593, "MULHSU": 594, "MULHU": 595, "DIV": 596, "DIVU": 597, "REM": 598, "REMU": 599}


def parse_instruction(text: str) -> list:
    """One line of the reference's instruction formatter (openvm-riscv/src/isa/instruction_formatter.rs:6-48) back into the wire
    format [opcode, a, b, c, d, e, f, g] (c reduced mod p): `ADD rd_ptr = 8, rs1_ptr = 8, rs2 = 1, rs2_as = 0`,
    `LOADW rd_rs2_ptr = 60, rs1_ptr = 56, imm = 0, mem_as = 2, needs_write = 1, imm_sign = 0`, `BLTU 44 48 -44 1 1`, `MUL 8 7 5 1 0`."""
    name, _, rest = text.strip().partition(" ")
    op = MNEMONICS[name]
    if "=" in rest:
        v = {k.strip(): int(x) for k, x in (kv.split("=") for kv in rest.split(","))}
        if 512 <= op <= 521:
            return [op, v["rd_ptr"], v["rs1_ptr"], v["rs2"] % P, 1, v["rs2_as"], 0, 0]
        return [op, v["rd_rs2_ptr"], v["rs1_ptr"], v["imm"] % P, 1, v["mem_as"], v["needs_write"], v["imm_sign"]]
    a, b, c, d, e = (int(x) for x in rest.split())
    return [op, a % P, b % P, c % P, d, e, 0, 0]  # the formatter prints five operands: f = g = 0 (symbolic_instruction_builder.rs:6-35)


def parse_apc_snapshot(text: str):
    """-> ([(pc, [opcode, a, b, c, d, e, f, g])], TextAir of the optimised machine). Column `<original column>_<k>` of the machine is
    the cell `<original column>` of the k-th instruction's row (its AIR: the opcode's); `is_valid` and optimiser-made columns
    (`free_var_*`, `inv_of_sum_*`: derived columns whose definitions the text does not carry) are the rest."""
    head, _, rest = text.partition("Symbolic machine using")
    instrs = []
    for line in head.split("Instructions:")[1].split("APC advantage:")[0].splitlines():
        if line.strip():
            pc, _, ins = line.partition(":")
            instrs.append((int(pc), parse_instruction(ins)))
    (air,) = parse_airs("# apc\nSymbolic machine using" + rest)
    return instrs, air


def postfix_to_wire(code, ref_of_column):
    """Post-fix code (this module's encoding, column-index operands) -> the serde wire format of an `AlgebraicExpression`
    (expression/src/lib.rs:209-246: a number, "name@id", [left, op, right] or ["-", e]); ref_of_column[c] = "name@id"."""
    st, i, code = [], 0, [int(x) for x in code]
    while i < len(code):
        op = code[i]
        if op == OP_PUSH_COL:
            st.append(ref_of_column[code[i + 1]])
            i += 2
        elif op == OP_PUSH_CONST:
            st.append(code[i + 1])
            i += 2
        elif op == OP_NEG:
            st.append(["-", st.pop()])
            i += 1
        else:
            r, l = st.pop(), st.pop()
            st.append([l, {OP_ADD: "+", OP_SUB: "-", OP_MUL: "*"}[op], r])
            i += 1
    if len(st) != 1:
        raise ValueError("malformed post-fix code")
    return st[0]
