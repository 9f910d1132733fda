"""TEST INFRASTRUCTURE — numpy restatement of the record -> row expansion of the thirteen original RV32IM chips an autoprecompile is
built from: BaseAlu, Shift, LoadStore, BranchEqual, JalLui (the five of a keccak block) and LessThan,
BranchLessThan, Jalr, LoadSignExtend, DivRem, MulH, Multiplication, Auipc (round 3: the rest of the reference's snapshot).

The chips themselves are EXTERNAL (openvm-circuit, `chip.generate_proving_ctx(record_arena)` at
/root/reference/openvm/src/powdr_extension/trace_generator/cuda/mod.rs:228-253); what IS in the checkout is the complete list of
their columns, constraints and bus interactions (openvm-riscv/tests/openvm_constraints.txt:1-1200, 13 AIRs). This file fills
every column of a row from a compact record so that
  * ALL of the algebraic constraints hold (`check_constraints`, evaluated from the parsed text itself), and
  * every bus interaction of the row is a legal one (`check_interactions`): range-checker values lie inside their ranges
    (buses 3, 6, 7), the bitwise lookup's XOR is a XOR, the PC-lookup tuple IS the instruction, the execution bridge moves
    (pc, timestamp) as the ISA says, and the memory bus reads what the record holds and writes what an independent word-level
    RV32IM model (`rv32_model`) computes for the instruction.
Algebraic constraints + lookups are exactly what the reference's proof enforces on these rows, so the two checks together pin
every cell the proof system pins. It is the checker for the device expanders in powdr_amd/csrc/original_chips.hip.

Record layout (ours; the reference's DenseRecordArena layouts are EXTERNAL): one block of u32 words per APC call, word-major on
the device (records[word * num_calls + call]); word 0 = the call's first timestamp, then per instruction with substitutions
  BaseAlu / Shift / LessThan / DivRem / MulH / Multiplication : b, c, prev_data (rd before the write), prev_ts(rs1), prev_ts(rs2), prev_ts(rd)
  LoadStore / LoadSignExtend : rs1_data, read_data (the aligned word that is read), prev_data (the word that is overwritten), prev_ts x 3
  BranchEqual / BranchLessThan : a, b, prev_ts(rs1), prev_ts(rs2)
  Jalr            : rs1_data, prev_data, prev_ts(rs1), prev_ts(rd)
  JalLui / Auipc  : prev_data, prev_ts(rd)
Everything else in a row follows from the record, the instruction's operands and the timestamp. A record of a real execution is
consistent by construction; random test records are made so by the expander itself where the ISA demands it (memory pointers are
brought below 2^29 and to the access's alignment by adjusting rs1, jump targets below 2^30)."""
from __future__ import annotations

import numpy as np

P = 0x78000001
(KIND_BASE_ALU, KIND_SHIFT, KIND_LOAD_STORE, KIND_BRANCH_EQ, KIND_JAL_LUI, KIND_LESS_THAN, KIND_BRANCH_LT, KIND_JALR, KIND_LOAD_SIGN_EXTEND,
 KIND_DIV_REM, KIND_MUL_H, KIND_MUL, KIND_AUIPC) = range(13)
KIND_NAMES = ["BaseAlu", "Shift", "LoadStore", "BranchEqual", "JalLui", "LessThan", "BranchLessThan", "Jalr", "LoadSignExtend", "DivRem", "MulH",
              "Multiplication", "Auipc"]
N_KINDS = 13
WIDTHS = [36, 53, 41, 26, 18, 37, 32, 28, 36, 59, 39, 31, 20]
RECORD_WORDS = [6, 6, 6, 4, 2, 6, 4, 4, 6, 6, 6, 6, 2]
N_PREV_TS = [3, 3, 3, 2, 1, 3, 2, 2, 3, 3, 3, 3, 1]  # trailing record words that are previous timestamps
TS_STEP = N_PREV_TS  # execution-bridge timestamp advance per instruction (bus 0 of each AIR) = its number of memory accesses
OPCODE_KIND = {}
for _lo, _hi, _k in [(512, 516, 0), (517, 519, 1), (520, 521, 5), (528, 533, 2), (534, 535, 8), (544, 545, 3), (549, 552, 6), (560, 561, 4),
                     (565, 565, 7), (576, 576, 12), (592, 592, 11), (593, 595, 10), (596, 599, 9)]:
    for _op in range(_lo, _hi + 1):
        OPCODE_KIND[_op] = _k
ALL_OPCODES = sorted(OPCODE_KIND)
INSTR_DTYPE = np.dtype([("kind", "<u4"), ("opcode", "<u4"), ("pc", "<u4"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("e", "<u4"), ("f", "<u4"),
                        ("g", "<u4"), ("ts_delta", "<u4"), ("air_row", "<u4"), ("rec_off", "<u4")])  # = PowdrOrigInstr (include/powdr_gpu.h)
M32 = 0xFFFFFFFF


def sanitise_instructions(instructions):
    """Synthetic APC blocks (powdr_amd/synth.py) draw their operands at random; bring them into the shape the program ROM holds for
    each chip (the PC-lookup tuple of every AIR, openvm_constraints.txt `// Bus 2` lines): register pointers multiples of 4 below 128,
    d = 1, rs2_as in {0, 1}, memory address space 2, needs_write = 1, 16-bit immediates with their sign in g, operands a chip does
    not have zero."""
    out = []
    for ins in instructions:
        op, a, b, c, d, e, f, g = (int(x) for x in ins)
        k = OPCODE_KIND[op]
        a, b = (a % 32) * 4, (b % 32) * 4
        if k in (KIND_BASE_ALU, KIND_SHIFT, KIND_LESS_THAN):
            e &= 1
            c = (c % 32) * 4 if e else c & 0xFF  # a register pointer, or a small non-negative immediate
            f = g = 0
        elif k in (KIND_LOAD_STORE, KIND_LOAD_SIGN_EXTEND):
            c, e, f, g = c & 0xFFFF, 2, 1, g & 1
        elif k in (KIND_BRANCH_EQ, KIND_BRANCH_LT):
            e, f, g = 1, 0, 0
        elif k == KIND_JAL_LUI:
            b, c, e, f, g = 0, c & 0xFFFFF, 0, 1, 0
        elif k == KIND_JALR:
            c, e, f, g = c & 0xFFFF, 0, 1, g & 1
        elif k == KIND_AUIPC:
            b, c, e, f, g = 0, c & 0xFFFFFF, 0, 0, 0
        else:  # DivRem, MulH, Multiplication: three registers
            c, e, f, g = (c % 32) * 4, 0, 0, 0
        out.append([op, a, b, c, 1, e, f, g])
    return out


def build_instruction_table(instructions, has_subs, start_pc=0x200000, pcs=None):
    """instructions: [[opcode, a, b, c, d, e, f, g], ...] of the APC block (autoprecompiles Instr wire format);
    has_subs[i]: the instruction keeps at least one cell (only those rows exist in the dummy traces, cuda/mod.rs:283-291);
    pcs: the instructions' pcs when the block is a superblock of several basic blocks (default: start_pc + 4 i).
    Returns (table[INSTR_DTYPE] of the instructions WITH substitutions in program order, index of each in `instructions`,
    row_block_size per kind, words per call record)."""
    rows, idx = [], []
    air_rows = [0] * N_KINDS
    rec_off, ts = 1, 0
    for i, ins in enumerate(instructions):
        kind = OPCODE_KIND[int(ins[0])]
        if has_subs[i]:
            rows.append((kind, int(ins[0]), start_pc + 4 * i if pcs is None else int(pcs[i]), int(ins[1]), int(ins[2]), int(ins[3]) % P, int(ins[5]), int(ins[6]), int(ins[7]), ts,
                         air_rows[kind], rec_off))
            idx.append(i)
            air_rows[kind] += 1
            rec_off += RECORD_WORDS[kind]
        ts += TS_STEP[kind]
    return np.array(rows, dtype=INSTR_DTYPE), np.array(idx, np.int64), air_rows, rec_off


def random_records(table, words_per_call, num_calls, seed=0):
    """Random but CONSISTENT records [words_per_call, num_calls] (u32): timestamps increase, previous timestamps lie before the
    access, comparison operands are equal half of the time (and differ in one limb only a quarter of the time)."""
    rng = np.random.default_rng(seed)
    rec = rng.integers(0, 1 << 32, size=(words_per_call, num_calls), dtype=np.uint64).astype(np.uint32)
    base = rng.integers(1 << 10, 1 << 26, size=num_calls, dtype=np.uint64).astype(np.uint32)
    rec[0] = base
    for ins in table:
        o, k = int(ins["rec_off"]), int(ins["kind"])
        ts = base.astype(np.int64) + int(ins["ts_delta"])
        n_prev = N_PREV_TS[k]
        first_prev = RECORD_WORDS[k] - n_prev
        for j in range(n_prev):
            gap = rng.integers(1, 1 << 28, size=num_calls)  # timestamp + j - prev - 1 in [0, 2^29)
            rec[o + first_prev + j] = np.maximum(ts + j - gap, 0).astype(np.uint32)
        if k in (KIND_BRANCH_EQ, KIND_BRANCH_LT, KIND_LESS_THAN, KIND_DIV_REM):
            u = rng.random(num_calls)
            limb = rng.integers(0, 4, size=num_calls)
            near = rec[o] ^ (rng.integers(1, 256, size=num_calls).astype(np.uint32) << (8 * limb).astype(np.uint32))
            rec[o + 1] = np.where(u < 0.5, rec[o], np.where(u < 0.75, near, rec[o + 1]))
        if k == KIND_DIV_REM:  # small divisors, so that quotients are not all 0 / 1
            small = rng.random(num_calls) < 0.5
            rec[o + 1] = np.where(small, (rec[o + 1].astype(np.int32) >> 20).astype(np.uint32), rec[o + 1])
    return rec


def _bytes(w):
    w = np.asarray(w).astype(np.int64)
    return [(w >> (8 * i)) & 0xFF for i in range(4)]


def _word(b):
    return sum(b[i] << (8 * i) for i in range(4))


def _ts_decomp(ts, prev):
    d = ts - prev.astype(np.int64) - 1
    return [prev.astype(np.int64), d & 0x1FFFF, d >> 17]


def _signed(w):
    w = np.asarray(w).astype(np.int64)
    return np.where(w >= 1 << 31, w - (1 << 32), w)


def _inv(v):
    """field inverse of every element (0 -> 0), int64 array"""
    return np.array([pow(int(x) % P, P - 2, P) for x in np.asarray(v).reshape(-1)], np.int64).reshape(np.shape(v))


def _imm_ext(ins):
    """the sign-extended 16-bit immediate of a load / store / jalr as a 32-bit word: operand c = low 16 bits, operand g = sign"""
    return (int(ins["c"]) & 0xFFFF) | (0xFFFF0000 if int(ins["g"]) & 1 else 0)


def _mem_ptr(ins, rs1_word, align_mask):
    """(ptr, rs1) of a memory access: rs1 + imm brought below 2^29 and to the access's alignment by adjusting rs1 (a record of a
    real execution already is; random test records are not)"""
    ext = _imm_ext(ins)
    ptr = (rs1_word.astype(np.int64) + ext) & 0x1FFFFFFF & ~np.int64(align_mask)
    return ptr, (ptr - ext) & M32


# the flags columns of LoadStoreCoreAir (openvm_constraints.txt:776: the opcode as a polynomial of flags in {0, 1, 2}^4) by (opcode, shift)
LOAD_STORE_FLAGS = {(528, 0): (2, 0, 0, 0), (530, 0): (0, 2, 0, 0), (530, 2): (0, 0, 2, 0), (529, 0): (0, 0, 0, 2), (529, 1): (1, 0, 0, 0),
                    (529, 2): (0, 1, 0, 0), (529, 3): (0, 0, 1, 0), (531, 0): (0, 0, 0, 1), (532, 0): (1, 1, 0, 0), (532, 2): (1, 0, 1, 0),
                    (533, 0): (1, 0, 0, 1), (533, 1): (0, 1, 1, 0), (533, 2): (0, 1, 0, 1), (533, 3): (0, 0, 1, 1)}
ACCESS_ALIGN = {528: 3, 531: 3, 530: 1, 532: 1, 535: 1, 529: 0, 533: 0, 534: 0}  # LOADW/STOREW words, LOADHU/STOREH/LOADH halves, bytes


def _load_store_write(op, shift, read, prev):
    """write_data of LoadStoreCoreAir (byte lists): loads place the selected bytes at the bottom (zero extended), stores merge into prev"""
    z = np.zeros_like(read[0])
    out = []
    for s in range(4):
        if op == 528 or op == 531:
            w = read
        elif op == 530:
            w = [read[s], read[s + 1], z, z] if s in (0, 2) else None
        elif op == 529:
            w = [read[s], z, z, z]
        elif op == 532:
            w = ([read[0], read[1], prev[2], prev[3]] if s == 0 else [prev[0], prev[1], read[0], read[1]]) if s in (0, 2) else None
        else:
            w = [read[0] if i == s else prev[i] for i in range(4)]
        out.append(w)
    res = []
    for i in range(4):
        v = z.copy()
        for s in range(4):
            if out[s] is not None:
                v = np.where(shift == s, out[s][i], v)
        res.append(v)
    return res


def _diff_marker(x, y, lt):
    """(marker[4], diff_val) of the LessThan cores: the most significant limb where x and y differ gets the marker, diff_val is the
    positive difference there (y - x if x < y else x - y); x, y: limb lists, limb 3 possibly signed (the *_msb_f columns)"""
    n = x[0].shape[0]
    marker = [np.zeros(n, np.int64) for _ in range(4)]
    val = np.zeros(n, np.int64)
    done = np.zeros(n, bool)
    for i in (3, 2, 1, 0):
        pick = (x[i] != y[i]) & ~done
        marker[i] = pick.astype(np.int64)
        val = np.where(pick, np.where(lt, y[i] - x[i], x[i] - y[i]), val)
        done |= pick
    return marker, val


def _divrem(op, bw, cw):
    """(q, r) words of DIV / DIVU / REM / REMU (596..599), RISC-V semantics (division by zero: q = all ones, r = b; overflow: q = b, r = 0)"""
    signed = op in (596, 598)
    b = _signed(bw) if signed else bw.astype(np.int64)
    c = _signed(cw) if signed else cw.astype(np.int64)
    zero = c == 0
    cs = np.where(zero, 1, c)
    qa = np.abs(b) // np.abs(cs)
    q = np.where((b < 0) != (cs < 0), -qa, qa)
    r = b - cs * q
    q = np.where(zero, -1, q)
    r = np.where(zero, b, r)
    return q & M32, r & M32


def rv32_model(ins, rec, base_ts):
    """Independent WORD-level model of one instruction on the records of all calls: the memory accesses in the chip's order
    [(enabled, address_space, pointer, word before, word after)], the next pc (field element) and the timestamp step."""
    k, op, pc = int(ins["kind"]), int(ins["opcode"]), int(ins["pc"])
    a, b, c, e, f = int(ins["a"]), int(ins["b"]), int(ins["c"]), int(ins["e"]), int(ins["f"])
    o = int(ins["rec_off"])
    n = rec.shape[1]
    W = lambda j: rec[o + j].astype(np.int64)
    full = lambda v: np.full(n, v, np.int64)
    next_pc = full((pc + 4) % P)
    if k in (KIND_BASE_ALU, KIND_SHIFT, KIND_LESS_THAN, KIND_DIV_REM, KIND_MUL_H, KIND_MUL):
        x = W(0)
        reg = e != 0 or k in (KIND_DIV_REM, KIND_MUL_H, KIND_MUL)
        y = W(1) if reg else full((c & 0xFFFFFF) | (((c >> 16) & 0xFF) << 24))
        sx, sy = _signed(x), _signed(y)
        sh = y & 31
        res = {512: lambda: x + y, 513: lambda: x - y, 514: lambda: x ^ y, 515: lambda: x | y, 516: lambda: x & y,
               517: lambda: x << sh, 518: lambda: x >> sh, 519: lambda: sx >> sh,
               520: lambda: (sx < sy).astype(np.int64), 521: lambda: (x < y).astype(np.int64),
               592: lambda: np.array([(int(p) * int(q)) & M32 for p, q in zip(x, y)], np.int64),
               593: lambda: np.array([((int(p) * int(q)) >> 32) & M32 for p, q in zip(sx, sy)], np.int64),
               594: lambda: np.array([((int(p) * int(q)) >> 32) & M32 for p, q in zip(sx, y)], np.int64),
               595: lambda: np.array([((int(p) * int(q)) >> 32) & M32 for p, q in zip(x, y)], np.int64),
               596: lambda: _divrem(op, x, y)[0], 597: lambda: _divrem(op, x, y)[0], 598: lambda: _divrem(op, x, y)[1],
               599: lambda: _divrem(op, x, y)[1]}[op]() & M32
        acc = [(1, 1, full(b), x, x), (1 if reg else 0, 1 if reg else 0, full(c), y, y), (1, 1, full(a), W(2), res)]
        return acc, next_pc, 3
    if k in (KIND_LOAD_STORE, KIND_LOAD_SIGN_EXTEND):
        ptr, rs1 = _mem_ptr(ins, W(0), ACCESS_ALIGN[op])
        shift = ptr & 3
        aligned = ptr - shift
        read, prev = W(1), W(2)
        rb, pb = _bytes(read), _bytes(prev)
        sel = lambda j: sum(np.where(shift == s, rb[s + j] if s + j < 4 else 0, 0) for s in range(4))
        if op in (528, 529, 530, 534, 535):
            if op == 528:
                val = read
            elif op == 529:
                val = sel(0)
            elif op == 530:
                val = sel(0) | (sel(1) << 8)
            elif op == 534:
                val = (sel(0) ^ 0x80) - 0x80
            else:
                val = ((sel(0) | (sel(1) << 8)) ^ 0x8000) - 0x8000
            acc = [(1, 1, full(b), rs1, rs1), (1, e, aligned, read, read), (f & 1, 1, full(a), prev, val & M32)]
        else:
            nbytes = {531: 4, 532: 2, 533: 1}[op]
            mask = ((1 << (8 * nbytes)) - 1) << (8 * shift)
            val = (prev & ~mask) | ((read << (8 * shift)) & mask)
            acc = [(1, 1, full(b), rs1, rs1), (1, 1, full(a), read, read), (1, e, aligned, prev, val & M32)]
        return acc, next_pc, 3
    if k in (KIND_BRANCH_EQ, KIND_BRANCH_LT):
        x, y = W(0), W(1)
        taken = {544: x == y, 545: x != y, 549: _signed(x) < _signed(y), 550: x < y, 551: _signed(x) >= _signed(y), 552: x >= y}[op]
        return [(1, 1, full(a), x, x), (1, 1, full(b), y, y)], np.where(taken, (pc + c) % P, (pc + 4) % P), 2
    if k == KIND_JAL_LUI:
        rd = full((pc + 4) if op == 560 else ((c << 12) & M32))
        return [(f & 1, 1, full(a), W(0), rd)], full((pc + c) % P if op == 560 else (pc + 4) % P), 1
    if k == KIND_JALR:
        ext = _imm_ext(ins)
        to_pc = (W(0) + ext) & 0x3FFFFFFF  # jump targets below 2^30
        rs1 = (to_pc - ext) & M32
        return [(1, 1, full(b), rs1, rs1), (f & 1, 1, full(a), W(1), full(pc + 4))], (to_pc & ~np.int64(1)) % P, 2
    assert k == KIND_AUIPC
    return [(1, 1, full(a), W(0), full((pc + (c << 8)) & M32))], next_pc, 1


def expand_rows(ins, rec, base_ts):
    """All cells (canonical, int64 arrays of num_calls) of the row instruction `ins` produces, in the AIR's column order."""
    k, op = int(ins["kind"]), int(ins["opcode"])
    n = rec.shape[1]
    const = lambda v: np.full(n, v % P, np.int64)
    zero = np.zeros(n, np.int64)
    o = int(ins["rec_off"])
    ts = base_ts.astype(np.int64) + int(ins["ts_delta"])
    pc = int(ins["pc"])
    A, B, C_ = int(ins["a"]), int(ins["b"]), int(ins["c"])
    if k in (KIND_BASE_ALU, KIND_SHIFT, KIND_LESS_THAN):
        rs2_as = int(ins["e"])
        b = _bytes(rec[o])
        if rs2_as:
            c = _bytes(rec[o + 1])
        else:  # immediate: 24-bit value, sign byte repeated (constraints (1 - rs2_as) * (rs2 - (c0 + 256 c1 + 65536 c2)), c2 = c3 in {0, 255})
            c = [const(C_ & 0xFF), const((C_ >> 8) & 0xFF), const((C_ >> 16) & 0xFF), const((C_ >> 16) & 0xFF)]
        bw, cw = _word(b), _word(c)
        head = [const(pc), ts, const(A), const(B), const(C_), const(rs2_as)]
        r0 = _ts_decomp(ts, rec[o + 3])
        r1 = _ts_decomp(ts + 1, rec[o + 4]) if rs2_as else [const(0)] * 3
        w = _ts_decomp(ts + 2, rec[o + 5])
        prev_data = _bytes(rec[o + 2])
        if k == KIND_BASE_ALU:
            aw = {512: (bw + cw) & M32, 513: (bw - cw) & M32, 514: bw ^ cw, 515: bw | cw, 516: bw & cw}[op]
            flags = [const(1 if op == 512 + j else 0) for j in range(5)]
            return head + r0 + r1 + w + prev_data + _bytes(aw) + b + c + flags
        if k == KIND_LESS_THAN:
            signed = op == 520
            bm = np.where(signed & (b[3] >= 128), b[3] - 256, b[3])
            cm = np.where(signed & (c[3] >= 128), c[3] - 256, c[3])
            lt = (_signed(bw) < _signed(cw)) if signed else (bw < cw)
            marker, val = _diff_marker(b[:3] + [bm], c[:3] + [cm], lt)
            return (head + r0 + r1 + w + prev_data + b + c + [lt.astype(np.int64), const(int(signed)), const(int(not signed)), bm % P, cm % P]
                    + marker + [val])
        shift = c[0] & 31
        bit, limb = shift & 7, shift >> 3
        sll, srl, sra = op == 517, op == 518, op == 519
        sign = (b[3] >> 7) if sra else zero
        if sll:
            aw = (bw << shift) & M32
            carry = [b[i] >> (8 - bit) for i in range(4)]
        else:
            fill = np.where(sign == 1, (M32 << (32 - shift)) & M32, 0) if sra else 0
            aw = (bw >> shift) | fill
            carry = [b[i] & ((1 << bit) - 1) for i in range(4)]
        mul_l = np.where(np.full(n, sll), 1 << bit, 0).astype(np.int64)
        mul_r = np.where(np.full(n, not sll), 1 << bit, 0).astype(np.int64)
        bit_marker = [(bit == j).astype(np.int64) for j in range(8)]
        limb_marker = [(limb == j).astype(np.int64) for j in range(4)]
        return (head + r0 + r1 + w + prev_data + _bytes(aw) + b + c + [const(int(sll)), const(int(srl)), const(int(sra)), mul_l, mul_r, sign]
                + bit_marker + limb_marker + carry)
    if k in (KIND_DIV_REM, KIND_MUL_H, KIND_MUL):
        b, c = _bytes(rec[o]), _bytes(rec[o + 1])
        bw, cw = _word(b), _word(c)
        head = ([const(pc), ts, const(A), const(B), const(C_)] + _ts_decomp(ts, rec[o + 3]) + _ts_decomp(ts + 1, rec[o + 4]) + _ts_decomp(ts + 2, rec[o + 5])
                + _bytes(rec[o + 2]))
        low = np.array([(int(x) * int(y)) & M32 for x, y in zip(bw, cw)], np.int64)
        if k == KIND_MUL:
            return head + _bytes(low) + b + c + [const(1)]
        if k == KIND_MUL_H:
            b_neg = (b[3] >> 7) if op in (593, 594) else zero  # MULH: both signed; MULHSU: b signed; MULHU: none
            c_neg = (c[3] >> 7) if op == 593 else zero
            sb = np.where(b_neg == 1, bw - (1 << 32), bw)
            sc = np.where(c_neg == 1, cw - (1 << 32), cw)
            high = np.array([((int(x) * int(y)) >> 32) & M32 for x, y in zip(sb, sc)], np.int64)
            return head + _bytes(high) + b + c + _bytes(low) + [b_neg * 255, c_neg * 255] + [const(int(op == 593 + j)) for j in range(3)]
        signed = op in (596, 598)
        qw, rw = _divrem(op, bw, cw)
        q, r = _bytes(qw), _bytes(rw)
        zero_div = (cw == 0).astype(np.int64)
        r_zero = ((rw == 0) & (cw != 0)).astype(np.int64)
        b_sign = (b[3] >> 7) if signed else zero
        c_sign = (c[3] >> 7) if signed else zero
        sign_xor = b_sign ^ c_sign
        # q_sign: the sign the quotient is EXTENDED with in the carry chain: sign_xor when q != 0, 0 when q == 0 (constraints at
        # openvm_constraints.txt:942-943); free when the divisor is zero, where q = -1 reads as negative exactly for the signed opcodes
        q_sign = np.where(zero_div == 1, int(signed), np.where(qw != 0, sign_xor, 0))
        r_prime_w = np.where(sign_xor == 1, (-rw) & M32, rw)
        rp = _bytes(r_prime_w)
        r_inv = [_inv((rp[i] - 256) % P) for i in range(4)]
        need_lt = (zero_div == 0) & (r_zero == 0)
        # |r| < |c|: compare r' with c under c's sign (c_sign = 0: c - r' > 0, c_sign = 1: r' - c > 0 at the first differing limb from the top)
        marker = [np.zeros(n, np.int64) for _ in range(4)]
        lt_diff = np.zeros(n, np.int64)
        done = ~need_lt
        for i in (3, 2, 1, 0):
            pick = (rp[i] != c[i]) & ~done
            marker[i] = pick.astype(np.int64)
            lt_diff = np.where(pick, np.where(c_sign == 1, rp[i] - c[i], c[i] - rp[i]), lt_diff)
            done |= pick
        c_sum_inv = _inv(c[0] + c[1] + c[2] + c[3])
        r_sum_inv = _inv(r[0] + r[1] + r[2] + r[3])
        return (head + b + c + q + r + [zero_div, r_zero, b_sign, c_sign, q_sign, sign_xor, c_sum_inv, r_sum_inv] + rp + r_inv + marker + [lt_diff]
                + [const(int(op == 596 + j)) for j in range(4)])
    if k in (KIND_LOAD_STORE, KIND_LOAD_SIGN_EXTEND):
        imm, imm_sign = C_ & 0xFFFF, int(ins["g"]) & 1  # operand c = 16-bit immediate, operand g = its sign
        ptr, rs1w = _mem_ptr(ins, rec[o], ACCESS_ALIGN[op])
        shift = ptr & 3
        rs1 = _bytes(rs1w)
        limbs = [ptr & 0xFFFF, ptr >> 16]
        read, prev = _bytes(rec[o + 1]), _bytes(rec[o + 2])
        mem_as = int(ins["e"])
        needs_write = int(ins["f"]) & 1  # (a load into x0 has f = 0; stores always write)
        t0 = _ts_decomp(ts, rec[o + 3])
        t1 = _ts_decomp(ts + 1, rec[o + 4])
        t2 = _ts_decomp(ts + 2, rec[o + 5]) if needs_write else [const(0)] * 3
        lead = ([const(pc), ts, const(B)] + rs1 + t0 + [const(A if needs_write else 0)] + t1 + [const(imm), const(imm_sign)] + limbs + [const(mem_as)]
                + t2 + [const(needs_write)])
        if k == KIND_LOAD_STORE:
            flags = [sum(np.where(shift == s, LOAD_STORE_FLAGS.get((op, s), (0, 0, 0, 0))[j], 0) for s in range(4)) for j in range(4)]
            is_load = op <= 530
            return lead + flags + [const(1), const(int(is_load))] + read + prev + _load_store_write(op, shift, read, prev)
        # LoadSignExtend: shifted_read_data = read_data rotated down by 2 * (shift >> 1) bytes; LOADB picks byte shift & 1 of it, LOADH bytes 0, 1
        msb_shift = shift >> 1
        sh_read = [np.where(msb_shift == 1, read[(i + 2) % 4], read[i]) for i in range(4)]
        loadb = op == 534
        flag1 = (shift & 1) if loadb else zero
        flag0 = (1 - flag1) if loadb else zero
        top = np.where(flag0 == 1, sh_read[0], sh_read[1]) if loadb else sh_read[1]
        return lead + [flag0, flag1, const(int(not loadb)), msb_shift, top >> 7] + sh_read + prev
    if k in (KIND_BRANCH_EQ, KIND_BRANCH_LT):
        a, b = _bytes(rec[o]), _bytes(rec[o + 1])
        head = [const(pc), ts, const(A), const(B)] + _ts_decomp(ts, rec[o + 2]) + _ts_decomp(ts + 1, rec[o + 3]) + a + b
        if k == KIND_BRANCH_EQ:
            eq = (rec[o] == rec[o + 1])
            beq = op == 544
            cmp = (eq if beq else ~eq).astype(np.int64)
            # diff_inv_marker: 1 / (a_i - b_i) at the first differing limb, zero elsewhere
            marker = [np.zeros(n, np.int64) for _ in range(4)]
            done = np.zeros(n, bool)
            for i in range(4):
                d = (a[i] - b[i]) % P
                pick = (d != 0) & ~done
                marker[i] = np.where(pick, _inv(d), 0)
                done |= pick
            return head + [cmp, const(C_), const(int(beq)), const(int(not beq))] + marker
        signed = op in (549, 551)
        is_lt_op = op in (549, 550)
        aw, bw = _word(a), _word(b)
        am = np.where(signed & (a[3] >= 128), a[3] - 256, a[3])
        bm = np.where(signed & (b[3] >= 128), b[3] - 256, b[3])
        lt = (_signed(aw) < _signed(bw)) if signed else (aw < bw)
        cmp = (lt if is_lt_op else ~lt).astype(np.int64)
        marker, val = _diff_marker(a[:3] + [am], b[:3] + [bm], lt)
        return head + [cmp, const(C_)] + [const(int(op == 549 + j)) for j in range(4)] + [am % P, bm % P, lt.astype(np.int64)] + marker + [val]
    if k == KIND_JAL_LUI:
        is_jal = op == 560
        needs_write = int(ins["f"]) & 1
        rdw = (pc + 4) if is_jal else ((C_ << 12) & M32)
        rd_data = [const((rdw >> (8 * i)) & 0xFF) for i in range(4)]
        t = _ts_decomp(ts, rec[o + 1]) if needs_write else [const(0)] * 3
        prev = _bytes(rec[o]) if needs_write else [const(0)] * 4
        return [const(pc), ts, const(A)] + t + prev + [const(needs_write), const(C_)] + rd_data + [const(int(is_jal)), const(int(not is_jal))]
    if k == KIND_JALR:
        needs_write = int(ins["f"]) & 1
        ext = _imm_ext(ins)
        to_pc = (rec[o].astype(np.int64) + ext) & 0x3FFFFFFF
        rs1 = _bytes((to_pc - ext) & M32)
        rd = pc + 4
        t1 = _ts_decomp(ts + 1, rec[o + 3]) if needs_write else [const(0)] * 3
        prev = _bytes(rec[o + 1]) if needs_write else [const(0)] * 4
        return ([const(pc), ts, const(B)] + _ts_decomp(ts, rec[o + 2]) + [const(A)] + t1 + prev + [const(needs_write), const(C_ & 0xFFFF)] + rs1
                + [const((rd >> 8) & 0xFF), const((rd >> 16) & 0xFF), const(rd >> 24), const(1), to_pc & 1, (to_pc & 0xFFFF) >> 1, to_pc >> 16,
                   const(int(ins["g"]) & 1)])
    assert k == KIND_AUIPC
    rd = (pc + (C_ << 8)) & M32
    return ([const(pc), ts, const(A)] + _ts_decomp(ts, rec[o + 1]) + _bytes(rec[o]) + [const(1), const(C_ & 0xFF), const((C_ >> 8) & 0xFF), const((C_ >> 16) & 0xFF),
                                                                                 const((pc >> 8) & 0xFF), const((pc >> 16) & 0xFF)]
            + [const((rd >> (8 * i)) & 0xFF) for i in range(4)])


def expand_dummy_traces(table, rec, row_block_size, pow2=True):
    """The column-major dummy traces {kind: u32[width, height]} the original chips would hand to the gather: row of instruction i of
    call r at air_row(i) + r * row_block_size (only kinds that occur)."""
    n = rec.shape[1]
    out = {}
    for k in range(N_KINDS):
        b = row_block_size[k]
        if not b:
            continue
        rows = b * n
        h = max(4, 1 << (rows - 1).bit_length()) if pow2 else rows
        out[k] = np.zeros((WIDTHS[k], h), np.uint32)
    for ins in table:
        k = int(ins["kind"])
        cells = expand_rows(ins, rec, rec[0])
        assert len(cells) == WIDTHS[k], (KIND_NAMES[k], len(cells))
        r = int(ins["air_row"]) + np.arange(n) * row_block_size[k]
        for c, v in enumerate(cells):
            out[k][c, r] = (v % P).astype(np.uint32)
    return out


# ---- the reference's constraints, evaluated directly on columns (numpy, exact mod p) ------------------------------------
def eval_postfix(bc, cols):
    """post-fix bytecode (powdr_amd/air_text.py encoding, column-index operands) on int64 column arrays."""
    st = []
    i = 0
    while i < len(bc):
        op = int(bc[i])
        i += 1
        if op == 0:
            st.append(cols[int(bc[i])])
            i += 1
        elif op == 1:
            st.append(np.int64(int(bc[i])))
            i += 1
        elif op == 5:
            st.append((-st.pop()) % P)
        else:
            y, x = st.pop(), st.pop()
            st.append((x + y) % P if op == 2 else (x - y) % P if op == 3 else (x * y) % P)
    assert len(st) == 1
    return st[0]


def check_constraints(cons_bc, cons_spans, trace_cols):
    """Number of (row, constraint) pairs that do not vanish and the first failing constraint index (or None)."""
    cols = [c.astype(np.int64) for c in trace_cols]
    bad, first = 0, None
    for k, (off, ln) in enumerate(np.asarray(cons_spans).reshape(-1, 2).tolist()):
        v = eval_postfix(cons_bc[off:off + ln], cols)
        nz = int(np.count_nonzero(v))
        if nz and first is None:
            first = k
        bad += nz
    return bad, first


BUS_EXECUTION, BUS_MEMORY, BUS_PC_LOOKUP, BUS_VAR_RANGE, BUS_BITWISE, BUS_TUPLE_RANGE = 0, 1, 2, 3, 6, 7


def check_interactions(interactions, ins, rec, rows):
    """The bus interactions of one instruction's rows (all calls) against the tables they are looked up in.
    interactions = (inter[n, 3] (bus, n_args, first span), spans, bytecode) of the instruction's AIR (openvm_airs.npz);
    rows: the row's cells as int64 arrays over the calls. Returns a list of human-readable failures (empty = fine)."""
    inter, spans, ibc = interactions
    cols = [np.asarray(c).astype(np.int64) % P for c in rows]
    n = rec.shape[1]
    ev = lambda s: np.broadcast_to(eval_postfix(ibc[int(spans[s][0]):int(spans[s][0]) + int(spans[s][1])], cols), (n,)).astype(np.int64)
    fails = []
    k, op, pc = int(ins["kind"]), int(ins["opcode"]), int(ins["pc"])
    ts = rec[0].astype(np.int64) + int(ins["ts_delta"])
    accesses, next_pc, step = rv32_model(ins, rec, rec[0])
    mem, exe = [], []
    for i, (bus, n_args, s0) in enumerate(np.asarray(inter).tolist()):
        mult = ev(s0)
        args = [ev(s0 + 1 + j) for j in range(n_args)]
        on = mult != 0
        say = lambda what: fails.append(f"{KIND_NAMES[k]} op {op} interaction {i} (bus {bus}): {what}")
        if bus in (BUS_VAR_RANGE, BUS_BITWISE, BUS_TUPLE_RANGE, BUS_PC_LOOKUP) and not np.isin(mult, (0, 1)).all():
            say("multiplicity outside {0, 1}")
        if bus == BUS_VAR_RANGE:
            if (on & (args[0] >= (1 << args[1]))).any():
                say(f"value outside {int(args[1][0])} bits")
        elif bus == BUS_BITWISE:
            if (on & ((args[0] >= 256) | (args[1] >= 256))).any():
                say("operand outside a byte")
            if (on & ~np.isin(args[3], (0, 1))).any() or (on & (args[3] == 0) & (args[2] != 0)).any() or (on & (args[3] == 1) & (args[2] != (args[0] ^ args[1]))).any():
                say("not a row of the bitwise table")
        elif bus == BUS_TUPLE_RANGE:
            if (on & ((args[0] >= 256) | (args[1] >= 2048))).any():
                say("outside 256 x 2048")
        elif bus == BUS_PC_LOOKUP:
            want = [pc, op, int(ins["a"]), int(ins["b"]), int(ins["c"]), 1, int(ins["e"]), int(ins["f"]), int(ins["g"])]
            if not on.all() or any((args[j] != want[j] % P).any() for j in range(9)):
                say(f"is not the instruction {want}: {[int(x[0]) for x in args]}")
        elif bus == BUS_EXECUTION:
            exe.append((mult, args))
        elif bus == BUS_MEMORY:
            mem.append((mult, args))
        else:
            say("unknown bus")
    if len(exe) != 2 or (exe[0][0] != P - 1).any() or (exe[1][0] != 1).any():
        fails.append(f"{KIND_NAMES[k]} op {op}: execution bridge is not one receive and one send")
    else:
        if (exe[0][1][0] != pc).any() or (exe[0][1][1] != ts % P).any():
            fails.append(f"{KIND_NAMES[k]} op {op}: execution bridge does not start at (pc, timestamp)")
        if (exe[1][1][0] != next_pc).any() or (exe[1][1][1] != (ts + step) % P).any():
            fails.append(f"{KIND_NAMES[k]} op {op}: execution bridge does not end at the model's (next pc, timestamp + {step})")
    if len(mem) != 2 * len(accesses):
        fails.append(f"{KIND_NAMES[k]} op {op}: {len(mem)} memory interactions for {len(accesses)} accesses")
        return fails
    for j, (enabled, space, ptr, before, after) in enumerate(accesses):
        (m0, a0), (m1, a1) = mem[2 * j], mem[2 * j + 1]
        what = f"{KIND_NAMES[k]} op {op}: memory access {j}"
        if not enabled:
            if m0.any() or m1.any():
                fails.append(what + " should be disabled")
            continue
        if (m0 != P - 1).any() or (m1 != 1).any():
            fails.append(what + " is not a receive followed by a send")
        for nm, a, word in (("old", a0, before), ("new", a1, after)):
            if (a[0] != space).any() or (a[1] != ptr % P).any():
                fails.append(what + f" {nm}: wrong address ({int(a[0][0])}, {int(a[1][0])}) instead of ({space}, {int(ptr[0])})")
            got = a[2] + (a[3] << 8) + (a[4] << 16) + (a[5] << 24)
            if any((a[2 + i] >= 256).any() for i in range(4)) or (got != (word & M32)).any():
                r = int(np.argmax((got != (word & M32)) | (a[2] >= 256)))
                fails.append(what + f" {nm}: data {int(got[r]):#x} instead of the model's {int(word[r]) & M32:#x} (call {r})")
        if (a1[6] != (ts + j) % P).any() or (a0[6] >= a1[6]).any():
            fails.append(what + " timestamps: not (previous < timestamp + access index)")
    return fails
