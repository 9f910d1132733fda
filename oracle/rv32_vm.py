"""TEST INFRASTRUCTURE — a small RV32IM executor that produces CONSISTENT call records for a block of instructions: what the
reference's preflight execution (`PowdrExecutor::execute`, /root/reference/openvm/src/powdr_extension/executor/mod.rs:457-528, SURVEY.md §8 row
a7: EXTERNAL executors fill the record arenas) hands to trace generation. Registers and memory are words keyed by (address space,
pointer) with the timestamp of their last access; an instruction's accesses happen at timestamp + 0, 1, 2 in the chip's order
(the memory-bus lines of openvm-riscv/tests/openvm_constraints.txt), results come from oracle/original_chips.py::rv32_model.

Used to check the reference's golden APC machines (openvm-riscv/tests/apc_snapshots/**, tests/golden/apc_snapshots.json.gz): an
optimised APC relies on the memory consistency of a real execution (a register read after a write sees the written word and the
write's timestamp), which random records do not have."""
from __future__ import annotations

import numpy as np

from . import original_chips as oc

M32 = 0xFFFFFFFF
MULT_KINDS = (oc.KIND_DIV_REM, oc.KIND_MUL_H, oc.KIND_MUL)


class Rejected(Exception):
    """the drawn initial state leaves the block's path (a branch goes the other way) or the ISA's domain (misaligned / too large a pointer);
    `repair` = ((space, ptr), word): an initial word that would make the offending equality branch go the block's way"""

    def __init__(self, why, repair=None):
        super().__init__(why)
        self.repair = repair


def access_list(ins, ptr_of_memory):
    """[(enabled, address space, pointer)] of an instruction's accesses in the chip's order; ptr_of_memory: the aligned memory pointer."""
    k, a, b, c, e, f = (int(ins[n]) for n in ("kind", "a", "b", "c", "e", "f"))
    if k in (oc.KIND_BASE_ALU, oc.KIND_SHIFT, oc.KIND_LESS_THAN) or k in MULT_KINDS:
        reg = e != 0 or k in MULT_KINDS
        return [(1, 1, b), (int(reg), 1, c), (1, 1, a)]
    if k in (oc.KIND_LOAD_STORE, oc.KIND_LOAD_SIGN_EXTEND):
        if int(ins["opcode"]) in (531, 532, 533):
            return [(1, 1, b), (1, 1, a), (1, e, ptr_of_memory)]
        return [(1, 1, b), (1, e, ptr_of_memory), (f & 1, 1, a)]
    if k in (oc.KIND_BRANCH_EQ, oc.KIND_BRANCH_LT):
        return [(1, 1, a), (1, 1, b)]
    if k == oc.KIND_JAL_LUI:
        return [(f & 1, 1, a)]
    if k == oc.KIND_JALR:
        return [(1, 1, b), (f & 1, 1, a)]
    return [(1, 1, a)]


def run_call(table, next_pcs, words_per_call, rng, value_pool, forced):
    """One call of the block from a random initial state -> (record column u32[words_per_call], initial {(as, ptr): (word, ts)},
    final {(as, ptr): (word, ts)}, pc after the block). next_pcs[i]: the pc the block continues at after instruction i (None: free);
    forced: initial words by location that override the draw."""
    start_ts = int(rng.integers(1 << 10, 1 << 26))
    state, initial, written = {}, {}, set()

    def draw_word():
        u = rng.random()
        if u < 0.35:
            return int(rng.integers(0, 1 << 32))
        if u < 0.6:
            return int(rng.integers(0, 4))
        if u < 0.8 and value_pool:
            return int(value_pool[int(rng.integers(0, len(value_pool)))])
        if state:
            keys = list(state)
            return state[keys[int(rng.integers(0, len(keys)))]][0]
        return 0

    def touch(space, ptr, want=None):
        key = (space, ptr)
        if key not in state:
            drawn = draw_word() if want is None else want  # (drawn even when overridden: a repaired call keeps the rest of its draws)
            word = 0 if key == (1, 0) else forced.get(key, drawn)  # x0 holds 0
            state[key] = initial[key] = (word & M32, int(rng.integers(0, start_ts)))
        return state[key]

    rec = np.zeros((words_per_call, 1), np.uint32)
    rec[0, 0] = start_ts
    base = rec[0]
    for i, ins in enumerate(table):
        k, op, o = int(ins["kind"]), int(ins["opcode"]), int(ins["rec_off"])
        ts = start_ts + int(ins["ts_delta"])
        n_data = oc.RECORD_WORDS[k] - oc.N_PREV_TS[k]
        mem_ptr = None
        if k in (oc.KIND_LOAD_STORE, oc.KIND_LOAD_SIGN_EXTEND):
            key = (1, int(ins["b"]))
            ext = oc._imm_ext(ins)
            if key not in state and key != (1, 0):  # a base register that is read for the first time: a legal pointer for this access
                target = int(rng.integers(1 << 10, 1 << 26)) * 4 + (int(rng.integers(0, 4)) & ~oc.ACCESS_ALIGN[op] & 3)
                touch(1, int(ins["b"]), (target - ext) & M32)
            ptr = (touch(*key)[0] + ext) & M32
            if ptr >= 1 << 29 or ptr & oc.ACCESS_ALIGN[op]:
                raise Rejected("pointer")
            mem_ptr = ptr & ~3
        if k == oc.KIND_JALR:
            key = (1, int(ins["b"]))
            ext = oc._imm_ext(ins)
            if key not in state and key != (1, 0):
                touch(1, int(ins["b"]), (int(rng.integers(0, 1 << 28)) * 4 - ext) & M32)
            if (touch(*key)[0] + ext) & M32 >= 1 << 30:
                raise Rejected("jump target")
        accesses = access_list(ins, mem_ptr)
        for j, (enabled, space, ptr) in enumerate(accesses):  # the record: what every access finds
            if enabled:
                word, last = touch(space, ptr)
                rec[o + j, 0] = word
                rec[o + n_data + j, 0] = last
                # an earlier access of THIS instruction to the same location moves its timestamp (rd = rs1: the write finds the read's)
                for jj in range(j):
                    if accesses[jj][0] and accesses[jj][1:] == (space, ptr):
                        rec[o + n_data + j, 0] = ts + jj
        model, next_pc, step = oc.rv32_model(ins, rec, base)
        assert step == len(accesses) and len(model) == len(accesses)
        for j, ((enabled, space, ptr), (m_en, m_space, m_ptr, before, after)) in enumerate(zip(accesses, model)):
            assert enabled == m_en and (not enabled or (space == m_space and ptr == int(m_ptr[0]) and int(before[0]) & M32 == int(rec[o + j, 0])))
            if enabled:
                if j == len(accesses) - 1 and k not in (oc.KIND_BRANCH_EQ, oc.KIND_BRANCH_LT):  # every chip but the branches writes last
                    written.add((space, ptr))
                state[(space, ptr)] = (int(after[0]) & M32, ts + j)
        if next_pcs[i] is not None and int(next_pc[0]) != next_pcs[i] % oc.P:
            repair = None
            if op in (544, 545):  # an equality the block's path needs: give one still-initial operand the other's word
                ka, kb = (1, int(ins["a"])), (1, int(ins["b"]))
                for mine, other in ((ka, kb), (kb, ka)):
                    if mine not in written and mine != (1, 0) and mine not in forced:
                        repair = (mine, state[other][0])
                        break
            raise Rejected("path", repair)
        last_pc = int(next_pc[0])
    return rec[:, 0], initial, dict(state), last_pc


def execute_block(table, pcs, words_per_call, calls, seed=0, max_tries=4000):
    """`calls` calls of a block whose instructions sit at `pcs` -> (records u32[words_per_call, calls], per-call (initial, final, exit pc)).
    A call follows the block: after instruction i the pc is pcs[i + 1] (a taken branch inside a superblock, a not-taken one inside a
    basic block); initial states that leave this path or the ISA's domain are redrawn."""
    pool = sorted({int(ins["c"]) for ins in table if int(ins["c"]) < (1 << 24)} | {0, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF})
    next_pcs = [int(pcs[i + 1]) for i in range(len(pcs) - 1)] + [None]
    cols, info = [], []
    tries = 0
    while len(cols) < calls:
        tries += 1
        if tries > max_tries:
            raise RuntimeError(f"only {len(cols)} of {calls} calls follow the block after {max_tries} draws")
        forced = {}
        for _ in range(len(table) + 1):  # the same draw again with the repairs the rejected run asked for
            try:
                rec, initial, final, exit_pc = run_call(table, next_pcs, words_per_call, np.random.default_rng([seed, tries]), pool, forced)
            except Rejected as r:
                if r.repair is None:
                    break
                forced[r.repair[0]] = r.repair[1]
                continue
            cols.append(rec)
            info.append((initial, final, exit_pc))
            break
    return np.stack(cols, axis=1), info
