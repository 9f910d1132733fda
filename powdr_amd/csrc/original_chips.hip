// SURVEY.md §8 row (f)-1, the PRODUCER half: the thirteen original RV32IM chips an autoprecompile is built from on the device, and
// the APC gather fused into them.
//
// The reference materialises, per original AIR, a full column-major dummy trace from the record arena
// (`chip.generate_proving_ctx(record_arena)`, /root/reference/openvm/src/powdr_extension/trace_generator/cuda/mod.rs:228-253)
// and then gathers the few percent of its cells an optimised APC keeps (apc_tracegen.cu:35-66): at BASELINE configs[1] 150 GB
// of dummy traces are written and re-read for 8.5 GB of APC trace. A chip computes every cell of a row from one small record
// anyway, so here the record -> row expansion runs INSIDE the gather:
//     powdr_apc_tracegen_records   records (16 KB per call) -> only the substituted cells, straight into the APC columns
//     powdr_original_airs_expand   the same expanders writing full dummy traces (the reference flow's producer; used to
//                                  check fused == gather-from-expanded and to mock-prove the chips' own constraints)
// The chips are EXTERNAL to the reference checkout; their columns, constraints and bus interactions are not
// (openvm-riscv/tests/openvm_constraints.txt: 1 BaseAlu, 95 LessThan, 194 Shift, 363 BranchEqual, 425 BranchLessThan, 511 JalLui,
// 564 Jalr, 633 LoadSignExtend, 719 LoadStore, 816 DivRem, 982 MulH, 1074 Multiplication, 1144 Auipc): every expander below fills
// all columns so that ALL of those constraints hold and every bus interaction is a legal one — checked on the device with
// pw_prover_check_constraints on the parsed text (tests/test_original_chips.py) and cell for cell against oracle/original_chips.py,
// whose rows are checked against the constraints, the lookup tables and a word-level RV32IM model.
// Record layout: include/powdr_gpu.h (PowdrOrigInstr); ours, since the reference's DenseRecordArena layouts are EXTERNAL.
#include "babybear.hpp"
#include "common.hpp"
#include "../../include/powdr_gpu.h"
#include "original_chips_tables.hpp"

#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

constexpr int kCalls = 128;      // calls per workgroup (= threads)
constexpr int kMaxWidth = 59;    // widest of the thirteen AIRs (DivRem)
constexpr int kInstrPerBlock = 16;
using orig::kKinds;
using orig::kWidths;
using orig::kOpcodeLo;
using orig::kOpcodeHi;

// Where an expander puts cell `c` of this lane's row (canonical value). Two sinks:
struct DenseSink {  // a full dummy trace: every cell, Montgomery form
    uint32_t* base;
    size_t stride;
    __device__ __forceinline__ void operator()(int c, uint32_t v) const { base[(size_t)c * stride] = bb::to_monty(v); }
};
struct SparseSink {  // the fused gather: only the cells the APC keeps, straight into their APC columns
    uint32_t* out_r;      // d_output + r
    size_t H;
    uint64_t wanted;      // bit c: column c of this instruction's AIR is substituted (wave-uniform: a scalar branch per cell)
    const uint32_t* dst;  // dst[c]: the APC column it goes to (wave-uniform: scalar loads)
    __device__ __forceinline__ void operator()(int c, uint32_t v) const {
        if ((wanted >> c) & 1ull) __builtin_nontemporal_store(bb::to_monty(v), out_r + (size_t)dst[c] * H);
    }
};

template <class Sink>
PW_HD void put_bytes(const Sink& o, int c, uint32_t w) {
    o(c, w & 0xffu); o(c + 1, (w >> 8) & 0xffu); o(c + 2, (w >> 16) & 0xffu); o(c + 3, w >> 24);
}
// prev_timestamp, and timestamp - prev - 1 split into 17 + 12 bits (the `timestamp_lt_aux` columns)
template <class Sink>
PW_HD void put_ts(const Sink& o, int c, uint32_t ts, uint32_t prev, bool enabled) {
    const uint32_t d = ts - prev - 1u;
    o(c, enabled ? prev : 0u); o(c + 1, enabled ? d & 0x1ffffu : 0u); o(c + 2, enabled ? d >> 17 : 0u);
}
PW_HD uint32_t field_inv(uint32_t canonical) { return bb::from_monty(bb::inv(bb::to_monty(canonical))); }
PW_HD uint32_t field_of(int32_t v) { return v < 0 ? bb::P - (uint32_t)(-v) : (uint32_t)v; }
PW_HD uint32_t byte_of(uint32_t w, uint32_t i) { return (w >> (8u * i)) & 0xffu; }

// `rec`: this call's record words of the instruction; `ts`: from_state.timestamp of the instruction in this call.
// Rv32BaseAluAdapter (BaseAlu, Shift, LessThan): columns 0..18
template <class Sink>
PW_HD void put_alu_adapter(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o, bool reg) {
    o(0, in.pc); o(1, ts); o(2, in.a); o(3, in.b); o(4, in.c); o(5, reg ? 1u : 0u);
    put_ts(o, 6, ts, rec[3], true);
    put_ts(o, 9, ts + 1u, rec[4], reg);
    put_ts(o, 12, ts + 2u, rec[5], true);
    put_bytes(o, 15, rec[2]);
}
// the most significant limb where x and y differ gets the marker, diff_val the positive difference there (LessThan cores)
template <class Sink>
PW_HD void put_diff_marker(const Sink& o, int c_marker, int c_val, const int32_t* x, const int32_t* y, bool lt) {
    bool done = false;
    uint32_t val = 0u;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        const bool pick = !done && x[i] != y[i];
        o(c_marker + i, pick ? 1u : 0u);
        if (pick) val = (uint32_t)(lt ? y[i] - x[i] : x[i] - y[i]);
        done = done || pick;
    }
    o(c_val, val);
}

template <class Sink>
PW_HD void expand_alu_shift_lt(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const bool reg = in.e != 0;
    const uint32_t bw = rec[0];
    // immediate: 24-bit value, its top byte repeated (openvm_constraints.txt:82-85)
    const uint32_t cw = reg ? rec[1] : ((in.c & 0xffffffu) | (((in.c >> 16) & 0xffu) << 24));
    put_alu_adapter(in, rec, ts, o, reg);
    if (in.kind == POWDR_ORIG_BASE_ALU) {
        const uint32_t op = in.opcode - 512u;
        const uint32_t aw = op == 0 ? bw + cw : op == 1 ? bw - cw : op == 2 ? bw ^ cw : op == 3 ? bw | cw : bw & cw;
        put_bytes(o, 19, aw); put_bytes(o, 23, bw); put_bytes(o, 27, cw);
#pragma unroll
        for (uint32_t j = 0; j < 5; ++j) o(31 + j, op == j ? 1u : 0u);
        return;
    }
    if (in.kind == POWDR_ORIG_LESS_THAN) {
        const bool sgn = in.opcode == 520u;
        const bool lt = sgn ? (int32_t)bw < (int32_t)cw : bw < cw;
        int32_t x[4], y[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) { x[i] = (int32_t)byte_of(bw, i); y[i] = (int32_t)byte_of(cw, i); }
        if (sgn) { x[3] = (int32_t)(int8_t)x[3]; y[3] = (int32_t)(int8_t)y[3]; }  // b_msb_f, c_msb_f: the top limb as a signed byte
        put_bytes(o, 19, bw); put_bytes(o, 23, cw);
        o(27, lt ? 1u : 0u); o(28, sgn ? 1u : 0u); o(29, sgn ? 0u : 1u); o(30, field_of(x[3])); o(31, field_of(y[3]));
        put_diff_marker(o, 32, 36, x, y, lt);
        return;
    }
    const uint32_t op = in.opcode - 517u;  // 0 SLL, 1 SRL, 2 SRA
    const uint32_t shift = cw & 31u, bit = shift & 7u, limb = shift >> 3;
    const uint32_t sign = op == 2 ? bw >> 31 : 0u;
    uint32_t aw;
    if (op == 0) aw = bw << shift;
    else aw = (bw >> shift) | ((sign && shift) ? 0xffffffffu << (32u - shift) : 0u);
    put_bytes(o, 19, aw); put_bytes(o, 23, bw); put_bytes(o, 27, cw);
    o(31, op == 0 ? 1u : 0u); o(32, op == 1 ? 1u : 0u); o(33, op == 2 ? 1u : 0u);
    o(34, op == 0 ? 1u << bit : 0u); o(35, op == 0 ? 0u : 1u << bit); o(36, sign);
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) o(37 + j, bit == j ? 1u : 0u);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) o(45 + j, limb == j ? 1u : 0u);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t byte = (bw >> (8 * j)) & 0xffu;
        o(49 + j, op == 0 ? byte >> (8u - bit) : byte & ((1u << bit) - 1u));
    }
}

// Rv32LoadStoreAdapter (LoadStore, LoadSignExtend): columns 0..22. The pointer rs1 + imm is taken modulo 2^29 and the access's
// alignment, rs1 adjusted to match (the record of a real execution satisfies both already). Returns the pointer.
template <class Sink>
PW_HD uint32_t put_load_store_adapter(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o, uint32_t align_mask) {
    const uint32_t imm = in.c & 0xffffu, imm_sign = in.g & 1u, ext = imm | (imm_sign ? 0xffff0000u : 0u);
    const uint32_t ptr = (rec[0] + ext) & 0x1fffffffu & ~align_mask;
    const uint32_t rs1 = ptr - ext;
    const uint32_t needs_write = in.f & 1u;
    o(0, in.pc); o(1, ts); o(2, in.b);
    put_bytes(o, 3, rs1);
    put_ts(o, 7, ts, rec[3], true);
    o(10, needs_write ? in.a : 0u);
    put_ts(o, 11, ts + 1u, rec[4], true);
    o(14, imm); o(15, imm_sign); o(16, ptr & 0xffffu); o(17, ptr >> 16); o(18, in.e);
    put_ts(o, 19, ts + 2u, rec[5], needs_write != 0);
    o(22, needs_write);
    return ptr;
}

template <class Sink>
PW_HD void expand_load_store(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const uint32_t op = in.opcode - 528u;  // 0 LOADW, 1 LOADBU, 2 LOADHU, 3 STOREW, 4 STOREH, 5 STOREB
    const uint32_t nbytes = (op == 0 || op == 3) ? 4u : (op == 2 || op == 4) ? 2u : 1u;
    const uint32_t s = put_load_store_adapter(in, rec, ts, o, nbytes - 1u) & 3u;
    const bool is_load = op < 3;
    // the opcode and the shift as flags in {0, 1, 2}^4 (openvm_constraints.txt:776: the opcode is a polynomial of them)
    uint32_t f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    if (op == 0) f0 = 2;
    else if (op == 2) { f1 = s == 0 ? 2 : 0; f2 = s == 2 ? 2 : 0; }
    else if (op == 1) { f3 = s == 0 ? 2 : 0; f0 = s == 1; f1 = s == 2; f2 = s == 3; }
    else if (op == 3) f3 = 1;
    else if (op == 4) { f0 = 1; f1 = s == 0; f2 = s == 2; }
    else { f0 = s == 0; f3 = s == 0 || s == 2 || s == 3; f1 = s == 1 || s == 2; f2 = s == 1 || s == 3; }
    o(23, f0); o(24, f1); o(25, f2); o(26, f3);
    o(27, 1u); o(28, is_load ? 1u : 0u);
    const uint32_t read = rec[1], prev = rec[2];
    const uint32_t mask = nbytes == 4 ? 0xffffffffu : ((1u << (8u * nbytes)) - 1u);
    // loads place the selected bytes at the bottom (zero extended), stores merge them into the overwritten word
    const uint32_t write = is_load ? (read >> (8u * s)) & mask : (prev & ~(mask << (8u * s))) | ((read & mask) << (8u * s));
    put_bytes(o, 29, read); put_bytes(o, 33, prev); put_bytes(o, 37, write);
}

template <class Sink>
PW_HD void expand_load_sign_extend(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const bool loadb = in.opcode == 534u;
    const uint32_t s = put_load_store_adapter(in, rec, ts, o, loadb ? 0u : 1u) & 3u;
    const uint32_t read = rec[1];
    const uint32_t rotated = (s & 2u) ? (read >> 16) | (read << 16) : read;  // shifted_read_data
    const uint32_t flag1 = loadb ? s & 1u : 0u;
    const uint32_t top = loadb ? byte_of(rotated, flag1) : byte_of(rotated, 1);
    o(23, loadb ? 1u - flag1 : 0u); o(24, flag1); o(25, loadb ? 0u : 1u); o(26, s >> 1); o(27, top >> 7);
    put_bytes(o, 28, rotated); put_bytes(o, 32, rec[2]);
}

// Rv32BranchAdapter (BranchEqual, BranchLessThan): columns 0..17
template <class Sink>
PW_HD void put_branch_adapter(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    o(0, in.pc); o(1, ts); o(2, in.a); o(3, in.b);
    put_ts(o, 4, ts, rec[2], true);
    put_ts(o, 7, ts + 1u, rec[3], true);
    put_bytes(o, 10, rec[0]); put_bytes(o, 14, rec[1]);
}

template <class Sink>
PW_HD void expand_branch_eq(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const uint32_t aw = rec[0], bw = rec[1];
    const bool beq = in.opcode == 544u, eq = aw == bw;
    put_branch_adapter(in, rec, ts, o);
    o(18, (eq == beq) ? 1u : 0u); o(19, in.c); o(20, beq ? 1u : 0u); o(21, beq ? 0u : 1u);
    // diff_inv_marker: the inverse of a_i - b_i at the first differing limb, zero elsewhere
    bool done = false;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t x = (aw >> (8 * j)) & 0xffu, y = (bw >> (8 * j)) & 0xffu;
        uint32_t m = 0u;
        if (!done && x != y) {
            m = field_inv(x > y ? x - y : bb::P - (y - x));
            done = true;
        }
        o(22 + j, m);
    }
}

template <class Sink>
PW_HD void expand_branch_lt(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const uint32_t op = in.opcode - 549u;  // 0 BLT, 1 BLTU, 2 BGE, 3 BGEU
    const uint32_t aw = rec[0], bw = rec[1];
    const bool sgn = (op & 1u) == 0, lt = sgn ? (int32_t)aw < (int32_t)bw : aw < bw;
    put_branch_adapter(in, rec, ts, o);
    int32_t x[4], y[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) { x[i] = (int32_t)byte_of(aw, i); y[i] = (int32_t)byte_of(bw, i); }
    if (sgn) { x[3] = (int32_t)(int8_t)x[3]; y[3] = (int32_t)(int8_t)y[3]; }
    o(18, (lt == (op < 2)) ? 1u : 0u); o(19, in.c);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) o(20 + j, op == j ? 1u : 0u);
    o(24, field_of(x[3])); o(25, field_of(y[3])); o(26, lt ? 1u : 0u);
    put_diff_marker(o, 27, 31, x, y, lt);
}

template <class Sink>
PW_HD void expand_jal_lui(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const bool is_jal = in.opcode == 560u;
    const uint32_t needs_write = in.f & 1u;
    const uint32_t rd = is_jal ? in.pc + 4u : in.c << 12;
    o(0, in.pc); o(1, ts); o(2, in.a);  // rd_ptr is the PC-lookup tuple's `a` whether or not rd is written
    put_ts(o, 3, ts, rec[1], needs_write != 0);
    put_bytes(o, 6, needs_write ? rec[0] : 0u);
    o(10, needs_write); o(11, in.c);
    put_bytes(o, 12, rd);
    o(16, is_jal ? 1u : 0u); o(17, is_jal ? 0u : 1u);
}

template <class Sink>
PW_HD void expand_jalr(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const uint32_t imm = in.c & 0xffffu, imm_sign = in.g & 1u, ext = imm | (imm_sign ? 0xffff0000u : 0u);
    const uint32_t to_pc = (rec[0] + ext) & 0x3fffffffu;  // jump targets lie below 2^30 (to_pc_limbs: 15 + 14 bits)
    const uint32_t needs_write = in.f & 1u, rd = in.pc + 4u;
    o(0, in.pc); o(1, ts); o(2, in.b);
    put_ts(o, 3, ts, rec[2], true);
    o(6, in.a);
    put_ts(o, 7, ts + 1u, rec[3], needs_write != 0);
    put_bytes(o, 10, needs_write ? rec[1] : 0u);
    o(14, needs_write); o(15, imm);
    put_bytes(o, 16, to_pc - ext);
    o(20, byte_of(rd, 1)); o(21, byte_of(rd, 2)); o(22, rd >> 24);
    o(23, 1u); o(24, to_pc & 1u); o(25, (to_pc & 0xffffu) >> 1); o(26, to_pc >> 16); o(27, imm_sign);
}

template <class Sink>
PW_HD void expand_auipc(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const uint32_t rd = in.pc + (in.c << 8);
    o(0, in.pc); o(1, ts); o(2, in.a);
    put_ts(o, 3, ts, rec[1], true);
    put_bytes(o, 6, rec[0]);
    o(10, 1u); o(11, byte_of(in.c, 0)); o(12, byte_of(in.c, 1)); o(13, byte_of(in.c, 2)); o(14, byte_of(in.pc, 1)); o(15, byte_of(in.pc, 2));
    put_bytes(o, 16, rd);
}

// Rv32MultAdapter (Multiplication, MulH, DivRem): columns 0..17
template <class Sink>
PW_HD void put_mult_adapter(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    o(0, in.pc); o(1, ts); o(2, in.a); o(3, in.b); o(4, in.c);
    put_ts(o, 5, ts, rec[3], true);
    put_ts(o, 8, ts + 1u, rec[4], true);
    put_ts(o, 11, ts + 2u, rec[5], true);
    put_bytes(o, 14, rec[2]);
}

template <class Sink>
PW_HD void expand_mul(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const uint32_t bw = rec[0], cw = rec[1];
    put_mult_adapter(in, rec, ts, o);
    if (in.kind == POWDR_ORIG_MUL) {
        put_bytes(o, 18, bw * cw); put_bytes(o, 22, bw); put_bytes(o, 26, cw);
        o(30, 1u);
        return;
    }
    const uint32_t op = in.opcode - 593u;  // 0 MULH (signed x signed), 1 MULHSU (signed x unsigned), 2 MULHU
    const bool b_neg = op < 2 && (bw >> 31), c_neg = op == 0 && (cw >> 31);
    const int64_t sb = b_neg ? (int64_t)(int32_t)bw : (int64_t)bw, sc = c_neg ? (int64_t)(int32_t)cw : (int64_t)cw;
    // |sb| <= 2^32, |sc| <= 2^32 and at most one of them reaches 2^32 - 1 unsigned while the other is signed: the product fits 64 bits
    // except for MULHU, which is an unsigned product
    const uint64_t prod = op == 2 ? (uint64_t)bw * (uint64_t)cw : (uint64_t)(sb * sc);
    put_bytes(o, 18, (uint32_t)(prod >> 32)); put_bytes(o, 22, bw); put_bytes(o, 26, cw); put_bytes(o, 30, (uint32_t)prod);
    o(34, b_neg ? 255u : 0u); o(35, c_neg ? 255u : 0u);
#pragma unroll
    for (uint32_t j = 0; j < 3; ++j) o(36 + j, op == j ? 1u : 0u);
}

template <class Sink>
PW_HD void expand_div_rem(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    const uint32_t op = in.opcode - 596u;  // 0 DIV, 1 DIVU, 2 REM, 3 REMU
    const bool sgn = (op & 1u) == 0;
    const uint32_t bw = rec[0], cw = rec[1];
    put_mult_adapter(in, rec, ts, o);
    uint32_t q, r;
    if (cw == 0u) { q = 0xffffffffu; r = bw; }                                        // division by zero
    else if (sgn && bw == 0x80000000u && cw == 0xffffffffu) { q = bw; r = 0u; }      // the one signed overflow
    else if (sgn) { q = (uint32_t)((int32_t)bw / (int32_t)cw); r = (uint32_t)((int32_t)bw % (int32_t)cw); }
    else { q = bw / cw; r = bw % cw; }
    const uint32_t zero_div = cw == 0u, r_zero = (r == 0u && cw != 0u);
    const uint32_t b_sign = sgn ? bw >> 31 : 0u, c_sign = sgn ? cw >> 31 : 0u, sign_xor = b_sign ^ c_sign;
    // q_sign: what the quotient is sign-extended with in the carry chain: sign_xor when q != 0, else 0 (openvm_constraints.txt:942-943);
    // free when the divisor is zero, where q = -1 reads as negative exactly for the signed opcodes
    const uint32_t q_sign = zero_div ? (sgn ? 1u : 0u) : (q != 0u ? sign_xor : 0u);
    const uint32_t rp = sign_xor ? 0u - r : r;  // r': the remainder brought to the divisor's sign
    put_bytes(o, 18, bw); put_bytes(o, 22, cw); put_bytes(o, 26, q); put_bytes(o, 30, r);
    o(34, zero_div); o(35, r_zero); o(36, b_sign); o(37, c_sign); o(38, q_sign); o(39, sign_xor);
    o(40, field_inv(byte_of(cw, 0) + byte_of(cw, 1) + byte_of(cw, 2) + byte_of(cw, 3)));
    o(41, field_inv(byte_of(r, 0) + byte_of(r, 1) + byte_of(r, 2) + byte_of(r, 3)));
    put_bytes(o, 42, rp);
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) o(46 + i, field_inv(bb::P - (256u - byte_of(rp, i))));
    // |r| < |c|: the most significant limb where r' and c differ, and the positive difference there under c's sign
    bool done = zero_div || r_zero;
    uint32_t lt_diff = 0u;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        const uint32_t x = byte_of(rp, i), y = byte_of(cw, i);
        const bool pick = !done && x != y;
        o(50 + i, pick ? 1u : 0u);
        if (pick) lt_diff = c_sign ? x - y : y - x;
        done = done || pick;
    }
    o(54, lt_diff);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) o(55 + j, op == j ? 1u : 0u);
}

PW_HD int record_words(uint32_t kind) {
    return (kind == POWDR_ORIG_JAL_LUI || kind == POWDR_ORIG_AUIPC) ? 2
         : (kind == POWDR_ORIG_BRANCH_EQ || kind == POWDR_ORIG_BRANCH_LT || kind == POWDR_ORIG_JALR) ? 4 : 6;
}

template <class Sink>
PW_HD void expand(const PowdrOrigInstr& in, const uint32_t* rec, uint32_t ts, const Sink& o) {
    switch (in.kind) {
        case POWDR_ORIG_BASE_ALU: case POWDR_ORIG_SHIFT: case POWDR_ORIG_LESS_THAN: expand_alu_shift_lt(in, rec, ts, o); break;
        case POWDR_ORIG_LOAD_STORE: expand_load_store(in, rec, ts, o); break;
        case POWDR_ORIG_BRANCH_EQ: expand_branch_eq(in, rec, ts, o); break;
        case POWDR_ORIG_JAL_LUI: expand_jal_lui(in, rec, ts, o); break;
        case POWDR_ORIG_BRANCH_LT: expand_branch_lt(in, rec, ts, o); break;
        case POWDR_ORIG_JALR: expand_jalr(in, rec, ts, o); break;
        case POWDR_ORIG_LOAD_SIGN_EXTEND: expand_load_sign_extend(in, rec, ts, o); break;
        case POWDR_ORIG_DIV_REM: expand_div_rem(in, rec, ts, o); break;
        case POWDR_ORIG_MUL_H: case POWDR_ORIG_MUL: expand_mul(in, rec, ts, o); break;
        default: expand_auipc(in, rec, ts, o); break;
    }
}

struct HostSink {  // the test hook below: canonical values into a plain row
    uint32_t* row;
    void operator()(int c, uint32_t v) const { row[c] = v; }
};

struct AirSlots { uint32_t* buffer[kKinds]; uint32_t height[kKinds]; uint32_t row_block[kKinds]; };

// full dummy traces: lane = call, blockIdx.y = instruction; cell (c) of the row at buffer[c * height + air_row + call * row_block]
__global__ __launch_bounds__(kCalls) void original_airs_expand_kernel(const uint32_t* __restrict__ records, size_t num_calls,
                                                                       const PowdrOrigInstr* __restrict__ instrs, AirSlots airs) {
    const size_t r = (size_t)blockIdx.x * kCalls + threadIdx.x;
    if (r >= num_calls) return;
    const PowdrOrigInstr in = instrs[blockIdx.y];
    uint32_t rec[6];
    const int n = record_words(in.kind);
#pragma unroll
    for (int w = 0; w < 6; ++w) rec[w] = w < n ? records[(size_t)(in.rec_off + w) * num_calls + r] : 0u;
    const uint32_t ts = records[r] + in.ts_delta;
    const size_t h = airs.height[in.kind];
    const DenseSink o{airs.buffer[in.kind] + in.air_row + r * airs.row_block[in.kind], h};
    expand(in, rec, ts, o);
}

constexpr int kDstStride = 64;  // dst table entries per instruction (>= kMaxWidth)

// fused: every workgroup takes kCalls calls and a run of instructions; a lane expands the row of its call and the cells the APC
// keeps leave straight for their APC columns (512-byte segments per column and workgroup); nothing else is stored anywhere
__global__ __launch_bounds__(kCalls) void apc_tracegen_records_kernel(uint32_t* __restrict__ out, size_t H, const uint32_t* __restrict__ records,
                                                                       size_t num_calls, const PowdrOrigInstr* __restrict__ instrs,
                                                                       const uint64_t* __restrict__ wanted, const uint32_t* __restrict__ dst,
                                                                       uint32_t n_instrs) {
    const size_t r = (size_t)blockIdx.x * kCalls + threadIdx.x;
    const bool live = r < num_calls;
    const bool in_trace = r < H;
    const uint32_t base_ts = live ? records[r] : 0u;
    const uint32_t i0 = blockIdx.y * kInstrPerBlock;
    const uint32_t i1 = min(n_instrs, i0 + kInstrPerBlock);
    for (uint32_t i = i0; i < i1; ++i) {
        const uint64_t w = wanted[i];
        if (w == 0) continue;
        const SparseSink o{out + r, H, w, dst + (size_t)i * kDstStride};
        if (live) {
            const PowdrOrigInstr in = instrs[i];
            uint32_t rec[6];
            const int n = record_words(in.kind);
#pragma unroll
            for (int k = 0; k < 6; ++k) rec[k] = k < n ? __builtin_nontemporal_load(records + (size_t)(in.rec_off + k) * num_calls + r) : 0u;
            expand(in, rec, base_ts + in.ts_delta, o);
        } else if (in_trace) {  // padding rows of the trace: zero
            for (uint64_t m = w; m; m &= m - 1) __builtin_nontemporal_store(0u, o.out_r + (size_t)o.dst[__builtin_ctzll(m)] * H);
        }
    }
}

// device copies of an instruction table (+ substitutions), cached by content
struct RecordPlan {
    PowdrOrigInstr* d_instrs = nullptr;
    uint64_t* d_wanted = nullptr;
    uint32_t* d_dst = nullptr;
    std::vector<PowdrOrigInstr> key_instrs;
    std::vector<PowdrRecordSubst> key_subs;
    int device = 0;
    uint64_t last_use = 0;
    ~RecordPlan() {
        for (void* q : {(void*)d_instrs, (void*)d_wanted, (void*)d_dst}) if (q) (void)hipFree(q);
    }
};
std::mutex g_mu;
std::unordered_map<uint64_t, std::shared_ptr<RecordPlan>> g_plans;
uint64_t g_clock = 0;
constexpr size_t kMaxPlans = 32;

uint64_t fnv(const void* p, size_t n, uint64_t h) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}

int check_instrs(const PowdrOrigInstr* h_instrs, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const PowdrOrigInstr& in = h_instrs[i];
        if (in.kind >= (uint32_t)kKinds || in.rec_off == 0) return (int)hipErrorInvalidValue;
        if (in.opcode < kOpcodeLo[in.kind] || in.opcode > kOpcodeHi[in.kind]) return (int)hipErrorInvalidValue;
    }
    return 0;
}

int get_plan(const PowdrOrigInstr* h_instrs, size_t n_instrs, const PowdrRecordSubst* h_subs, size_t n_subs, std::shared_ptr<RecordPlan>& out) {
    int device = 0;
    PW_HIP_TRY(hipGetDevice(&device));
    uint64_t key = fnv(h_instrs, n_instrs * sizeof(PowdrOrigInstr), 1469598103934665603ull);
    key = fnv(h_subs, n_subs * sizeof(PowdrRecordSubst), key);
    key = fnv(&device, sizeof device, key);
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) {
        const RecordPlan& c = *it->second;
        if (c.device != device || c.key_instrs.size() != n_instrs || c.key_subs.size() != n_subs ||
            memcmp(c.key_instrs.data(), h_instrs, n_instrs * sizeof(PowdrOrigInstr)) != 0 ||
            (n_subs && memcmp(c.key_subs.data(), h_subs, n_subs * sizeof(PowdrRecordSubst)) != 0)) { g_plans.erase(it); it = g_plans.end(); }
    }
    if (it == g_plans.end()) {
        if (g_plans.size() >= kMaxPlans) {
            auto victim = g_plans.begin();
            for (auto j = g_plans.begin(); j != g_plans.end(); ++j) if (j->second->last_use < victim->second->last_use) victim = j;
            g_plans.erase(victim);
        }
        auto p = std::make_shared<RecordPlan>();
        // per instruction: which columns leave (bit mask) and where to (dst[instr * 64 + col]). Duplicate destinations resolve like the
        // sequential reference loop: the last wins; one source cell has one destination (the reference's Subst maps an original column to
        // at most one APC column, cuda/mod.rs:272-328)
        std::unordered_map<int32_t, size_t> last;
        for (size_t i = 0; i < n_subs; ++i) last[h_subs[i].apc_col] = i;
        std::vector<uint64_t> wanted(n_instrs + 1, 0);
        std::vector<uint32_t> dst((n_instrs + 1) * (size_t)kDstStride, 0);
        for (size_t i = 0; i < n_subs; ++i) {
            if (last[h_subs[i].apc_col] != i) continue;
            const size_t ins = (size_t)h_subs[i].instr;
            const uint64_t bit = 1ull << h_subs[i].col;
            if (wanted[ins] & bit) return (int)hipErrorInvalidValue;  // the same cell to two APC columns
            wanted[ins] |= bit;
            dst[ins * kDstStride + (size_t)h_subs[i].col] = (uint32_t)h_subs[i].apc_col;
        }
        PW_HIP_TRY(hipMalloc(&p->d_instrs, (n_instrs + 1) * sizeof(PowdrOrigInstr)));
        PW_HIP_TRY(hipMalloc(&p->d_wanted, wanted.size() * 8));
        PW_HIP_TRY(hipMalloc(&p->d_dst, dst.size() * 4));
        if (n_instrs) PW_HIP_TRY(hipMemcpy(p->d_instrs, h_instrs, n_instrs * sizeof(PowdrOrigInstr), hipMemcpyHostToDevice));
        PW_HIP_TRY(hipMemcpy(p->d_wanted, wanted.data(), wanted.size() * 8, hipMemcpyHostToDevice));
        PW_HIP_TRY(hipMemcpy(p->d_dst, dst.data(), dst.size() * 4, hipMemcpyHostToDevice));
        p->key_instrs.assign(h_instrs, h_instrs + n_instrs);
        p->key_subs.assign(h_subs, h_subs + n_subs);
        p->device = device;
        it = g_plans.emplace(key, std::move(p)).first;
    }
    out = it->second;
    out->last_use = ++g_clock;
    return 0;
}

}  // namespace

extern "C" int powdr_original_airs_expand(const uint32_t* d_records, size_t num_calls, const PowdrOrigInstr* h_instrs, size_t n_instrs,
                                          const OriginalAir* h_airs) {
    (void)hipGetLastError();
    if (!num_calls || !n_instrs) return 0;
    if (!d_records || !h_instrs || !h_airs) return (int)hipErrorInvalidValue;
    if (int rc = check_instrs(h_instrs, n_instrs)) return rc;
    AirSlots airs{};
    for (int k = 0; k < kKinds; ++k) {
        airs.buffer[k] = const_cast<uint32_t*>(h_airs[k].buffer);
        airs.height[k] = (uint32_t)h_airs[k].height;
        airs.row_block[k] = (uint32_t)h_airs[k].row_block_size;
    }
    for (size_t i = 0; i < n_instrs; ++i) {
        const PowdrOrigInstr& in = h_instrs[i];
        const OriginalAir& a = h_airs[in.kind];
        if (!a.buffer || a.width != kWidths[in.kind] || in.air_row >= (uint32_t)a.row_block_size ||
            (size_t)a.row_block_size * num_calls > (size_t)a.height)
            return (int)hipErrorInvalidValue;
    }
    std::shared_ptr<RecordPlan> plan;
    if (int rc = get_plan(h_instrs, n_instrs, nullptr, 0, plan)) return rc;
    pw::ScopedKernelTimer t("original_airs_expand_kernel");
    for (size_t i0 = 0; i0 < n_instrs; i0 += 65535) {
        const unsigned cnt = (unsigned)std::min<size_t>(65535, n_instrs - i0);
        hipLaunchKernelGGL(original_airs_expand_kernel, dim3(pw::div_up(num_calls, kCalls), cnt), dim3(kCalls), 0, pw::stream(), d_records, num_calls,
                           plan->d_instrs + i0, airs);
    }
    return (int)hipGetLastError();
}

extern "C" int powdr_apc_tracegen_records(PowdrFp* d_output, size_t output_height, const uint32_t* d_records, size_t num_apc_calls,
                                          const PowdrOrigInstr* h_instrs, size_t n_instrs, const PowdrRecordSubst* h_subs, size_t n_subs) {
    (void)hipGetLastError();
    const size_t H = output_height;
    if ((H & (H - 1)) != 0) return (int)hipErrorInvalidValue;
    if (H == 0 || n_subs == 0) return (int)hipGetLastError();
    if (!d_output || !h_instrs || !h_subs || (num_apc_calls && !d_records)) return (int)hipErrorInvalidValue;
    // the records are word-major with stride num_apc_calls: clamping the count would read them with the wrong stride
    if (num_apc_calls > H) return (int)hipErrorInvalidValue;
    if (int rc = check_instrs(h_instrs, n_instrs)) return rc;
    for (size_t i = 0; i < n_subs; ++i)
        if (h_subs[i].instr < 0 || (size_t)h_subs[i].instr >= n_instrs || h_subs[i].col < 0 || h_subs[i].col >= kWidths[h_instrs[h_subs[i].instr].kind] ||
            h_subs[i].apc_col < 0)
            return (int)hipErrorInvalidValue;
    std::shared_ptr<RecordPlan> plan;
    if (int rc = get_plan(h_instrs, n_instrs, h_subs, n_subs, plan)) return rc;
    pw::ScopedKernelTimer t("apc_tracegen_records_kernel");
    static_assert(kMaxWidth <= kDstStride && kDstStride <= 64, "one mask bit and one table entry per column");
    hipLaunchKernelGGL(apc_tracegen_records_kernel, dim3(pw::div_up(H, kCalls), pw::div_up(n_instrs, kInstrPerBlock)), dim3(kCalls), 0, pw::stream(),
                       d_output, H, d_records, num_apc_calls, plan->d_instrs, plan->d_wanted, plan->d_dst, (uint32_t)n_instrs);
    return (int)hipGetLastError();
}

// Test hook: the row ONE instruction produces from its record words, computed on the HOST by the very expander code the kernels run
// (the expanders are host-device templates over their sink). row_out receives the canonical cells; returns the AIR's width or -1.
extern "C" int powdr_original_row_expand_host(const PowdrOrigInstr* instr, const uint32_t* record_words6, uint32_t timestamp, uint32_t* row_out) {
    if (!instr || !record_words6 || !row_out || check_instrs(instr, 1)) return -1;
    uint32_t rec[6];
    const int n = record_words(instr->kind);
    for (int w = 0; w < 6; ++w) rec[w] = w < n ? record_words6[w] : 0u;
    for (int c = 0; c < kWidths[instr->kind]; ++c) row_out[c] = 0u;
    expand(*instr, rec, timestamp, HostSink{row_out});
    return kWidths[instr->kind];
}
