// "xbc": the accumulator-style expression code the evaluation kernels actually run.
//
// The reference encodes expressions as post-fix stack code (cuda/mod.rs:49-81, expr_eval.cuh:12-89):
// every leaf is a PUSH, every operator pops two stack slots. On a GPU the stack traffic and the
// per-opcode scalar work (fetch, decode, branch — the interpreter is scalar-unit bound: SQ_INSTS_SALU
// in profiles/r01_pmc_sq_counters_h18.tsv) dominate. At plan time the post-fix code is parsed back
// into a tree, simplified (constant folding, x+0, x*1, x*0, --x) and re-emitted with the deeper
// subtree first so that operators take a LEAF operand directly:
//     top = top (+|-|*) T[a]      top = T[a] - top      top = top (+|*) const      top = const - top
// Constants are converted to Montgomery form on the host. Field arithmetic is exact, so the value of
// every expression is unchanged; the stack (LDS, one column per lane) is only used when both operands
// of an operator are compound.
//
// One instruction = two u32 words {opcode, operand}.
#pragma once
#include "babybear.hpp"

#include <stdint.h>

namespace xbc {

enum Op : uint32_t {
    SET_COL = 0,     // top = T[a]                      (first instruction of an expression)
    SET_CONST = 1,   // top = m
    PUSH_COL = 2,    // push top; top = T[a]
    PUSH_CONST = 3,  // push top; top = m
    ADD_COL = 4,     // top = top + T[a]
    SUB_COL = 5,     // top = top - T[a]
    RSUB_COL = 6,    // top = T[a] - top
    MUL_COL = 7,     // top = top * T[a]
    ADD_CONST = 8,   // top = top + m
    RSUB_CONST = 9,  // top = m - top
    MUL_CONST = 10,  // top = top * m
    ADD = 11,        // top = pop + top
    SUB = 12,        // top = pop - top
    RSUB = 13,       // top = top - pop
    MUL = 14,        // top = pop * top
    NEG = 15,
    INV = 16,        // top = top == 0 ? 0 : top^-1
};

#if defined(__HIPCC__)
// NR rows per lane: every decoded instruction is applied to NR rows, so the scalar work of the interpreter (fetch,
// decode, dispatch — one scalar unit serves a CU's four SIMDs) is shared. Pays for long programs (the quotient's
// constraints); for the 1-3-instruction programs of the bus replay it does not (profiles/r01_pipeline_experiments.txt).
// `stk`: slot k of row n at stk[(k * NR + n) * STRIDE].
template <int STRIDE, bool COLUMN_OPERANDS, int NR>
__device__ __forceinline__ void eval_rows(const uint32_t* __restrict__ code, uint32_t n_instr, const uint32_t* __restrict__ trace,
                                          const size_t (&r)[NR], uint32_t* __restrict__ stk, size_t col_stride, uint32_t (&top)[NR]) {
#pragma unroll
    for (int n = 0; n < NR; ++n) top[n] = 0u;
    int sp = 0;
    const uint2* ins = reinterpret_cast<const uint2*>(code);
    for (uint32_t ip = 0; ip < n_instr; ++ip) {
        const uint2 in = ins[ip];
        const uint32_t op = in.x, a = in.y;
        const uint32_t* src = COLUMN_OPERANDS ? trace + (size_t)a * col_stride : trace + (size_t)a;
        if (op <= MUL_COL) {
            if (op <= PUSH_CONST) {
                if (op >= PUSH_COL) {
#pragma unroll
                    for (int n = 0; n < NR; ++n) stk[(sp * NR + n) * STRIDE] = top[n];
                    ++sp;
                }
                if (op & 1u) {
#pragma unroll
                    for (int n = 0; n < NR; ++n) top[n] = a;
                } else {
#pragma unroll
                    for (int n = 0; n < NR; ++n) top[n] = src[r[n]];
                }
            } else {
                uint32_t v[NR];
#pragma unroll
                for (int n = 0; n < NR; ++n) v[n] = src[r[n]];
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    if (op == ADD_COL) top[n] = bb::add(top[n], v[n]);
                    else if (op == SUB_COL) top[n] = bb::sub(top[n], v[n]);
                    else if (op == RSUB_COL) top[n] = bb::sub(v[n], top[n]);
                    else top[n] = bb::mul(top[n], v[n]);
                }
            }
        } else if (op <= MUL_CONST) {
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                if (op == ADD_CONST) top[n] = bb::add(top[n], a);
                else if (op == RSUB_CONST) top[n] = bb::sub(a, top[n]);
                else top[n] = bb::mul(top[n], a);
            }
        } else if (op <= MUL) {
            --sp;
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const uint32_t s_ = stk[(sp * NR + n) * STRIDE];
                if (op == ADD) top[n] = bb::add(s_, top[n]);
                else if (op == SUB) top[n] = bb::sub(s_, top[n]);
                else if (op == RSUB) top[n] = bb::sub(top[n], s_);
                else top[n] = bb::mul(s_, top[n]);
            }
        } else if (op == NEG) {
#pragma unroll
            for (int n = 0; n < NR; ++n) top[n] = bb::neg(top[n]);
        } else {
#pragma unroll
            for (int n = 0; n < NR; ++n) top[n] = bb::inv_or_zero(top[n]);
        }
    }
}

// `stk` = this thread's LDS column (slot k at stk[k * STRIDE]). COLUMN_OPERANDS: operand is a column
// index, T[a] = trace[a * col_stride + r]; otherwise an element offset, T[a] = trace[a + r].
template <int STRIDE, bool COLUMN_OPERANDS>
__device__ __forceinline__ uint32_t eval(const uint32_t* __restrict__ code, uint32_t n_instr,
                                         const uint32_t* __restrict__ trace, size_t r, uint32_t* __restrict__ stk,
                                         size_t col_stride = 1) {
    uint32_t top = 0u;
    int sp = 0;
    const uint2* ins = reinterpret_cast<const uint2*>(code);
    for (uint32_t ip = 0; ip < n_instr; ++ip) {
        const uint2 in = ins[ip];
        const uint32_t op = in.x, a = in.y;
        if (op <= MUL_COL) {
            if (op <= PUSH_CONST) {
                if (op >= PUSH_COL) { stk[sp * STRIDE] = top; ++sp; }
                if (op & 1u) top = a;
                else top = COLUMN_OPERANDS ? trace[(size_t)a * col_stride + r] : trace[(size_t)a + r];
            } else {
                const uint32_t v = COLUMN_OPERANDS ? trace[(size_t)a * col_stride + r] : trace[(size_t)a + r];
                if (op == ADD_COL) top = bb::add(top, v);
                else if (op == SUB_COL) top = bb::sub(top, v);
                else if (op == RSUB_COL) top = bb::sub(v, top);
                else top = bb::mul(top, v);
            }
        } else if (op <= MUL_CONST) {
            if (op == ADD_CONST) top = bb::add(top, a);
            else if (op == RSUB_CONST) top = bb::sub(a, top);
            else top = bb::mul(top, a);
        } else if (op <= MUL) {
            --sp;
            const uint32_t s = stk[sp * STRIDE];
            if (op == ADD) top = bb::add(s, top);
            else if (op == SUB) top = bb::sub(s, top);
            else if (op == RSUB) top = bb::sub(top, s);
            else top = bb::mul(s, top);
        } else if (op == NEG) {
            top = bb::neg(top);
        } else {
            top = bb::inv_or_zero(top);
        }
    }
    return top;
}
#endif

}  // namespace xbc
