// Post-fix expression evaluator for APC rows on gfx950.
//
// Same instruction set and semantics as the reference's device evaluator
// (/root/reference/openvm/cuda/src/expr_eval.cuh:12-89; encoder at
// openvm/src/powdr_extension/trace_generator/cuda/mod.rs:49-81):
//   PUSH_APC off   push trace[off + r]        (off = apc_col * H, Montgomery)
//   PUSH_CONST u   push Fp(u)                 (u canonical < p)
//   ADD SUB MUL NEG INV_OR_ZERO
// with a 16-entry stack (expr_eval.cuh:22).
//
// Design for CDNA4: every lane of a wave evaluates the SAME bytecode on a
// different row, so the instruction stream is wave-uniform: opcodes/operands
// come through the scalar cache (s_load) and the dispatch is scalar branching;
// only the field arithmetic and the row loads are vector work. The reference's
// runtime-indexed `Fp stack[16]` would live in scratch memory on AMD hardware
// (dynamic VGPR indexing); here the stack is a per-thread column of LDS
// (stack[slot * blockDim + tid], one bank per lane, conflict-free), and the top
// of stack is cached in a register so unary ops and the second operand of
// binary ops never touch LDS.
//
// Malformed bytecode (overflow/underflow/unknown opcode) traps in the reference
// through assert(); here the evaluator clamps the stack pointer so that it
// never leaves its LDS column and returns an unspecified value.
#pragma once
#include "babybear.hpp"
#include "../../include/powdr_gpu.h"

namespace pw {

constexpr int kStackCap = POWDR_EXPR_STACK_CAPACITY;

// `stk` points at this thread's column: slot k lives at stk[k * stride].
// Returns the value in Montgomery form.
// COLUMN_OPERANDS = false: PUSH_APC operand is an element offset (reference encoding);
//                   true:  operand is a column index, cell = trace[operand * col_stride + r]
//                          (used by the quotient kernel, whose matrices exceed 2^32 elements).
template <int STRIDE, bool COLUMN_OPERANDS = false>
__device__ __forceinline__ uint32_t eval_expr(const uint32_t* __restrict__ bc, uint32_t len,
                                              const uint32_t* __restrict__ trace, size_t r,
                                              uint32_t* __restrict__ stk, size_t col_stride = 1) {
    uint32_t top = 0;  // cached top of stack (valid when sp > 0)
    int sp = 0;        // number of live entries, including `top`
    for (uint32_t ip = 0; ip < len;) {
        const uint32_t op = bc[ip++];
        if (op <= POWDR_OP_PUSH_CONST) {
            const uint32_t operand = bc[ip++];
            if (sp > 0 && sp < kStackCap) stk[(sp - 1) * STRIDE] = top;
            sp = sp < kStackCap ? sp + 1 : sp;
            if (op == POWDR_OP_PUSH_APC)
                top = COLUMN_OPERANDS ? trace[(size_t)operand * col_stride + r] : trace[(size_t)operand + r];
            else
                top = bb::to_monty(operand);
        } else if (op <= POWDR_OP_MUL) {
            // binary: a = second from top, b = top
            sp = sp > 1 ? sp - 1 : sp;
            const uint32_t a = stk[(sp > 0 ? sp - 1 : 0) * STRIDE];  // sp == 0 (malformed code): slot 0, never below the column
            if (op == POWDR_OP_ADD) top = bb::add(a, top);
            else if (op == POWDR_OP_SUB) top = bb::sub(a, top);
            else top = bb::mul(a, top);
        } else if (op == POWDR_OP_NEG) {
            top = bb::neg(top);
        } else {  // POWDR_OP_INV_OR_ZERO
            top = bb::inv_or_zero(top);
        }
    }
    return top;
}

}  // namespace pw
