/*
 * TEST INFRASTRUCTURE — CPU oracle for APC trace generation. Not part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link or call this file.
 *
 * Two restatements of the same computation, on canonical BabyBear values:
 *
 *  (A) the GPU convention (column-major output, `Subst`/bytecode tables), each
 *      function following one reference kernel line by line in meaning:
 *        or_apc_tracegen        <- openvm/cuda/src/apc_tracegen.cu:35-66
 *        or_apc_apply_derived   <- openvm/cuda/src/apc_tracegen.cu:72-100
 *        or_apc_apply_bus       <- openvm/cuda/src/apc_apply_bus.cu:23-113
 *        or_eval_expr           <- openvm/cuda/src/expr_eval.cuh:36-89
 *
 *  (B) the CPU convention (row-major output, dummy rows addressed through
 *      `generate_trace`'s per-instruction offsets):
 *        or_generate_witness    <- openvm/src/powdr_extension/trace_generator/cpu/mod.rs:156-228
 *                                  autoprecompiles/src/trace_handler.rs:68-124
 *                                  openvm/src/powdr_extension/trace_generator/cpu/periphery.rs:176-237
 *
 * tests/ check (A) == transpose(B) on identical inputs, and the HIP library
 * against (A) through the C ABI.
 *
 * PARITY STATUS: the reference holds no golden vectors for cell values or
 * histogram contents on this path; the oracle is pinned only to
 * the structural pins the reference tests do hold (column/bus/constraint
 * counts, modulus, serde format) — see oracle/README.md.
 */
#include "babybear.h"
#include <stddef.h>
#include <stdint.h>
#include <string.h>

enum { OP_PUSH_APC = 0, OP_PUSH_CONST = 1, OP_ADD = 2, OP_SUB = 3, OP_MUL = 4, OP_NEG = 5, OP_INV_OR_ZERO = 6 };
#define OR_STACK_CAP 16

/* expr_eval.cuh:36-89. `trace[operand + r]` is the APC cell. Returns -1 on a
 * malformed program (the reference asserts), 0 otherwise. */
int or_eval_expr(const uint32_t* bc, uint32_t len, const uint32_t* trace, size_t r, uint32_t* result) {
    uint32_t st[OR_STACK_CAP];
    int sp = 0;
    uint32_t ip = 0;
    while (ip < len) {
        uint32_t op = bc[ip++];
        switch (op) {
            case OP_PUSH_APC:
                if (sp >= OR_STACK_CAP || ip >= len) return -1;
                st[sp++] = trace[(size_t)bc[ip++] + r];
                break;
            case OP_PUSH_CONST:
                if (sp >= OR_STACK_CAP || ip >= len) return -1;
                st[sp++] = bc[ip++] % OR_P;
                break;
            case OP_ADD: case OP_SUB: case OP_MUL: {
                if (sp < 2) return -1;
                uint32_t b = st[--sp], a = st[--sp];
                st[sp++] = op == OP_ADD ? or_add(a, b) : op == OP_SUB ? or_sub(a, b) : or_mul(a, b);
                break;
            }
            case OP_NEG:
                if (sp < 1) return -1;
                st[sp - 1] = or_neg(st[sp - 1]);
                break;
            case OP_INV_OR_ZERO:
                if (sp < 1) return -1;
                st[sp - 1] = st[sp - 1] == 0 ? 0 : or_inv(st[sp - 1]);
                break;
            default:
                return -1;
        }
    }
    if (sp != 1) return -1;
    *result = st[0];
    return 0;
}

/* ------------------------------------------------------------------ (A) GPU convention */

/* apc_tracegen.cu:35-66. airs_buf[k] = host pointer to the column-major dummy
 * trace of AIR k; subs = n_subs x {air_index, col, row, apc_col}. */
void or_apc_tracegen(uint32_t* out, size_t H, const uint32_t* const* airs_buf, const int32_t* airs_height,
                     const int32_t* airs_row_block, const int32_t* subs, size_t n_subs, int num_calls) {
    for (size_t r = 0; r < H; ++r) {
        int in_range = r < (size_t)(num_calls < 0 ? 0 : num_calls);
        for (size_t i = 0; i < n_subs; ++i) {
            const int32_t* s = subs + 4 * i;
            size_t dst = (size_t)s[3] * H + r;
            if (!in_range) { out[dst] = 0; continue; }
            size_t a = (size_t)s[0];
            size_t src = (size_t)s[1] * (size_t)airs_height[a] + (size_t)s[2] + r * (size_t)airs_row_block[a];
            out[dst] = airs_buf[a][src];
        }
    }
}

/* Extension (include/powdr_gpu.h powdr_apc_tracegen_callmajor): the same gather from call-major compacted sources,
 * airs_buf[a][r * cells_per_call[a] + slot]; subs = n x {air, slot, apc_col}; sequential like the reference loop
 * (a later substitution to the same column overwrites an earlier one). */
void or_apc_tracegen_callmajor(uint32_t* out, size_t H, const uint32_t* const* airs_buf, const int32_t* cells_per_call,
                               const int32_t* subs, size_t n_subs, int num_calls) {
    for (size_t r = 0; r < H; ++r) {
        int in_range = r < (size_t)(num_calls < 0 ? 0 : num_calls);
        for (size_t i = 0; i < n_subs; ++i) {
            const int32_t* s = subs + 3 * i;
            size_t a = (size_t)s[0];
            out[(size_t)s[2] * H + r] = in_range ? airs_buf[a][r * (size_t)cells_per_call[a] + (size_t)s[1]] : 0;
        }
    }
}

/* apc_tracegen.cu:72-100. specs = n_cols x {col_base(u64 as 2 x u32 lo,hi), off, len}. */
int or_apc_apply_derived(uint32_t* out, size_t H, int num_calls, const uint64_t* col_base,
                         const uint32_t* span_off, const uint32_t* span_len, size_t n_cols,
                         const uint32_t* bytecode) {
    for (size_t r = 0; r < H; ++r) {
        if (r < (size_t)(num_calls < 0 ? 0 : num_calls)) {
            for (size_t i = 0; i < n_cols; ++i) {
                uint32_t v;
                if (or_eval_expr(bytecode + span_off[i], span_len[i], out, r, &v)) return -1;
                out[col_base[i] + r] = v;
            }
        } else {
            for (size_t i = 0; i < n_cols; ++i) out[col_base[i] + r] = 0;
        }
    }
    return 0;
}

/* apc_apply_bus.cu:23-113. interactions = n x {bus_id, num_args, args_index_off};
 * spans = {off, len} pairs. Histogram arithmetic is the reference's 32-bit
 * arithmetic; indices outside a table are dropped (the reference would write
 * out of bounds / assert); shift counts >= 32 give 0 as in the CUDA build.
 * bitwise table layout: [2^16 range | 2^16 xor], index x*256+y. */
int or_apc_apply_bus(const uint32_t* trace, int num_calls, const uint32_t* bytecode,
                     const uint32_t* interactions, size_t n_interactions, const uint32_t* spans,
                     uint32_t var_bus, uint32_t* var_hist, size_t var_bins, uint32_t tuple_bus,
                     uint32_t* tuple_hist, uint32_t sz0, uint32_t sz1, uint32_t bitwise_bus,
                     uint32_t* bitwise_hist) {
    for (int r = 0; r < num_calls; ++r) {
        for (size_t i = 0; i < n_interactions; ++i) {
            uint32_t bus = interactions[3 * i], off = interactions[3 * i + 2];
            const uint32_t* sp = spans + 2 * (size_t)off;
            uint32_t m;
            if (or_eval_expr(bytecode + sp[0], sp[1], trace, (size_t)r, &m)) return -1;
            if (m == 0) continue;
            if (bus == var_bus) {
                uint32_t v, bits;
                if (or_eval_expr(bytecode + sp[2], sp[3], trace, (size_t)r, &v)) return -1;
                if (or_eval_expr(bytecode + sp[4], sp[5], trace, (size_t)r, &bits)) return -1;
                uint32_t idx = (bits < 32 ? (1u << bits) : 0u) + v - 1u;
                if (idx < var_bins) var_hist[idx] += m;
            } else if (bus == tuple_bus) {
                uint32_t v0, v1;
                if (or_eval_expr(bytecode + sp[2], sp[3], trace, (size_t)r, &v0)) return -1;
                if (or_eval_expr(bytecode + sp[4], sp[5], trace, (size_t)r, &v1)) return -1;
                uint32_t idx = v0 * sz1 + v1;
                if (idx < sz0 * sz1) tuple_hist[idx] += m;
            } else if (bus == bitwise_bus) {
                uint32_t x, y, sel;
                if (or_eval_expr(bytecode + sp[2], sp[3], trace, (size_t)r, &x)) return -1;
                if (or_eval_expr(bytecode + sp[4], sp[5], trace, (size_t)r, &y)) return -1;
                if (or_eval_expr(bytecode + sp[8], sp[9], trace, (size_t)r, &sel)) return -1;
                if (sel <= 1 && x < 256 && y < 256) bitwise_hist[sel * 65536u + x * 256u + y] += m;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ (B) CPU convention */

/*
 * cpu/mod.rs:156-228 + trace_handler.rs:68-124 + cpu/periphery.rs:176-237.
 *
 * Inputs (all host, canonical):
 *   dummy[k], dummy_width[k]      row-major dummy trace of AIR k
 *   n_instr                       instructions with non-empty subs (trace_handler.rs:53-57)
 *   instr_air[j]                  AIR of instruction j
 *   instr_table_offset[j]         #earlier instructions on the same AIR (trace_handler.rs:77-88)
 *   air_occurrences[k]            instructions per call on AIR k (trace_handler.rs:68)
 *   sub_begin[j..j+1], sub_pairs  (dummy_trace_index, apc_index) pairs (trace_handler.rs:90-102)
 *   derived: apc column, kind (0 Constant, 1 QuotientOrZero), constant, bytecode spans of
 *            e1 (numerator) and e2 (denominator); operands of PUSH_APC are APC column indices
 *   buses:   per interaction bus id + spans [mult, args...]; same operand convention
 * Output: values[height*width] row-major (zero for rows >= num_calls), histograms.
 * Returns 0, -1 malformed bytecode, -2 unknown bus id (reference: unreachable!,
 * cpu/periphery.rs:232-234).
 */
int or_generate_witness(uint32_t* values, size_t height, size_t width, size_t num_calls,
                        const uint32_t* const* dummy, const int32_t* dummy_width, size_t n_instr,
                        const int32_t* instr_air, const int32_t* instr_table_offset,
                        const int32_t* air_occurrences, const uint32_t* sub_begin,
                        const uint32_t* sub_pairs, size_t n_derived, const uint32_t* derived_col,
                        const uint32_t* derived_kind, const uint32_t* derived_const,
                        const uint32_t* derived_spans /* e1.off,e1.len,e2.off,e2.len */,
                        const uint32_t* derived_bc, size_t n_inter, const uint32_t* inter_bus,
                        const uint32_t* inter_nargs, const uint32_t* inter_span_off,
                        const uint32_t* bus_spans, const uint32_t* bus_bc, uint32_t var_bus,
                        uint32_t* var_hist, size_t var_bins, int has_tuple, uint32_t tuple_bus,
                        uint32_t* tuple_hist, uint32_t sz0, uint32_t sz1, int has_bitwise,
                        uint32_t bitwise_bus, uint32_t* bitwise_hist) {
    memset(values, 0, height * width * sizeof(uint32_t));
    for (size_t row = 0; row < num_calls && row < height; ++row) {
        uint32_t* row_slice = values + row * width;
        /* copy substituted cells (cpu/mod.rs:170-178) */
        for (size_t j = 0; j < n_instr; ++j) {
            int a = instr_air[j];
            size_t w = (size_t)dummy_width[a];
            size_t start = (row * (size_t)air_occurrences[a] + (size_t)instr_table_offset[j]) * w; /* trace_handler.rs:115 */
            const uint32_t* dummy_row = dummy[a] + start;
            for (uint32_t s = sub_begin[j]; s < sub_begin[j + 1]; ++s)
                row_slice[sub_pairs[2 * s + 1]] = dummy_row[sub_pairs[2 * s]];
        }
        /* derived columns, in order (cpu/mod.rs:182-202) */
        for (size_t d = 0; d < n_derived; ++d) {
            uint32_t v;
            if (derived_kind[d] == 0) {
                v = derived_const[d] % OR_P;
            } else {
                const uint32_t* sp = derived_spans + 4 * d;
                uint32_t den, num;
                if (or_eval_expr(derived_bc + sp[2], sp[3], row_slice, 0, &den)) return -1;
                if (den == 0) v = 0;
                else {
                    if (or_eval_expr(derived_bc + sp[0], sp[1], row_slice, 0, &num)) return -1;
                    v = or_mul(or_inv(den), num);
                }
            }
            row_slice[derived_col[d]] = v;
        }
        /* replay bus interactions on the shared periphery (cpu/mod.rs:204-224) */
        for (size_t i = 0; i < n_inter; ++i) {
            const uint32_t* sp = bus_spans + 2 * (size_t)inter_span_off[i];
            uint32_t m, args[8];
            uint32_t na = inter_nargs[i];
            if (or_eval_expr(bus_bc + sp[0], sp[1], row_slice, 0, &m)) return -1;
            for (uint32_t k = 0; k < na && k < 8; ++k)
                if (or_eval_expr(bus_bc + sp[2 + 2 * k], sp[3 + 2 * k], row_slice, 0, &args[k])) return -1;
            uint32_t id = inter_bus[i];
            if (has_bitwise && id == bitwise_bus) {
                if (na < 4) return -1;
                uint32_t x = args[0], y = args[1], sel = args[3];
                if (m && sel > 1) return -3; /* unreachable!("Invalid selector") */
                if (m && (x >= 256 || y >= 256)) return -3;
                if (m) bitwise_hist[sel * 65536u + x * 256u + y] += m;
            } else if (id == var_bus) {
                if (na < 2) return -1;
                uint32_t idx = (1u << args[1]) + args[0] - 1u; /* VariableRangeChecker::add_count */
                if (m && (args[1] >= 32 || idx >= var_bins)) return -3;
                if (m) var_hist[idx] += m;
            } else if (has_tuple && id == tuple_bus) {
                if (na != 2) return -1;
                if (m && (args[0] >= sz0 || args[1] >= sz1)) return -3;
                if (m) tuple_hist[args[0] * sz1 + args[1]] += m;
            } else if (id <= 2) {
                /* execution bridge, memory, pc lookup: nothing (cpu/periphery.rs:228-231) */
            } else {
                return -2;
            }
        }
    }
    return 0;
}
