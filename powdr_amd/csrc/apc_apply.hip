// _apc_apply_derived_expr and _apc_apply_bus for gfx950.
//
// Reference semantics:
//   derived columns  /root/reference/openvm/cuda/src/apc_tracegen.cu:72-124
//                    (CPU twin: openvm/src/powdr_extension/trace_generator/cpu/mod.rs:182-202)
//   bus -> histogram /root/reference/openvm/cuda/src/apc_apply_bus.cu:23-169
//                    (CPU twin: .../cpu/periphery.rs:176-237)
//
// One lane per APC row; the bytecode is wave-uniform (see expr_eval.hpp).
// Differences from the reference launch, none of them observable in the result:
//   * interactions whose bus id is none of the three periphery buses are
//     skipped before their multiplicity is evaluated (the reference evaluates
//     it and then ignores it, apc_apply_bus.cu:58-63,111);
//   * a multiplicity m is applied as one atomicAdd(bin, m) instead of m single
//     increments (apc_apply_bus.cu:77,91,106) — same sum mod 2^32;
//   * the interaction list is additionally split over blockIdx.y so that short
//     traces still fill 256 CUs (histogram sums are order independent).
// Defined behaviour where the reference traps or writes out of bounds: a
// histogram index >= the bin count is dropped (index arithmetic is the
// reference's 32-bit wrap-around arithmetic); bitwise operands >= 256 or a
// selector other than 0/1 are ignored (reference: assert(false),
// apc_apply_bus.cu:109).
#include "babybear.hpp"
#include "common.hpp"
#include "expr_eval.hpp"
#include "../../include/powdr_gpu.h"

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void apc_apply_derived_expr_kernel(
    uint32_t* d_output, size_t H, int num_calls, const DerivedExprSpec* __restrict__ specs,
    size_t n_cols, const uint32_t* __restrict__ bytecode) {
    __shared__ uint32_t stack_lds[pw::kStackCap * kBlock];
    uint32_t* stk = stack_lds + threadIdx.x;
    const size_t total = (size_t)gridDim.x * kBlock;
    for (size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x; r < H; r += total) {
        if (r < (size_t)num_calls) {
            for (size_t i = 0; i < n_cols; ++i) {
                const DerivedExprSpec spec = specs[i];
                // later derived columns may read earlier ones of the same row: plain
                // (non-restrict) accesses by the same thread keep program order.
                uint32_t v = pw::eval_expr<kBlock>(bytecode + spec.span.off, spec.span.len,
                                                  d_output, r, stk);
                d_output[spec.col_base + r] = v;
            }
        } else {
            for (size_t i = 0; i < n_cols; ++i) d_output[specs[i].col_base + r] = 0u;
        }
    }
}

struct BusParams {
    uint32_t var_bus, tuple_bus, bitwise_bus;
    uint32_t* var_hist;
    uint32_t* tuple_hist;
    uint32_t* bitwise_hist;
    uint32_t var_bins, tuple_sz0, tuple_sz1;
};

// Layout of BitwiseOperationLookup<8>'s count buffer (EXTERNAL to the reference
// checkout, assumption A3 of SURVEY.md): 2^16 range counters followed by 2^16 xor
// counters, both indexed by x * 256 + y.
__device__ __forceinline__ uint32_t bitwise_index(uint32_t x, uint32_t y, uint32_t selector) {
    return selector * (1u << (2 * POWDR_BITWISE_NUM_BITS)) + (x << POWDR_BITWISE_NUM_BITS) + y;
}

__global__ __launch_bounds__(kBlock) void apc_apply_bus_kernel(
    const uint32_t* __restrict__ trace, int num_calls, const uint32_t* __restrict__ bytecode,
    const DevInteraction* __restrict__ interactions, uint32_t n_interactions,
    const ExprSpan* __restrict__ spans, BusParams p, uint32_t per_chunk) {
    __shared__ uint32_t stack_lds[pw::kStackCap * kBlock];
    uint32_t* stk = stack_lds + threadIdx.x;
    const int r_i = blockIdx.x * kBlock + threadIdx.x;
    if (r_i >= num_calls) return;
    const size_t r = (size_t)r_i;
    const uint32_t i0 = blockIdx.y * per_chunk;
    const uint32_t i1 = min(n_interactions, i0 + per_chunk);

    for (uint32_t i = i0; i < i1; ++i) {
        const DevInteraction intr = interactions[i];
        int kind;
        if (intr.bus_id == p.var_bus) kind = 0;
        else if (intr.bus_id == p.tuple_bus) kind = 1;
        else if (intr.bus_id == p.bitwise_bus) kind = 2;
        else continue;  // execution bridge / memory / pc lookup: no periphery side effect

        const ExprSpan* sp = spans + intr.args_index_off;
        const ExprSpan ms = sp[0];
        const uint32_t m = bb::from_monty(pw::eval_expr<kBlock>(bytecode + ms.off, ms.len, trace, r, stk));
        if (m == 0u) continue;

        const ExprSpan s0 = sp[1], s1 = sp[2];
        const uint32_t a0 = bb::from_monty(pw::eval_expr<kBlock>(bytecode + s0.off, s0.len, trace, r, stk));
        const uint32_t a1 = bb::from_monty(pw::eval_expr<kBlock>(bytecode + s1.off, s1.len, trace, r, stk));
        if (kind == 0) {
            // [value, max_bits] -> bin (1 << max_bits) + value - 1   (apc_apply_bus.cu:74)
            // (shift counts >= 32 give 0, as PTX shl.b32 does for the reference build)
            const uint32_t idx = (a1 < 32u ? (1u << a1) : 0u) + a0 - 1u;
            if (idx < p.var_bins) atomicAdd(p.var_hist + idx, m);
        } else if (kind == 1) {
            // [v0, v1] -> bin v0 * sz1 + v1                         (apc_apply_bus.cu:89)
            const uint32_t idx = a0 * p.tuple_sz1 + a1;
            if (idx < p.tuple_sz0 * p.tuple_sz1) atomicAdd(p.tuple_hist + idx, m);
        } else {
            // [x, y, x_xor_y, selector]; arg 2 is never read (apc_apply_bus.cu:94-99)
            const ExprSpan s3 = sp[4];
            const uint32_t sel = bb::from_monty(pw::eval_expr<kBlock>(bytecode + s3.off, s3.len, trace, r, stk));
            if (sel <= 1u && a0 < 256u && a1 < 256u)
                atomicAdd(p.bitwise_hist + bitwise_index(a0, a1, sel), m);
        }
    }
}

}  // namespace

extern "C" int _apc_apply_derived_expr(PowdrFp* d_output, size_t H, int num_apc_calls,
                                       const DerivedExprSpec* d_specs, size_t n_cols,
                                       const uint32_t* d_bytecode) {
    if (n_cols == 0) return 0;  // apc_tracegen.cu:114
    (void)hipGetLastError();    // do not report a stale error of an unrelated earlier call
    if (H == 0) return (int)hipGetLastError();
    unsigned g = pw::div_up(H, kBlock);
    if (g > 65535u * 16u) g = 65535u * 16u;
    pw::ScopedKernelTimer t("apc_apply_derived_expr_kernel");
    hipLaunchKernelGGL(apc_apply_derived_expr_kernel, dim3(g), dim3(kBlock), 0, pw::stream(),
                       d_output, H, num_apc_calls, d_specs, n_cols, d_bytecode);
    return (int)hipGetLastError();
}

extern "C" int _apc_apply_bus(const PowdrFp* d_output, int num_apc_calls,
                              const uint32_t* d_bytecode, size_t bytecode_len,
                              const DevInteraction* d_interactions, size_t n_interactions,
                              const ExprSpan* d_arg_spans, size_t n_arg_spans,
                              uint32_t var_range_bus_id, uint32_t* d_var_hist, size_t var_num_bins,
                              uint32_t tuple2_bus_id, uint32_t* d_tuple2_hist, uint32_t tuple2_sz0,
                              uint32_t tuple2_sz1, uint32_t bitwise_bus_id,
                              uint32_t* d_bitwise_hist) {
    (void)bytecode_len;
    (void)n_arg_spans;
    if (num_apc_calls <= 0) return 0;  // apc_apply_bus.cu:146
    (void)hipGetLastError();
    if (n_interactions == 0) return (int)hipGetLastError();
    const unsigned row_blocks = pw::div_up((size_t)num_apc_calls, kBlock);
    // Enough workgroups to cover 256 CUs x 8 blocks even for short traces.
    unsigned chunks = 1;
    const unsigned want_blocks = 256u * 8u;
    if (row_blocks < want_blocks) {
        chunks = (want_blocks + row_blocks - 1) / row_blocks;
        const unsigned max_chunks = (unsigned)((n_interactions + 15) / 16);  // >= 16 interactions each
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks > 65535u) chunks = 65535u;
        if (chunks == 0) chunks = 1;
    }
    const uint32_t per_chunk = (uint32_t)((n_interactions + chunks - 1) / chunks);
    chunks = (unsigned)((n_interactions + per_chunk - 1) / per_chunk);
    BusParams p;
    p.var_bus = var_range_bus_id; p.tuple_bus = tuple2_bus_id; p.bitwise_bus = bitwise_bus_id;
    p.var_hist = d_var_hist; p.tuple_hist = d_tuple2_hist; p.bitwise_hist = d_bitwise_hist;
    p.var_bins = (uint32_t)var_num_bins; p.tuple_sz0 = tuple2_sz0; p.tuple_sz1 = tuple2_sz1;
    pw::ScopedKernelTimer t("apc_apply_bus_kernel");
    hipLaunchKernelGGL(apc_apply_bus_kernel, dim3(row_blocks, chunks), dim3(kBlock), 0, pw::stream(),
                       d_output, num_apc_calls, d_bytecode, d_interactions,
                       (uint32_t)n_interactions, d_arg_spans, p, per_chunk);
    return (int)hipGetLastError();
}
