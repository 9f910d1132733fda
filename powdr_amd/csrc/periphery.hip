// Traces of the shared periphery chips from the lookup histograms _apc_apply_bus fills: the receive side of the
// variable-range, tuple-range and bitwise buses.
//
// The chips themselves are EXTERNAL (openvm-circuit-primitives: VariableRangeCheckerAir, RangeTupleCheckerAir<2>,
// BitwiseOperationLookupAir<8>; instantiated in /root/reference/openvm/src/powdr_extension/trace_generator/cuda/
// periphery.rs:33-85). What IS in the reference is how a lookup becomes a histogram index
// (/root/reference/openvm/cuda/src/apc_apply_bus.cu:74,89,104; cpu/periphery.rs:176-237), and a chip's trace is that
// map inverted: row i carries the tuple whose index is i and the count as its multiplicity. Column layouts below are
// this library's (the chips' own keep the tuple in preprocessed columns): every matrix is column-major, Montgomery.
//
// Pure streaming: one u32 read and 3-5 u32 writes per row, coalesced; <= 2^19 rows, so these are launch-latency sized.
#include "babybear.hpp"
#include "common.hpp"
#include "../../include/powdr_gpu.h"

namespace {

constexpr int kBlock = 256;

// row i <-> (bits, value) with i = (1 << bits) + value - 1, value < 2^bits  (apc_apply_bus.cu:74)
__global__ __launch_bounds__(kBlock) void var_range_trace_kernel(const uint32_t* __restrict__ hist, size_t n, uint32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = (uint32_t)(i + 1);
    const uint32_t bits = 31u - (uint32_t)__clz(k);
    out[i] = bb::to_monty(k - (1u << bits));
    out[n + i] = bb::to_monty(bits);
    out[2 * n + i] = bb::to_monty(hist[i] % bb::P);
}

// row i <-> (i / sz1, i % sz1)  (apc_apply_bus.cu:89)
__global__ __launch_bounds__(kBlock) void tuple2_trace_kernel(const uint32_t* __restrict__ hist, uint32_t sz1, size_t n, uint32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    out[i] = bb::to_monty((uint32_t)(i / sz1));
    out[n + i] = bb::to_monty((uint32_t)(i % sz1));
    out[2 * n + i] = bb::to_monty(hist[i] % bb::P);
}

// row i <-> (x, y) = (i >> 8, i & 255); histogram = [range counts | xor counts]
__global__ __launch_bounds__(kBlock) void bitwise_trace_kernel(const uint32_t* __restrict__ hist, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= 65536u) return;
    const uint32_t x = i >> 8, y = i & 255u;
    out[i] = bb::to_monty(x);
    out[65536u + i] = bb::to_monty(y);
    out[2 * 65536u + i] = bb::to_monty(x ^ y);
    out[3 * 65536u + i] = bb::to_monty(hist[i] % bb::P);
    out[4 * 65536u + i] = bb::to_monty(hist[65536u + i] % bb::P);
}

}  // namespace

extern "C" int powdr_periphery_var_range_trace(const uint32_t* d_hist, size_t n_bins, PowdrFp* d_out) {
    (void)hipGetLastError();
    if (!d_hist || !d_out || n_bins == 0 || (n_bins & (n_bins - 1)) || n_bins > ((size_t)1 << 31)) return (int)hipErrorInvalidValue;
    pw::ScopedKernelTimer t("var_range_trace_kernel");
    hipLaunchKernelGGL(var_range_trace_kernel, dim3(pw::div_up(n_bins, kBlock)), dim3(kBlock), 0, pw::stream(), d_hist, n_bins, d_out);
    return (int)hipGetLastError();
}

extern "C" int powdr_periphery_tuple2_trace(const uint32_t* d_hist, uint32_t sz0, uint32_t sz1, PowdrFp* d_out) {
    (void)hipGetLastError();
    const size_t n = (size_t)sz0 * sz1;
    if (!d_hist || !d_out || n == 0 || (n & (n - 1)) || n > ((size_t)1 << 31)) return (int)hipErrorInvalidValue;
    pw::ScopedKernelTimer t("tuple2_trace_kernel");
    hipLaunchKernelGGL(tuple2_trace_kernel, dim3(pw::div_up(n, kBlock)), dim3(kBlock), 0, pw::stream(), d_hist, sz1, n, d_out);
    return (int)hipGetLastError();
}

extern "C" int powdr_periphery_bitwise_trace(const uint32_t* d_hist, PowdrFp* d_out) {
    (void)hipGetLastError();
    if (!d_hist || !d_out) return (int)hipErrorInvalidValue;
    pw::ScopedKernelTimer t("bitwise_trace_kernel");
    hipLaunchKernelGGL(bitwise_trace_kernel, dim3(65536 / kBlock), dim3(kBlock), 0, pw::stream(), d_hist, d_out);
    return (int)hipGetLastError();
}
