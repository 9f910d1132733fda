// Synthetic INPUT generator of the benchmarks and tests (libpowdr_synth.so) — not part of the product library and not behind
// include/*.h: the original chips' dummy traces of a segment, refilled on the device from a seed, so that the segments of a
// multi-segment run differ in their values like the segments of one execution do (/root/reference/openvm-riscv/src/lib.rs:585-592:
// one execution is cut into segments; same chips, different rows) while only ONE segment's inputs are resident at a time.
// One write-only pass: word (col, r) of a column-major w x h matrix = a counter-based hash of (seed, col, r) — below P for free
// cells (any word < P is a valid Montgomery representation), below the cell's bound and converted to Montgomery form for cells that
// feed bounded APC columns (bytes, range-checked limbs, flags): bounds[col * b + (r mod b)], 0 = unbounded.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../csrc/babybear.hpp"

namespace {
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t word(uint64_t key, uint32_t bound) {
    const uint64_t x = mix64(key);
    if (bound) return bb::to_monty((uint32_t)(((x >> 32) * bound) >> 32));  // uniform below the bound (bound < 2^31)
    return (uint32_t)(((x >> 32) * (uint64_t)bb::P) >> 32);                  // uniform below P
}

__global__ __launch_bounds__(256) void synth_fill_kernel(uint32_t* __restrict__ out, size_t h, uint32_t b, const uint32_t* __restrict__ bounds,
                                                         uint64_t seed) {
    const size_t r0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (r0 >= h) return;
    const uint32_t col = blockIdx.y;
    const uint64_t base = mix64(seed ^ ((uint64_t)col << 40));
    const uint32_t* bc = bounds + (size_t)col * b;
    uint32_t v[4];
    uint32_t row = (uint32_t)(r0 % b);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = word(base + r0 + k, bc[row]);
        row = row + 1 == b ? 0 : row + 1;
    }
    *reinterpret_cast<uint4*>(out + (size_t)col * h + r0) = make_uint4(v[0], v[1], v[2], v[3]);
}
}  // namespace

// d_out: w x h words, column-major, h a multiple of 4; d_bounds: w x b words. Runs on `stream` (nullptr: the null stream).
extern "C" int powdr_synth_fill_sources(uint32_t* d_out, uint32_t w, size_t h, uint32_t b, const uint32_t* d_bounds, uint64_t seed, void* stream) {
    if (!d_out || !d_bounds || !w || !h || (h & 3) || !b) return -1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)((h / 4 + 255) / 256), w), dim3(256), 0, (hipStream_t)stream, d_out, h, b, d_bounds, seed);
    return (int)hipGetLastError();
}
