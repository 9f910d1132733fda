// Montgomery-product formulations for BabyBear on gfx950 (which instruction mix is cheapest?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 2048, CH = 8;
constexpr uint32_t P = 0x78000001u, NPINV = 0x77ffffffu, PINV = 0x88000001u;

__device__ __forceinline__ uint32_t mA(uint32_t a, uint32_t b) {  // additive, 2x mad_u64
    uint64_t t = (uint64_t)a * b; uint32_t m = (uint32_t)t * NPINV; uint64_t u = t + (uint64_t)m * P;
    uint32_t r = (uint32_t)(u >> 32); return min(r, r - P);
}
__device__ __forceinline__ uint32_t mB(uint32_t a, uint32_t b) {  // subtractive with mul_lo/mul_hi
    uint32_t lo = a * b, hi = __umulhi(a, b); uint32_t m = lo * PINV; uint32_t u = __umulhi(m, P);
    uint32_t r = hi - u; return min(r, r + P);
}
__device__ __forceinline__ uint32_t mC(uint32_t a, uint32_t b) {  // 64-bit product, subtractive
    uint64_t t = (uint64_t)a * b; uint32_t m = (uint32_t)t * PINV; uint32_t u = __umulhi(m, P);
    uint32_t r = (uint32_t)(t >> 32) - u; return min(r, r + P);
}
__device__ __forceinline__ uint32_t mD(uint32_t a, uint32_t b) {  // lazy: result in [0, 2p), no final reduce
    uint64_t t = (uint64_t)a * b; uint32_t m = (uint32_t)t * NPINV; uint64_t u = t + (uint64_t)m * P;
    return (uint32_t)(u >> 32);
}
__device__ __forceinline__ uint32_t mE(uint32_t a, uint32_t b) {  // shifts instead of mul_lo for m
    uint64_t t = (uint64_t)a * b; uint32_t lo = (uint32_t)t;
    uint32_t m = (lo << 31) - (lo << 27) - lo;  // lo * 0x77ffffff
    uint64_t u = t + (uint64_t)m * P; uint32_t r = (uint32_t)(u >> 32); return min(r, r - P);
}
template <int MODE> __global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = (seed + threadIdx.x * 977u + c * 131u + blockIdx.x) % P;
    uint32_t y = seed % P;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (MODE == 0) x[c] = mA(x[c], y);
            else if (MODE == 1) x[c] = mB(x[c], y);
            else if (MODE == 2) x[c] = mC(x[c], y);
            else if (MODE == 3) { uint32_t r = mD(x[c], y); x[c] = min(r, r - P); }
            else if (MODE == 4) x[c] = mE(x[c], y);
            else if (MODE == 5) { uint32_t s = x[c] + y; x[c] = min(s, s - P); }
            else if (MODE == 6) { x[c] = x[c] + y; x[c] = x[c] ^ (x[c] >> 3); }   // 3 plain int ops
            else if (MODE == 7) { float f = __uint_as_float(x[c]); f = fmaf(f, 1.0001f, 0.5f); x[c] = __float_as_uint(f); }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc ^= x[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> void run(const char* n) {
    int blocks = 256 * 16; uint32_t* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 12345u); hipDeviceSynchronize();
    hipEventRecord(a); k<MODE><<<blocks, 256>>>(out, 12345u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * ITER * CH;
    printf("%-44s %7.3f ms  %8.2f Gop/s\n", n, ms, ops / ms * 1e-6);
    hipFree(out);
}
int main() {
    run<7>("v_fma_f32 (reference: 1 full-rate op)");
    run<6>("add, shift, xor (3 plain int ops)");
    run<5>("mod add (add, sub, min)");
    run<0>("monty A: 2x mad_u64_u32 + mul_lo + sub + min");
    run<1>("monty B: mul_lo, mul_hi, mul_lo, mul_hi, sub, add, min");
    run<2>("monty C: mad_u64, mul_lo, mul_hi, sub, add, min");
    run<3>("monty D: A with separate final reduce");
    run<4>("monty E: shifts for m");
    return 0;
}
