// Integer-ALU microbenchmarks for gfx950: issue rates of the instructions a BabyBear
// Montgomery multiply is made of. Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int CHAINS = 8;
constexpr uint32_t P = 0x78000001u;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = seed + threadIdx.x * 977u + c * 131u + blockIdx.x;
    uint32_t y = seed | 1u;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (MODE == 0) x[c] = x[c] * y + 1u;                          // v_mul_lo_u32 (+add)
            else if (MODE == 1) x[c] = __umulhi(x[c], y) + y;              // v_mul_hi_u32
            else if (MODE == 2) { uint64_t t = (uint64_t)x[c] * y + x[c]; x[c] = (uint32_t)(t >> 32) ^ (uint32_t)t; }  // v_mad_u64_u32
            else if (MODE == 3) {                                          // Montgomery mul
                uint64_t t = (uint64_t)x[c] * y;
                uint32_t m = (uint32_t)t * 0x77ffffffu;
                uint64_t u = t + (uint64_t)m * P;
                uint32_t r = (uint32_t)(u >> 32);
                x[c] = min(r, r - P);
            } else if (MODE == 4) { uint32_t s = x[c] + y; x[c] = min(s, s - P); }   // mod add
            else if (MODE == 5) x[c] = __mul24(x[c], y) + 1u;              // v_mul_u32_u24 / mad24
            else if (MODE == 6) { x[c] = x[c] + y; }                       // plain add
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc ^= x[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void atomics(uint32_t* hist, uint32_t mask, int per_thread, uint32_t mul) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = i * 2654435761u;
    for (int k = 0; k < per_thread; ++k) {
        s = s * mul + 12345u;
        atomicAdd(&hist[(s >> 8) & mask], 1u);
    }
}

template <int MODE>
int run(const char* name, int ops_per_iter) {
    int blocks = 256 * 16;
    uint32_t* out;
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)blocks * 256 * ITER * CHAINS;
    printf("%-28s %8.3f ms  %8.2f Gop/s (x%d instr)  -> %.2f cycles/wave-op/SIMD @2.4GHz\n", name, ms, n / ms * 1e-6, ops_per_iter,
           (1024.0 * 2.4e9) / (n / 64 / (ms * 1e-3)));
    hipFree(out);
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs %d, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    run<6>("v_add_u32", 1);
    run<0>("v_mul_lo_u32 + add", 2);
    run<1>("v_mul_hi_u32 + add", 2);
    run<2>("v_mad_u64_u32 + xor", 2);
    run<5>("v_mad_u32_u24", 1);
    run<4>("mod add (add,sub,min)", 3);
    run<3>("montgomery mul", 5);
    // atomics: 2^18-bin histogram (var range), random bins, and a hot-bin variant
    uint32_t* hist; CHECK(hipMalloc(&hist, (1u << 20) * 4)); CHECK(hipMemset(hist, 0, (1u << 20) * 4));
    for (uint32_t mask : {0xFFFFFu, 0x3FFFFu, 0xFFFFu, 0xFFu, 0x0u}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        int blocks = 4096, per = 64;
        atomics<<<blocks, 256>>>(hist, mask, per, 1664525u);
        hipDeviceSynchronize();
        hipEventRecord(a);
        atomics<<<blocks, 256>>>(hist, mask, per, 1664525u);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("atomicAdd u32, %8u bins: %8.3f ms  %8.2f G atomics/s\n", mask + 1, ms, (double)blocks * 256 * per / ms * 1e-6);
    }
    return 0;
}
