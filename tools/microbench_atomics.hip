// Atomic-add throughput on gfx950 by memory scope and table size (histogram design input).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int SCOPE, bool PRIVATE>
__global__ void k(uint32_t* hist, uint32_t mask, int per_thread, uint32_t copy_stride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = i * 2654435761u;
    uint32_t* h = hist;
    if (PRIVATE) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        h = hist + (size_t)(xcc & 15u) * copy_stride;
    }
    for (int kk = 0; kk < per_thread; ++kk) {
        s = s * 1664525u + 12345u;
        __hip_atomic_fetch_add(&h[(s >> 8) & mask], 1u, __ATOMIC_RELAXED, SCOPE);
    }
}
__global__ void sum_k(const uint32_t* h, size_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += h[i];
    atomicAdd(out, acc);
}

template <int SCOPE, bool PRIVATE>
int run(const char* name, uint32_t* hist, uint32_t mask, size_t total_words, unsigned long long* d_sum) {
    int blocks = 8192, per = 64;
    CHECK(hipMemset(hist, 0, total_words * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<SCOPE, PRIVATE><<<blocks, 256>>>(hist, mask, per, mask + 1);
    hipDeviceSynchronize();
    CHECK(hipMemset(hist, 0, total_words * 4));
    hipEventRecord(a);
    k<SCOPE, PRIVATE><<<blocks, 256>>>(hist, mask, per, mask + 1);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    CHECK(hipMemset(d_sum, 0, 8));
    sum_k<<<1024, 256>>>(hist, total_words, d_sum);
    unsigned long long s; CHECK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
    unsigned long long expect = (unsigned long long)blocks * 256 * per;
    printf("%-34s bins %8u: %8.3f ms %8.2f G atomics/s  sum %s (%llu/%llu)\n", name, mask + 1, ms, (double)expect / ms * 1e-6,
           s == expect ? "OK" : "LOST UPDATES", s, expect);
    return 0;
}

int main() {
    uint32_t* hist; size_t words = (size_t)16 << 20;  // 16 copies x 1M bins
    CHECK(hipMalloc(&hist, words * 4));
    unsigned long long* d_sum; CHECK(hipMalloc(&d_sum, 8));
    for (uint32_t mask : {0xFFFFFu, 0x3FFFFu, 0xFFFFu, 0xFFFu, 0xFFu}) {
        run<__HIP_MEMORY_SCOPE_AGENT, false>("agent scope, shared table", hist, mask, words, d_sum);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, false>("workgroup scope, shared table", hist, mask, words, d_sum);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope, per-XCD table", hist, mask, words, d_sum);
        run<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope, per-XCD table", hist, mask, words, d_sum);
        run<__HIP_MEMORY_SCOPE_WAVEFRONT, true>("wavefront scope, per-XCD table", hist, mask, words, d_sum);
    }
    return 0;
}
