// The leaf hash in isolation: a (rows x cols) column-major matrix of canonical words, one lane per row, the production
// permutation (powdr_amd/csrc/poseidon2.hpp). Variants are chosen at compile time so that one GPU visit can time several
// builds of the same source against each other:
//   -DHB_MINW=6|8      __launch_bounds__ second argument (VGPR cap)
//   -DHB_TWO_LOOPS     full chunks in one loop, the short last chunk apart (two copies of the permutation)
//   -DHB_NO_SPONGE     every permutation canonicalises its output
//   -DPW_NO_REDUCE_MAD reduce_wide_loose with mul_lo + sub instead of one multiply-add
// prints ms per launch, permutations per second and an XOR/sum checksum of the digests (equal across variants).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I powdr_amd/csrc [-D...] tools/microbench_hash.hip -o /tmp/hb_x
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "poseidon2.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#ifndef HB_MINW
#define HB_MINW 6
#endif
#ifdef HB_NO_SPONGE
constexpr bool kSponge = false;
#else
constexpr bool kSponge = true;
#endif
constexpr int kBlock = 256;
__constant__ p2::Params c_params;

__global__ __launch_bounds__(kBlock, HB_MINW) void hash_kernel(const uint32_t* __restrict__ m, size_t height, uint32_t width,
                                                              uint32_t* __restrict__ digests) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= height) return;
    uint32_t st[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0u;
    const uint32_t* col = m + j;
#ifdef HB_TWO_LOOPS
    uint32_t c0 = 0;
    for (; c0 + 8 <= width; c0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) st[k] = col[(size_t)(c0 + k) * height];
        p2::permute<kSponge>(st, c_params);
    }
    if (c0 < width) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c0 + k < width) st[k] = col[(size_t)(c0 + k) * height];
        p2::permute<kSponge>(st, c_params);
    }
#else
#pragma unroll 1
    for (uint32_t c0 = 0; c0 < width; c0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c0 + k < width) st[k] = col[(size_t)(c0 + k) * height];
        p2::permute<kSponge>(st, c_params);
    }
#endif
#pragma unroll
    for (int k = 0; k < 8; ++k) digests[j * 8 + k] = bb::reduce_2p(st[k]);
}

int main(int argc, char** argv) {
    const int log_h = argc > 1 ? atoi(argv[1]) : 19;
    const uint32_t width = argc > 2 ? (uint32_t)atoi(argv[2]) : 512;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const size_t H = (size_t)1 << log_h;
    p2::Params P;
    p2::generate_params(P);
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_params), &P, sizeof(P)));
    std::vector<uint32_t> h((size_t)width * H);
    uint64_t s = 12345;
    for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint32_t)((s >> 33) % bb::P); }
    uint32_t *d_m, *d_dig;
    CHECK(hipMalloc(&d_m, h.size() * 4));
    CHECK(hipMalloc(&d_dig, H * 32));
    CHECK(hipMemcpy(d_m, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const dim3 grid((unsigned)((H + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(hash_kernel, grid, dim3(kBlock), 0, 0, d_m, H, width, d_dig);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(hash_kernel, grid, dim3(kBlock), 0, 0, d_m, H, width, d_dig);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<uint32_t> dig(H * 8);
    CHECK(hipMemcpy(dig.data(), d_dig, H * 32, hipMemcpyDeviceToHost));
    uint64_t x = 0, sum = 0;
    for (size_t i = 0; i < dig.size(); ++i) { x ^= (uint64_t)dig[i] << (i % 32); sum += dig[i]; if (dig[i] >= bb::P) { printf("NON-CANONICAL digest word\n"); return 2; } }
    const double perms = (double)H * ((width + 7) / 8);
    printf("rows 2^%d cols %u: %.3f ms, %.2f G permutations/s, checksum %016llx %016llx\n", log_h, width, best, perms / best / 1e6,
           (unsigned long long)x, (unsigned long long)sum);
    return 0;
}
