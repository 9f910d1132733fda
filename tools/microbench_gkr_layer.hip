// what would a bus argument WITHOUT committed columns cost on this machine? The two kernels a fractional-sum GKR
// (LogUp-GKR: the sum of m_i / d_i as a binary tree of fractions (p, q), one sumcheck per layer) spends its time in, measured in
// isolation with the product's own field arithmetic (csrc/babybear.hpp, csrc/ext.hpp):
//
//   build   one layer from the one below: node x = (pL qR + pR qL, qL qR) of its children (2x, 2x + 1) — 64 B read, 32 B written,
//           3 extension products per node
//   round   one sumcheck round of the layer's GKR step over tables PL, PR, QL, QR, EQ of n extension elements each:
//           lane = index pair (x, x + n/2); the round polynomial s(t) = sum_x eq(t) [ (pL(t) + lambda qL(t)) qR(t) + pR(t) qL(t) ]
//           at t = 0, 2, 3 (s(1) follows from the claim), wave + workgroup reduction, one partial per workgroup; then the five tables
//           folded with the round's challenge: 160 B read, 80 B written, 16 extension products per index pair
//
// A layer of 2^k nodes costs one `build` over 2^k nodes and rounds over 2^(k-1) + 2^(k-2) + ... = 2^k index pairs, so per node of the
// tree: build + round, i.e. the two rates below price the whole argument (profiles/r06_gkr_pricing.txt does the sum for C2:
// 1 734 interactions x 2^20 rows -> 2^31 leaves).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipowdr_amd/csrc -Iinclude tools/microbench_gkr_layer.hip -o tools/microbench_gkr_layer
// usage: tools/microbench_gkr_layer [log2 n = 27] [repeats = 5]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "ext.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using bb::Ext;
constexpr int kBlock = 256;

__device__ __forceinline__ Ext ld(const uint4* p, size_t i) { const uint4 v = p[i]; return Ext{{v.x, v.y, v.z, v.w}}; }
__device__ __forceinline__ void st(uint4* p, size_t i, const Ext& e) { p[i] = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]); }

__global__ __launch_bounds__(kBlock) void fill_kernel(uint4* t, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) { z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32; w[k] = (uint32_t)(z % bb::P); }
        t[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// parents[x] = (pL qR + pR qL, qL qR), children interleaved (p, q) pairs: child[2 * (2x) ..] = pL, qL, pR, qR
__global__ __launch_bounds__(kBlock) void build_kernel(const uint4* __restrict__ child, uint4* __restrict__ parent, size_t n_parents) {
    for (size_t x = (size_t)blockIdx.x * kBlock + threadIdx.x; x < n_parents; x += (size_t)gridDim.x * kBlock) {
        const Ext pl = ld(child, 4 * x), ql = ld(child, 4 * x + 1), pr = ld(child, 4 * x + 2), qr = ld(child, 4 * x + 3);
        st(parent, 2 * x, bb::ext_add(bb::ext_mul(pl, qr), bb::ext_mul(pr, ql)));
        st(parent, 2 * x + 1, bb::ext_mul(ql, qr));
    }
}

struct Sum3 { Ext s[3]; };
__device__ __forceinline__ uint32_t shfl_down(uint32_t v, int d) { return __shfl_down(v, d, 64); }

// one sumcheck round: tables of n entries, folded into out tables of n/2 entries with challenge r; partial[blockIdx.x] = the block's
// contribution to s(0), s(2), s(3)
__global__ __launch_bounds__(kBlock) void round_kernel(const uint4* __restrict__ PL, const uint4* __restrict__ PR, const uint4* __restrict__ QL,
                                                      const uint4* __restrict__ QR, const uint4* __restrict__ EQ, uint4* __restrict__ oPL,
                                                      uint4* __restrict__ oPR, uint4* __restrict__ oQL, uint4* __restrict__ oQR, uint4* __restrict__ oEQ,
                                                      size_t half, Ext lambda, Ext r, uint4* __restrict__ partial) {
    __shared__ uint32_t red[kBlock / 64][12];
    Ext acc[3] = {bb::ext_zero(), bb::ext_zero(), bb::ext_zero()};
    for (size_t x = (size_t)blockIdx.x * kBlock + threadIdx.x; x < half; x += (size_t)gridDim.x * kBlock) {
        const Ext pl0 = ld(PL, x), pl1 = ld(PL, x + half), pr0 = ld(PR, x), pr1 = ld(PR, x + half);
        const Ext ql0 = ld(QL, x), ql1 = ld(QL, x + half), qr0 = ld(QR, x), qr1 = ld(QR, x + half);
        const Ext e0 = ld(EQ, x), e1 = ld(EQ, x + half);
        const Ext dpl = bb::ext_sub(pl1, pl0), dpr = bb::ext_sub(pr1, pr0), dql = bb::ext_sub(ql1, ql0), dqr = bb::ext_sub(qr1, qr0), de = bb::ext_sub(e1, e0);
        // a(t) = pL(t) + lambda qL(t): linear in t as well
        const Ext a0 = bb::ext_add(pl0, bb::ext_mul(lambda, ql0)), da = bb::ext_add(dpl, bb::ext_mul(lambda, dql));
        Ext a = a0, pr = pr0, ql = ql0, qr = qr0, e = e0;
#pragma unroll
        for (int t = 0, k = 0; t <= 3; ++t) {
            if (t != 1) {
                const Ext g = bb::ext_add(bb::ext_mul(a, qr), bb::ext_mul(pr, ql));
                acc[k] = bb::ext_add(acc[k], bb::ext_mul(e, g));
                ++k;
            }
            a = bb::ext_add(a, da); pr = bb::ext_add(pr, dpr); ql = bb::ext_add(ql, dql); qr = bb::ext_add(qr, dqr); e = bb::ext_add(e, de);
        }
        st(oPL, x, bb::ext_add(pl0, bb::ext_mul(r, dpl)));
        st(oPR, x, bb::ext_add(pr0, bb::ext_mul(r, dpr)));
        st(oQL, x, bb::ext_add(ql0, bb::ext_mul(r, dql)));
        st(oQR, x, bb::ext_add(qr0, bb::ext_mul(r, dqr)));
        st(oEQ, x, bb::ext_add(e0, bb::ext_mul(r, de)));
    }
    // 12 words: wave reduction by shuffles, then the block's waves through LDS
    uint32_t v[12];
    for (int k = 0; k < 3; ++k) for (int c = 0; c < 4; ++c) v[4 * k + c] = acc[k].c[c];
    for (int d = 32; d >= 1; d >>= 1)
        for (int i = 0; i < 12; ++i) v[i] = bb::add(v[i], shfl_down(v[i], d));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) for (int i = 0; i < 12; ++i) red[wave][i] = v[i];
    __syncthreads();
    if (threadIdx.x < 12) {
        uint32_t s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s = bb::add(s, red[w][threadIdx.x]);
        reinterpret_cast<uint32_t*>(partial + 3 * (size_t)blockIdx.x)[threadIdx.x] = s;
    }
}

int main(int argc, char** argv) {
    const int logn = argc > 1 ? atoi(argv[1]) : 27;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const size_t n = (size_t)1 << logn, half = n / 2;
    uint4 *t[5], *o[5], *partial;
    for (int k = 0; k < 5; ++k) { CHECK(hipMalloc(&t[k], n * 16)); CHECK(hipMalloc(&o[k], half * 16)); }
    const unsigned grid = 256 * 16;
    CHECK(hipMalloc(&partial, (size_t)grid * 48));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(kBlock), 0, 0, t[k], n, 1000u + k);
    CHECK(hipDeviceSynchronize());
    const Ext lambda{{12345u, 678u, 91011u, 1213u}}, r{{424242u, 171717u, 99u, 31337u}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    // ---- round
    hipLaunchKernelGGL(round_kernel, dim3(grid), dim3(kBlock), 0, 0, t[0], t[1], t[2], t[3], t[4], o[0], o[1], o[2], o[3], o[4], half, lambda, r, partial);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(round_kernel, dim3(grid), dim3(kBlock), 0, 0, t[0], t[1], t[2], t[3], t[4], o[0], o[1], o[2], o[3], o[4], half, lambda, r, partial);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double round_ms = ms / reps;
    uint32_t chk[12];
    CHECK(hipMemcpy(chk, partial, 48, hipMemcpyDeviceToHost));
    printf("round  n = 2^%d table entries (%zu index pairs): %.3f ms  = %.2f G index pairs/s, %.0f GB/s algorithmic (240 B per pair)  [checksum %08x]\n", logn,
           half, round_ms, half / round_ms / 1e6, half * 240.0 / round_ms / 1e6, chk[0] ^ chk[5] ^ chk[11]);
    // ---- build: children = 4 tables' worth of (p, q) pairs, parents = n/2 ... use t[0..3] as 2n child entries? keep it simple:
    // n_parents = n / 4 nodes read 4 entries each from t[0] and write 2 entries each to o[0]
    const size_t n_parents = n / 4;
    hipLaunchKernelGGL(build_kernel, dim3(grid), dim3(kBlock), 0, 0, t[0], o[0], n_parents);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(build_kernel, dim3(grid), dim3(kBlock), 0, 0, t[0], o[0], n_parents);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double build_ms = ms / reps;
    printf("build  %zu parent nodes: %.3f ms  = %.2f G nodes/s, %.0f GB/s algorithmic (96 B per node)\n", n_parents, build_ms, n_parents / build_ms / 1e6,
           n_parents * 96.0 / build_ms / 1e6);
    // ---- the price of the argument for BASELINE configs[1] (C2): 1 734 interactions padded to 2^11 per row x 2^20 rows = 2^31 leaves
    const double leaves = 2147483648.0;
    const double pair_s = round_ms * 1e-3 / half, node_s = build_ms * 1e-3 / n_parents;
    printf("model  C2: 2^31 leaves -> %.0f ms of layer builds (2^31 nodes) + %.0f ms of sumcheck rounds (2^31 index pairs) = %.0f ms, before the leaf layer, the\n"
           "       31 x ~30 host round trips of the round polynomials and the main-trace openings at the final point; the committed-column phase it would replace\n"
           "       costs 186 ms of the 307 ms step (307.0 - 121.0)\n",
           leaves * node_s * 1e3, leaves * pair_s * 1e3, leaves * (node_s + pair_s) * 1e3);
    return 0;
}
