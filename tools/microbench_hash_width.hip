// VERDICT r3 #8, as an experiment: what would a rate-16 leaf sponge (Poseidon2 width 24) buy over the rate-8 one (width 16)?
// Both permutations are written here in ONE plain style — unsigned Montgomery words, x^7 by four bb::mul, linear layers in 64-bit
// accumulators with one reduction per output, internal layer s_i <- mu_i s_i + sum — so that the RATIO of their costs per
// absorbed word is measured like for like (the production width-16 kernel, poseidon2.hpp, is 1.6x faster than this style; its
// tricks — signed representatives, per-stage scales — carry over to either width). Shapes: width 16 = 8 external + 13 partial
// rounds, width 24 = 8 external + 21 partial rounds (the round numbers of the BabyBear x^7 instances), external layer
// circ(2 M4, M4, ..., M4), internal layer 1 1^T + diag(mu). Round constants and mu: a splitmix stream (timing does not depend on them).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I powdr_amd/csrc tools/microbench_hash_width.hip -o tools/microbench_hash_width
// run:   tools/microbench_hash_width [log_rows=19] [cols=512] [reps=5]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "babybear.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kBlock = 256;

template <int W, int RP>
struct Consts {
    uint32_t ext_rc[8][W];
    uint32_t int_rc[RP];
    uint32_t mu[W];
};
__constant__ Consts<16, 13> c16;
__constant__ Consts<24, 21> c24;

__device__ __forceinline__ uint32_t sbox(uint32_t x) {
    const uint32_t x2 = bb::mul(x, x), x3 = bb::mul(x2, x), x4 = bb::mul(x2, x2);
    return bb::mul(x3, x4);
}

// circ(2 M4, M4, ..., M4) with M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]: M4 per block in 64-bit accumulators, then every
// output adds the sum of its column over the blocks; one reduction per output (all sums < 128 p)
template <int W>
__device__ __forceinline__ void external_layer(uint32_t* s) {
    uint64_t t[W];
    uint64_t col[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < W / 4; ++b) {
        const uint32_t x0 = s[4 * b], x1 = s[4 * b + 1], x2 = s[4 * b + 2], x3 = s[4 * b + 3];
        const uint64_t a01 = (uint64_t)x0 + x1, a23 = (uint64_t)x2 + x3, all = a01 + a23;
        t[4 * b + 0] = all + a01 + 2ull * x1;       // 2 x0 + 3 x1 + x2 + x3
        t[4 * b + 1] = all + (uint64_t)x1 + 2ull * x2;  // x0 + 2 x1 + 3 x2 + x3
        t[4 * b + 2] = all + a23 + 2ull * x3;       // x0 + x1 + 2 x2 + 3 x3
        t[4 * b + 3] = all + (uint64_t)x3 + 2ull * x0;  // 3 x0 + x1 + x2 + 2 x3
#pragma unroll
        for (int i = 0; i < 4; ++i) col[i] += t[4 * b + i];
    }
#pragma unroll
    for (int i = 0; i < W; ++i) s[i] = bb::reduce_wide(t[i] + col[i & 3]);
}

template <int W, int RP>
__device__ __forceinline__ void permute(uint32_t* s, const Consts<W, RP>& c) {
    external_layer<W>(s);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < W; ++i) s[i] = sbox(bb::add(s[i], c.ext_rc[r][i]));
        external_layer<W>(s);
    }
#pragma unroll 1
    for (int r = 0; r < RP; ++r) {
        s[0] = sbox(bb::add(s[0], c.int_rc[r]));
        uint64_t sum = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) sum += s[i];
        const uint32_t sm = bb::reduce_wide(sum);
#pragma unroll
        for (int i = 0; i < W; ++i) s[i] = bb::add(bb::mul(s[i], c.mu[i]), sm);
    }
#pragma unroll 1
    for (int r = 4; r < 8; ++r) {
#pragma unroll
        for (int i = 0; i < W; ++i) s[i] = sbox(bb::add(s[i], c.ext_rc[r][i]));
        external_layer<W>(s);
    }
}

template <int W, int RP, int RATE>
__global__ __launch_bounds__(kBlock) void hash_kernel(const uint32_t* __restrict__ m, size_t height, uint32_t width, uint32_t* __restrict__ digests) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= height) return;
    uint32_t st[W];
#pragma unroll
    for (int i = 0; i < W; ++i) st[i] = 0u;
    const uint32_t* col = m + j;
#pragma unroll 1
    for (uint32_t c0 = 0; c0 < width; c0 += RATE) {
#pragma unroll
        for (int k = 0; k < RATE; ++k)
            if (c0 + k < width) st[k] = col[(size_t)(c0 + k) * height];
        if (W == 16) permute<16, 13>(st, c16); else permute<24, 21>(st, c24);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) digests[j * 8 + k] = st[k];
}

template <int W, int RP>
void fill(Consts<W, RP>& c, uint64_t seed) {
    uint64_t s = seed;
    auto next = [&]() -> uint32_t {
        for (;;) {
            s += 0x9E3779B97F4A7C15ull;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            const uint32_t v = (uint32_t)(z & 0x7fffffffu);
            if (v < bb::P && v) return v;
        }
    };
    for (auto& r : c.ext_rc) for (auto& x : r) x = next();
    for (auto& x : c.int_rc) x = next();
    for (auto& x : c.mu) x = next();
}

template <int W, int RP, int RATE>
int run(const char* name, const uint32_t* d_m, size_t H, uint32_t width, uint32_t* d_dig, int reps, double* ns_per_word) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const dim3 grid((unsigned)((H + kBlock - 1) / kBlock));
    hipLaunchKernelGGL((hash_kernel<W, RP, RATE>), grid, dim3(kBlock), 0, 0, d_m, H, width, d_dig);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((hash_kernel<W, RP, RATE>), grid, dim3(kBlock), 0, 0, d_m, H, width, d_dig);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<uint32_t> dig(64);
    CHECK(hipMemcpy(dig.data(), d_dig, 256, hipMemcpyDeviceToHost));
    uint64_t x = 0;
    for (size_t i = 0; i < dig.size(); ++i) x ^= (uint64_t)dig[i] << (i % 32);
    const double perms = (double)H * ((width + RATE - 1) / RATE);
    *ns_per_word = (double)best * 1e6 / ((double)H * width);
    printf("%-28s rows %zu cols %u: %8.3f ms, %6.2f G permutations/s, %.4f ns per absorbed word, checksum %016llx\n", name, H, width, best,
           perms / best / 1e6, *ns_per_word, (unsigned long long)x);
    return 0;
}

int main(int argc, char** argv) {
    const int log_h = argc > 1 ? atoi(argv[1]) : 19;
    const uint32_t width = argc > 2 ? (uint32_t)atoi(argv[2]) : 512;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const size_t H = (size_t)1 << log_h;
    static Consts<16, 13> h16;
    static Consts<24, 21> h24;
    fill(h16, 1);
    fill(h24, 2);
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c16), &h16, sizeof h16));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c24), &h24, sizeof h24));
    std::vector<uint32_t> h((size_t)width * H);
    uint64_t s = 12345;
    for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint32_t)((s >> 33) % bb::P); }
    uint32_t *d_m, *d_dig;
    CHECK(hipMalloc(&d_m, h.size() * 4));
    CHECK(hipMalloc(&d_dig, H * 32));
    CHECK(hipMemcpy(d_m, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    double a = 0, b = 0;
    if (run<16, 13, 8>("width 16, rate 8 (8+13)", d_m, H, width, d_dig, reps, &a)) return 1;
    if (run<24, 21, 16>("width 24, rate 16 (8+21)", d_m, H, width, d_dig, reps, &b)) return 1;
    printf("cost per absorbed word, width 24 / width 16: %.3f  (a rate-16 leaf sponge would take %.1f %% off the leaf hash in this style)\n", b / a,
           (1.0 - b / a) * 100.0);
    return 0;
}
