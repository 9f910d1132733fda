// The Poseidon2 external (MDS) layer on the matrix cores — the experiment SURVEY H2 / VERDICT r1 item 5 ask for.
//
// The layer is  out = M s  (mod p) with M = circ(2 M4, M4, M4, M4), entries in {1,2,3,4,6}, row sums 35; s = 16 BabyBear
// words per row. Two implementations on the same states, both returning canonical words, compared word for word:
//
//  (A) the production form at the time of the measurement (external_layer_valu below): lane = row, 64-bit multiply-add
//      accumulators, one reduce_wide per output — integer VALU only.
//  (B) v_mfma_i32_16x16x64_i8 on byte planes. Out^T = M S^T per tile of 16 rows: A = M (out index x word index), B = the
//      bytes of S^T. The i8 inputs force the 31-bit words into 4 byte planes; the planes cannot share one MFMA because
//      their weights 2^(8b) do not fit an i8 matrix entry, so a tile takes 4 MFMAs (A masked to k-block b; only 16 of the
//      64 k-slots carry data each time) and a wave (64 rows) takes 16. Around the MFMAs, in VALU instructions per tile and
//      lane: 8 v_perm_b32 (4x4 byte transpose of the lane's 4 words into planes), 4 v_xor (u8 -> i8 offset), 4
//      v_permlane{32,16}_swap (4x4 transpose across the four 16-lane groups: the D layout of one layer is not the B layout
//      of the next), and per output word 2 v_lshl_add + 2 v_add + 1 v_mad_u64_u32 to recombine the four i32 planes into
//      a 64-bit value + reduce_wide (5): 10 per word.
//  (C) the price of entering / leaving the tile layout from the kernels' lane-per-row layout through LDS (the 13 partial
//      rounds of a permutation need lane-per-row: one S-box per row), two switches per permutation at least.
//
// build: hipcc --offload-arch=gfx950 -O3 -I powdr_amd/csrc tools/microbench_mfma_mds.hip -o tools/microbench_mfma_mds
// output -> profiles/r02_microbench_mfma_mds.txt
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "poseidon2.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;

// ---- (A) production form ------------------------------------------------------------------------------------------
// The VALU form this experiment was measured against (round 2, before the permutation moved to signed representatives and
// shared M4 partial sums — powdr_amd/csrc/poseidon2.hpp has the current one): four multiply-add chains per block in unsigned
// 64-bit accumulators, the column sums, one reduce_wide per output. 140 instructions.
__device__ __forceinline__ void external_layer_valu(uint32_t* s) {
    uint64_t y[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x0 = s[4 * b + i], x1 = s[4 * b + ((i + 1) & 3)], x2 = s[4 * b + ((i + 2) & 3)], x3 = s[4 * b + ((i + 3) & 3)];
            uint64_t a = bb::wide_mul(x0, 2);
            a = bb::wide_fma(a, x1, 3);
            a = bb::wide_add(a, x2);
            y[4 * b + i] = bb::wide_add(a, x3);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t col = (y[i] + y[4 + i]) + (y[8 + i] + y[12 + i]);
#pragma unroll
        for (int b = 0; b < 4; ++b) s[4 * b + i] = bb::reduce_wide(y[4 * b + i] + col);
    }
}

__global__ __launch_bounds__(kBlock) void mds_valu_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int layers) {
    const size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = in[row * 16 + i];
#pragma unroll 1
    for (int l = 0; l < layers; ++l) external_layer_valu(s);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[row * 16 + i] = s[i];
}

// ---- (B) MFMA form ------------------------------------------------------------------------------------------------
// tile layout of a wave's 64 rows: lane (n = lane % 16, g = lane / 16), tile t < 4: w[t][r] = word 4g + r of row 16 t + n
__device__ __forceinline__ uint32_t mds_entry(int i, int j) {
    const int M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    return (uint32_t)M4[i & 3][j & 3] * ((i >> 2) == (j >> 2) ? 2u : 1u);
}

__device__ __forceinline__ void mds_mfma_tile(uint32_t* w, const v4i* A) {
    // 1. planes: pk[b] = byte b of (w0, w1, w2, w3)
    const uint32_t x0 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), x1 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
    const uint32_t y0 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), y1 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
    uint32_t pk0 = __builtin_amdgcn_perm(y0, x0, 0x05040100u), pk1 = __builtin_amdgcn_perm(y0, x0, 0x07060302u);
    uint32_t pk2 = __builtin_amdgcn_perm(y1, x1, 0x05040100u), pk3 = __builtin_amdgcn_perm(y1, x1, 0x07060302u);
    // u8 -> i8: u - 128 (the matrix unit multiplies signed bytes); the offset comes back as 128 * 35 per plane below
    pk0 ^= 0x80808080u; pk1 ^= 0x80808080u; pk2 ^= 0x80808080u; pk3 ^= 0x80808080u;
    // 2. 4x4 transpose over (lane group, register): lane group b ends up with plane b of all 16 words of its row
    {
        auto a = __builtin_amdgcn_permlane32_swap(pk0, pk2, false, false); pk0 = a[0]; pk2 = a[1];
        auto c = __builtin_amdgcn_permlane32_swap(pk1, pk3, false, false); pk1 = c[0]; pk3 = c[1];
        auto d = __builtin_amdgcn_permlane16_swap(pk0, pk1, false, false); pk0 = d[0]; pk1 = d[1];
        auto e = __builtin_amdgcn_permlane16_swap(pk2, pk3, false, false); pk2 = e[0]; pk3 = e[1];
    }
    const v4i B = {(int)pk0, (int)pk1, (int)pk2, (int)pk3};
    const v4i Z = {0, 0, 0, 0};
    // 3. one MFMA per plane (A[b] = M in k-block b, zero elsewhere)
    const v4i D0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], B, Z, 0, 0, 0);
    const v4i D1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], B, Z, 0, 0, 0);
    const v4i D2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], B, Z, 0, 0, 0);
    const v4i D3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[3], B, Z, 0, 0, 0);
    // 4. recombine: value = sum_b (D_b + 4480) 2^(8b) < 35 p, reduced once
    constexpr uint32_t kOff = 128u * 35u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t t = ((uint32_t)D1[r] << 8) + (uint32_t)D0[r];
        t = ((uint32_t)D2[r] << 16) + t;
        t += kOff * (1u + 256u + 65536u);
        const uint32_t e3 = (uint32_t)D3[r] + kOff;
        const uint64_t v = (uint64_t)e3 * (1u << 24) + t;
        w[r] = bb::reduce_wide(v);
    }
}

// lane-per-row -> tile layout and back through LDS (one wave: 64 rows x 16 words, row pitch 20 words)
constexpr int kPitch = 20;
__device__ __forceinline__ void to_tiles(uint32_t* lds, const uint32_t* s, uint32_t w[4][4], int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(lds + lane * kPitch + 4 * q) = make_uint4(s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]);
    __builtin_amdgcn_wave_barrier();
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint4 v = *reinterpret_cast<const uint4*>(lds + (16 * t + n) * kPitch + 4 * g);
        w[t][0] = v.x; w[t][1] = v.y; w[t][2] = v.z; w[t][3] = v.w;
    }
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void from_tiles(uint32_t* lds, uint32_t* s, const uint32_t w[4][4], int lane) {
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<uint4*>(lds + (16 * t + n) * kPitch + 4 * g) = make_uint4(w[t][0], w[t][1], w[t][2], w[t][3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = *reinterpret_cast<const uint4*>(lds + lane * kPitch + 4 * q);
        s[4 * q] = v.x; s[4 * q + 1] = v.y; s[4 * q + 2] = v.z; s[4 * q + 3] = v.w;
    }
    __builtin_amdgcn_wave_barrier();
}

// MODE 0: layout switches once, `layers` MFMA layers in between. MODE 1: only the switches, `layers` round trips.
template <int MODE>
__global__ __launch_bounds__(kBlock) void mds_mfma_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int layers) {
    __shared__ uint32_t lds_all[(kBlock / 64) * 64 * kPitch];
    const int lane = threadIdx.x & 63;
    uint32_t* lds = lds_all + (threadIdx.x >> 6) * 64 * kPitch;
    const size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = in[row * 16 + i];
    // A operands: lane (m = lane % 16 = output index, k-block lane / 16) holds M[m][0..15] when its k-block is b
    v4i A[4];
    {
        const int m = lane & 15, kb = lane >> 4;
        uint32_t rowb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            rowb[q] = mds_entry(m, 4 * q) | (mds_entry(m, 4 * q + 1) << 8) | (mds_entry(m, 4 * q + 2) << 16) | (mds_entry(m, 4 * q + 3) << 24);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const bool on = kb == b;
            A[b] = v4i{on ? (int)rowb[0] : 0, on ? (int)rowb[1] : 0, on ? (int)rowb[2] : 0, on ? (int)rowb[3] : 0};
        }
    }
    uint32_t w[4][4];
    if (MODE == 0) {
        to_tiles(lds, s, w, lane);
#pragma unroll 1
        for (int l = 0; l < layers; ++l) {
#pragma unroll
            for (int t = 0; t < 4; ++t) mds_mfma_tile(w[t], A);
        }
        from_tiles(lds, s, w, lane);
    } else {
#pragma unroll 1
        for (int l = 0; l < layers; ++l) {
            to_tiles(lds, s, w, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t][l & 3] ^= (uint32_t)l;  // keep the round trip alive
            from_tiles(lds, s, w, lane);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[row * 16 + i] = s[i];
}

template <class F>
static float time_ms(F launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(a);
        launch();
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t rows = (size_t)256 * 8 * kBlock;  // 8 waves per SIMD
    const int blocks = (int)(rows / kBlock);
    std::vector<uint32_t> h(rows * 16);
    uint64_t st = 12345;
    for (auto& v : h) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = (uint32_t)(st >> 33) % bb::P; }
    uint32_t *d_in, *d_a, *d_b;
    CHECK(hipMalloc(&d_in, h.size() * 4)); CHECK(hipMalloc(&d_a, h.size() * 4)); CHECK(hipMalloc(&d_b, h.size() * 4));
    CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // correctness: 3 layers both ways
    hipLaunchKernelGGL(mds_valu_kernel, dim3(blocks), dim3(kBlock), 0, 0, d_in, d_a, 3);
    hipLaunchKernelGGL(mds_mfma_kernel<0>, dim3(blocks), dim3(kBlock), 0, 0, d_in, d_b, 3);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> ra(h.size()), rb(h.size());
    CHECK(hipMemcpy(ra.data(), d_a, h.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(rb.data(), d_b, h.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); ++i) bad += ra[i] != rb[i];
    // and against plain arithmetic on the host for the first rows
    size_t bad_host = 0;
    for (size_t r = 0; r < 64; ++r) {
        uint64_t s[16];
        for (int i = 0; i < 16; ++i) s[i] = h[r * 16 + i];
        for (int l = 0; l < 3; ++l) {
            uint64_t o[16];
            for (int i = 0; i < 16; ++i) {
                uint64_t acc = 0;
                const int M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
                for (int j = 0; j < 16; ++j) acc += (uint64_t)M4[i & 3][j & 3] * ((i >> 2) == (j >> 2) ? 2 : 1) * s[j];
                o[i] = acc % bb::P;
            }
            for (int i = 0; i < 16; ++i) s[i] = o[i];
        }
        for (int i = 0; i < 16; ++i) bad_host += s[i] != ra[r * 16 + i];
    }
    printf("correctness over %zu rows x 3 layers: MFMA form vs VALU form %zu mismatching words; VALU form vs host arithmetic (64 rows) %zu\n",
           rows, bad, bad_host);
    const int L = 512;
    const float ms_a = time_ms([&] { hipLaunchKernelGGL(mds_valu_kernel, dim3(blocks), dim3(kBlock), 0, 0, d_in, d_a, L); });
    const float ms_b = time_ms([&] { hipLaunchKernelGGL(mds_mfma_kernel<0>, dim3(blocks), dim3(kBlock), 0, 0, d_in, d_b, L); });
    const float ms_c = time_ms([&] { hipLaunchKernelGGL((mds_mfma_kernel<1>), dim3(blocks), dim3(kBlock), 0, 0, d_in, d_b, L); });
    const double layer_rows = (double)rows * L;
    printf("%zu rows (8 waves/SIMD), %d layers per launch\n", rows, L);
    printf("(A) VALU 64-bit accumulator form : %8.3f ms  %7.2f G row-layers/s  %6.1f cycles per wave-layer per SIMD @2.4 GHz\n", ms_a,
           layer_rows / ms_a * 1e-6, ms_a * 1e-3 * 2.4e9 * 1024 / (layer_rows / 64));
    printf("(B) MFMA i8 byte-plane form      : %8.3f ms  %7.2f G row-layers/s  %6.1f cycles per wave-layer per SIMD\n", ms_b,
           layer_rows / ms_b * 1e-6, ms_b * 1e-3 * 2.4e9 * 1024 / (layer_rows / 64));
    printf("(C) LDS layout round trip         : %8.3f ms  %7.2f G row-switch-pairs/s  %6.1f cycles per wave round trip per SIMD\n", ms_c,
           layer_rows / ms_c * 1e-6, ms_c * 1e-3 * 2.4e9 * 1024 / (layer_rows / 64));
    printf("ratio B/A = %.2f (MFMA form %s)\n", ms_b / ms_a, ms_b > ms_a ? "loses" : "wins");
    return bad || bad_host ? 2 : 0;
}
