// Kernels of the STREAMED proof path (prover_stream.hpp): traces whose low-degree extension does not fit in HBM are proven from
// their coefficient arrays, one sub-coset of the extended domain at a time (ntt.hip subcoset_lde). What is here:
//   * part_scatter:   the quotient's partial sums of one sub-coset (rows r + 2^b i of the domain) -> their rows of the N-row vector
//   * ext_lincomb:    sum_k gamma^k column_k over coefficient arrays — the DEEP numerator as a POLYNOMIAL (its LDE is 4 columns,
//                     instead of a pass over the LDE of every column)
//   * deep_from_combo: the reduced-opening vector from those combinations
//   * ext_to_cols:    an Ext vector as four base columns
// Lane = row everywhere; gamma powers are wave-uniform (scalar loads).
#include "prover_internal.hpp"

namespace pw {

namespace {

constexpr int kBlock = 256;
using bb::Ext;

// out[k * N + r + (i << b)] = sum_c part[(c * 4 + k) * m + i]  (canonical words; the chunks' sums stay below 2^64)
__global__ __launch_bounds__(kBlock) void part_scatter_kernel(const uint32_t* __restrict__ part, uint32_t n_chunks, size_t m, int b,
                                                               uint32_t r, size_t N, uint32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const size_t j = r + (i << b);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t a = 0;
        for (uint32_t c = 0; c < n_chunks; ++c) a += part[((size_t)c * 4 + k) * m + i];
        out[(size_t)k * N + j] = (uint32_t)(a % bb::P);
    }
}

// out[k * len + q] (k < 4) = coordinate k of sum_{c < wa} g[c] ma_c[q] + sum_{c < wb} g[wa + c] mb_c[q]
// out[(4 + k) * len + q]   = coordinate k of sum_{c < wb} g[second + c] mb_c[q]                          (second != 0 only)
// g: CENTRED gamma powers (the host centres them for the DEEP kernels); cells are Montgomery words, centred here; signed 64-bit
// accumulators folded every fourth column (bb::ExtCentredAcc) — the inner loops of deep_logup_kernel over `len` rows of
// arbitrary matrices. A lane owns TWO neighbouring rows (one 8-byte load per column: half the load instructions, 512 contiguous
// bytes per wave and column); len is even (a power of two >= 2).
__global__ __launch_bounds__(kBlock) void ext_lincomb_kernel(const uint32_t* __restrict__ ma, uint32_t wa, const uint32_t* __restrict__ mb,
                                                              uint32_t wb, size_t len, const Ext* __restrict__ gpow, uint32_t second,
                                                              uint32_t* __restrict__ out) {
    const size_t q = 2 * ((size_t)blockIdx.x * kBlock + threadIdx.x);
    if (q >= len) return;
    bb::ExtCentredAcc w1[2], w2[2];
    const int32_t (*g)[4] = reinterpret_cast<const int32_t (*)[4]>(gpow);
    auto ld2 = [&](const uint32_t* m, uint32_t c) {  // (one 8-byte, non-temporal load: the matrices are read once)
        const unsigned long long w = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(m + (size_t)c * len + q));
        return make_uint2((uint32_t)w, (uint32_t)(w >> 32));
    };
    uint32_t k = 0;
    for (; k + 4 <= wa; k += 4) {
        uint2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld2(ma, k + u);
#pragma unroll
        for (int u = 0; u < 4; ++u) { w1[0].fma_uniform(g[k + u], bb::centred(v[u].x)); w1[1].fma_uniform(g[k + u], bb::centred(v[u].y)); }
        w1[0].fold(); w1[1].fold();
    }
    for (; k < wa; ++k) { const uint2 v = ld2(ma, k); w1[0].fma_uniform(g[k], bb::centred(v.x)); w1[1].fma_uniform(g[k], bb::centred(v.y)); }
    w1[0].fold(); w1[1].fold();
    for (k = 0; k + 4 <= wb; k += 4) {
        uint2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld2(mb, k + u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t x0 = bb::centred(v[u].x), x1 = bb::centred(v[u].y);
            w1[0].fma_uniform(g[wa + k + u], x0); w1[1].fma_uniform(g[wa + k + u], x1);
            if (second) { w2[0].fma_uniform(g[second + k + u], x0); w2[1].fma_uniform(g[second + k + u], x1); }
        }
        w1[0].fold(); w1[1].fold();
        w2[0].fold(); w2[1].fold();
    }
    for (; k < wb; ++k) {
        const uint2 v = ld2(mb, k);
        const int32_t x0 = bb::centred(v.x), x1 = bb::centred(v.y);
        w1[0].fma_uniform(g[wa + k], x0); w1[1].fma_uniform(g[wa + k], x1);
        if (second) { w2[0].fma_uniform(g[second + k], x0); w2[1].fma_uniform(g[second + k], x1); }
    }
    const Ext a10 = w1[0].result(), a11 = w1[1].result();
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<uint2*>(out + (size_t)c * len + q) = make_uint2(a10.c[c], a11.c[c]);
    if (second) {
        const Ext a20 = w2[0].result(), a21 = w2[1].result();
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<uint2*>(out + (size_t)(4 + c) * len + q) = make_uint2(a20.c[c], a21.c[c]);
    }
}

// v[j] = (G1[j] + sum_{k < 8} gq[k] Q_k[j] - sum1) / (x_j - zeta) [+ (G2[j] - sum2) / (x_j - g zeta)]
// glde: 8 columns of N (G1's coordinates, then G2's); gq: the (centred) gamma powers of the eight quotient columns
__global__ __launch_bounds__(kBlock) void deep_from_combo_kernel(const uint32_t* __restrict__ glde, const uint32_t* __restrict__ qlde, size_t N,
                                                                  const Ext* __restrict__ gq, Ext sum1, Ext sum2, Ext zeta, Ext gzeta, int two,
                                                                  uint32_t shift, uint32_t wN, Ext* __restrict__ v) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= N) return;
    bb::ExtCentredAcc w;
    const int32_t (*g)[4] = reinterpret_cast<const int32_t (*)[4]>(gq);
    for (int k = 0; k < 8; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) w.fma_uniform(g[k + u], bb::centred(qlde[(size_t)(k + u) * N + j]));
        w.fold();
    }
    const Ext g1 = {{glde[j], glde[N + j], glde[2 * N + j], glde[3 * N + j]}};
    const Ext a1 = bb::ext_add(w.result(), g1);
    const uint32_t xj = bb::mul(shift, bb::pow_u32(wN, (uint32_t)j));
    const Ext xe = bb::ext_from_base(xj);
    Ext t = bb::ext_mul(bb::ext_sub(a1, sum1), bb::ext_inv(bb::ext_sub(xe, zeta)));
    if (two) {
        const Ext a2 = {{glde[4 * N + j], glde[5 * N + j], glde[6 * N + j], glde[7 * N + j]}};
        t = bb::ext_add(t, bb::ext_mul(bb::ext_sub(a2, sum2), bb::ext_inv(bb::ext_sub(xe, gzeta))));
    }
    v[j] = t;
}

__global__ __launch_bounds__(kBlock) void ext_to_cols_kernel(const Ext* __restrict__ in, size_t len, uint32_t* __restrict__ cols4) {
    const size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= len) return;
    const Ext e = in[q];
#pragma unroll
    for (int k = 0; k < 4; ++k) cols4[(size_t)k * len + q] = e.c[k];
}

}  // namespace

int part_scatter(const uint32_t* part, uint32_t n_chunks, size_t m, int b, uint32_t r, size_t N, uint32_t* out) {
    ScopedKernelTimer t("quotient_part_scatter_kernel");
    hipLaunchKernelGGL(part_scatter_kernel, dim3(div_up(m, kBlock)), dim3(kBlock), 0, stream(), part, n_chunks, m, b, r, N, out);
    return (int)hipGetLastError();
}

int ext_lincomb(const uint32_t* ma, uint32_t wa, const uint32_t* mb, uint32_t wb, size_t len, const bb::Ext* d_gpow, uint32_t second,
                uint32_t* out) {
    ScopedKernelTimer t("ext_lincomb_kernel");
    if (len & 1) return (int)hipErrorInvalidValue;  // (traces have at least two rows)
    // the kernel reads two words per load: the matrices (and with an even column length every column of them) must be 8-byte aligned —
    // the resident provers fall back to deep_quotient* for a caller's trace that is not, the streamed ones refuse it
    if ((((uintptr_t)ma) | ((uintptr_t)mb) | ((uintptr_t)out)) & 7) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(ext_lincomb_kernel, dim3(div_up(len / 2, kBlock)), dim3(kBlock), 0, stream(), ma, wa, mb, wb, len, d_gpow, second, out);
    return (int)hipGetLastError();
}

int deep_from_combo(const uint32_t* glde, const uint32_t* qlde, size_t N, int logN, const bb::Ext* d_gpow_quotient, bb::Ext sum1, bb::Ext sum2,
                    bb::Ext zeta, bb::Ext gzeta, int two, bb::Ext* v) {
    ScopedKernelTimer t("deep_from_combo_kernel");
    hipLaunchKernelGGL(deep_from_combo_kernel, dim3(div_up(N, kBlock)), dim3(kBlock), 0, stream(), glde, qlde, N, d_gpow_quotient, sum1, sum2, zeta,
                       gzeta, two, bb::to_monty(field::kCosetShift), field::root_of_unity(logN), v);
    return (int)hipGetLastError();
}

int ext_to_cols(const bb::Ext* in, size_t len, uint32_t* cols4) {
    hipLaunchKernelGGL(ext_to_cols_kernel, dim3(div_up(len, kBlock)), dim3(kBlock), 0, stream(), in, len, cols4);
    return (int)hipGetLastError();
}

}  // namespace pw
