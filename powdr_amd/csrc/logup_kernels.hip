// Kernels of the LogUp extension of pw-stark v0 (protocol: oracle/stark_oracle.cpp, "pw-stark v0 + LogUp"):
// permutation-trace generation, running-sum scan, the extended quotient and DEEP kernels.
// One lane per row everywhere; interaction programs are xbc code with column-index operands.
#include "prover_internal.hpp"
#include "xbc.hpp"
#include "expr_eval.hpp"
#include "jit_device.hpp"

namespace pw {

namespace {

constexpr int kBlock = 256;
using bb::Ext;

// d_i = al + bus_i + sum_j bl^(j+1) a_ij on row r of matrix `m` (column stride `stride`).
// Every coordinate is a sum of products accumulated RAW in a signed 64-bit register (one v_mad_i64_i32 per term) and reduced
// by signed Montgomery reductions (bb::smont, no conditional subtraction): the challenge powers are wave-uniform, so their
// centred representatives (|b| <= p/2) come from the scalar unit for free, an argument a is a canonical word, a product is
// below p^2 / 2 and two of them fit the reduction's domain (1.209 p^2) on top of what is already there (<= 0.134 p^2);
// after two products the accumulator is reduced and re-enters as r * (R mod p). Per coordinate and argument ~3.5
// instructions where a Montgomery product and a modular addition took 8.
// One multiplicity / argument on row r. FAST: the span's small form (k0 + k1 A + k2 B + k3 A B, fixed code, both loads issued at
// once, ~10 scalar instructions); else the xbc interpreter (a scalar decode of ~15 instructions per xbc instruction on the
// CU's one scalar unit, which is what bounds these kernels when it runs: PMC profiles/r02_pmc_logup_kernels.txt).
template <bool FAST>
__device__ __forceinline__ uint32_t eval_span(const LogupProgram& lp, uint32_t span, const uint32_t* __restrict__ m, size_t stride,
                                              size_t r, uint32_t* stk) {
    if (FAST && !(lp.d_forms[span].flags & SmallForm::NOT_SMALL)) {
        const SmallForm f = lp.d_forms[span];
        const uint32_t ta = (f.flags & SmallForm::USES_A) ? m[(size_t)f.a * stride + r] : 0u;
        const uint32_t tb = (f.flags & SmallForm::USES_B) ? m[(size_t)f.b * stride + r] : 0u;
        return f.eval(ta, tb);
    }
    const uint32_t off = lp.d_xspans[2 * span], len = lp.d_xspans[2 * span + 1];
    return xbc::eval<kBlock, true>(lp.d_code + 2 * (size_t)off, len, m, r, stk, stride);
}

using pwj::DenominatorSeeds;
using pwj::denominator_seeds;
// (the accumulation itself lives in jit_device.hpp, shared with the run-time specialised kernels)
template <bool FAST>
__device__ __forceinline__ Ext interaction_denominator(const LogupInteraction& it, const LogupProgram& lp, const uint32_t* __restrict__ m,
                                                       size_t stride, size_t r, uint32_t* stk, const DenominatorSeeds& sd,
                                                       const Ext* __restrict__ blpow) {
    pwj::DenominatorAcc acc(sd, it.bus_monty);
    for (uint32_t j = 0; j < it.n_args; ++j) acc.add(eval_span<FAST>(lp, it.first_span + 1 + j, m, stride, r, stk), blpow[j + 1]);
    return acc.result();
}
template <bool FAST>
__device__ __forceinline__ uint32_t interaction_mult(const LogupInteraction& it, const LogupProgram& lp, const uint32_t* __restrict__ m,
                                                     size_t stride, size_t r, uint32_t* stk) {
    return eval_span<FAST>(lp, it.first_span, m, stride, r, stk);
}

// perm[(4g+k)*H + r] = coordinate k of q_g(r) = sum_{i in g} m_i(r) / d_i(r);  rowsum[r] = sum_g q_g(r).
// Only interactions with a non-zero multiplicity on the row contribute (padding rows cost no inversion at all); the
// active ones of a group share one inversion: q = num / den, (num, den) <- (num d_i + m_i den, den d_i).
template <bool FAST>
__global__ __launch_bounds__(kBlock) void logup_perm_kernel(const uint32_t* __restrict__ trace, size_t H, LogupProgram lp, Ext al,
                                                             const Ext* __restrict__ blpow, uint32_t* __restrict__ perm,
                                                             Ext* __restrict__ rowsum) {
    __shared__ uint32_t stack_lds[kStackCap * kBlock];
    uint32_t* stk = stack_lds + threadIdx.x;
    const size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= H) return;
    const DenominatorSeeds sd = denominator_seeds(al);
    Ext acc = bb::ext_zero();
    for (uint32_t g = 0; g < lp.n_groups; ++g) {
        Ext num = bb::ext_zero(), den = bb::ext_one();
        bool any = false;
        for (uint32_t i = lp.d_gstarts[g]; i < lp.d_gstarts[g + 1]; ++i) {
            const LogupInteraction it = lp.d_inter[i];
            const uint32_t m = interaction_mult<FAST>(it, lp, trace, H, r, stk);
            if (m == 0u) continue;
            const Ext d = interaction_denominator<FAST>(it, lp, trace, H, r, stk, sd, blpow);
            if (any) {
                num = bb::ext_add(bb::ext_mul(num, d), bb::ext_scale(den, m));
                den = bb::ext_mul(den, d);
            } else {
                num = bb::ext_from_base(m);
                den = d;
                any = true;
            }
        }
        const Ext q = any ? bb::ext_mul(num, bb::ext_inv(den)) : bb::ext_zero();
        uint32_t* out = perm + (size_t)(4 * g) * H + r;
        out[0] = q.c[0]; out[H] = q.c[1]; out[2 * H] = q.c[2]; out[3 * H] = q.c[3];
        acc = bb::ext_add(acc, q);
    }
    rowsum[r] = acc;
}

// ---- inclusive scan of an Ext vector (length H, power of two) into the 4 phi columns -------------------
constexpr int kScanPerThread = 16;
constexpr int kScanChunk = kBlock * kScanPerThread;  // rows per workgroup

__device__ __forceinline__ Ext block_exclusive_scan(Ext v, Ext* lds, Ext& total) {
    // Hillis-Steele over 256 threads in LDS
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        Ext x = lds[t];
        if (t >= off) x = bb::ext_add(x, lds[t - off]);
        __syncthreads();
        lds[t] = x;
        __syncthreads();
    }
    total = lds[kBlock - 1];
    Ext incl = lds[t];
    __syncthreads();
    return bb::ext_sub(incl, v);
}

__global__ __launch_bounds__(kBlock) void scan_block_totals_kernel(const Ext* __restrict__ in, size_t H, Ext* __restrict__ totals) {
    __shared__ Ext lds[kBlock];
    const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanPerThread;
    Ext s = bb::ext_zero();
    for (int k = 0; k < kScanPerThread; ++k)
        if (base + k < H) s = bb::ext_add(s, in[base + k]);
    Ext total;
    (void)block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}
__global__ void scan_totals_kernel(Ext* totals, uint32_t n) {  // tiny: n <= 2^22 / 4096 = 1024 blocks
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        Ext run = bb::ext_zero();
        for (uint32_t i = 0; i < n; ++i) { Ext t = totals[i]; totals[i] = run; run = bb::ext_add(run, t); }
    }
}
__global__ __launch_bounds__(kBlock) void scan_write_kernel(const Ext* __restrict__ in, size_t H, const Ext* __restrict__ offsets,
                                                             uint32_t* __restrict__ phi_cols /* 4 columns of H */) {
    __shared__ Ext lds[kBlock];
    const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanPerThread;
    Ext s = bb::ext_zero();
    for (int k = 0; k < kScanPerThread; ++k)
        if (base + k < H) s = bb::ext_add(s, in[base + k]);
    Ext total;
    Ext run = bb::ext_add(block_exclusive_scan(s, lds, total), offsets[blockIdx.x]);
    for (int k = 0; k < kScanPerThread; ++k) {
        if (base + k < H) {
            run = bb::ext_add(run, in[base + k]);
            phi_cols[base + k] = run.c[0]; phi_cols[H + base + k] = run.c[1];
            phi_cols[2 * H + base + k] = run.c[2]; phi_cols[3 * H + base + k] = run.c[3];
        }
    }
}

// ---- quotient with the LogUp constraints ---------------------------------------------------------------
// acc = sum_k apow[k] C_k + sum_g apow[nc+g] (q_g den_g - num_g) + apow[nc+G] is_first (phi - sum q)
//       + apow[nc+G+1] is_trans (phi' - phi - sum q') + apow[nc+G+2] is_last (phi - S);  q = acc / Z_H
// with den_g = prod_{i in g} d_i, num_g = sum_{i in g} m_i prod_{j != i} d_j, G = number of groups
template <bool XBC, bool FAST>
__global__ __launch_bounds__(kBlock) void quotient_logup_kernel(const uint32_t* __restrict__ lde, const uint32_t* __restrict__ plde,
                                                                 size_t N, const uint32_t* __restrict__ bytecode,
                                                                 const uint32_t* __restrict__ spans, uint32_t nc, LogupProgram lp,
                                                                 const Ext* __restrict__ apow, Ext al, const Ext* __restrict__ blpow,
                                                                 Ext S, uint32_t zval_even, uint32_t zval_odd, uint32_t shift,
                                                                 uint32_t wN, uint32_t ginv, uint32_t* __restrict__ q, int main_only) {
    __shared__ uint32_t stack_lds[kStackCap * kBlock];
    uint32_t* stk = stack_lds + threadIdx.x;
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= N) return;
    const size_t jn = (j + 2) & (N - 1);
    const DenominatorSeeds sd = denominator_seeds(al);
    // sum_k apow[k] * (constraint or group value): raw products in 96-bit accumulators, reduced once per row
    bb::ExtProductAcc wide;
    for (uint32_t c = 0; c < nc; ++c) {
        const uint32_t off = spans[2 * c], len = spans[2 * c + 1];
        const uint32_t v = XBC ? xbc::eval<kBlock, true>(bytecode + 2 * (size_t)off, len, lde, j, stk, N)
                               : eval_expr<kBlock, true>(bytecode + off, len, lde, j, stk, N);
        wide.fma_base(apow[c], v);
    }
    Ext sumq = bb::ext_zero(), sumq_next = bb::ext_zero();
    for (uint32_t g = 0; g < lp.n_groups; ++g) {
        const uint32_t* pc = plde + (size_t)(4 * g) * N;
        const Ext qi = {{pc[j], pc[N + j], pc[2 * N + j], pc[3 * N + j]}};
        if (!main_only) {
            const Ext qn = {{pc[jn], pc[N + jn], pc[2 * N + jn], pc[3 * N + jn]}};
            sumq = bb::ext_add(sumq, qi);
            sumq_next = bb::ext_add(sumq_next, qn);
        }
        const uint32_t i0 = lp.d_gstarts[g], i1 = lp.d_gstarts[g + 1];
        Ext num, den;
        for (uint32_t i = i0; i < i1; ++i) {
            const LogupInteraction it = lp.d_inter[i];
            const Ext d = interaction_denominator<FAST>(it, lp, lde, N, j, stk, sd, blpow);
            const uint32_t m = interaction_mult<FAST>(it, lp, lde, N, j, stk);
            if (i == i0) {
                num = bb::ext_from_base(m);
                den = d;
            } else {
                num = bb::ext_add(bb::ext_mul(num, d), bb::ext_scale(den, m));
                den = bb::ext_mul(den, d);
            }
        }
        wide.fma(apow[nc + g], bb::ext_sub(bb::ext_mul(qi, den), num));
    }
    Ext acc = wide.result();
    if (main_only) {  // the streamed path: `lde` / `plde` hold one sub-coset, the boundary terms are added by quotient_logup_tail_kernel
#pragma unroll
        for (int k = 0; k < 4; ++k) q[(size_t)k * N + j] = acc.c[k];
        return;
    }
    const uint32_t* pp = plde + (size_t)(4 * lp.n_groups) * N;
    const Ext phi = {{pp[j], pp[N + j], pp[2 * N + j], pp[3 * N + j]}};
    const Ext phin = {{pp[jn], pp[N + jn], pp[2 * N + jn], pp[3 * N + jn]}};
    const uint32_t x = bb::mul(shift, bb::pow_u32(wN, (uint32_t)j));
    const uint32_t Z = (j & 1) ? zval_odd : zval_even;
    const uint32_t one = bb::R_MOD_P;
    const uint32_t is_first = bb::mul(Z, bb::inv(bb::sub(x, one)));
    const uint32_t is_last = bb::mul(Z, bb::inv(bb::sub(x, ginv)));
    const uint32_t is_trans = bb::sub(x, ginv);
    acc = bb::ext_add(acc, bb::ext_mul(apow[nc + lp.n_groups], bb::ext_scale(bb::ext_sub(phi, sumq), is_first)));
    acc = bb::ext_add(acc, bb::ext_mul(apow[nc + lp.n_groups + 1], bb::ext_scale(bb::ext_sub(bb::ext_sub(phin, phi), sumq_next), is_trans)));
    acc = bb::ext_add(acc, bb::ext_mul(apow[nc + lp.n_groups + 2], bb::ext_scale(bb::ext_sub(phi, S), is_last)));
    const uint32_t zi = bb::inv(Z);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[(size_t)k * N + j] = bb::mul(acc.c[k], zi);
}

// ---- the same quotient when the constraint / group terms come from run-time specialised kernels (jit_codegen.hpp) -----------
// part[(c * 4 + k) * N + j]: the partial sums of the chunks; the boundary terms need sum_g q_g at rows j and j + 2: by linearity
// of the LDE that is the LDE of the per-row sums the permutation kernels computed anyway (4 extra columns next to the
// committed ones, `plde_sumq`), so the perm matrix is read once per group and not a second time at the next row.
__global__ __launch_bounds__(kBlock) void quotient_logup_tail_kernel(const uint32_t* __restrict__ part, uint32_t n_chunks,
                                                                      const uint32_t* __restrict__ plde_phi, const uint32_t* __restrict__ plde_sumq,
                                                                      size_t N, const Ext* __restrict__ apow_tail, Ext S, uint32_t zval_even,
                                                                      uint32_t zval_odd, uint32_t shift, uint32_t wN, uint32_t ginv,
                                                                      uint32_t* __restrict__ q) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= N) return;
    const size_t jn = (j + 2) & (N - 1);
    Ext acc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t a = 0;  // < 2^32 chunks of canonical words
        for (uint32_t c = 0; c < n_chunks; ++c) a += part[((size_t)c * 4 + k) * N + j];
        acc.c[k] = (uint32_t)(a % bb::P);
    }
    const Ext phi = {{plde_phi[j], plde_phi[N + j], plde_phi[2 * N + j], plde_phi[3 * N + j]}};
    const Ext phin = {{plde_phi[jn], plde_phi[N + jn], plde_phi[2 * N + jn], plde_phi[3 * N + jn]}};
    const Ext sumq = {{plde_sumq[j], plde_sumq[N + j], plde_sumq[2 * N + j], plde_sumq[3 * N + j]}};
    const Ext sumq_next = {{plde_sumq[jn], plde_sumq[N + jn], plde_sumq[2 * N + jn], plde_sumq[3 * N + jn]}};
    const uint32_t x = bb::mul(shift, bb::pow_u32(wN, (uint32_t)j));
    const uint32_t Z = (j & 1) ? zval_odd : zval_even;
    const uint32_t one = bb::R_MOD_P;
    const uint32_t is_first = bb::mul(Z, bb::inv(bb::sub(x, one)));
    const uint32_t is_last = bb::mul(Z, bb::inv(bb::sub(x, ginv)));
    const uint32_t is_trans = bb::sub(x, ginv);
    acc = bb::ext_add(acc, bb::ext_mul(apow_tail[0], bb::ext_scale(bb::ext_sub(phi, sumq), is_first)));
    acc = bb::ext_add(acc, bb::ext_mul(apow_tail[1], bb::ext_scale(bb::ext_sub(bb::ext_sub(phin, phi), sumq_next), is_trans)));
    acc = bb::ext_add(acc, bb::ext_mul(apow_tail[2], bb::ext_scale(bb::ext_sub(phi, S), is_last)));
    const uint32_t zi = bb::inv(Z);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[(size_t)k * N + j] = bb::mul(acc.c[k], zi);
}

// rowsum[r] = sum over the chunks of a specialised permutation kernel's partial row sums; also as four columns (the input of
// the extra LDE columns quotient_logup_tail_kernel reads)
__global__ __launch_bounds__(kBlock) void rowsum_combine_kernel(const uint32_t* __restrict__ part, uint32_t n_chunks, size_t H,
                                                                 Ext* __restrict__ rowsum, uint32_t* __restrict__ cols4) {
    const size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= H) return;
    Ext s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t a = 0;
        for (uint32_t c = 0; c < n_chunks; ++c) a += part[((size_t)c * 4 + k) * H + r];
        s.c[k] = (uint32_t)(a % bb::P);
        cols4[(size_t)k * H + r] = s.c[k];
    }
    rowsum[r] = s;
}

// ---- DEEP with two opening points ------------------------------------------------------------------------
// v[j] = (sum_{k<K1} g^k f_k(x_j) - sum1) / (x_j - zeta) + (sum_{k<Wp} g^(K1+k) p_k(x_j) - sum2) / (x_j - g zeta)
__global__ __launch_bounds__(kBlock) void deep_logup_kernel(const uint32_t* __restrict__ lde, uint32_t W, const uint32_t* __restrict__ plde,
                                                             uint32_t Wp, const uint32_t* __restrict__ qlde, size_t N,
                                                             const Ext* __restrict__ gpow, Ext sum1, Ext sum2, Ext zeta, Ext gzeta,
                                                             uint32_t shift, uint32_t wN, Ext* __restrict__ v) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= N) return;
    // centred gamma powers (host) x centred cells in signed 64-bit accumulators, folded every fourth column (bb::ExtCentredAcc)
    bb::ExtCentredAcc w1, w2;
    const int32_t (*g)[4] = reinterpret_cast<const int32_t (*)[4]>(gpow);
    const uint32_t K1 = W + Wp + 8;
    uint32_t k = 0;
    for (; k + 4 <= W; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) w1.fma_uniform(g[k + u], bb::centred(lde[(size_t)(k + u) * N + j]));
        w1.fold();
    }
    for (; k < W; ++k) w1.fma_uniform(g[k], bb::centred(lde[(size_t)k * N + j]));
    w1.fold();
    for (k = 0; k + 4 <= Wp; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t x = bb::centred(plde[(size_t)(k + u) * N + j]);
            w1.fma_uniform(g[W + k + u], x);
            w2.fma_uniform(g[K1 + k + u], x);
        }
        w1.fold();
        w2.fold();
    }
    for (; k < Wp; ++k) {
        const int32_t x = bb::centred(plde[(size_t)k * N + j]);
        w1.fma_uniform(g[W + k], x);
        w2.fma_uniform(g[K1 + k], x);
    }
    w1.fold();
    w2.fold();
    for (k = 0; k < 8; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) w1.fma_uniform(g[W + Wp + k + u], bb::centred(qlde[(size_t)(k + u) * N + j]));
        w1.fold();
    }
    const Ext a1 = w1.result(), a2 = w2.result();
    const uint32_t xj = bb::mul(shift, bb::pow_u32(wN, (uint32_t)j));
    const Ext xe = bb::ext_from_base(xj);
    const Ext t1 = bb::ext_mul(bb::ext_sub(a1, sum1), bb::ext_inv(bb::ext_sub(xe, zeta)));
    const Ext t2 = bb::ext_mul(bb::ext_sub(a2, sum2), bb::ext_inv(bb::ext_sub(xe, gzeta)));
    v[j] = bb::ext_add(t1, t2);
}

}  // namespace

int logup_scan(const bb::Ext* d_rowsum, size_t H, bb::Ext* d_block_totals, uint32_t* phi_cols) {
    const uint32_t blocks = div_up(H, kScanChunk);
    ScopedKernelTimer t("logup_scan_kernels");
    hipLaunchKernelGGL(scan_block_totals_kernel, dim3(blocks), dim3(kBlock), 0, stream(), d_rowsum, H, d_block_totals);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(64), 0, stream(), d_block_totals, blocks);
    hipLaunchKernelGGL(scan_write_kernel, dim3(blocks), dim3(kBlock), 0, stream(), d_rowsum, H, d_block_totals, phi_cols);
    return (int)hipGetLastError();
}

int logup_perm_trace(const uint32_t* trace, size_t H, const LogupProgram& lp, bb::Ext al, const bb::Ext* d_blpow, uint32_t* perm,
                     bb::Ext* d_rowsum, bb::Ext* d_block_totals) {
    {
        ScopedKernelTimer t("logup_perm_kernel");
        call_stats()[kStatInterpreterKernelLaunches] += 1;
        if (lp.d_forms) hipLaunchKernelGGL(logup_perm_kernel<true>, dim3(div_up(H, kBlock)), dim3(kBlock), 0, stream(), trace, H, lp, al, d_blpow, perm, d_rowsum);
        else hipLaunchKernelGGL(logup_perm_kernel<false>, dim3(div_up(H, kBlock)), dim3(kBlock), 0, stream(), trace, H, lp, al, d_blpow, perm, d_rowsum);
    }
    return logup_scan(d_rowsum, H, d_block_totals, perm + (size_t)(4 * lp.n_groups) * H);
}

int logup_rowsum_combine(const uint32_t* part, uint32_t n_chunks, size_t H, bb::Ext* d_rowsum, uint32_t* cols4) {
    ScopedKernelTimer t("logup_rowsum_combine_kernel");
    hipLaunchKernelGGL(rowsum_combine_kernel, dim3(div_up(H, kBlock)), dim3(kBlock), 0, stream(), part, n_chunks, H, d_rowsum, cols4);
    return (int)hipGetLastError();
}

int quotient_logup_tail(const uint32_t* part, uint32_t n_chunks, const uint32_t* plde_phi, const uint32_t* plde_sumq, size_t N, int logN,
                        const bb::Ext* d_apow_tail, bb::Ext S, uint32_t zval_even, uint32_t zval_odd, uint32_t* q) {
    const uint32_t shift = bb::to_monty(field::kCosetShift), wN = field::root_of_unity(logN);
    const uint32_t ginv = bb::inv(field::root_of_unity(logN - 1));
    ScopedKernelTimer t("quotient_logup_tail_kernel");
    hipLaunchKernelGGL(quotient_logup_tail_kernel, dim3(div_up(N, kBlock)), dim3(kBlock), 0, stream(), part, n_chunks, plde_phi, plde_sumq, N,
                       d_apow_tail, S, zval_even, zval_odd, shift, wN, ginv, q);
    return (int)hipGetLastError();
}

int quotient_eval_logup(const uint32_t* lde, const uint32_t* plde, size_t N, int logN, const ConstraintProgram& prog,
                        const LogupProgram& lp, const bb::Ext* d_apow, bb::Ext al, const bb::Ext* d_blpow, bb::Ext S,
                        uint32_t zval_even, uint32_t zval_odd, uint32_t* q, bool main_only) {
    const uint32_t shift = bb::to_monty(field::kCosetShift), wN = field::root_of_unity(logN);
    const uint32_t ginv = bb::inv(field::root_of_unity(logN - 1));
    const int mo = main_only ? 1 : 0;
    ScopedKernelTimer t("quotient_logup_kernel");
    call_stats()[kStatInterpreterKernelLaunches] += 1;
#define PW_LAUNCH_QL(X, F) hipLaunchKernelGGL((quotient_logup_kernel<X, F>), dim3(div_up(N, kBlock)), dim3(kBlock), 0, stream(), lde, plde, N, \
                                            prog.d_bytecode, prog.d_spans, prog.n_constraints, lp, d_apow, al, d_blpow, S, zval_even, zval_odd, shift, wN, ginv, q, mo)
    if (prog.is_xbc) { if (lp.d_forms) PW_LAUNCH_QL(true, true); else PW_LAUNCH_QL(true, false); }
    else { if (lp.d_forms) PW_LAUNCH_QL(false, true); else PW_LAUNCH_QL(false, false); }
#undef PW_LAUNCH_QL
    return (int)hipGetLastError();
}

int deep_quotient_logup(const uint32_t* lde, uint32_t W, const uint32_t* plde, uint32_t Wp, const uint32_t* qlde, size_t N, int logN,
                        const bb::Ext* d_gpow, bb::Ext sum1, bb::Ext sum2, bb::Ext zeta, bb::Ext gzeta, bb::Ext* v) {
    ScopedKernelTimer t("deep_logup_kernel");
    hipLaunchKernelGGL(deep_logup_kernel, dim3(div_up(N, kBlock)), dim3(kBlock), 0, stream(), lde, W, plde, Wp, qlde, N, d_gpow, sum1,
                       sum2, zeta, gzeta, bb::to_monty(field::kCosetShift), field::root_of_unity(logN), v);
    return (int)hipGetLastError();
}

}  // namespace pw
