// pw-stark v0 device kernels other than NTT and Merkle hashing: constraint/quotient
// evaluation on the extended domain, quotient chunking, openings at zeta, the DEEP
// (reduced-opening) vector, FRI folding, row gathers for query answers.
// Every kernel maps one lane to one row of a column-major matrix, so a wave reads
// 256 contiguous bytes per column; per-column coefficients (alpha^i, gamma^k) and
// constraint bytecode are wave-uniform and arrive through scalar loads.
#include "prover_internal.hpp"
#include "expr_eval.hpp"
#include "xbc.hpp"

#include <algorithm>

namespace pw {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ bb::Ext load_ext_uniform(const bb::Ext* p) { return *p; }

// ---- quotient ---------------------------------------------------------------------------
// kQRows LDE rows per lane (rows t, t + kBlock, ... of a kQRows * kBlock-row block): the constraint interpreter is
// bound by its scalar work, which is shared by the rows of a lane (xbc::eval_rows).
constexpr int kQRows = 2;
template <bool XBC>
__global__ __launch_bounds__(kBlock) void quotient_kernel(const uint32_t* __restrict__ lde, size_t N,
                                                           const uint32_t* __restrict__ bytecode,
                                                           const uint32_t* __restrict__ spans, uint32_t n_constraints,
                                                           const bb::Ext* __restrict__ alpha_pows, uint32_t zinv_even,
                                                           uint32_t zinv_odd, uint32_t* __restrict__ q, uint32_t per_chunk) {
    __shared__ uint32_t stack_lds[kStackCap * kQRows * kBlock];
    uint32_t* stk = stack_lds + threadIdx.x;
    size_t rows[kQRows];
    bool live[kQRows];
#pragma unroll
    for (int n = 0; n < kQRows; ++n) {
        const size_t j = ((size_t)blockIdx.x * kQRows + n) * kBlock + threadIdx.x;
        live[n] = j < N;
        rows[n] = live[n] ? j : 0;  // rows past the end evaluate row 0 and are not stored
    }
    if (!live[0]) return;
    bb::ExtWideAcc acc[kQRows];
    // gridDim.y > 1 (short traces with many constraints): this block folds only its chunk of the constraints and
    // leaves the partial sum in q[(4 * chunk + k) * N + j]; quotient_combine_kernel adds the chunks up
    const uint32_t c_begin = blockIdx.y * per_chunk;
    const uint32_t c_end = min(n_constraints, c_begin + per_chunk);
    for (uint32_t c = c_begin; c < c_end; ++c) {
        const uint32_t off = spans[2 * c], len = spans[2 * c + 1];
        uint32_t v[kQRows];
        if (XBC) {
            xbc::eval_rows<kBlock, true, kQRows>(bytecode + 2 * (size_t)off, len, lde, rows, stk, N, v);
        } else {
#pragma unroll
            for (int n = 0; n < kQRows; ++n) v[n] = eval_expr<kBlock, true>(bytecode + off, len, lde, rows[n], stk, N);
        }
        const bb::Ext a = alpha_pows[c];
#pragma unroll
        for (int n = 0; n < kQRows; ++n) acc[n].fma(a, v[n]);
    }
#pragma unroll
    for (int n = 0; n < kQRows; ++n) {
        if (!live[n]) continue;
        const size_t j = rows[n];
        const bb::Ext r = acc[n].result();
        if (gridDim.y > 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[((size_t)blockIdx.y * 4 + k) * N + j] = r.c[k];
        } else {
            const uint32_t zi = (j & 1) ? zinv_odd : zinv_even;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[(size_t)k * N + j] = bb::mul(r.c[k], zi);
        }
    }
}

__global__ __launch_bounds__(kBlock) void quotient_combine_kernel(const uint32_t* __restrict__ part, uint32_t n_chunks, size_t N,
                                                                   uint32_t zinv_even, uint32_t zinv_odd, uint32_t* __restrict__ q) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= N) return;
    const uint32_t zi = (j & 1) ? zinv_odd : zinv_even;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t a = 0u;
        for (uint32_t c = 0; c < n_chunks; ++c) a = bb::add(a, part[((size_t)c * 4 + k) * N + j]);
        q[(size_t)k * N + j] = bb::mul(a, zi);
    }
}

// "mock prover": evaluate every constraint on every TRACE row; first[0] <- min over violations of row * nc + c
template <bool XBC>
__global__ __launch_bounds__(kBlock) void check_constraints_kernel(const uint32_t* __restrict__ trace, size_t H,
                                                                    const uint32_t* __restrict__ bytecode,
                                                                    const uint32_t* __restrict__ spans, uint32_t n_constraints,
                                                                    unsigned long long* __restrict__ first,
                                                                    unsigned long long* __restrict__ count) {
    __shared__ uint32_t stack_lds[kStackCap * kBlock];
    uint32_t* stk = stack_lds + threadIdx.x;
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= H) return;
    for (uint32_t c = 0; c < n_constraints; ++c) {
        const uint32_t off = spans[2 * c], len = spans[2 * c + 1];
        const uint32_t v = XBC ? xbc::eval<kBlock, true>(bytecode + 2 * (size_t)off, len, trace, j, stk, H)
                               : eval_expr<kBlock, true>(bytecode + off, len, trace, j, stk, H);
        if (v != 0u) {
            atomicMin(first, (unsigned long long)j * n_constraints + c);
            atomicAdd(count, 1ull);
        }
    }
}

// out[(4*ch + k)*H + q'] = cbr[k*N + 2q' + ch] * s^-(bitrev(q') + ch*H) / 2
__global__ __launch_bounds__(kBlock) void quotient_split_kernel(const uint32_t* __restrict__ cbr, size_t H, int log_h,
                                                                 uint32_t sinv, uint32_t half_m, uint32_t* __restrict__ out) {
    const size_t qp = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (qp >= H) return;
    const uint32_t kk = log_h ? (__brev((uint32_t)qp) >> (32 - log_h)) : 0u;
    const uint32_t f_lo = bb::mul(bb::pow_u32(sinv, kk), half_m);
    const uint32_t f_hi = bb::mul(f_lo, bb::pow_u32(sinv, (uint32_t)H));
    const size_t N = 2 * H;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint2 pr = *reinterpret_cast<const uint2*>(cbr + (size_t)k * N + 2 * qp);
        out[(size_t)k * H + qp] = bb::mul(pr.x, f_lo);
        out[(size_t)(4 + k) * H + qp] = bb::mul(pr.y, f_hi);
    }
}

// ---- openings ----------------------------------------------------------------------------
// Both weight kernels store CENTRED representatives (|w_k| <= p / 2 as int32 bit patterns): the only consumer is
// ext_dot_partial_kernel, where a weight is wave-uniform and centring it there would run on the scalar unit once per row and wave.
__device__ __forceinline__ bb::Ext centred_words(const bb::Ext& e) {
    bb::Ext r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.c[k] = (uint32_t)bb::centred(e.c[k]);
    return r;
}
__global__ __launch_bounds__(kBlock) void zeta_weights_kernel(bb::Ext z, int log_h, uint32_t ninv, bb::Ext* __restrict__ w) {
    const size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= ((size_t)1 << log_h)) return;
    const uint32_t k = log_h ? (__brev((uint32_t)q) >> (32 - log_h)) : 0u;
    w[q] = centred_words(bb::ext_scale(bb::ext_pow(z, k), ninv));
}

// w[i] = scale * g^i / (zeta - g^i): Lagrange weights of the order-2^n subgroup at zeta, so that
// f(zeta) = sum_i f(g^i) w[i] for deg f < 2^n, with scale = (zeta^H - 1) / H.
__global__ __launch_bounds__(kBlock) void barycentric_weights_kernel(bb::Ext zeta, bb::Ext scale, int log_h, uint32_t g,
                                                                      bb::Ext* __restrict__ w) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= ((size_t)1 << log_h)) return;
    const uint32_t gi = bb::pow_u32(g, (uint32_t)i);
    const bb::Ext den = bb::ext_sub(zeta, bb::ext_from_base(gi));
    w[i] = centred_words(bb::ext_mul(bb::ext_scale(scale, gi), bb::ext_inv(den)));
}

constexpr int kDotRowsPerBlock = 8192;
constexpr int kDotTile = 64;  // columns per workgroup (= lanes of a wave) and rows per LDS tile
// sum_q w(q) * col_c(q) for the columns of a matrix (the openings at zeta). A lane owns a COLUMN: the weight of a row is then
// wave-uniform — a tile's 64 weights come in once (1 KB, coalesced), sit in LDS and are read back by broadcast — and is fetched
// once per 64 columns; nothing is reduced across lanes. (The first form gave a lane rows and a workgroup 4-8 columns: every group
// re-read the 16-byte weights — 28 GB of weight traffic for 23 GB of cells at C2 with LogUp, PMC r03: 51 GB fetched — and finished
// with 6 shuffle steps per sum: 11.8 ms. Lane = column with the weights through scalar loads: still 11.8 ms, the scalar path had
// become the limit; weights through LDS and the next tile prefetched into registers: 4.9 ms, 4.7 TB/s. DESIGN.md §7d.)
// The cells come in coalesced, 64 rows x 64 columns at a time, and turn through LDS (padded: conflict-free both ways).
// NW = 2: two weight vectors in one pass (the permutation matrix is opened at zeta AND at g zeta: read once instead of twice);
// partial sums of the second vector go to partial + second_off. Centred weights x centred cells in signed 64-bit accumulators,
// folded every fourth row (bb::ExtCentredAcc).
template <int NW>
__global__ __launch_bounds__(kBlock) void ext_dot_partial_kernel(const uint32_t* __restrict__ cols, size_t stride, size_t len,
                                                                  uint32_t n_cols, const bb::Ext* __restrict__ weights,
                                                                  const bb::Ext* __restrict__ weights2, bb::Ext* __restrict__ partial,
                                                                  size_t second_off, uint32_t n_chunks) {
    constexpr int kWaves = kBlock / 64, kRowsPerWave = kDotTile / kWaves;
    __shared__ uint32_t tile[kDotTile][kDotTile + 1];
    __shared__ uint4 wts[NW][kDotTile];  // the tile's weights: read back at a wave-uniform address (an LDS broadcast)
    __shared__ uint32_t red[NW][4][kWaves][kDotTile];
    const uint32_t c0 = blockIdx.y * kDotTile;
    const uint32_t nc = n_cols - c0 < (uint32_t)kDotTile ? n_cols - c0 : (uint32_t)kDotTile;  // block-uniform
    const uint32_t* col = cols + (size_t)c0 * stride;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const size_t q0 = (size_t)blockIdx.x * kDotRowsPerBlock;
    const size_t q1 = q0 + kDotRowsPerBlock < len ? q0 + kDotRowsPerBlock : len;
    bb::ExtCentredAcc acc[NW];
    // in: wave w brings columns w*16 .. w*16+15, a lane one row of each (256 contiguous bytes per wave and column), and wave v < NW the
    // tile's 64 weights of vector v (1 KB); the NEXT tile is requested before the current one is consumed (registers as the second buffer)
    uint32_t cells[kRowsPerWave];
    uint4 wq = make_uint4(0u, 0u, 0u, 0u);
    auto fetch = [&](size_t t0) {
#pragma unroll
        for (int j = 0; j < kRowsPerWave; ++j) {
            const uint32_t c = (uint32_t)(wave * kRowsPerWave + j);
            cells[j] = (c < nc && t0 + lane < q1) ? __builtin_nontemporal_load(col + (size_t)c * stride + t0 + lane) : 0u;
        }
        if (wave < NW) {
            const uint4* src = reinterpret_cast<const uint4*>(wave == 0 ? weights : weights2);
            wq = t0 + lane < q1 ? src[t0 + lane] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    fetch(q0);
    for (size_t t0 = q0; t0 < q1; t0 += kDotTile) {
#pragma unroll
        for (int j = 0; j < kRowsPerWave; ++j) tile[lane][wave * kRowsPerWave + j] = cells[j];
        if (wave < NW) wts[wave][lane] = wq;
        __syncthreads();
        if (t0 + kDotTile < q1) fetch(t0 + kDotTile);
        // out: wave w takes rows w*16 .. w*16+15 of the tile, a lane its column
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            const int row = wave * kRowsPerWave + r;  // wave-uniform; rows beyond q1 hold zero cells and zero weights
            const int32_t x = bb::centred(tile[row][lane]);
#pragma unroll
            for (int v = 0; v < NW; ++v) {
                const uint4 e = wts[v][row];  // centred already (the weight kernels above)
                const int32_t w[4] = {(int32_t)e.x, (int32_t)e.y, (int32_t)e.z, (int32_t)e.w};
                acc[v].fma(w, x);
            }
            if ((r & 3) == 3) {
#pragma unroll
                for (int v = 0; v < NW; ++v) acc[v].fold();
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < NW; ++v) {
        const bb::Ext part = acc[v].result();
#pragma unroll
        for (int k = 0; k < 4; ++k) red[v][k][wave][lane] = part.c[k];
    }
    __syncthreads();
    if (threadIdx.x < NW * kDotTile) {
        const uint32_t v = threadIdx.x / kDotTile, c = threadIdx.x % kDotTile;
        if (c < nc) {
            bb::Ext o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t s_ = red[v][k][0][c];
                for (int w = 1; w < kWaves; ++w) s_ = bb::add(s_, red[v][k][w][c]);
                o.c[k] = s_;
            }
            partial[(v ? second_off : 0) + (size_t)(c0 + c) * n_chunks + blockIdx.x] = o;
        }
    }
}
__global__ void ext_dot_final_kernel(const bb::Ext* __restrict__ partial, uint32_t n_cols, uint32_t n_chunks, bb::Ext* __restrict__ out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    bb::Ext acc = bb::ext_zero();
    for (uint32_t i = 0; i < n_chunks; ++i) acc = bb::ext_add(acc, partial[(size_t)c * n_chunks + i]);
    out[c] = acc;
}

// out[k] = base^k (reversed: base^(n - 1 - k)), k < n; centre: coordinates as CENTRED words — what the DEEP kernels take (deep_kernel
// below). Lane k multiplies the squares base^(2^i) its bits select (wave-uniform kernel arguments); field arithmetic is exact, so the
// words equal those of n - 1 successive multiplications on the host.
struct ExtSquares { bb::Ext s[24]; };
__global__ __launch_bounds__(kBlock) void ext_powers_kernel(ExtSquares g, uint32_t n, int reversed, int centre, bb::Ext* __restrict__ out) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const uint32_t e = reversed ? n - 1u - k : k;
    bb::Ext r = bb::ext_one();
#pragma unroll 1
    for (int i = 0; i < 24; ++i)
        if ((e >> i) & 1u) r = bb::ext_mul(r, g.s[i]);
    if (centre) {
#pragma unroll
        for (int c = 0; c < 4; ++c) r.c[c] = (uint32_t)bb::centred(r.c[c]);
    }
    out[k] = r;
}

// out[j.out_off + q * j.width + c] = j.m[c * j.height + idx[j.idx_off + q]]: the queried rows of MANY matrices in one launch (a
// segment's query phase: one job per AIR and tree, each far shorter than a launch takes to issue)
__global__ __launch_bounds__(kBlock) void gather_rows_multi_kernel(const GatherRowsJob* __restrict__ jobs, const uint32_t* __restrict__ idx,
                                                                    uint32_t* __restrict__ out) {
    const GatherRowsJob j = jobs[blockIdx.z];
    const size_t row = idx[j.idx_off + blockIdx.y];
    uint32_t* o = out + j.out_off + (size_t)blockIdx.y * j.width;
    for (uint32_t c = blockIdx.x * kBlock + threadIdx.x; c < j.width; c += gridDim.x * kBlock) o[c] = j.m[(size_t)c * j.height + row];
}

// ---- DEEP / reduced opening ----------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void deep_kernel(const uint32_t* __restrict__ ma, uint32_t wa,
                                                       const uint32_t* __restrict__ mb, uint32_t wb, size_t N,
                                                       const bb::Ext* __restrict__ gpow, bb::Ext opened_sum, bb::Ext zeta,
                                                       uint32_t shift, uint32_t wN, bb::Ext* __restrict__ v) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= N) return;
    // gpow[k] is wave-uniform (scalar loads) and CENTRED by the host (deep_gamma_powers); the centred column value is the only
    // vector operand of the multiply-adds; signed 64-bit accumulators folded every fourth column (bb::ExtCentredAcc)
    bb::ExtCentredAcc wide;
    const int32_t (*g)[4] = reinterpret_cast<const int32_t (*)[4]>(gpow);
    const uint32_t* pa = ma + j;
    uint32_t k = 0;
    for (; k + 4 <= wa; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) wide.fma_uniform(g[k + u], bb::centred(pa[(size_t)(k + u) * N]));
        wide.fold();
    }
    for (; k < wa; ++k) wide.fma_uniform(g[k], bb::centred(pa[(size_t)k * N]));
    wide.fold();
    const uint32_t* pb = mb + j;
    for (k = 0; k + 4 <= wb; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) wide.fma_uniform(g[wa + k + u], bb::centred(pb[(size_t)(k + u) * N]));
        wide.fold();
    }
    for (; k < wb; ++k) wide.fma_uniform(g[wa + k], bb::centred(pb[(size_t)k * N]));
    const bb::Ext acc = wide.result();
    const uint32_t xj = bb::mul(shift, bb::pow_u32(wN, (uint32_t)j));
    const bb::Ext den = bb::ext_sub(bb::ext_from_base(xj), zeta);
    v[j] = bb::ext_mul(bb::ext_sub(acc, opened_sum), bb::ext_inv(den));
}

// ---- FRI fold ------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void fri_fold_kernel(const bb::Ext* __restrict__ v, size_t half, uint32_t shift_inv_half,
                                                           uint32_t w_inv, uint32_t inv2, bb::Ext beta, bb::Ext* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= half) return;
    const bb::Ext a = v[i], b = v[i + half];
    // 1 / (2 x_i) = (1 / (2 shift)) * w^-i
    const uint32_t xinv = bb::mul(shift_inv_half, bb::pow_u32(w_inv, (uint32_t)i));
    const bb::Ext s = bb::ext_scale(bb::ext_add(a, b), inv2);
    const bb::Ext d = bb::ext_scale(bb::ext_sub(a, b), xinv);
    out[i] = bb::ext_add(s, bb::ext_mul(beta, d));
}

// w[i] <- canonical representative of the Montgomery word w[i] (query answers are converted on the device: the host would
// spend ~7 ns per proof word on it, 21 ms for the 3 M words of a reth-shaped segment's proof)
__global__ __launch_bounds__(kBlock) void canonicalize_kernel(uint32_t* __restrict__ w, size_t n) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) w[i] = bb::from_monty(w[i]);
}

// y[i] += a * x[i]
__global__ __launch_bounds__(kBlock) void ext_axpy_kernel(bb::Ext* __restrict__ y, bb::Ext a, const bb::Ext* __restrict__ x, size_t n, int a_is_one) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    y[i] = bb::ext_add(y[i], a_is_one ? x[i] : bb::ext_mul(a, x[i]));
}
// out[i] = *ptrs[i]
__global__ void gather_words_kernel(const uint32_t* const* __restrict__ ptrs, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = *ptrs[i];
}

__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const uint32_t* __restrict__ m, size_t height, uint32_t width,
                                                              const uint32_t* __restrict__ idx, uint32_t* __restrict__ out) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= width) return;
    out[(size_t)blockIdx.y * width + c] = m[(size_t)c * height + idx[blockIdx.y]];
}

}  // namespace

uint32_t quotient_chunks(size_t N, uint32_t n_constraints) {
    // enough workgroups for 256 CUs x 8 even when the trace is short; at least 32 constraints per chunk
    const unsigned row_blocks = div_up(N, kBlock * kQRows), want = 256u * 8u;
    if (row_blocks >= want || n_constraints < 64) return 1;
    uint32_t chunks = (want + row_blocks - 1) / row_blocks;
    if (chunks > n_constraints / 32) chunks = n_constraints / 32;
    if (chunks > 64) chunks = 64;
    return chunks < 1 ? 1 : chunks;
}

int quotient_eval(const uint32_t* lde, size_t N, const ConstraintProgram& prog, const bb::Ext* d_alpha_pows,
                  uint32_t zinv_even, uint32_t zinv_odd, uint32_t* q, uint32_t* part, uint32_t n_chunks) {
    ScopedKernelTimer t("quotient_kernel");
    call_stats()[kStatInterpreterKernelLaunches] += 1;
    if (n_chunks < 1 || !part) n_chunks = 1;
    const uint32_t per_chunk = n_chunks > 1 ? (prog.n_constraints + n_chunks - 1) / n_chunks : (prog.n_constraints ? prog.n_constraints : 1);
    if (n_chunks > 1) n_chunks = (prog.n_constraints + per_chunk - 1) / per_chunk;
    uint32_t* out = n_chunks > 1 ? part : q;
    const dim3 grid(div_up(N, kBlock * kQRows), n_chunks);
    if (prog.is_xbc)
        hipLaunchKernelGGL(quotient_kernel<true>, grid, dim3(kBlock), 0, stream(), lde, N, prog.d_bytecode, prog.d_spans,
                           prog.n_constraints, d_alpha_pows, zinv_even, zinv_odd, out, per_chunk);
    else
        hipLaunchKernelGGL(quotient_kernel<false>, grid, dim3(kBlock), 0, stream(), lde, N, prog.d_bytecode, prog.d_spans,
                           prog.n_constraints, d_alpha_pows, zinv_even, zinv_odd, out, per_chunk);
    if (n_chunks > 1)
        hipLaunchKernelGGL(quotient_combine_kernel, dim3(div_up(N, kBlock)), dim3(kBlock), 0, stream(), part, n_chunks, N, zinv_even,
                           zinv_odd, q);
    return (int)hipGetLastError();
}

int quotient_combine(const uint32_t* part, uint32_t n_chunks, size_t N, uint32_t zinv_even, uint32_t zinv_odd, uint32_t* q) {
    ScopedKernelTimer t("quotient_combine_kernel");
    hipLaunchKernelGGL(quotient_combine_kernel, dim3(div_up(N, kBlock)), dim3(kBlock), 0, stream(), part, n_chunks, N, zinv_even, zinv_odd, q);
    return (int)hipGetLastError();
}

int check_constraints(const uint32_t* trace, size_t H, const ConstraintProgram& prog, unsigned long long* d_first_and_count) {
    ScopedKernelTimer t("check_constraints_kernel");
    if (prog.is_xbc)
        hipLaunchKernelGGL(check_constraints_kernel<true>, dim3(div_up(H, kBlock)), dim3(kBlock), 0, stream(), trace, H,
                           prog.d_bytecode, prog.d_spans, prog.n_constraints, d_first_and_count, d_first_and_count + 1);
    else
        hipLaunchKernelGGL(check_constraints_kernel<false>, dim3(div_up(H, kBlock)), dim3(kBlock), 0, stream(), trace, H,
                           prog.d_bytecode, prog.d_spans, prog.n_constraints, d_first_and_count, d_first_and_count + 1);
    return (int)hipGetLastError();
}

int quotient_split(const uint32_t* cbr, size_t H, int log_h, uint32_t* out) {
    const uint32_t sinv = bb::inv(bb::to_monty(field::kCosetShift));
    const uint32_t half_m = bb::inv(bb::to_monty(2));
    ScopedKernelTimer t("quotient_split_kernel");
    hipLaunchKernelGGL(quotient_split_kernel, dim3(div_up(H, kBlock)), dim3(kBlock), 0, stream(), cbr, H, log_h, sinv, half_m, out);
    return (int)hipGetLastError();
}

int zeta_weights(bb::Ext z, int log_h, bb::Ext* weights) {
    const uint32_t ninv = bb::inv(bb::to_monty((uint32_t)(((uint64_t)1 << log_h) % bb::P)));
    ScopedKernelTimer t("zeta_weights_kernel");
    hipLaunchKernelGGL(zeta_weights_kernel, dim3(div_up((size_t)1 << log_h, kBlock)), dim3(kBlock), 0, stream(), z, log_h, ninv, weights);
    return (int)hipGetLastError();
}

int barycentric_weights(bb::Ext zeta, int log_h, bb::Ext* weights) {
    const size_t H = (size_t)1 << log_h;
    const uint32_t hinv = bb::inv(bb::to_monty((uint32_t)(H % bb::P)));
    const bb::Ext scale = bb::ext_scale(bb::ext_sub(bb::ext_pow(zeta, H), bb::ext_one()), hinv);
    ScopedKernelTimer t("barycentric_weights_kernel");
    hipLaunchKernelGGL(barycentric_weights_kernel, dim3(div_up(H, kBlock)), dim3(kBlock), 0, stream(), zeta, scale, log_h,
                       field::root_of_unity(log_h), weights);
    return (int)hipGetLastError();
}

int ext_dot_columns(const uint32_t* cols, size_t stride, uint32_t n_cols, size_t len, const bb::Ext* weights, bb::Ext* out,
                    bb::Ext* scratch) {
    const uint32_t n_chunks = div_up(len, kDotRowsPerBlock);
    const uint32_t max_cols = 65535u * kDotTile;
    for (uint32_t c0 = 0; c0 < n_cols; c0 += max_cols) {
        uint32_t cc = n_cols - c0 < max_cols ? n_cols - c0 : max_cols;
        ScopedKernelTimer t("ext_dot_partial_kernel");
        hipLaunchKernelGGL(ext_dot_partial_kernel<1>, dim3(n_chunks, div_up(cc, kDotTile)), dim3(kBlock), 0, stream(),
                           cols + (size_t)c0 * stride, stride, len, cc, weights, weights, scratch + (size_t)c0 * n_chunks, (size_t)0, n_chunks);
    }
    hipLaunchKernelGGL(ext_dot_final_kernel, dim3(div_up(n_cols, 256)), dim3(256), 0, stream(), scratch, n_cols, n_chunks, out);
    return (int)hipGetLastError();
}

int ext_dot_columns2(const uint32_t* cols, size_t stride, uint32_t n_cols, size_t len, const bb::Ext* weights, const bb::Ext* weights2,
                     bb::Ext* out, bb::Ext* out2, bb::Ext* scratch) {
    const uint32_t n_chunks = div_up(len, kDotRowsPerBlock);
    const uint32_t max_cols = 65535u * kDotTile;
    const size_t second = (size_t)n_cols * n_chunks;
    for (uint32_t c0 = 0; c0 < n_cols; c0 += max_cols) {
        uint32_t cc = n_cols - c0 < max_cols ? n_cols - c0 : max_cols;
        ScopedKernelTimer t("ext_dot_partial_kernel");
        hipLaunchKernelGGL(ext_dot_partial_kernel<2>, dim3(n_chunks, div_up(cc, kDotTile)), dim3(kBlock), 0, stream(),
                           cols + (size_t)c0 * stride, stride, len, cc, weights, weights2, scratch + (size_t)c0 * n_chunks, second, n_chunks);
    }
    hipLaunchKernelGGL(ext_dot_final_kernel, dim3(div_up(n_cols, 256)), dim3(256), 0, stream(), scratch, n_cols, n_chunks, out);
    hipLaunchKernelGGL(ext_dot_final_kernel, dim3(div_up(n_cols, 256)), dim3(256), 0, stream(), scratch + second, n_cols, n_chunks, out2);
    return (int)hipGetLastError();
}

int ext_powers(bb::Ext base, uint32_t n, bool reversed, bool centred, bb::Ext* d_out) {
    if (!n) return 0;
    if (n > (1u << 24)) return (int)hipErrorInvalidValue;
    ExtSquares g;
    g.s[0] = base;
    for (int i = 1; i < 24; ++i) g.s[i] = bb::ext_sqr(g.s[i - 1]);
    hipLaunchKernelGGL(ext_powers_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0, stream(), g, n, reversed ? 1 : 0, centred ? 1 : 0, d_out);
    return (int)hipGetLastError();
}

int gamma_powers(bb::Ext gamma, uint32_t K, bb::Ext* d_gpow) { return ext_powers(gamma, K, false, true, d_gpow); }

int gather_rows_multi(const GatherRowsJob* d_jobs, uint32_t n_jobs, uint32_t max_width, const uint32_t* d_indices, uint32_t n_idx, uint32_t* out) {
    if (!n_jobs || !n_idx || !max_width) return 0;
    if (n_jobs > 65535u || n_idx > 65535u) return (int)hipErrorInvalidValue;
    ScopedKernelTimer t("gather_rows_kernel");
    const uint32_t gx = std::min<uint32_t>(div_up(max_width, kBlock), 8u);
    hipLaunchKernelGGL(gather_rows_multi_kernel, dim3(gx, n_idx, n_jobs), dim3(kBlock), 0, stream(), d_jobs, d_indices, out);
    return (int)hipGetLastError();
}

int deep_quotient(const uint32_t* lde_a, uint32_t wa, const uint32_t* lde_b, uint32_t wb, size_t N, int logN,
                  const bb::Ext* d_gpow, bb::Ext opened_sum, bb::Ext zeta, bb::Ext* v) {
    ScopedKernelTimer t("deep_kernel");
    hipLaunchKernelGGL(deep_kernel, dim3(div_up(N, kBlock)), dim3(kBlock), 0, stream(), lde_a, wa, lde_b, wb, N, d_gpow,
                       opened_sum, zeta, bb::to_monty(field::kCosetShift), field::root_of_unity(logN), v);
    return (int)hipGetLastError();
}

int fri_fold(const bb::Ext* v, size_t half, int log_size, uint32_t shift, bb::Ext beta, bb::Ext* out) {
    const uint32_t inv2 = bb::inv(bb::to_monty(2));
    const uint32_t shift_inv_half = bb::mul(bb::inv(shift), inv2);
    const uint32_t w_inv = bb::inv(field::root_of_unity(log_size));
    ScopedKernelTimer t("fri_fold_kernel");
    hipLaunchKernelGGL(fri_fold_kernel, dim3(div_up(half, kBlock)), dim3(kBlock), 0, stream(), v, half, shift_inv_half, w_inv, inv2, beta, out);
    return (int)hipGetLastError();
}

int ext_axpy(bb::Ext* y, const bb::Ext* a_or_null, const bb::Ext* x, size_t n) {
    if (!n) return 0;
    ScopedKernelTimer t("ext_axpy_kernel");
    hipLaunchKernelGGL(ext_axpy_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0, stream(), y, a_or_null ? *a_or_null : bb::ext_one(), x, n,
                       a_or_null ? 0 : 1);
    return (int)hipGetLastError();
}

int canonicalize_words(uint32_t* d_words, size_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL(canonicalize_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0, stream(), d_words, n);
    return (int)hipGetLastError();
}

int gather_words(const uint32_t* const* d_ptrs, uint32_t n, uint32_t* d_out) {
    if (!n) return 0;
    hipLaunchKernelGGL(gather_words_kernel, dim3(div_up(n, 256)), dim3(256), 0, stream(), d_ptrs, n, d_out);
    return (int)hipGetLastError();
}

int gather_rows(const uint32_t* m, size_t height, uint32_t width, const uint32_t* d_indices, uint32_t n_idx, uint32_t* out) {
    if (!n_idx || !width) return 0;
    ScopedKernelTimer t("gather_rows_kernel");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(div_up(width, kBlock), n_idx), dim3(kBlock), 0, stream(), m, height, width, d_indices, out);
    return (int)hipGetLastError();
}

}  // namespace pw
