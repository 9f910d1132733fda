// Radix-2 NTT / low-degree extension over BabyBear for gfx950, LDS-staged.
//
// Column-major matrices of 2^n-row columns; every column is an independent
// transform (blockIdx.y = column). A transform is cut into "stage groups": one
// kernel launch runs k consecutive butterfly stages on tiles of 2^(k+c) elements
// held in LDS, so the data crosses HBM once per group instead of once per stage.
// A tile gathers the 2^k elements whose indices differ in the group's k active bits,
// times 2^c neighbouring columns-of-the-4-step-matrix (c low index bits) so that
// every global access is a >= 64-byte contiguous segment.
//
//   inverse  = DIF (Gentleman-Sande), natural-order input -> bit-reversed output
//   forward  = DIT (Cooley-Tukey),    bit-reversed input  -> natural-order output
// so iNTT followed by the coset NTT needs no bit-reversal pass. Zero-padding the
// coefficient vector from H to 2H in bit-reversed order is a duplication
// (padded[2q] = c[q], padded[2q+1] = 0, and the first DIT stage maps (a, 0) to
// (a, a)), which the expand kernel fuses with the coset scaling s^k / H.
//
// LDE definition (natural order, as oracle/stark_oracle.cpp `lde_column`):
//   L[j] = T(s * g_{n+1}^j),  j < 2H,  s = 31.
#include "babybear.hpp"
#include "common.hpp"
#include "prover_internal.hpp"

#include <map>
#include <mutex>
#include <vector>

namespace pw {

namespace {

constexpr int kBlock = 256;
constexpr int kTileLog = 13;        // 8192 elements = 32 KB of LDS per workgroup
constexpr int kStridedC = 4;        // 16 neighbouring elements = 64-byte segments
constexpr int kStridedK = kTileLog - kStridedC;

__global__ void fill_powers_kernel(uint32_t* out, uint32_t base, uint32_t scale, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bb::mul(bb::pow_u32(base, (uint32_t)i), scale);
}

// One stage group. DIF: active bits [n-s0-k, n-s0), stage order high bit -> low bit.
//                  DIT: active bits [s0, s0+k),     stage order low bit -> high bit.
// `lowbits` = number of index bits below the active bits; c = min(lowbits, cmax) of them
// ride along in the tile.
template <bool DIF>
__global__ __launch_bounds__(kBlock) void ntt_group_kernel(
    const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t in_stride, size_t out_stride,
    int n, int s0, int k, int c, const uint32_t* __restrict__ tw /* w^j, j < 2^(n-1) */) {
    extern __shared__ uint32_t tile[];
    const int lowbits = DIF ? n - s0 - k : s0;
    const int E = 1 << (k + c);
    const uint32_t cmask = (1u << c) - 1u;
    // decompose the tile id into (hi, lo_hi)
    const int lo_hi_bits = lowbits - c;
    const size_t tile_id = blockIdx.x;
    const size_t lo_hi = tile_id & (((size_t)1 << lo_hi_bits) - 1);
    const size_t hi = tile_id >> lo_hi_bits;
    const size_t base = (hi << (lowbits + k)) | (lo_hi << c);
    const uint32_t* src = in + (size_t)blockIdx.y * in_stride;
    uint32_t* dst = out + (size_t)blockIdx.y * out_stride;

    for (int e = threadIdx.x; e < E; e += kBlock) {
        size_t t = (size_t)(e >> c), lo = (size_t)(e & cmask);
        tile[e] = src[base | (t << lowbits) | lo];
    }
    __syncthreads();
    const size_t lo_base = lo_hi << c;
    for (int j = 0; j < k; ++j) {
        const int pos = DIF ? (k - 1 - j) : j;  // active bit handled by this stage
        const int s = s0 + j;                   // global stage number
        for (int b = threadIdx.x; b < (E >> 1); b += kBlock) {
            const uint32_t lo = (uint32_t)b & cmask;
            const uint32_t tt = (uint32_t)b >> c;
            const uint32_t t0 = ((tt >> pos) << (pos + 1)) | (tt & ((1u << pos) - 1u));
            const uint32_t t1 = t0 | (1u << pos);
            const uint32_t i0 = (t0 << c) | lo, i1 = (t1 << c) | lo;
            // p mod d with d = distance of this stage
            const size_t pm = ((size_t)(t0 & ((1u << pos) - 1u)) << lowbits) | lo_base | lo;
            const size_t ex = DIF ? (pm << s) : (pm << (n - s - 1));
            const uint32_t w = tw[ex];
            uint32_t a = tile[i0], bq = tile[i1];
            if (DIF) {
                tile[i0] = bb::add(a, bq);
                tile[i1] = bb::mul(bb::sub(a, bq), w);
            } else {
                bq = bb::mul(bq, w);
                tile[i0] = bb::add(a, bq);
                tile[i1] = bb::sub(a, bq);
            }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < E; e += kBlock) {
        size_t t = (size_t)(e >> c), lo = (size_t)(e & cmask);
        dst[base | (t << lowbits) | lo] = tile[e];
    }
}

// out[2q] = out[2q+1] = in[q] * scale[bitrev_n(q)]   (zero-pad + first DIT stage + coset scaling)
__global__ __launch_bounds__(kBlock) void expand_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                        size_t in_stride, size_t out_stride, int n,
                                                        const uint32_t* __restrict__ scale) {
    size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= ((size_t)1 << n)) return;
    uint32_t k = n ? (__brev((uint32_t)q) >> (32 - n)) : 0u;
    uint32_t v = bb::mul(in[(size_t)blockIdx.y * in_stride + q], scale[k]);
    uint2* o = reinterpret_cast<uint2*>(out + (size_t)blockIdx.y * out_stride + 2 * q);
    *o = make_uint2(v, v);
}

struct Tables {
    uint32_t* tw_fwd = nullptr;  // g_n^j, j < 2^(n-1)
    uint32_t* tw_inv = nullptr;  // g_n^-j
    uint32_t* shift = nullptr;   // s^k / 2^n, k < 2^n
};
std::mutex g_mu;
std::map<int, Tables> g_tables;

const Tables* tables(int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_tables.find(n);
    if (it != g_tables.end()) return &it->second;
    Tables t;
    size_t half = n ? (size_t)1 << (n - 1) : 1, full = (size_t)1 << n;
    if (hipMalloc(&t.tw_fwd, half * 4) != hipSuccess) return nullptr;
    if (hipMalloc(&t.tw_inv, half * 4) != hipSuccess) return nullptr;
    if (hipMalloc(&t.shift, full * 4) != hipSuccess) return nullptr;
    uint32_t w = field::root_of_unity(n);
    uint32_t one = bb::R_MOD_P;
    hipLaunchKernelGGL(fill_powers_kernel, dim3(div_up(half, 256)), dim3(256), 0, stream(), t.tw_fwd, w, one, half);
    hipLaunchKernelGGL(fill_powers_kernel, dim3(div_up(half, 256)), dim3(256), 0, stream(), t.tw_inv, bb::inv(w), one, half);
    uint32_t ninv = bb::inv(bb::to_monty((uint32_t)(((uint64_t)1 << n) % bb::P)));
    hipLaunchKernelGGL(fill_powers_kernel, dim3(div_up(full, 256)), dim3(256), 0, stream(), t.shift,
                       bb::to_monty(field::kCosetShift), ninv, full);
    return &g_tables.emplace(n, t).first->second;
}

struct Group { int s0, k, c; };

// Split stages [first, n) into groups. Contiguous groups (no low bits, or fewer than
// kStridedC of them) may take up to kTileLog - c stages, strided groups kStridedK.
std::vector<Group> plan_groups(bool dif, int n, int first) {
    std::vector<Group> g;
    int s = first;
    while (s < n) {
        int remaining = n - s;
        // low bits available if this group takes `k` stages
        auto lowbits = [&](int k) { return dif ? n - s - k : s; };
        int k = remaining < kTileLog ? remaining : kTileLog;
        for (; k >= 1; --k) {
            int lb = lowbits(k);
            int c = lb < kStridedC ? lb : kStridedC;
            if (k + c <= kTileLog) break;
        }
        // balance: avoid a tiny trailing group (e.g. 9 + 1): cap k so that the rest is not < 3
        int rest = remaining - k;
        if (rest > 0 && rest < 3 && k > 4) k -= (3 - rest);
        int lb = lowbits(k);
        int c = lb < kStridedC ? lb : kStridedC;
        g.push_back({s, k, c});
        s += k;
    }
    return g;
}

template <bool DIF>
void run_groups(const uint32_t* in, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n,
                int first_stage, const uint32_t* tw, const char* name) {
    auto groups = plan_groups(DIF, n, first_stage);
    const uint32_t* src = in;
    size_t src_stride = in_stride;
    for (auto& g : groups) {
        size_t tiles = ((size_t)1 << n) >> (g.k + g.c);
        for (uint32_t c0 = 0; c0 < cols; c0 += 65535u) {
            uint32_t cc = cols - c0 < 65535u ? cols - c0 : 65535u;
            ScopedKernelTimer t(name);
            hipLaunchKernelGGL(ntt_group_kernel<DIF>, dim3((unsigned)tiles, cc), dim3(kBlock), (size_t)4 << (g.k + g.c),
                               stream(), src + (size_t)c0 * src_stride, out + (size_t)c0 * out_stride, src_stride,
                               out_stride, n, g.s0, g.k, g.c, tw);
        }
        src = out;
        src_stride = out_stride;
    }
    if (groups.empty() && in != out) {
        // n == first_stage: nothing to do but copy
        for (uint32_t c = 0; c < cols; ++c)
            (void)hipMemcpyAsync(out + (size_t)c * out_stride, in + (size_t)c * in_stride, (size_t)4 << n,
                                 hipMemcpyDeviceToDevice, stream());
    }
}

}  // namespace

// Unscaled inverse NTT: natural-order evaluations on <g_n> -> bit-reversed coefficient order,
// out[q] = n * coefficient[bitrev(q)].
int intt_dif(const uint32_t* in, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n) {
    const Tables* t = tables(n);
    if (!t) return (int)hipErrorOutOfMemory;
    run_groups<true>(in, out, in_stride, out_stride, cols, n, 0, t->tw_inv, "ntt_group_kernel<dif>");
    return (int)hipGetLastError();
}

// coeffs: bit-reversed order, scaled by 2^n (as intt_dif leaves them). out: 2^(n+1) natural-order
// evaluations on the coset s*<g_{n+1}>.
int coset_lde_from_coeffs(const uint32_t* coeffs, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n) {
    const Tables* tn = tables(n);
    const Tables* t1 = tables(n + 1);
    if (!tn || !t1) return (int)hipErrorOutOfMemory;
    for (uint32_t c0 = 0; c0 < cols; c0 += 65535u) {
        uint32_t cc = cols - c0 < 65535u ? cols - c0 : 65535u;
        ScopedKernelTimer t("expand_kernel");
        hipLaunchKernelGGL(expand_kernel, dim3(div_up((size_t)1 << n, kBlock), cc), dim3(kBlock), 0, stream(),
                           coeffs + (size_t)c0 * in_stride, out + (size_t)c0 * out_stride, in_stride, out_stride, n, tn->shift);
    }
    run_groups<false>(out, out, out_stride, out_stride, cols, n + 1, 1, t1->tw_fwd, "ntt_group_kernel<dit>");
    return (int)hipGetLastError();
}

const uint32_t* shift_table(int n) { const Tables* t = tables(n); return t ? t->shift : nullptr; }

}  // namespace pw
