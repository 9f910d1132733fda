// Radix-2 NTT / low-degree extension over BabyBear for gfx950: LDS-staged stage groups with
// register-resident radix-16 rounds.
//
// Column-major matrices of 2^n-row columns; every column is an independent transform
// (blockIdx.y = column). A transform of n stages is cut into "stage groups": one launch runs k
// consecutive butterfly stages on a tile of 2^(k+c) elements, so the data crosses HBM once per
// group, not once per stage. A tile gathers the 2^k elements whose indices differ in the
// group's k active bits, times 2^c neighbouring elements (c low index bits) so that every global
// access of a strided group is a 64-byte contiguous segment.
//
// Inside a group a thread owns 16 (or 32) elements in VGPRs and runs up to 4 stages on them
// without touching LDS ("round", a radix-16 butterfly network); rounds exchange data through a
// padded LDS tile (one extra word per 32: conflict-free for every round's access pattern). The
// first round of a group loads straight from HBM into registers and the last round stores
// straight from registers to HBM. Twiddles: the factor of a butterfly splits into a per-thread
// base (one table lookup per round, squared from stage to stage) times a constant 16th/8th/4th
// root of unity, so no per-butterfly table traffic exists.
//
//   inverse  = DIF (Gentleman-Sande), natural-order input -> bit-reversed output
//   forward  = DIT (Cooley-Tukey),    bit-reversed input  -> natural-order output
// so iNTT followed by the coset NTT needs no bit-reversal pass. Zero-padding the coefficient
// vector from H to 2H in bit-reversed order is a duplication (padded[2q] = c[q], padded[2q+1] = 0,
// and the first DIT stage maps (a, 0) to (a, a)); the first DIT group of the LDE reads the
// H-sized coefficient array directly, applies the coset scaling s^k/H and duplicates in
// registers (EXPAND), so the 2H-sized vector is written exactly once per group.
//
// LDE definition (natural order, as oracle/stark_oracle.cpp `lde_column`):
//   L[j] = T(s * g_{n+1}^j),  j < 2H,  s = 31.
#include "babybear.hpp"
#include "common.hpp"
#include "prover_internal.hpp"

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace pw {

namespace {

constexpr int kBlock = 256;
// 2^c neighbouring elements ride along in strided tiles: 64-byte segments (c = 4) up to 2^17 rows, 128-byte ones (c = 5) from
// 2^18 rows on, where the strided passes run at 3.5 TB/s with 64-byte segments (profiles/r02_bench_ntt.txt); POWDR_NTT_C forces one

__constant__ uint32_t c_roots16[2][8];  // [0]: w16^r forward, [1]: inverse (Montgomery), r < 8

__global__ void fill_powers_kernel(uint32_t* out, uint32_t base, uint32_t scale, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bb::mul(bb::pow_u32(base, (uint32_t)i), scale);
}
__global__ void bitrev_copy_kernel(const uint32_t* in, uint32_t* out, int n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << n)) return;
    uint32_t k = n ? (__brev((uint32_t)i) >> (32 - n)) : 0u;
    out[i] = in[k];
}

struct GroupParams {
    int n;        // log2 of the transform size
    int s0;       // first stage of the group
    int k;        // stages in the group
    int c;        // passive low bits carried in the tile
    int lowbits;  // index bits below the active bits (DIF: n-s0-k, DIT: s0)
    int B;        // k + c: in-tile index bits
    int n_rounds;
    int rb[4];    // window start (tile bit) per round, in execution order
    int logr[4];  // window width per round
    unsigned long long n_tiles;  // tiles per column = 2^(n-B)
    int canonical_out;  // forward transform: the last group reduces its [0, 2p) values to [0, p) when it stores
    unsigned twt_off[4];  // per round: offset of its twiddle table (contiguous groups of the fused LDE kernel, see FusedTwiddles)
    // sub-coset evaluation (subcoset_lde): FOLD loads — element g of the transform's input is the sum of 2^fold coefficients times the
    // wave-uniform constants foldk[] — and a COSET transform: the twiddles of stage s carry the constant c^(2^(n-1-s)). Tile-invariant
    // groups read them from a table derived for the sub-coset; the others multiply a round's top twiddle base by cst[round] (the lower
    // stages' constants follow from the squaring chain).
    int logt;     // log2 of the LDS tile the group's kernel is instantiated for (0: the plan's common size)
    int fold;
    int coset;
    uint32_t foldk[16];
    uint32_t cst[4];
    // SELECT (the query phase of a streamed proof, subcoset_query_rows): the group's outputs are not stored — of every tile only the
    // n_sel positions sel_pos[q] are used: the term position * sel_xpow[q * n_tiles + tile] of output q goes to
    // sel_part[(column * n_tiles + tile) * n_sel + q] (one plain store per term; select_reduce_kernel sums a column's tiles) — or, with
    // sel_part == nullptr (POWDR_QUERY_SELECT=2, round 5's form), is added to sel_acc[column * n_sel + q] by a 64-bit atomic.
    int n_sel;
    const uint32_t* sel_pos;
    const uint32_t* sel_xpow;
    unsigned long long* sel_acc;
    uint32_t* sel_part;
};

__device__ __forceinline__ uint32_t lds_phys(uint32_t l) { return l + (l >> 5); }

struct IndexMap {
    int B, c, lowbits, k;
    uint32_t cmask;
    size_t tile0;
    size_t n_tiles;
    // global index of local element l; valid = tile exists
    __device__ __forceinline__ size_t global(uint32_t l, bool& valid) const {
        const uint32_t i = l & ((1u << B) - 1u);
        const size_t tile = tile0 + (l >> B);
        valid = tile < n_tiles;
        const int lo_hi_bits = lowbits - c;
        const size_t lo_hi = tile & (((size_t)1 << lo_hi_bits) - 1);
        const size_t hi = tile >> lo_hi_bits;
        return (hi << (lowbits + k)) | ((size_t)(i >> c) << lowbits) | (lo_hi << c) | (size_t)(i & cmask);
    }
    // global index bits below tile bit `rb` for the slot whose local index (with the window zeroed) is l
    __device__ __forceinline__ size_t glow(uint32_t l, int rb) const {
        const uint32_t i = l & ((1u << rb) - 1u);
        const size_t tile = tile0 + (l >> B);
        if (tile >= n_tiles) return 0;  // padding slot of a short transform: any in-range twiddle
        const int lo_hi_bits = lowbits - c;
        const size_t lo_hi = tile & (((size_t)1 << lo_hi_bits) - 1);
        return ((size_t)(i >> c) << lowbits) | (lo_hi << c) | (size_t)(i & cmask);
    }
};

// One slot of one round: LOGR stages on the window [rb, rb+LOGR) of the tile index, on the
// R = 2^LOGR elements v[rho] (fully reduced arithmetic: see the note at bb::mul_lazy).
// The twiddle base of slot m of a round: w^(g << shift) for the slot's low global index bits g and the round's
// smallest shift (the one of window bit logr-1). A table lookup with global-memory latency on the critical path
// of the slot's butterflies, so callers fetch it one slot (or one round) ahead.
template <bool DIF, int NT = kBlock>
__device__ __forceinline__ uint32_t load_twiddle_base(const IndexMap& im, const GroupParams& gp, int round, int m, int tid,
                                                      const uint32_t* __restrict__ tw) {
    const int rb = gp.rb[round], logr = gp.logr[round];
    const uint32_t sigma = (uint32_t)tid + (uint32_t)NT * m;
    const uint32_t l0 = ((sigma >> rb) << (rb + logr)) | (sigma & ((1u << rb) - 1u));
    const size_t g = im.glow(l0, rb);
    const int sh_top = DIF ? gp.s0 + gp.k - 1 - (rb + (logr - 1) - gp.c)       // stage number s_q
                           : gp.n - 1 - (gp.s0 + rb + (logr - 1) - gp.c);       // n - 1 - s_q
    return tw[g << sh_top];
}

// `tt` (TWT): the slot's 2^LOGR - 1 twiddles, stage q at [2^q - 1, 2^(q+1) - 1), loaded from a table instead of being derived
// from `base_top` by LOGR - 1 squarings and 2^LOGR - 1 - LOGR products (70 of a radix-16 slot's 390 instructions).
template <bool DIF, int LOGR, bool TWT = false>
__device__ __forceinline__ void slot_butterflies(uint32_t* v, uint32_t base_top, const uint32_t* tt = nullptr) {
    constexpr int R = 1 << LOGR;
    const uint32_t* roots = c_roots16[DIF ? 1 : 0];
    // base[q] = w^(g << shift_q); the smallest shift belongs to q = LOGR-1 and base[q-1] = base[q]^2
    uint32_t base[LOGR];
    if (!TWT) {
        base[LOGR - 1] = base_top;
#pragma unroll
        for (int q = LOGR - 2; q >= 0; --q) base[q] = bb::sqr(base[q + 1]);
    }
#pragma unroll
    for (int qi = 0; qi < LOGR; ++qi) {
        const int q = DIF ? (LOGR - 1 - qi) : qi;
        // twiddles of this stage: base[q] * w_{2^(q+1)}^r, r < 2^q;  w_{2^(q+1)}^r = w16^(r << (3 - q))
        uint32_t t[1 << (LOGR - 1)], nt[1 << (LOGR - 1)];
#pragma unroll
        for (int r = 0; r < (1 << q); ++r) {
            if (TWT) t[r] = tt[(1 << q) - 1 + r];
            else t[r] = r == 0 ? base[q] : bb::mul(base[q], roots[r << (3 - q)]);
            if (DIF) nt[r] = bb::P - t[r];  // a representative of -t in (0, p]: fine as a factor
        }
#pragma unroll
        for (int u = 0; u < R / 2; ++u) {
            const int lowp = u & ((1 << q) - 1);
            const int i0 = ((u >> q) << (q + 1)) | lowp;
            const int i1 = i0 | (1 << q);
            const uint32_t a = v[i0], b = v[i1];
            if (DIF) {
                v[i0] = bb::add(a, b);
                v[i1] = bb::mul2(a, t[lowp], b, nt[lowp]);  // (a - b) t as a t + b (p - t): one reduction, no subtraction
            } else {
                // forward transform: all values are representatives in [0, 2p) between the first load and the last store
                const uint32_t bt = bb::mul_lazy(b, t[lowp]);
                v[i0] = bb::add_2p(a, bt);
                v[i1] = bb::sub_2p(a, bt);
            }
        }
    }
}

// FOLD load (sub-coset evaluation, subcoset_lde below): input element g of a size-2^n' transform is
//   sum_{t < 2^f} src[(g << f) + t] * K[t]
// — the polynomial reduced modulo x^(2^n') - c^(2^n') on the sub-coset, in bit-reversed order: the 2^f coefficients that fold onto one
// position are contiguous, and their factors (c^(2^n') to the power bitrev(t), over H) are the same for every position. Canonical.
__device__ __forceinline__ uint32_t fold_load(const uint32_t* __restrict__ src, const uint32_t* K, size_t g, int f) {
    const size_t q = g << f;
    if (f == 0) return bb::mul(src[q], K[0]);
    if (f == 1) {
        const uint2 a = *reinterpret_cast<const uint2*>(src + q);
        return bb::mul2(a.x, K[0], a.y, K[1]);
    }
    const uint4* pa = reinterpret_cast<const uint4*>(src + q);
    uint32_t acc = 0u;
    for (int t = 0; t < (1 << (f - 2)); ++t) {
        const uint4 a = pa[t];
        acc = bb::add(acc, bb::add(bb::mul2(a.x, K[4 * t], a.y, K[4 * t + 1]), bb::mul2(a.z, K[4 * t + 2], a.w, K[4 * t + 3])));
    }
    return acc;
}

// One round of a group, slot by slot (a slot = the R elements a thread combines). A round is an
// in-place network: every slot reads and writes the same R tile positions, so slots need no
// barrier among themselves; barriers separate rounds only. The first round of a group loads from
// HBM, the last one stores to HBM.
// EXPAND: `src` is the H-sized bit-reversed coefficient array (n = log2(2H)); element g of the
// 2H-sized vector is src[g >> 1] * scale_br[g >> 1].
// `tw_base` enters as the twiddle base of this round's slot 0 and leaves as the one of the next round's slot 0.
// MODE: 0 = plain loads, 1 = EXPAND, 2 = FOLD (fold_load above).
// NC (two-column experiment): a thread runs its slot for NC columns — column cc in tile0 + cc * tile_cs, src0 + cc * src_cs,
// dst0 + cc * dst_cs — with ONE set of slot indices and twiddles (the table loads and the index arithmetic are shared).
template <bool DIF, int LOGR, int EPT, int MODE, bool TWT = false, int NC = 1, int NT = kBlock>
__device__ __forceinline__ void run_round(uint32_t* tile0, const IndexMap& im, const GroupParams& gp, int round,
                                          const uint32_t* __restrict__ src0, uint32_t* __restrict__ dst0,
                                          const uint32_t* __restrict__ tw, const uint32_t* __restrict__ scale_br, int tid,
                                          uint32_t& tw_base, bool io_first = true, bool io_last = true, size_t src_cs = 0, size_t dst_cs = 0,
                                          uint32_t tile_cs = 0) {
    static_assert(NC == 1 || MODE == 0, "several columns per thread: plain loads only");
    constexpr bool EXPAND = MODE == 1, FOLD = MODE == 2;
    const int rb = gp.rb[round];
    // io_first / io_last = false: the group's first round reads / its last round writes the LDS tile instead of HBM
    // (the fused LDE kernel chains two groups through LDS)
    const bool last_round = round == gp.n_rounds - 1;
    const bool first = round == 0 && io_first, last = last_round && io_last;
    constexpr int R = 1 << LOGR;
    constexpr int SLOTS = EPT / R;
    const bool vec_plain = rb == 0 && gp.c == 0 && LOGR >= 2;   // the slot's elements are contiguous in HBM
    const bool vec_expand = EXPAND && rb == 1 && gp.c == 1 && LOGR >= 2;
    const int gshift = rb - gp.c + gp.lowbits;
    // padded LDS position of element rho of a slot, relative to the slot's base: the window bits and the
    // slot's bits are disjoint, so lds_phys(l0 | rho << rb) = lds_phys(l0) + lds_off(rho); wave-uniform
    auto lds_off = [rb](int rho) -> uint32_t {
        return ((uint32_t)rho << rb) + (rb >= 5 ? ((uint32_t)rho << (rb - 5)) : ((uint32_t)rho >> (5 - rb)));
    };
#pragma unroll 1
    for (int m = 0; m < SLOTS; ++m) {
        const uint32_t sigma = (uint32_t)tid + (uint32_t)NT * m;
        const uint32_t l0 = ((sigma >> rb) << (rb + LOGR)) | (sigma & ((1u << rb) - 1u));
        const uint32_t p0 = lds_phys(l0);
        // fetch the next slot's (or the next round's first) twiddle base while this slot computes
        const uint32_t cur_base = (!TWT && gp.coset) ? bb::mul(tw_base, gp.cst[round]) : tw_base;
        uint32_t tt[R];  // TWT: the slot's twiddles from the group's table (`tw`), [stage-major index][g], g = the index bits below the window
        if (TWT) {
            const uint32_t* tab = tw + gp.twt_off[round] + (l0 & ((1u << rb) - 1u));
#pragma unroll
            for (int j = 0; j < R - 1; ++j) tt[j] = tab[(uint32_t)j << rb];
        } else {
            if (m + 1 < SLOTS) tw_base = load_twiddle_base<DIF, NT>(im, gp, round, m + 1, tid, tw);
            else if (!last_round) tw_base = load_twiddle_base<DIF, NT>(im, gp, round + 1, 0, tid, tw);
        }
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        uint32_t x[R];
        uint32_t* tile = tile0 + (size_t)cc * tile_cs;
        const uint32_t* src = src0 + (size_t)cc * src_cs;
        uint32_t* dst = dst0 + (size_t)cc * dst_cs;
        // ---- load ----
        if (first) {
            if (FOLD) {
                bool valid;
                const size_t g0 = im.global(l0, valid);
                if (valid) {
#pragma unroll
                    for (int rho = 0; rho < R; ++rho) x[rho] = fold_load(src, gp.foldk, g0 + ((size_t)rho << gshift), gp.fold);
                } else {
#pragma unroll
                    for (int rho = 0; rho < R; ++rho) x[rho] = 0u;
                }
            } else if (EXPAND ? vec_expand : vec_plain) {
                bool valid;
                const size_t g0 = im.global(l0, valid);
                const uint4* pc = reinterpret_cast<const uint4*>(src + (EXPAND ? (g0 >> 1) : g0));
                const uint4* ps = reinterpret_cast<const uint4*>(scale_br + (g0 >> 1));
#pragma unroll
                for (int v4 = 0; v4 < R / 4; ++v4) {
                    uint4 d = make_uint4(0, 0, 0, 0);
                    if (valid) d = pc[v4];
                    if (EXPAND) {
                        uint4 sc = make_uint4(0, 0, 0, 0);
                        if (valid) sc = ps[v4];
                        d.x = bb::mul(d.x, sc.x); d.y = bb::mul(d.y, sc.y); d.z = bb::mul(d.z, sc.z); d.w = bb::mul(d.w, sc.w);
                    }
                    x[4 * v4 + 0] = d.x; x[4 * v4 + 1] = d.y; x[4 * v4 + 2] = d.z; x[4 * v4 + 3] = d.w;
                }
            } else {
                // window bit (rb + q) of the tile index is bit gshift + q of the global index
                bool valid;
                const size_t g0 = im.global(l0, valid);
                if (EXPAND) {
                    const size_t q0 = g0 >> 1;
                    const int qshift = gshift - 1;  // the EXPAND group has lowbits = c = 1, so gshift = rb >= 1
                    // one test per slot, not one predicated load per element: the R loads issue back to back
                    if (valid) {
                        uint32_t cf[R], sc[R];
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) { const size_t q = q0 + ((size_t)rho << qshift); cf[rho] = src[q]; sc[rho] = scale_br[q]; }
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) x[rho] = bb::mul(cf[rho], sc[rho]);
                    } else {
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) x[rho] = 0u;
                    }
                } else {
                    const uint32_t* ps = src + g0;
                    if (valid) {
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) x[rho] = ps[(size_t)rho << gshift];
                    } else {
#pragma unroll
                        for (int rho = 0; rho < R; ++rho) x[rho] = 0u;
                    }
                }
            }
        } else {
#pragma unroll
            for (int rho = 0; rho < R; ++rho) x[rho] = tile[p0 + lds_off(rho)];
        }
        // ---- butterflies ----
        slot_butterflies<DIF, LOGR, TWT>(x, cur_base, tt);
        // ---- store ----
        if (!DIF && last && gp.canonical_out) {
#pragma unroll
            for (int rho = 0; rho < R; ++rho) x[rho] = bb::reduce_2p(x[rho]);
        }
        if (last) {
            if (vec_plain) {
                bool valid;
                const size_t g0 = im.global(l0, valid);
                if (valid) {
                    uint4* pd = reinterpret_cast<uint4*>(dst + g0);
#pragma unroll
                    for (int v4 = 0; v4 < R / 4; ++v4) pd[v4] = make_uint4(x[4 * v4], x[4 * v4 + 1], x[4 * v4 + 2], x[4 * v4 + 3]);
                }
            } else {
                bool valid;
                const size_t g0 = im.global(l0, valid);
                if (valid) {
                    uint32_t* pd = dst + g0;
#pragma unroll
                    for (int rho = 0; rho < R; ++rho) pd[(size_t)rho << gshift] = x[rho];
                }
            }
        } else {
#pragma unroll
            for (int rho = 0; rho < R; ++rho) tile[p0 + lds_off(rho)] = x[rho];
        }
      }
    }
}

// TWT: `tw` is the group's twiddle TABLE (tile-invariant groups, see group_twiddles) instead of the transform's root table
// NT: threads per workgroup (256; 1 024 for the 2^14-element tiles of the two-pass plan: the same 16 elements per lane, 8 waves per SIMD)
template <bool DIF, int LOGT, int MODE, bool TWT = false, int NT = kBlock>
__global__ __launch_bounds__(NT) void ntt_group_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                       size_t in_stride, size_t out_stride, GroupParams gp,
                                                       const uint32_t* __restrict__ tw,
                                                       const uint32_t* __restrict__ scale_br) {
    static_assert(NT == kBlock || MODE == 0, "wide workgroups: plain strided groups only");
    constexpr int EPT = (1 << LOGT) / NT;
    __shared__ uint32_t tile[(1 << LOGT) + ((1 << LOGT) >> 5)];
    const int tid = threadIdx.x;
    IndexMap im;
    im.B = gp.B; im.c = gp.c; im.lowbits = gp.lowbits; im.k = gp.k;
    im.cmask = (1u << gp.c) - 1u;
    im.tile0 = (size_t)blockIdx.x << (LOGT - gp.B);
    im.n_tiles = (size_t)gp.n_tiles;
    const uint32_t* src = in + (size_t)blockIdx.y * in_stride;
    uint32_t* dst = out + (size_t)blockIdx.y * out_stride;
    uint32_t tw_base = TWT ? 0u : load_twiddle_base<DIF, NT>(im, gp, 0, 0, tid, tw);
    // FOLD (sub-coset evaluation): the folded inputs of the workgroup's tile(s) are staged through LDS first — lane = output, so a wave
    // reads 2^fold * 256 CONTIGUOUS bytes per instruction. (Loaded slot by slot like the other modes, a lane would read the 16 neighbouring
    // outputs of its slot: 64 lanes x 16 instructions walking 64 cache lines side by side, which L1 does not hold for a CU's worth of waves.)
    constexpr bool kStageFold = MODE == 2;
    const bool select = MODE == 2 && gp.n_sel > 0;
    if (kStageFold) {
        if (gp.B == LOGT && gp.c == 0 && gp.lowbits == 0 && gp.fold <= 1) {
            // The whole-tile case of a tall transform (one tile per workgroup, element l of the tile = element tile * 2^LOGT + l of the
            // transform's input): every lane issues ALL its loads — 16 bytes each — before the first product, instead of four 4-byte
            // loads at a time (round 5: the group ran at 56 % of the issue rate with 30 % of the HBM rate, waiting for its own loads).
            const size_t g0 = (size_t)blockIdx.x << LOGT;
            constexpr int kVec = (1 << LOGT) / (4 * kBlock);  // uint4 loads per lane and coefficient word of an output
            if (gp.fold == 0) {
                const uint4* pv = reinterpret_cast<const uint4*>(src + g0);
                uint4 d[kVec];
#pragma unroll
                for (int m = 0; m < kVec; ++m) d[m] = pv[(uint32_t)tid + kBlock * m];
                const uint32_t k0 = gp.foldk[0];
#pragma unroll
                for (int m = 0; m < kVec; ++m) {
                    const uint32_t l = 4u * ((uint32_t)tid + kBlock * m);
                    tile[lds_phys(l)] = bb::mul(d[m].x, k0); tile[lds_phys(l + 1)] = bb::mul(d[m].y, k0);
                    tile[lds_phys(l + 2)] = bb::mul(d[m].z, k0); tile[lds_phys(l + 3)] = bb::mul(d[m].w, k0);
                }
            } else {  // two coefficient words per output: a uint4 holds two outputs
                const uint4* pv = reinterpret_cast<const uint4*>(src + 2 * g0);
                uint4 d[2 * kVec];
#pragma unroll
                for (int m = 0; m < 2 * kVec; ++m) d[m] = pv[(uint32_t)tid + kBlock * m];
                const uint32_t k0 = gp.foldk[0], k1 = gp.foldk[1];
#pragma unroll
                for (int m = 0; m < 2 * kVec; ++m) {
                    const uint32_t l = 2u * ((uint32_t)tid + kBlock * m);
                    tile[lds_phys(l)] = bb::mul2(d[m].x, k0, d[m].y, k1);
                    tile[lds_phys(l + 1)] = bb::mul2(d[m].z, k0, d[m].w, k1);
                }
            }
        } else {
#pragma unroll 4
            for (uint32_t l = (uint32_t)tid; l < (1u << LOGT); l += kBlock) {
                bool valid;
                const size_t g = im.global(l, valid);
                tile[lds_phys(l)] = valid ? fold_load(src, gp.foldk, g, gp.fold) : 0u;
            }
        }
        __syncthreads();
    }
    for (int r = 0; r < gp.n_rounds; ++r) {
        switch (gp.logr[r]) {
            case 1: run_round<DIF, 1, EPT, MODE, TWT, 1, NT>(tile, im, gp, r, src, dst, tw, scale_br, tid, tw_base, !kStageFold, !select); break;
            case 2: run_round<DIF, 2, EPT, MODE, TWT, 1, NT>(tile, im, gp, r, src, dst, tw, scale_br, tid, tw_base, !kStageFold, !select); break;
            case 3: run_round<DIF, 3, EPT, MODE, TWT, 1, NT>(tile, im, gp, r, src, dst, tw, scale_br, tid, tw_base, !kStageFold, !select); break;
            default: run_round<DIF, 4, EPT, MODE, TWT, 1, NT>(tile, im, gp, r, src, dst, tw, scale_br, tid, tw_base, !kStageFold, !select); break;
        }
        if (r + 1 < gp.n_rounds) __syncthreads();  // the next round reads what this round wrote
    }
    if (select) {
        // the tile stays in LDS (the forward network's [0, 2p) representatives): this tile's term of every selected output
        __syncthreads();
        const size_t t = blockIdx.x;  // (one tile per workgroup: the host only selects then)
        uint32_t* part = gp.sel_part ? gp.sel_part + ((size_t)blockIdx.y * gp.n_tiles + t) * (size_t)gp.n_sel : nullptr;
        for (int q = tid; q < gp.n_sel; q += kBlock) {
            const uint32_t v = bb::reduce_2p(tile[lds_phys(gp.sel_pos[q])]);
            const uint32_t term = bb::mul(v, gp.sel_xpow[(size_t)q * gp.n_tiles + t]);
            if (part) part[q] = term;  // n_sel consecutive words per workgroup
            else atomicAdd(gp.sel_acc + (size_t)blockIdx.y * gp.n_sel + q, (unsigned long long)term);
        }
    }
}

// all rounds of one stage group on the workgroup's tile(s). TWT: `tw` is the group's twiddle TABLE (FusedTwiddles).
template <bool DIF, int EPT, bool TWT = false, int NC = 1>
__device__ __forceinline__ void run_group(uint32_t* tile, const IndexMap& im, const GroupParams& gp, const uint32_t* __restrict__ src,
                                          uint32_t* __restrict__ dst, const uint32_t* __restrict__ tw, int tid, bool io_first, bool io_last,
                                          size_t src_cs = 0, size_t dst_cs = 0, uint32_t tile_cs = 0) {
    uint32_t tw_base = TWT ? 0u : load_twiddle_base<DIF>(im, gp, 0, 0, tid, tw);
    for (int r = 0; r < gp.n_rounds; ++r) {
        switch (gp.logr[r]) {
            case 1: run_round<DIF, 1, EPT, 0, TWT, NC>(tile, im, gp, r, src, dst, tw, nullptr, tid, tw_base, io_first, io_last, src_cs, dst_cs, tile_cs); break;
            case 2: run_round<DIF, 2, EPT, 0, TWT, NC>(tile, im, gp, r, src, dst, tw, nullptr, tid, tw_base, io_first, io_last, src_cs, dst_cs, tile_cs); break;
            case 3: run_round<DIF, 3, EPT, 0, TWT, NC>(tile, im, gp, r, src, dst, tw, nullptr, tid, tw_base, io_first, io_last, src_cs, dst_cs, tile_cs); break;
            default: run_round<DIF, 4, EPT, 0, TWT, NC>(tile, im, gp, r, src, dst, tw, nullptr, tid, tw_base, io_first, io_last, src_cs, dst_cs, tile_cs); break;
        }
        if (r + 1 < gp.n_rounds) __syncthreads();
    }
}

// The middle of the LDE in ONE pass over HBM: the last (contiguous) stage group of the inverse transform, the coset
// scaling s^k / H, the zero-padding to 2H (a duplication in bit-reversed order) and the first (contiguous) stage group of
// the forward transform all act on the same 2^ka bit-reversed coefficients, so a workgroup loads them once, runs the DIF
// rounds, expands through LDS and runs the DIT rounds on the 2^(ka+1) results: the H-sized coefficient array is neither
// written nor re-read (8 of the 44 bytes the unfused schedule moves per trace cell). For H <= 2^12 the whole LDE of a
// column is this one launch.
template <int NC>
__device__ __forceinline__ void lde_fused_body(uint32_t* tile, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t in_stride,
                                               size_t out_stride, const GroupParams& ga, const GroupParams& gd, const uint32_t* __restrict__ tw_inv,
                                               const uint32_t* __restrict__ tw_fwd, const uint32_t* __restrict__ scale_br) {
    constexpr int LOGA = 12, LOGD = 13;
    constexpr uint32_t kTileWords = (1u << LOGD) + ((1u << LOGD) >> 5);
    const int tid = threadIdx.x;
    const uint32_t* src = in + (size_t)blockIdx.y * NC * in_stride;
    uint32_t* dst = out + (size_t)blockIdx.y * NC * out_stride;
    IndexMap ia;
    ia.B = ga.B; ia.c = ga.c; ia.lowbits = ga.lowbits; ia.k = ga.k;
    ia.cmask = (1u << ga.c) - 1u;
    ia.tile0 = (size_t)blockIdx.x << (LOGA - ga.B);
    ia.n_tiles = (size_t)ga.n_tiles;
    run_group<true, (1 << LOGA) / kBlock, true, NC>(tile, ia, ga, src, nullptr, tw_inv, tid, true, false, in_stride, 0, kTileWords);
    __syncthreads();
    // scale and duplicate: element l of the coefficient tile becomes elements 2l, 2l + 1 of the forward tile
    uint32_t v[NC][(1 << LOGA) / kBlock];
#pragma unroll
    for (int m = 0; m < (1 << LOGA) / kBlock; ++m) {
        const uint32_t l = (uint32_t)tid + 256u * m;
        bool valid;
        const size_t q = ia.global(l, valid);
        const uint32_t sc = valid ? scale_br[q] : 0u;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) v[cc][m] = bb::mul(tile[cc * kTileWords + lds_phys(l)], sc);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < (1 << LOGA) / kBlock; ++m) {
        const uint32_t l = (uint32_t)tid + 256u * m;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            tile[cc * kTileWords + lds_phys(2 * l)] = v[cc][m];
            tile[cc * kTileWords + lds_phys(2 * l + 1)] = v[cc][m];
        }
    }
    __syncthreads();
    IndexMap id;
    id.B = gd.B; id.c = gd.c; id.lowbits = gd.lowbits; id.k = gd.k;
    id.cmask = (1u << gd.c) - 1u;
    id.tile0 = (size_t)blockIdx.x << (LOGD - gd.B);
    id.n_tiles = (size_t)gd.n_tiles;
    run_group<false, (1 << LOGD) / kBlock, true, NC>(tile, id, gd, nullptr, dst, tw_fwd, tid, false, true, 0, out_stride, kTileWords);
}

__global__ __launch_bounds__(kBlock) void lde_fused_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t in_stride,
                                                           size_t out_stride, GroupParams ga, GroupParams gd,
                                                           const uint32_t* __restrict__ tw_inv, const uint32_t* __restrict__ tw_fwd,
                                                           const uint32_t* __restrict__ scale_br) {
    __shared__ uint32_t tile[(1 << 13) + ((1 << 13) >> 5)];
    lde_fused_body<1>(tile, in, out, in_stride, out_stride, ga, gd, tw_inv, tw_fwd, scale_br);
}

// Two columns per workgroup (POWDR_NTT_TWO_COL=1; VERDICT r4 #5, measured in profiles/r05_ntt_two_column.txt): both columns' tiles in
// LDS (67.6 KB, dynamic: two workgroups per CU instead of four), one set of twiddle-table loads and slot indices for the pair.
__global__ __launch_bounds__(kBlock) void lde_fused2_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t in_stride,
                                                            size_t out_stride, GroupParams ga, GroupParams gd,
                                                            const uint32_t* __restrict__ tw_inv, const uint32_t* __restrict__ tw_fwd,
                                                            const uint32_t* __restrict__ scale_br) {
    extern __shared__ uint32_t tile2[];
    lde_fused_body<2>(tile2, in, out, in_stride, out_stride, ga, gd, tw_inv, tw_fwd, scale_br);
}

struct Tables {
    uint32_t* tw_fwd = nullptr;    // g_n^j, j < 2^(n-1)
    uint32_t* tw_inv = nullptr;    // g_n^-j
    uint32_t* shift = nullptr;     // s^k / 2^n, k < 2^n
    uint32_t* shift_br = nullptr;  // shift[bitrev_n(q)]
};
std::mutex g_mu;
std::map<std::pair<int, int>, Tables> g_tables;  // (device, n): twiddle tables live on the device that was current when they were built
uint64_t g_roots_uploaded = 0;                   // bit d: device d's __constant__ copy of the 16th roots is initialised

int upload_roots() {
    int device = 0;
    PW_HIP_TRY(hipGetDevice(&device));
    if (device < 64 && (g_roots_uploaded >> device) & 1) return 0;
    uint32_t h[2][8];
    const uint32_t w16 = field::root_of_unity(4), w16i = bb::inv(w16);
    uint32_t a = bb::R_MOD_P, b = bb::R_MOD_P;
    for (int r = 0; r < 8; ++r) { h[0][r] = a; h[1][r] = b; a = bb::mul(a, w16); b = bb::mul(b, w16i); }
    PW_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_roots16), h, sizeof h));
    if (device < 64) g_roots_uploaded |= 1ull << device;
    return 0;
}

const Tables* tables(int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (upload_roots()) return nullptr;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    const std::pair<int, int> key{device, n};
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return &it->second;
    Tables t;
    size_t half = n ? (size_t)1 << (n - 1) : 1, full = (size_t)1 << n;
    if (hipMalloc(&t.tw_fwd, half * 4) != hipSuccess) return nullptr;
    if (hipMalloc(&t.tw_inv, half * 4) != hipSuccess) return nullptr;
    if (hipMalloc(&t.shift, full * 4) != hipSuccess) return nullptr;
    if (hipMalloc(&t.shift_br, full * 4) != hipSuccess) return nullptr;
    uint32_t w = field::root_of_unity(n);
    uint32_t one = bb::R_MOD_P;
    hipLaunchKernelGGL(fill_powers_kernel, dim3(div_up(half, 256)), dim3(256), 0, stream(), t.tw_fwd, w, one, half);
    hipLaunchKernelGGL(fill_powers_kernel, dim3(div_up(half, 256)), dim3(256), 0, stream(), t.tw_inv, bb::inv(w), one, half);
    uint32_t ninv = bb::inv(bb::to_monty((uint32_t)(((uint64_t)1 << n) % bb::P)));
    hipLaunchKernelGGL(fill_powers_kernel, dim3(div_up(full, 256)), dim3(256), 0, stream(), t.shift,
                       bb::to_monty(field::kCosetShift), ninv, full);
    hipLaunchKernelGGL(bitrev_copy_kernel, dim3(div_up(full, 256)), dim3(256), 0, stream(), t.shift, t.shift_br, n);
    // other host threads (other streams) may use the tables as soon as they are published
    if (hipStreamSynchronize(stream()) != hipSuccess) return nullptr;
    return &g_tables.emplace(key, t).first->second;
}

// Split stages [first, n) into groups of at most `LOGT - c` stages and each group into rounds.
// `end` < n: only the stages [first, end) (the fused LDE kernel takes the rest); `balance`: the groups get nearly equal numbers
// of stages instead of greedy-full groups followed by a short one.
std::vector<GroupParams> plan_groups(bool dif, int n, int first, int& logt_out, int end = -1, bool balance = false) {
    std::vector<GroupParams> out;
    int kStridedC = (n - (dif ? 0 : 1)) >= 18 ? 5 : 4;  // n = log2 of the transform: the LDE's forward half has one bit more
    if (const char* e = getenv("POWDR_NTT_C")) { int v = atoi(e); if (v >= 0 && v <= 6) kStridedC = v; }
    if (end < 0) end = n;
    const int total = end - first;
    // tile size: 2^13 when it saves a pass or the transform is large, else 2^12
    int logt = 12;
    int n_groups = 0;
    {
        auto passes = [&](int lt) {
            int s = first, p = 0;
            while (s < end) {
                int rem = end - s, k = rem < lt ? rem : lt;
                for (; k >= 1; --k) { int lb = dif ? n - s - k : s; int c = lb < kStridedC ? lb : kStridedC; if (k + c <= lt) break; }
                s += k; ++p;
            }
            return p;
        };
        if (total > 0 && passes(13) < passes(12)) logt = 13;
        n_groups = total > 0 ? passes(logt) : 0;
    }
    logt_out = logt;
    auto make_group = [&](int s, int k, int c, int lt) {
        GroupParams g{};
        g.n = n; g.s0 = s; g.k = k; g.c = c; g.lowbits = dif ? n - s - k : s; g.B = k + c; g.logt = lt;
        g.n_tiles = 1ull << (n - g.B);
        // rounds: ceil(k/4) windows of nearly equal width
        int nr = (k + 3) / 4;
        g.n_rounds = nr;
        int widths[4];
        for (int r = 0; r < nr; ++r) widths[r] = k / nr + (r < k % nr ? 1 : 0);
        if (dif) {  // from the top active bit downwards
            int top = c + k;
            for (int r = 0; r < nr; ++r) { top -= widths[r]; g.rb[r] = top; g.logr[r] = widths[r]; }
        } else {    // from the lowest active bit upwards
            int bot = c;
            for (int r = 0; r < nr; ++r) { g.rb[r] = bot; g.logr[r] = widths[r]; bot += widths[r]; }
        }
        return g;
    };
    int s = first;
    int groups_left = n_groups;
    while (s < end) {
        int rem = end - s, k = rem < logt ? rem : logt;
        if (balance && groups_left > 0) { const int target = (rem + groups_left - 1) / groups_left; if (k > target) k = target; }
        --groups_left;
        int c = 0;
        for (; k >= 1; --k) { int lb = dif ? n - s - k : s; c = lb < kStridedC ? lb : kStridedC; if (k + c <= logt) break; }
        out.push_back(make_group(s, k, c, logt));
        s += k;
    }
    // A tall transform whose greedy plan is three passes over HBM (2^22 points: 12 contiguous + 7 + 3 stages forward, 7 + 7 + 8 inverse)
    // as TWO: the 12 contiguous stages on a 2^12 tile and ALL strided stages (at most 10) as one group on a 2^14-element tile with
    // 64-byte segments (c = 4) — the strided groups run at the box's copy rate, so a pass less is the only way to make them faster
    // (round 5; POWDR_NTT_TILE14=0 keeps the three-pass plan).
    static const int tile14 = [] { const char* e = getenv("POWDR_NTT_TILE14"); return e ? atoi(e) : 0; }();
    if (tile14 && !balance && out.size() == 3 && end == n && (dif || first == 0) && total - 12 >= 1 && total - 12 <= 10) {
        const int ks = total - 12;
        out.clear();
        if (dif) { out.push_back(make_group(first, ks, 4, 14)); out.push_back(make_group(first + ks, 12, 0, 12)); }
        else     { out.push_back(make_group(first, 12, 0, 12)); out.push_back(make_group(first + 12, ks, 4, 14)); }
        logt_out = 12;
    }
    return out;
}

// Twiddle table of ONE tile-invariant stage group (lowbits == c: the group's factors depend on the tile-local index only —
// the last stages of an inverse / the first stages of a forward transform). Per round [stage-major twiddle index j <
// 2^logr - 1][g < 2^rb]: the roots of order 2^(rb + logr) to the power g, times the constant 2^(q+1)-th roots of the stage.
void append_group_table(std::vector<uint32_t>& h, const GroupParams& gp, bool dif, unsigned* offs, size_t base) {
    const uint32_t w16 = field::root_of_unity(4), w16i = bb::inv(w16);
    uint32_t roots[8];
    { uint32_t a = bb::R_MOD_P; for (int r = 0; r < 8; ++r) { roots[r] = a; a = bb::mul(a, dif ? w16i : w16); } }
    for (int r = 0; r < gp.n_rounds; ++r) {
        const int rb = gp.rb[r], logr = gp.logr[r];
        offs[r] = (unsigned)(h.size() - base);
        const size_t G = (size_t)1 << rb, J = ((size_t)1 << logr) - 1;
        h.resize(h.size() + J * G);
        uint32_t* tab = h.data() + base + offs[r];
        uint32_t om = field::root_of_unity(rb + logr);
        if (dif) om = bb::inv(om);
        uint32_t bt = bb::R_MOD_P;  // om^g
        for (size_t g = 0; g < G; ++g) {
            uint32_t bq[4];
            bq[logr - 1] = bt;
            for (int q = logr - 2; q >= 0; --q) bq[q] = bb::sqr(bq[q + 1]);
            for (int q = 0; q < logr; ++q)
                for (int rr = 0; rr < (1 << q); ++rr)
                    tab[(((size_t)1 << q) - 1 + rr) * G + g] = rr == 0 ? bq[q] : bb::mul(bq[q], roots[rr << (3 - q)]);
            bt = bb::mul(bt, om);
        }
    }
}
struct GroupTwiddles { uint32_t* d = nullptr; unsigned off[4] = {0, 0, 0, 0}; };
std::map<std::tuple<int, int, int, int>, GroupTwiddles> g_group_tw;  // (device, dif, c, k)
// nullptr: the group is not tile-invariant (or the table could not be built): the kernel derives its twiddles
const GroupTwiddles* group_twiddles(const GroupParams& gp, bool dif) {
    if (gp.lowbits != gp.c || getenv("POWDR_NTT_NO_TABLES")) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    const auto key = std::make_tuple(device, dif ? 1 : 0, gp.c, gp.k);
    auto it = g_group_tw.find(key);
    if (it != g_group_tw.end()) return &it->second;
    GroupTwiddles gt;
    std::vector<uint32_t> h;
    append_group_table(h, gp, dif, gt.off, 0);
    if (h.empty()) return nullptr;
    if (hipMalloc(&gt.d, h.size() * 4) != hipSuccess) return nullptr;
    if (hipMemcpy(gt.d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;  // synchronous: published complete
    return &g_group_tw.emplace(key, gt).first->second;
}

// A tile-invariant group's twiddle table with a sub-coset's constants folded in: entry (round r, stage q of the round) times cq[r][q]
struct CosetTableArgs { unsigned off[4]; int rb[4], logr[4]; int n_rounds; uint32_t cq[4][4]; unsigned total; };
__global__ void coset_table_kernel(const uint32_t* __restrict__ tab, uint32_t* __restrict__ out, CosetTableArgs a) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    int r = 0;
    while (r + 1 < a.n_rounds && i >= a.off[r + 1]) ++r;
    const unsigned j = (i - a.off[r]) >> a.rb[r];  // stage-major twiddle index: stage q holds j in [2^q - 1, 2^(q+1) - 1)
    const int q = 31 - __clz((int)(j + 1));
    out[i] = bb::mul(tab[i], a.cq[r][q]);
}

// Sub-coset evaluation (subcoset_lde): the first group loads with FOLD, every stage's twiddles carry the coset constant C[stage]
struct CosetSpec {
    int fold_log;
    uint32_t foldk[16];
    uint32_t C[32];       // C[s] = c^(2^(n-1-s)), s < n
    uint32_t* d_scratch;  // room for a group's derived twiddle table (< 2^13 words)
    // SELECT (GroupParams): only with max_groups = 1 and one tile per workgroup
    int n_sel = 0;
    const uint32_t* sel_pos = nullptr;
    const uint32_t* sel_xpow = nullptr;
    unsigned long long* sel_acc = nullptr;
    uint32_t* sel_part = nullptr;
};

template <bool DIF>
void run_groups(const uint32_t* in, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n,
                int first_stage, const uint32_t* tw, const uint32_t* expand_scale_br, const char* name, const CosetSpec* cs = nullptr,
                int max_groups = -1, int* stages_done = nullptr) {
    int logt = 12;
    auto groups = plan_groups(DIF, n, first_stage, logt);
    // max_groups: only the first groups of the plan (subcoset_lde_first_group: the rest is evaluated for a few outputs only)
    if (max_groups >= 0 && (size_t)max_groups < groups.size()) groups.resize((size_t)max_groups);
    if (stages_done) { *stages_done = first_stage; for (auto& g : groups) *stages_done += g.k; }
    const bool complete = !stages_done || *stages_done == n;
    const uint32_t* src = in;
    size_t src_stride = in_stride;
    int mode = cs ? 2 : expand_scale_br != nullptr ? 1 : 0;
    if (!groups.empty() && complete) groups.back().canonical_out = 1;
    for (auto& g : groups) {
        const int glt = g.logt ? g.logt : logt;
        const size_t tiles = (size_t)1 << (n - g.B);
        const size_t per_wg = (size_t)1 << (glt - g.B);
        const unsigned wgs = (unsigned)((tiles + per_wg - 1) / per_wg);
        const GroupTwiddles* gt = group_twiddles(g, DIF);  // tile-invariant groups read their twiddles from a table
        if (gt) for (int r = 0; r < 4; ++r) g.twt_off[r] = gt->off[r];
        const uint32_t* table = gt ? gt->d : nullptr;
        if (cs) {
            g.coset = 1;
            g.fold = mode == 2 ? cs->fold_log : 0;
            for (int k = 0; k < 16; ++k) g.foldk[k] = cs->foldk[k];
            if (mode == 2 && cs->n_sel > 0) { g.n_sel = cs->n_sel; g.sel_pos = cs->sel_pos; g.sel_xpow = cs->sel_xpow; g.sel_acc = cs->sel_acc; g.sel_part = cs->sel_part; }
            for (int r = 0; r < g.n_rounds; ++r) g.cst[r] = cs->C[g.s0 + g.rb[r] + g.logr[r] - 1 - g.c];
            if (gt) {
                CosetTableArgs a{};
                a.n_rounds = g.n_rounds;
                for (int r = 0; r < g.n_rounds; ++r) {
                    a.off[r] = gt->off[r]; a.rb[r] = g.rb[r]; a.logr[r] = g.logr[r];
                    for (int q = 0; q < g.logr[r]; ++q) a.cq[r][q] = cs->C[g.s0 + g.rb[r] - g.c + q];
                }
                const int lr = g.n_rounds - 1;
                a.total = gt->off[lr] + (((1u << g.logr[lr]) - 1u) << g.rb[lr]);
                hipLaunchKernelGGL(coset_table_kernel, dim3(div_up(a.total, 256)), dim3(256), 0, stream(), gt->d, cs->d_scratch, a);
                table = cs->d_scratch;
            }
        }
        for (uint32_t c0 = 0; c0 < cols; c0 += 65535u) {
            uint32_t cc = cols - c0 < 65535u ? cols - c0 : 65535u;
            // (the sub-coset transforms' first group — FOLD loads, the sub-coset's own twiddle table — has a timer of its own)
            ScopedKernelTimer t(mode == 2 ? "ntt_subcoset_first_group_kernel" : name);
            const uint32_t* s_ = src + (size_t)c0 * src_stride;
            uint32_t* d_ = out + (size_t)c0 * out_stride;
            dim3 grid(wgs, cc), block(kBlock);
#define PW_LAUNCH_NTT(LT, MD) do { if (gt) hipLaunchKernelGGL((ntt_group_kernel<DIF, LT, MD, true>), grid, block, 0, stream(), s_, d_, src_stride, out_stride, g, table, expand_scale_br); \
                                   else hipLaunchKernelGGL((ntt_group_kernel<DIF, LT, MD, false>), grid, block, 0, stream(), s_, d_, src_stride, out_stride, g, tw, expand_scale_br); } while (0)
            if (glt == 14) hipLaunchKernelGGL((ntt_group_kernel<DIF, 14, 0, false, 1024>), grid, dim3(1024), 0, stream(), s_, d_, src_stride, out_stride, g, tw, expand_scale_br);  // (a strided group: never the first of a coset / expanding transform)
            else if (glt == 13) { if (mode == 2) PW_LAUNCH_NTT(13, 2); else if (mode == 1) PW_LAUNCH_NTT(13, 1); else PW_LAUNCH_NTT(13, 0); }
            else            { if (mode == 2) PW_LAUNCH_NTT(12, 2); else if (mode == 1) PW_LAUNCH_NTT(12, 1); else PW_LAUNCH_NTT(12, 0); }
#undef PW_LAUNCH_NTT
        }
        src = out;
        src_stride = out_stride;
        mode = 0;
        expand_scale_br = nullptr;
    }
    if (groups.empty() && in != out) {
        for (uint32_t c = 0; c < cols; ++c)
            (void)hipMemcpyAsync(out + (size_t)c * out_stride, in + (size_t)c * in_stride, (size_t)4 << n,
                                 hipMemcpyDeviceToDevice, stream());
    }
}

// out[2q] = out[2q+1] = in[q] * scale_br[q]  (only used when the forward transform has no DIT stage
// left after the duplication, i.e. H = 1)
__global__ void expand_small_kernel(const uint32_t* in, uint32_t* out, size_t in_stride, size_t out_stride, size_t h,
                                    const uint32_t* scale_br) {
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= h) return;
    uint32_t v = bb::mul(in[(size_t)blockIdx.y * in_stride + q], scale_br[q]);
    out[(size_t)blockIdx.y * out_stride + 2 * q] = v;
    out[(size_t)blockIdx.y * out_stride + 2 * q + 1] = v;
}

template <bool DIF>
void launch_groups(std::vector<GroupParams>& groups, int logt, const uint32_t* in, uint32_t* out, size_t in_stride, size_t out_stride,
                   uint32_t cols, int n, const uint32_t* tw, const char* name) {
    const uint32_t* src = in;
    size_t src_stride = in_stride;
    for (auto& g : groups) {
        const size_t tiles = (size_t)1 << (n - g.B);
        const size_t per_wg = (size_t)1 << (logt - g.B);  // (balanced plans: every group on the plan's tile size)
        const unsigned wgs = (unsigned)((tiles + per_wg - 1) / per_wg);
        for (uint32_t c0 = 0; c0 < cols; c0 += 65535u) {
            const uint32_t cc = cols - c0 < 65535u ? cols - c0 : 65535u;
            ScopedKernelTimer t(name);
            const uint32_t* s_ = src + (size_t)c0 * src_stride;
            uint32_t* d_ = out + (size_t)c0 * out_stride;
            dim3 grid(wgs, cc), block(kBlock);
            if (logt == 13) hipLaunchKernelGGL((ntt_group_kernel<DIF, 13, 0>), grid, block, 0, stream(), s_, d_, src_stride, out_stride, g, tw, (const uint32_t*)nullptr);
            else hipLaunchKernelGGL((ntt_group_kernel<DIF, 12, 0>), grid, block, 0, stream(), s_, d_, src_stride, out_stride, g, tw, (const uint32_t*)nullptr);
        }
        src = out;
        src_stride = out_stride;
    }
}

GroupParams contiguous_group(bool dif, int n, int s0, int k, int c) {
    GroupParams g{};
    g.n = n; g.s0 = s0; g.k = k; g.c = c; g.lowbits = dif ? n - s0 - k : s0; g.B = k + c;
    g.n_tiles = 1ull << (n - g.B);
    const int nr = (k + 3) / 4;
    g.n_rounds = nr;
    int widths[4];
    for (int r = 0; r < nr; ++r) widths[r] = k / nr + (r < k % nr ? 1 : 0);
    if (dif) { int top = c + k; for (int r = 0; r < nr; ++r) { top -= widths[r]; g.rb[r] = top; g.logr[r] = widths[r]; } }
    else { int bot = c; for (int r = 0; r < nr; ++r) { g.rb[r] = bot; g.logr[r] = widths[r]; bot += widths[r]; } }
    return g;
}

}  // namespace

// Twiddle tables of the fused kernel's two contiguous groups. In a contiguous group the twiddle of a butterfly depends on
// the tile-local index only — the stages are the last ones of the inverse / the first ones of the forward transform, their
// factors are the roots of order 2^(rb + logr) of a round with window [rb, rb + logr) whatever the transform size — so a
// table per round, [stage-major twiddle index j < 2^logr - 1][g < 2^rb], serves every tile, column and size: <= 46 KB for
// ka = 12, L2-resident, each of a slot's 15 loads one coalesced 256-byte read per wave. Replaces 3 squarings + 11 products
// per radix-16 slot (70 of its 390 instructions).
struct FusedTwiddles {
    uint32_t* d = nullptr;
    unsigned dif_off[4] = {0, 0, 0, 0}, dit_off[4] = {0, 0, 0, 0};  // relative to d / to d + dit_base
    size_t dit_base = 0;
};
std::map<std::pair<int, int>, FusedTwiddles> g_fused_tw;  // (device, ka)

const FusedTwiddles* fused_twiddles(int ka, const GroupParams& ga, const GroupParams& gd) {
    std::lock_guard<std::mutex> lk(g_mu);
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    auto it = g_fused_tw.find({device, ka});
    if (it != g_fused_tw.end()) return &it->second;
    FusedTwiddles ft;
    std::vector<uint32_t> h;
    append_group_table(h, ga, true, ft.dif_off, 0);
    ft.dit_base = h.size();
    append_group_table(h, gd, false, ft.dit_off, ft.dit_base);
    if (hipMalloc(&ft.d, h.size() * 4) != hipSuccess) return nullptr;
    if (hipMemcpy(ft.d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;  // synchronous: published complete
    return &g_fused_tw.emplace(std::make_pair(device, ka), ft).first->second;
}

// The whole LDE of `cols` columns: natural-order evaluations on <g_n> (in) -> natural-order evaluations on the coset
// s <g_(n+1)> (out), through the fused middle kernel: strided DIF groups (stages 0 .. n-13, into `tmp`, which needs
// cols x 2^n words and is untouched when n <= 12), lde_fused_kernel (the 12 contiguous DIF stages, scaling, duplication,
// 12 contiguous DIT stages), strided DIT groups in place on `out`. Same values as intt_dif + coset_lde_from_coeffs.
int lde_fused(const uint32_t* in, uint32_t* tmp, uint32_t* out, size_t in_stride, size_t tmp_stride, size_t out_stride, uint32_t cols, int n) {
    if (n == 0) {
        int rc = intt_dif(in, tmp, in_stride, tmp_stride, cols, n);
        return rc ? rc : coset_lde_from_coeffs(tmp, out, tmp_stride, out_stride, cols, n);
    }
    const Tables* tn = tables(n);
    const Tables* t1 = tables(n + 1);
    if (!tn || !t1) return (int)hipErrorOutOfMemory;
    const int ka = n < 12 ? n : 12;
    const uint32_t* src = in;
    size_t src_stride = in_stride;
    if (n > ka) {
        int logt = 12;
        auto groups = plan_groups(true, n, 0, logt, n - ka, true);
        launch_groups<true>(groups, logt, in, tmp, in_stride, tmp_stride, cols, n, tn->tw_inv, "ntt_group_kernel<dif>");
        src = tmp;
        src_stride = tmp_stride;
    }
    GroupParams ga = contiguous_group(true, n, n - ka, ka, 0);
    GroupParams gd = contiguous_group(false, n + 1, 1, ka, 1);
    gd.canonical_out = ka == n ? 1 : 0;
    const FusedTwiddles* ft = fused_twiddles(ka, ga, gd);
    if (!ft) return (int)hipErrorOutOfMemory;
    for (int r = 0; r < 4; ++r) { ga.twt_off[r] = ft->dif_off[r]; gd.twt_off[r] = ft->dit_off[r]; }
    {
        const size_t tiles = (size_t)1 << (n - ka);
        const size_t per_wg = (size_t)1 << (12 - ka);
        const unsigned wgs = (unsigned)((tiles + per_wg - 1) / per_wg);
        uint32_t done = 0;
        static const bool two_col = [] { const char* e = getenv("POWDR_NTT_TWO_COL"); return e && atoi(e) != 0; }();
        if (two_col && cols >= 2) {
            constexpr size_t kLds2 = 2 * ((1u << 13) + ((1u << 13) >> 5)) * sizeof(uint32_t);
            // the attribute belongs to the current device's code object: set it once per device (worker threads of pw_prove_segments_multi)
            static std::atomic<uint64_t> attr_ok{0}, attr_tried{0};
            int dev = 0;
            (void)hipGetDevice(&dev);
            const uint64_t bit = 1ull << (dev & 63);
            if (!(attr_tried.load() & bit)) {
                if (hipFuncSetAttribute((const void*)lde_fused2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds2) == hipSuccess) attr_ok |= bit;
                else (void)hipGetLastError();
                attr_tried |= bit;
            }
            if (attr_ok.load() & bit) {
                const uint32_t pairs = cols / 2;
                for (uint32_t p0 = 0; p0 < pairs; p0 += 65535u) {
                    const uint32_t pc = pairs - p0 < 65535u ? pairs - p0 : 65535u;
                    ScopedKernelTimer t("lde_fused_kernel");
                    hipLaunchKernelGGL(lde_fused2_kernel, dim3(wgs, pc), dim3(kBlock), kLds2, stream(), src + (size_t)(2 * p0) * src_stride,
                                       out + (size_t)(2 * p0) * out_stride, src_stride, out_stride, ga, gd, ft->d, ft->d + ft->dit_base, tn->shift_br);
                }
                done = 2 * pairs;
            }
        }
        for (uint32_t c0 = done; c0 < cols; c0 += 65535u) {
            const uint32_t cc = cols - c0 < 65535u ? cols - c0 : 65535u;
            ScopedKernelTimer t("lde_fused_kernel");
            hipLaunchKernelGGL(lde_fused_kernel, dim3(wgs, cc), dim3(kBlock), 0, stream(), src + (size_t)c0 * src_stride,
                               out + (size_t)c0 * out_stride, src_stride, out_stride, ga, gd, ft->d, ft->d + ft->dit_base, tn->shift_br);
        }
    }
    if (n > ka) {
        int logt = 12;
        auto groups = plan_groups(false, n + 1, ka + 1, logt, n + 1, true);
        if (!groups.empty()) groups.back().canonical_out = 1;
        launch_groups<false>(groups, logt, out, out, out_stride, out_stride, cols, n + 1, t1->tw_fwd, "ntt_group_kernel<dit>");
    }
    return (int)hipGetLastError();
}

// Unscaled inverse NTT: natural-order evaluations on <g_n> -> bit-reversed coefficient order,
// out[q] = 2^n * coefficient[bitrev(q)].
int intt_dif(const uint32_t* in, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n) {
    const Tables* t = tables(n);
    if (!t) return (int)hipErrorOutOfMemory;
    run_groups<true>(in, out, in_stride, out_stride, cols, n, 0, t->tw_inv, nullptr, "ntt_group_kernel<dif>");
    return (int)hipGetLastError();
}

// coeffs: bit-reversed order, scaled by 2^n (as intt_dif leaves them). out: 2^(n+1) natural-order
// evaluations on the coset s*<g_{n+1}>.
int coset_lde_from_coeffs(const uint32_t* coeffs, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n) {
    const Tables* tn = tables(n);
    const Tables* t1 = tables(n + 1);
    if (!tn || !t1) return (int)hipErrorOutOfMemory;
    if (n == 0) {
        ScopedKernelTimer t("expand_small_kernel");
        hipLaunchKernelGGL(expand_small_kernel, dim3(1, cols), dim3(64), 0, stream(), coeffs, out, in_stride, out_stride,
                           (size_t)1, tn->shift_br);
        return (int)hipGetLastError();
    }
    // stage 0 of the size-2^(n+1) DIT is the duplication done by the EXPAND load of the first group
    run_groups<false>(coeffs, out, in_stride, out_stride, cols, n + 1, 1, t1->tw_fwd, tn->shift_br, "ntt_group_kernel<dit>");
    return (int)hipGetLastError();
}

// ---- sub-coset evaluation (the streamed prover, prover.hip "streamed proofs") ---------------------------------------------
// The LDE domain s <g_(n+1)> (2^(n+1) points) is the union of 2^b sub-cosets c <g_m>, c = s g_(n+1)^r, m = 2^(n+1-b), r < 2^b: rows
// j = r + 2^b i of the LDE. On one of them x^m is the constant c^m, so a polynomial of degree < 2^n is its remainder modulo
// x^m - c^m there: b_i = sum_t a_(i + t m) c^(t m), 2^(b-1) coefficients per position — contiguous in the bit-reversed order of the
// coefficient arrays, their factors the same for every position (kernel arguments) — and the remainder is evaluated on c <g_m> by a
// size-m COSET transform: the DIT network with the twiddles of stage s multiplied by c^(2^(log2 m - 1 - s)) (P(x) = E(x^2) + x O(x^2)
// level by level), i.e. no scaling pass and no per-element scale table. Work per column and sub-coset: 2^n multiply-adds +
// (m / 2) log2 m butterflies; all 2^b sub-cosets together read the coefficients 2^(b-1) times for one forward transform's butterflies.
namespace {
// the sub-coset's constants: c = s g_(n+1)^r; C[s] = c^(2^(nm-1-s)); fold factors (c^m)^bitrev(t) / H
bool subcoset_spec(int n, int b, uint32_t r, uint32_t* d_scratch, CosetSpec& cs) {
    const int nm = n + 1 - b;
    if (b < 1 || b > 5 || nm < 1 || (r >> b)) return false;
    cs = CosetSpec{};
    cs.fold_log = b - 1;
    cs.d_scratch = d_scratch;
    uint32_t pw[32];
    pw[0] = bb::mul(bb::to_monty(field::kCosetShift), bb::pow_u32(field::root_of_unity(n + 1), r));
    for (int e = 1; e <= nm; ++e) pw[e] = bb::sqr(pw[e - 1]);
    for (int s = 0; s < nm; ++s) cs.C[s] = pw[nm - 1 - s];
    const uint32_t hinv = bb::inv(bb::to_monty((uint32_t)(((uint64_t)1 << n) % bb::P)));  // the coefficient arrays are H-scaled
    const int f = b - 1;
    for (uint32_t tp = 0; tp < (1u << f); ++tp) {
        const uint32_t t = f ? (__builtin_bitreverse32(tp) >> (32 - f)) : 0u;
        cs.foldk[tp] = bb::mul(bb::pow_u32(pw[nm], t), hinv);
    }
    return true;
}

// After the first `k1` DIT stages block t of 2^k1 contiguous elements holds P_u(y), u = bitrev(t), on the points y = x^(2^k2) of its
// own coset, where P(x) = sum_u x^u P_u(x^(2^k2)), k2 = nm - k1. Output i of the whole transform is then a 2^k2-term sum:
//   out[i] = sum_u x_i^u part[bitrev_k2(u) * 2^k1 + (i mod 2^k1)],  x_i = c w_m^i
// — what the remaining (strided) stages compute for EVERY i; the query phase needs ~a hundred of them. One thread per (row, column).
// The partial values are the forward network's [0, 2p) representatives.
__global__ __launch_bounds__(256) void subcoset_rows_kernel(const uint32_t* __restrict__ part, size_t stride, uint32_t cols, int k1, int k2, uint32_t c0,
                                                            uint32_t wm, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ slot,
                                                            uint32_t* __restrict__ out) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= cols) return;
    const uint32_t i = idx[blockIdx.y];
    const uint32_t x = bb::mul(c0, bb::pow_u32(wm, i));
    const uint32_t* col = part + (size_t)c * stride + (i & ((1u << k1) - 1u));
    uint32_t acc = 0u, xp = bb::R_MOD_P;
    for (uint32_t u = 0; u < (1u << k2); ++u) {
        const uint32_t t = k2 ? (__brev(u) >> (32 - k2)) : 0u;
        acc = bb::add(acc, bb::mul(bb::reduce_2p(col[(size_t)t << k1]), xp));
        xp = bb::mul(xp, x);
    }
    out[(size_t)(slot ? slot[blockIdx.y] : blockIdx.y) * cols + c] = acc;
}
// xpow[q * n_tiles + t] = x_q^(bitrev_k2(t)), x_q = c0 wm^(idx[q]): the factor tile t's element contributes to output idx[q] with
__global__ __launch_bounds__(256) void select_xpow_kernel(const uint32_t* __restrict__ idx, uint32_t n_idx, int k2, uint32_t c0, uint32_t wm,
                                                         uint32_t* __restrict__ xpow, uint32_t* __restrict__ pos, uint32_t pos_mask) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n_tiles = (size_t)1 << k2;
    if (gid >= (size_t)n_idx * n_tiles) return;
    const uint32_t q = (uint32_t)(gid >> k2), t = (uint32_t)(gid & (n_tiles - 1));
    const uint32_t i = idx[q];
    const uint32_t x = bb::mul(c0, bb::pow_u32(wm, i));
    const uint32_t u = k2 ? (__brev(t) >> (32 - k2)) : 0u;
    xpow[gid] = bb::pow_u32(x, u);
    if (t == 0) pos[q] = i & pos_mask;
}

// rows_out[slot(q) * cols + c] = acc[c * n + q] mod p, as a Montgomery word in [0, p)
__global__ __launch_bounds__(256) void select_finish_kernel(const unsigned long long* __restrict__ acc, uint32_t n, uint32_t cols,
                                                           const uint32_t* __restrict__ slot, uint32_t* __restrict__ out) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= cols) return;
    const uint32_t q = blockIdx.y;
    out[(size_t)(slot ? slot[q] : q) * cols + c] = (uint32_t)(acc[(size_t)c * n + q] % bb::P);
}
// rows_out[slot(q) * cols + c] = sum_t part[(c * n_tiles + t) * n + q] mod p: lane = output q (a tile's n terms are consecutive words),
// the tiles of the column one after the other. 2^k2 <= 2^14 terms below p: the sum fits 64 bits with room to spare.
__global__ __launch_bounds__(128) void select_reduce_kernel(const uint32_t* __restrict__ part, uint32_t n, uint32_t cols, size_t n_tiles,
                                                           const uint32_t* __restrict__ slot, uint32_t* __restrict__ out) {
    const uint32_t q = blockIdx.x * 128u + threadIdx.x, c = blockIdx.y;
    if (q >= n) return;
    const uint32_t* col = part + (size_t)c * n_tiles * n + q;
    unsigned long long acc = 0;
#pragma unroll 8
    for (size_t t = 0; t < n_tiles; ++t) acc += col[t * n];
    out[(size_t)(slot ? slot[q] : q) * cols + c] = (uint32_t)(acc % bb::P);
}
}  // namespace

// The rows d_local_idx[q] (q < n_idx) of sub-coset r's LDE WITHOUT storing the partial transform: the first stage group with SELECT
// (the tile stays in LDS, every selected output gets this tile's term through a 64-bit atomic sum) — the query phase of a streamed
// proof reads the coefficients once and writes a few megabytes. Tall transforms only (one 2^12-tile per workgroup and at least one
// strided stage left); returns 1 when the shape does not qualify and nothing was done (the caller then takes
// subcoset_lde_first_group + subcoset_rows). d_work: 2^13 + n_idx * (2^k2 + 1) words + cols * n_idx 64-bit sums.
int subcoset_query_rows(const uint32_t* coeffs, size_t in_stride, uint32_t cols, int n, int b, uint32_t r, const uint32_t* d_local_idx,
                        uint32_t n_idx, const uint32_t* d_slot, uint32_t* rows_out, uint32_t* d_work, size_t work_words) {
    const int nm = n + 1 - b;
    if (!n_idx || !cols) return 0;
    if (cols > 65535u) return 1;  // (one launch: blockIdx.y is the column)
    // POWDR_QUERY_SELECT: 0 = never (the stored partial transform + subcoset_rows), 2 = round 5's 64-bit atomic sums, default (1) =
    // every workgroup STORES its n_idx terms and select_reduce_kernel adds a column's tiles: the same words, no atomics — tiles x cols x
    // n_idx of them per pass (4.3e8 at configs[2]), all tiles of a column on the same n_idx addresses — and a kernel that profiles
    // under rocprofv3 --pmc (the atomic form's FETCH_SIZE pass did not return: DESIGN §3.8 round 6, tools/repro_select_atomics.py)
    const int sel_mode = getenv("POWDR_QUERY_SELECT") ? atoi(getenv("POWDR_QUERY_SELECT")) : 1;
    if (sel_mode == 0) return 1;
    CosetSpec cs;
    if (!subcoset_spec(n, b, r, d_work, cs)) return (int)hipErrorInvalidValue;
    int logt = 12;
    auto groups = plan_groups(false, nm, 0, logt);
    if (groups.size() < 2 || (groups[0].logt ? groups[0].logt : logt) != 12 || groups[0].B != 12 || groups[0].c != 0 || groups[0].lowbits != 0 || cs.fold_log > 1) return 1;
    const int k1 = groups[0].k, k2 = nm - k1;
    const size_t n_tiles = (size_t)1 << k2;
    uint32_t* d_pos = d_work + (1u << 13);
    uint32_t* d_xpow = d_pos + ((n_idx + 1u) & ~1u);
    unsigned long long* d_acc = reinterpret_cast<unsigned long long*>(d_xpow + (((size_t)n_idx * n_tiles + 1) & ~(size_t)1));
    uint32_t* d_part = reinterpret_cast<uint32_t*>(d_acc + (size_t)cols * n_idx);
    bool partial = sel_mode != 2;
    size_t need = (size_t)(d_part - d_work) + (partial ? (size_t)cols * n_tiles * n_idx : 0);
    if (partial && need > work_words) { partial = false; need = (size_t)(d_part - d_work); }  // (no room for the terms: the atomic sums)
    if (need > work_words || ((uintptr_t)d_work & 7)) return 1;
    const Tables* tm = tables(nm);
    if (!tm) return (int)hipErrorOutOfMemory;
    const uint32_t c0 = bb::mul(bb::to_monty(field::kCosetShift), bb::pow_u32(field::root_of_unity(n + 1), r));
    {
        ScopedKernelTimer t("subcoset_rows_kernel");
        hipLaunchKernelGGL(select_xpow_kernel, dim3((unsigned)div_up((size_t)n_idx * n_tiles, 256)), dim3(256), 0, stream(), d_local_idx, n_idx, k2, c0,
                           field::root_of_unity(nm), d_xpow, d_pos, (1u << k1) - 1u);
    }
    if (!partial) PW_HIP_TRY(hipMemsetAsync(d_acc, 0, (size_t)cols * n_idx * sizeof(unsigned long long), stream()));
    cs.n_sel = (int)n_idx; cs.sel_pos = d_pos; cs.sel_xpow = d_xpow; cs.sel_acc = d_acc; cs.sel_part = partial ? d_part : nullptr;
    int done = 0;
    // (`out` of the group is never written in SELECT mode: the coefficient array stands in for it)
    run_groups<false>(coeffs, const_cast<uint32_t*>(coeffs), in_stride, in_stride, cols, nm, 0, tm->tw_fwd, nullptr, "ntt_group_kernel<dit>", &cs, 1, &done);
    {
        ScopedKernelTimer t("subcoset_rows_kernel");
        if (partial) hipLaunchKernelGGL(select_reduce_kernel, dim3(div_up(n_idx, 128), cols), dim3(128), 0, stream(), d_part, n_idx, cols, n_tiles, d_slot, rows_out);
        else hipLaunchKernelGGL(select_finish_kernel, dim3(div_up(cols, 256), n_idx), dim3(256), 0, stream(), d_acc, n_idx, cols, d_slot, rows_out);
    }
    return (int)hipGetLastError();
}

int subcoset_lde(const uint32_t* coeffs, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n, int b, uint32_t r,
                 uint32_t* d_scratch) {
    const int nm = n + 1 - b;  // log2 of the sub-coset's size
    CosetSpec cs;
    if (!subcoset_spec(n, b, r, d_scratch, cs)) return (int)hipErrorInvalidValue;
    const Tables* tm = tables(nm);
    if (!tm) return (int)hipErrorOutOfMemory;
    run_groups<false>(coeffs, out, in_stride, out_stride, cols, nm, 0, tm->tw_fwd, nullptr, "ntt_group_kernel<dit>", &cs);
    return (int)hipGetLastError();
}

// Coefficient arrays as intt_dif leaves them (bit-reversed, H-scaled) back to the values on <g_n>, natural order, canonical words: the
// forward network of subcoset_lde with the trivial coset (every stage constant 1) and the single "fold" factor 1/H. In place or not.
int values_from_coefficients(const uint32_t* coeffs, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n, uint32_t* d_scratch) {
    if (n < 1 || n > 26) return (int)hipErrorInvalidValue;
    CosetSpec cs{};
    cs.fold_log = 0;
    cs.d_scratch = d_scratch;
    for (int s = 0; s < 32; ++s) cs.C[s] = bb::R_MOD_P;
    cs.foldk[0] = bb::inv(bb::to_monty((uint32_t)(((uint64_t)1 << n) % bb::P)));
    const Tables* tn = tables(n);
    if (!tn) return (int)hipErrorOutOfMemory;
    run_groups<false>(coeffs, out, in_stride, out_stride, cols, n, 0, tn->tw_fwd, nullptr, "ntt_group_kernel<dit>", &cs);
    return (int)hipGetLastError();
}

// Only the first stage group of subcoset_lde (the contiguous stages, with the FOLD loads): `out` holds the PARTIAL transform,
// *stages_done stages of log2 m. subcoset_rows finishes it for chosen rows.
int subcoset_lde_first_group(const uint32_t* coeffs, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n, int b, uint32_t r,
                             uint32_t* d_scratch, int* stages_done) {
    const int nm = n + 1 - b;
    CosetSpec cs;
    if (!subcoset_spec(n, b, r, d_scratch, cs) || !stages_done) return (int)hipErrorInvalidValue;
    const Tables* tm = tables(nm);
    if (!tm) return (int)hipErrorOutOfMemory;
    run_groups<false>(coeffs, out, in_stride, out_stride, cols, nm, 0, tm->tw_fwd, nullptr, "ntt_group_kernel<dit>", &cs, 1, stages_done);
    return (int)hipGetLastError();
}

// out[slot(q) * cols + c] = row d_local_idx[q] of the sub-coset's LDE (canonical Montgomery words), from the partial transform;
// slot(q) = d_slot[q], or q when d_slot is null
int subcoset_rows(const uint32_t* part, size_t stride, uint32_t cols, int n, int b, uint32_t r, int stages_done, const uint32_t* d_local_idx,
                  uint32_t n_idx, const uint32_t* d_slot, uint32_t* out) {
    const int nm = n + 1 - b;
    if (!n_idx || !cols) return 0;
    if (stages_done < 0 || stages_done > nm) return (int)hipErrorInvalidValue;
    const uint32_t c0 = bb::mul(bb::to_monty(field::kCosetShift), bb::pow_u32(field::root_of_unity(n + 1), r));
    ScopedKernelTimer t("subcoset_rows_kernel");
    hipLaunchKernelGGL(subcoset_rows_kernel, dim3(div_up(cols, 256), n_idx), dim3(256), 0, stream(), part, stride, cols, stages_done, nm - stages_done, c0,
                       field::root_of_unity(nm), d_local_idx, d_slot, out);
    return (int)hipGetLastError();
}

const uint32_t* shift_table(int n) { const Tables* t = tables(n); return t ? t->shift : nullptr; }

}  // namespace pw
