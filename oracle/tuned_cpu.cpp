/* TEST / BASELINE INFRASTRUCTURE — not part of the product (only tests/ and bench.py's cpu_baseline leg use it).
 *
 * A TUNED CPU implementation of the two stages that dominate the proof — low-degree extension and Poseidon2 Merkle
 * commitment — so that bench.py's `cpu_baseline` can quote a figure a production CPU prover (Plonky3-style: Montgomery
 * arithmetic, AVX-512, one matrix batch per core, no per-row allocation) could plausibly reach, next to the deliberately
 * naive `u64 %` oracle that serves as the parity checker. Same results as the oracle
 * (oracle/stark_oracle.cpp or_lde / or_merkle_commit), checked in tests/test_tuned_cpu.py.
 *
 *   LDE       like Plonky3's radix-2 DIT on row-major matrices: vector lane = COLUMN. 16 columns are transposed into an
 *             interleaved [row][16] buffer, inverse DFT (DIF, natural -> bit-reversed), coset scaling, zero extension in
 *             bit-reversed order (= duplication), forward DFT (DIT, bit-reversed -> natural), transposed back. Every butterfly
 *             of every stage is one 16-lane Montgomery product + add + sub; twiddles are broadcast scalars from per-size tables.
 *   Merkle    vector lane = ROW: 16 consecutive rows of the column-major matrix are 16 sponges; inner nodes 16 at a time.
 * Build: g++ -O3 -march=native -fopenmp (AVX-512F required for the vector path; a scalar Montgomery path otherwise). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#if defined(__AVX512F__)
#include <immintrin.h>
#define TC_AVX512 1
#else
#define TC_AVX512 0
#endif

namespace {

typedef uint32_t u32;
typedef uint64_t u64;
constexpr u32 P = 0x78000001u;
constexpr u32 MU = 0x88000001u;      /* p^-1 mod 2^32 */
constexpr u32 R1 = 0x0ffffffeu;      /* 2^32 mod p */
constexpr u32 R2 = 1172168163u;      /* 2^64 mod p */
constexpr u32 TWO_ADIC_GEN = 0x1a427a41u, COSET_SHIFT = 31u;

inline u32 sadd(u32 a, u32 b) { u32 s = a + b; return s >= P ? s - P : s; }
inline u32 ssub(u32 a, u32 b) { return a >= b ? a - b : a + P - b; }
inline u32 smul(u32 a, u32 b) {  /* Montgomery product */
    u64 t = (u64)a * b;
    u32 q = (u32)t * MU;
    int64_t d = (int64_t)(t >> 32) - (int64_t)(((u64)q * P) >> 32);
    return d < 0 ? (u32)(d + P) : (u32)d;
}
inline u32 to_m(u32 c) { return smul(c, R2); }
inline u32 from_m(u32 m) { return smul(m, 1u); }
u32 spow(u32 a, u64 e) { u32 r = R1; while (e) { if (e & 1) r = smul(r, a); a = smul(a, a); e >>= 1; } return r; }
inline u32 sinv(u32 a) { return spow(a, P - 2); }
u32 root_of_unity_m(int log_n) { u32 g = to_m(TWO_ADIC_GEN); for (int i = log_n; i < 27; ++i) g = smul(g, g); return g; }
inline size_t bitrev(size_t x, int bits) { size_t r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

#if TC_AVX512
typedef __m512i V;
inline V vset(u32 x) { return _mm512_set1_epi32((int)x); }
inline V vadd(V a, V b) { V s = _mm512_add_epi32(a, b); return _mm512_min_epu32(s, _mm512_sub_epi32(s, vset(P))); }
inline V vsub(V a, V b) { V d = _mm512_sub_epi32(a, b); return _mm512_min_epu32(d, _mm512_add_epi32(d, vset(P))); }
inline V vmul(V a, V b) {
    const V pv = vset(P), mu = vset(MU);
    V ae = a, ao = _mm512_srli_epi64(a, 32), be = b, bo = _mm512_srli_epi64(b, 32);
    V pe = _mm512_mul_epu32(ae, be), po = _mm512_mul_epu32(ao, bo);
    V qe = _mm512_mul_epu32(pe, mu), qo = _mm512_mul_epu32(po, mu);
    V qpe = _mm512_mul_epu32(qe, pv), qpo = _mm512_mul_epu32(qo, pv);
    V de = _mm512_sub_epi64(pe, qpe), d_o = _mm512_sub_epi64(po, qpo);  /* multiples of 2^32, signed, |.| < p 2^32 */
    V r = _mm512_mask_blend_epi32(0xAAAA, _mm512_srli_epi64(de, 32), d_o);
    return _mm512_min_epu32(r, _mm512_add_epi32(r, pv));  /* negative lanes (huge as unsigned) get + p */
}
#else
struct V { u32 l[16]; };
inline V vset(u32 x) { V v; for (auto& e : v.l) e = x; return v; }
inline V vadd(V a, V b) { V r; for (int i = 0; i < 16; ++i) r.l[i] = sadd(a.l[i], b.l[i]); return r; }
inline V vsub(V a, V b) { V r; for (int i = 0; i < 16; ++i) r.l[i] = ssub(a.l[i], b.l[i]); return r; }
inline V vmul(V a, V b) { V r; for (int i = 0; i < 16; ++i) r.l[i] = smul(a.l[i], b.l[i]); return r; }
#endif
inline V vload(const u32* p) {
#if TC_AVX512
    return _mm512_loadu_si512((const void*)p);
#else
    V v; memcpy(v.l, p, 64); return v;
#endif
}
inline void vstore(u32* p, V v) {
#if TC_AVX512
    _mm512_storeu_si512((void*)p, v);
#else
    memcpy(p, v.l, 64);
#endif
}

/* ---- NTT on an interleaved [n][16] buffer: lane = column ------------------------------------------------------------ */
struct Twiddles {
    int log_n;
    std::vector<u32> w;  /* w[k] = g^k, k < n/2 (Montgomery), g = primitive n-th root (or its inverse) */
};
Twiddles make_twiddles(int log_n, bool inverse) {
    Twiddles t;
    t.log_n = log_n;
    const size_t half = log_n ? (size_t)1 << (log_n - 1) : 1;
    t.w.resize(half);
    u32 g = root_of_unity_m(log_n);
    if (inverse) g = sinv(g);
    u32 x = R1;
    for (size_t k = 0; k < half; ++k) { t.w[k] = x; x = smul(x, g); }
    return t;
}
/* DIF: natural order in, bit-reversed order out (inverse transform uses inverse roots; unscaled) */
void dif16(u32* a, int log_n, const Twiddles& tw) {
    const size_t n = (size_t)1 << log_n;
    for (int s = log_n; s >= 1; --s) {
        const size_t m = (size_t)1 << (s - 1), step = n >> s;
        for (size_t blk = 0; blk < n; blk += 2 * m)
            for (size_t j = 0; j < m; ++j) {
                u32* x = a + (blk + j) * 16; u32* y = x + m * 16;
                const V u = vload(x), v = vload(y);
                vstore(x, vadd(u, v));
                vstore(y, vmul(vsub(u, v), vset(tw.w[j * step])));
            }
    }
}
/* DIT: bit-reversed order in, natural order out */
void dit16(u32* a, int log_n, const Twiddles& tw) {
    const size_t n = (size_t)1 << log_n;
    for (int s = 1; s <= log_n; ++s) {
        const size_t m = (size_t)1 << (s - 1), step = n >> s;
        for (size_t blk = 0; blk < n; blk += 2 * m)
            for (size_t j = 0; j < m; ++j) {
                u32* x = a + (blk + j) * 16; u32* y = x + m * 16;
                const V u = vload(x), v = vmul(vload(y), vset(tw.w[j * step]));
                vstore(x, vadd(u, v));
                vstore(y, vsub(u, v));
            }
    }
}

/* ---- Poseidon2 (width 16), 16 states at a time: lane = state ---------------------------------------------------------- */
struct P2Consts { u32 ext_rc[8][16], int_rc[13], diag[16]; bool ready = false; };
P2Consts g_p2;  /* Montgomery */

inline V sbox7(V x) { V x2 = vmul(x, x), x3 = vmul(x2, x), x4 = vmul(x2, x2); return vmul(x3, x4); }
inline void external_layer(V* s) {
    /* M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] with shared sums, then every word gets its column sum over the blocks */
    for (int b = 0; b < 4; ++b) {
        V x0 = s[4 * b], x1 = s[4 * b + 1], x2 = s[4 * b + 2], x3 = s[4 * b + 3];
        V t01 = vadd(x0, x1), t23 = vadd(x2, x3), t = vadd(t01, t23);
        V ta = vadd(t, x1), tb = vadd(t, x3);
        s[4 * b] = vadd(ta, t01);
        s[4 * b + 1] = vadd(ta, vadd(x2, x2));
        s[4 * b + 2] = vadd(tb, t23);
        s[4 * b + 3] = vadd(tb, vadd(x0, x0));
    }
    for (int i = 0; i < 4; ++i) {
        V col = vadd(vadd(s[i], s[4 + i]), vadd(s[8 + i], s[12 + i]));
        for (int b = 0; b < 4; ++b) s[4 * b + i] = vadd(s[4 * b + i], col);
    }
}
inline void permute16(V* s) {
    const P2Consts& C = g_p2;
    external_layer(s);
    for (int r = 0; r < 4; ++r) {
        for (int i = 0; i < 16; ++i) s[i] = sbox7(vadd(s[i], vset(C.ext_rc[r][i])));
        external_layer(s);
    }
    for (int r = 0; r < 13; ++r) {
        s[0] = sbox7(vadd(s[0], vset(C.int_rc[r])));
        V sum = s[0];
        for (int i = 1; i < 16; ++i) sum = vadd(sum, s[i]);
        for (int i = 0; i < 16; ++i) s[i] = vadd(sum, vmul(s[i], vset(C.diag[i])));
    }
    for (int r = 4; r < 8; ++r) {
        for (int i = 0; i < 16; ++i) s[i] = sbox7(vadd(s[i], vset(C.ext_rc[r][i])));
        external_layer(s);
    }
}

}  // namespace

extern "C" {

int tc_has_avx512(void) { return TC_AVX512; }

/* canonical tables (as or_poseidon2_constants returns them) */
void tc_set_poseidon2_constants(const uint32_t* ext_rc, const uint32_t* int_rc, const uint32_t* diag) {
    for (int r = 0; r < 8; ++r) for (int i = 0; i < 16; ++i) g_p2.ext_rc[r][i] = to_m(ext_rc[16 * r + i]);
    for (int r = 0; r < 13; ++r) g_p2.int_rc[r] = to_m(int_rc[r]);
    for (int i = 0; i < 16; ++i) g_p2.diag[i] = to_m(diag[i]);
    g_p2.ready = true;
}

/* out (width x 2H, column-major, canonical) = evaluations on 31 * <g_{n+1}> of the columns of trace (width x H, canonical) */
void tc_lde(const uint32_t* trace, uint32_t width, int log_h, uint32_t* out) {
    const size_t H = (size_t)1 << log_h, N = 2 * H;
    const Twiddles inv = make_twiddles(log_h, true), fwd = make_twiddles(log_h + 1, false);
    /* per row q (bit-reversed coefficient index): s^bitrev(q) / H, Montgomery */
    std::vector<u32> scale(H);
    {
        const u32 hinv = sinv(to_m((u32)(H % P))), s = to_m(COSET_SHIFT);
        std::vector<u32> pw(H);
        u32 x = hinv;
        for (size_t k = 0; k < H; ++k) { pw[k] = x; x = smul(x, s); }
        for (size_t q = 0; q < H; ++q) scale[q] = pw[bitrev(q, log_h)];
    }
    const V r2 = vset(R2), one = vset(1u);
#pragma omp parallel
    {
        std::vector<u32> buf(N * 16 + 16);
#pragma omp for schedule(dynamic)
        for (long c0 = 0; c0 < (long)width; c0 += 16) {
            const int nc = (int)((long)width - c0 < 16 ? (long)width - c0 : 16);
            /* transpose in (and to Montgomery form) */
            for (size_t r = 0; r < H; ++r) {
                u32 row[16] = {0};
                for (int c = 0; c < nc; ++c) row[c] = trace[(size_t)(c0 + c) * H + r];
                vstore(&buf[r * 16], vmul(vload(row), r2));
            }
            dif16(buf.data(), log_h, inv);
            /* scale, move to the bit-reversed positions of the zero-extended vector (2q, and 2q + 1 = the duplicate the first DIT stage would produce) */
            for (size_t q = H; q-- > 0;) {
                const V v = vmul(vload(&buf[q * 16]), vset(scale[q]));
                vstore(&buf[(2 * q) * 16], v);
                vstore(&buf[(2 * q + 1) * 16], v);
            }
            /* forward DIT on 2H points, first stage already done (pairs hold (c, c)): the butterfly of stage 1 has twiddle 1: (c + 0, c - 0) */
            {
                const size_t n = N;
                for (int s = 2; s <= log_h + 1; ++s) {
                    const size_t m = (size_t)1 << (s - 1), step = n >> s;
                    for (size_t blk = 0; blk < n; blk += 2 * m)
                        for (size_t j = 0; j < m; ++j) {
                            u32* x = &buf[(blk + j) * 16]; u32* y = x + m * 16;
                            const V u = vload(x), v = vmul(vload(y), vset(fwd.w[j * step]));
                            vstore(x, vadd(u, v));
                            vstore(y, vsub(u, v));
                        }
                }
            }
            /* transpose out (canonical) */
            for (size_t r = 0; r < N; ++r) {
                u32 row[16];
                vstore(row, vmul(vload(&buf[r * 16]), one));
                for (int c = 0; c < nc; ++c) out[(size_t)(c0 + c) * N + r] = row[c];
            }
        }
    }
}

/* Poseidon2 Merkle root of a column-major canonical matrix (height a power of two >= 16): overwrite-mode sponge of rate 8
 * over each row, binary tree of 2-to-1 compressions (oracle/stark_oracle.cpp hash_row / compress). root8 canonical. */
void tc_merkle_root(const uint32_t* m, size_t height, uint32_t width, uint32_t* root8) {
    std::vector<u32> level(height * 8);  /* digest-major: level[k * height + row] (Montgomery) */
    const V r2 = vset(R2);
#pragma omp parallel for schedule(static)
    for (long r0 = 0; r0 < (long)height; r0 += 16) {
        V st[16];
        for (auto& v : st) v = vset(0u);
        for (uint32_t off = 0; off < width; off += 8) {
            const uint32_t k = width - off < 8 ? width - off : 8;
            for (uint32_t i = 0; i < k; ++i) st[i] = vmul(vload(m + (size_t)(off + i) * height + r0), r2);
            permute16(st);
        }
        for (int k = 0; k < 8; ++k) vstore(&level[(size_t)k * height + r0], st[k]);
    }
    size_t n = height;
    std::vector<u32> next;
    while (n > 1) {
        const size_t half = n / 2;
        next.assign(half * 8 + 16, 0u);
        if (half >= 16) {
#pragma omp parallel for schedule(static)
            for (long i0 = 0; i0 < (long)half; i0 += 16) {
                V st[16];
                u32 l[16], r[16];
                for (int k = 0; k < 8; ++k) {
                    for (int j = 0; j < 16; ++j) { l[j] = level[(size_t)k * n + 2 * (i0 + j)]; r[j] = level[(size_t)k * n + 2 * (i0 + j) + 1]; }
                    st[k] = vload(l); st[8 + k] = vload(r);
                }
                permute16(st);
                for (int k = 0; k < 8; ++k) vstore(&next[(size_t)k * half + i0], st[k]);
            }
        } else {
            V st[16];
            u32 l[16] = {0}, r[16] = {0}, o[16];
            for (int k = 0; k < 8; ++k) {
                for (size_t j = 0; j < half; ++j) { l[j] = level[(size_t)k * n + 2 * j]; r[j] = level[(size_t)k * n + 2 * j + 1]; }
                st[k] = vload(l); st[8 + k] = vload(r);
            }
            permute16(st);
            for (int k = 0; k < 8; ++k) { vstore(o, st[k]); for (size_t j = 0; j < half; ++j) next[(size_t)k * half + j] = o[j]; }
        }
        level.swap(next);
        n = half;
    }
    for (int k = 0; k < 8; ++k) root8[k] = from_m(level[k]);
}

}  /* extern "C" */
