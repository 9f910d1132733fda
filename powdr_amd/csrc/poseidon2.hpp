// Poseidon2 over BabyBear, width 16, x^7, 8 external + 13 internal rounds, on
// Montgomery words. Used by the Merkle kernels (device) and by the Fiat-Shamir
// transcript (host). Shape as restated in oracle/stark_oracle.cpp (Plonky3's
// BabyBear instance shape; the round-constant table is EXTERNAL to the reference
// checkout and therefore generated here by the documented splitmix64 stream —
// one swappable table, see DESIGN.md "pw-stark v0").
//
// Cost per permutation: S-boxes 8*16*4 + 13*4 = 564 Montgomery products, the internal diagonal adds 7 products
// per internal round; the linear layers are sums with coefficients <= 4, done as v_mad_u64_u32 multiply-adds into
// 64-bit accumulators with one reduction per output (bb::wide_fma / reduce_wide). The MDS layer is not a dense
// contraction worth an MFMA.
#pragma once
#include "babybear.hpp"

#ifndef PW_P2_UNROLL
#define PW_P2_UNROLL 1
#endif
#define PW_P2_STR2(x) #x
#define PW_P2_STR(x) PW_P2_STR2(x)
#define PW_P2_ROUND_LOOP _Pragma(PW_P2_STR(unroll PW_P2_UNROLL))
#ifndef PW_P2_UNROLL_PARTIAL
#define PW_P2_UNROLL_PARTIAL 1
#endif
#define PW_P2_PARTIAL_LOOP _Pragma(PW_P2_STR(unroll PW_P2_UNROLL_PARTIAL))

namespace p2 {

struct Params {
    uint32_t ext_rc[8][16];
    uint32_t int_rc[13];
    uint32_t diag[16];
    // ext_rc[r] folded into the external linear layer that precedes round r (external_layer_fold). The layer adds to every
    // output the sum of its column over the four blocks, so block q, column i must carry f[q][i] = c[q][i] - (sum_q' c[q'][i]) / 5
    // before the column sums. The M4 network shares its partial sums, so a block takes its constants as two seeds
    // (a into x0 + x1, b into x2 + x3: outputs get 2a + b, a + b, a + 2b, a + b) and two corrections:
    // ext_fold[r][4q + {0,1,2,3}] = {a, b, f[q][1] - a - b, f[q][3] - a - b} with a = (2 f0 - f2) / 3, b = (2 f2 - f0) / 3;
    // 64-bit words because they enter 64-bit accumulators from a scalar register pair
    uint64_t ext_fold[8][16];
    // the constant s_0 meets next, as a raw product c * (R mod p) that joins s_0's multiply-add of partial round r
    // (internal_layer): int_rc[r + 1] for r < 12, ext_rc[4][0] for the last one
    uint64_t int_fold[13];
};

// host-side generation (Montgomery form)
inline void generate_params(Params& p) {
    uint64_t s = 0x506F736569646F6Eull;  // "Poseidon"
    auto next = [&]() -> uint32_t {
        for (;;) {
            s += 0x9E3779B97F4A7C15ull;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            uint32_t v = (uint32_t)(z & 0x7fffffffu);
            if (v < bb::P) return bb::to_monty(v);
        }
    };
    for (int r = 0; r < 8; ++r)
        for (int i = 0; i < 16; ++i) p.ext_rc[r][i] = next();
    for (int r = 0; r < 13; ++r) p.int_rc[r] = next();
    auto m = [](uint32_t c) { return bb::to_monty(c); };
    auto inv2k = [&](int k) { return bb::inv(bb::pow_u32(m(2), (uint32_t)k)); };
    const uint32_t d[16] = {bb::neg(m(2)), m(1), m(2), inv2k(1), m(3), m(4), bb::neg(inv2k(1)), bb::neg(m(3)),
                            bb::neg(m(4)), inv2k(8), inv2k(2), inv2k(3), inv2k(27), bb::neg(inv2k(8)),
                            bb::neg(inv2k(4)), bb::neg(inv2k(27))};
    for (int i = 0; i < 16; ++i) p.diag[i] = d[i];
    const uint32_t inv5 = bb::inv(m(5));
    for (int r = 0; r < 8; ++r)
        for (int i = 0; i < 4; ++i) {
            uint32_t col = 0;
            for (int q = 0; q < 4; ++q) col = bb::add(col, p.ext_rc[r][4 * q + i]);
            const uint32_t t = bb::mul(col, inv5);
            for (int q = 0; q < 4; ++q) p.ext_fold[r][4 * q + i] = bb::sub(p.ext_rc[r][4 * q + i], t);
        }
    const uint32_t inv3 = bb::inv(m(3));
    for (int r = 0; r < 8; ++r)
        for (int q = 0; q < 4; ++q) {
            const uint32_t f0 = (uint32_t)p.ext_fold[r][4 * q], f1 = (uint32_t)p.ext_fold[r][4 * q + 1], f2 = (uint32_t)p.ext_fold[r][4 * q + 2],
                           f3 = (uint32_t)p.ext_fold[r][4 * q + 3];
            const uint32_t a = bb::mul(bb::sub(bb::double_(f0), f2), inv3), b = bb::mul(bb::sub(bb::double_(f2), f0), inv3);
            const uint32_t ab = bb::add(a, b);
            p.ext_fold[r][4 * q] = a;
            p.ext_fold[r][4 * q + 1] = b;
            p.ext_fold[r][4 * q + 2] = bb::sub(f1, ab);
            p.ext_fold[r][4 * q + 3] = bb::sub(f3, ab);
        }
    for (int r = 0; r < 13; ++r) p.int_fold[r] = (uint64_t)(r < 12 ? p.int_rc[r + 1] : p.ext_rc[4][0]) * bb::R_MOD_P;
}

// x^7 with every product lazy (bb::mul_lazy: (a b + m p) >> 32 < a b / 2^32 + p, valid while a b < 2^64 - 2^32 p = 2.418 p^2)
// and ONE conditional subtraction, on x^4. Ranges, x in [0, 1.032 p) — what reduce_wide_loose and the canonical additions
// of the round constants deliver (exact bounds: tools/poseidon2_bounds.py):
//   x2 = x*x   < 1.499 p      x3 = x2*x  < 1.725 p      x4 = x2*x2 < 2.053 p (< 2^32 = 2.133 p), after the subtraction < 1.053 p
//   x3*x4 < 1.82 p^2: the last product is valid; lazy it is < 1.851 p, reduced it is canonical.
// 14 instructions (sbox7_lazy) / 16 (sbox7) instead of 16 / 18 with two fully reduced squarings. The input may be as
// large as 1.088 p before x4 leaves 32 bits.
PW_HD uint32_t sbox7(uint32_t x) {
    const uint32_t x2 = bb::mul_lazy(x, x);
    const uint32_t x3 = bb::mul_lazy(x2, x);
    const uint32_t x4 = bb::reduce_2p(bb::mul_lazy(x2, x2));
    return bb::mul(x3, x4);
}

// x^7 left in [0, 1.851 p), for consumers that only multiply-accumulate it (external_layer_fold)
PW_HD uint32_t sbox7_lazy(uint32_t x) {
    const uint32_t x2 = bb::mul_lazy(x, x);
    const uint32_t x3 = bb::mul_lazy(x2, x);
    const uint32_t x4 = bb::reduce_2p(bb::mul_lazy(x2, x2));
    return bb::mul_lazy(x3, x4);
}

// External linear layer: M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on each block of four words, then every word
// gets the sum of its column over the four blocks added — followed by the addition of the next round's constants.
// All in 64-bit accumulators (one instruction per 32+64-bit or 64+64-bit addition, per small-constant multiply-add),
// with M4's shared partial sums: t01 = x0 + x1, t23 = x2 + x3, t = t01 + t23, ta = t + x1, tb = t + x3,
// y0 = ta + t01, y1 = ta + 2 x2, y2 = tb + t23, y3 = tb + 2 x0 — 11 instructions per block, 13 with the constants
// (Params::ext_fold) — then the column sums (12) and one reduction per output: 124 + 16 reductions where separate
// multiply-add chains took 140 and modular additions 72 * 3 + 16 * 3.
// Inputs may be lazy S-box outputs in [0, 1.86 p): an output is < 5 * (7 * 1.86 p + 4 p) = 85 p, inside reduce_wide's 128 p.
// LOOSE: the outputs only feed an S-box or the partial rounds, so the last conditional subtraction is skipped ([0, 1.032 p)).
template <bool FOLD, bool LOOSE>
PW_HD void external_layer_fold(uint32_t* s, const uint64_t* fold) {
    uint64_t y[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t x0 = s[4 * b], x1 = s[4 * b + 1], x2 = s[4 * b + 2], x3 = s[4 * b + 3];
        const uint64_t t01 = bb::wide_add(FOLD ? bb::wide_fma_uniform(fold[4 * b], x0, 1) : bb::wide_mul(x0, 1), x1);
        const uint64_t t23 = bb::wide_add(FOLD ? bb::wide_fma_uniform(fold[4 * b + 1], x2, 1) : bb::wide_mul(x2, 1), x3);
        const uint64_t t = t01 + t23;
        const uint64_t ta = bb::wide_add(t, x1), tb = bb::wide_add(t, x3);
        y[4 * b] = ta + t01;
        y[4 * b + 1] = bb::wide_fma(ta, x2, 2) + (FOLD ? fold[4 * b + 2] : 0ull);
        y[4 * b + 2] = tb + t23;
        y[4 * b + 3] = bb::wide_fma(tb, x0, 2) + (FOLD ? fold[4 * b + 3] : 0ull);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t col = (y[i] + y[4 + i]) + (y[8 + i] + y[12 + i]);
#pragma unroll
        for (int b = 0; b < 4; ++b) s[4 * b + i] = LOOSE ? bb::reduce_wide_loose(y[4 * b + i] + col) : bb::reduce_wide(y[4 * b + i] + col);
    }
}

// s_i <- sum + mu_i * s_i with mu = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27, -2^-8, -1/16, -2^-27]
// (diag[i] = mu_i in Montgomery form). Each output is ONE Montgomery reduction of sum * R + mu_i * s_i: the raw
// product sum * (R mod p) is shared and every element adds its own product to it with a single multiply-add.
// Between partial rounds only s_0 has to be small (it enters the S-box), so the other fifteen words stay LAZY: their
// reduction drops the conditional subtraction (3 instructions per word instead of 5, s_1 included: mu_1 = 1 is a
// product like the others), and the 16-term sum is reduced with reduce_wide_loose (4 instructions instead of 5).
// Ranges (exact: tools/poseidon2_bounds.py): sum < 1.032 p, sum * R < 0.138 p^2; a word below B p gives
// (0.138 + B) p^2 / 2^32 + p, whose fixed point is B = 2.0037 (< 2^32 / p = 2.133; the product stays below the
// reduction's 2.418 p^2); the 16-term sum is below 1.86 p + 15 * 2.0037 p < 32 p, inside reduce_wide_loose's 128 p.
// s_0 arrives as a lazy S-box output (< 1.86 p) and leaves canonical, with the constant it meets next already added
// (`next_c` = that constant times R mod p, < 0.134 p^2: the product stays below 2.13 p^2).
PW_HD void internal_layer(uint32_t* s, const uint32_t* diag, uint64_t next_c) {
    uint64_t wide = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) wide = bb::wide_add(wide, s[i]);
    const uint32_t sum = bb::reduce_wide_loose(wide);
    const uint64_t sum_r = (uint64_t)sum * bb::R_MOD_P;
    s[0] = bb::monty_reduce(bb::wide_mad_uniform(sum_r + next_c, s[0], diag[0]));
#pragma unroll
    for (int i = 1; i < 16; ++i) s[i] = bb::monty_reduce_lazy(bb::wide_mad_uniform(sum_r, s[i], diag[i]));
}

// The round loops are deliberately NOT unrolled: one full round + one partial round is
// ~1 K instructions (8 KB) and stays resident in the instruction cache shared by a CU pair;
// the fully unrolled permutation (~50 KB of code) would thrash it. Round constants are
// indexed by the (wave-uniform) round counter and arrive through scalar loads.
// Round constants of the external rounds are added by the linear layer that PRECEDES the round (external_layer_fold),
// except for round 4, which follows an internal layer.
// SPONGE: the permutation of an absorbing sponge whose output is not read — the rate words are overwritten and the
// capacity words only enter the next permutation's first linear layer, so the last layer leaves them in [0, 1.032 p).
template <bool SPONGE = false>
PW_HD void permute(uint32_t* s, const Params& P) {
    external_layer_fold<true, true>(s, P.ext_fold[0]);
PW_P2_ROUND_LOOP
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sbox7_lazy(s[i]);
        external_layer_fold<true, true>(s, P.ext_fold[r + 1]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = sbox7_lazy(s[i]);
    external_layer_fold<false, true>(s, nullptr);  // [0, 1.032 p): what the partial rounds take
    s[0] = bb::add_loose(s[0], P.int_rc[0]);
PW_P2_PARTIAL_LOOP
    for (int r = 0; r < 13; ++r) {
        s[0] = sbox7_lazy(s[0]);
        internal_layer(s, P.diag, P.int_fold[r]);
    }
    // the partial rounds leave s_0 canonical (constant of the next round included) and the others in [0, 2.004 p)
#pragma unroll
    for (int i = 1; i < 16; ++i) s[i] = bb::add(bb::reduce_2p(s[i]), P.ext_rc[4][i]);
PW_P2_ROUND_LOOP
    for (int r = 4; r < 7; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sbox7_lazy(s[i]);
        external_layer_fold<true, true>(s, P.ext_fold[r + 1]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = sbox7_lazy(s[i]);
    external_layer_fold<false, SPONGE>(s, nullptr);
}

}  // namespace p2
