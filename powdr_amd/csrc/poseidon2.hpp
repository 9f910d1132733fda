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

namespace p2 {

struct Params {
    uint32_t ext_rc[8][16];
    uint32_t int_rc[13];
    uint32_t diag[16];
    // ext_rc[r] folded into the external linear layer that precedes round r (external_layer_fold): the layer adds
    // to every output the sum of its column over the four blocks, so the constant that enters block q, column i is
    // c[q][i] - (sum_q' c[q'][i]) / 5; 64-bit words because they seed 64-bit accumulators from a scalar register pair
    uint64_t ext_fold[8][16];
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
}

// x^7 with every product lazy (bb::mul_lazy: (a b + m p) >> 32 < a b / 2^32 + p, valid while a b < 2^64 - 2^32 p = 2.418 p^2)
// and ONE conditional subtraction, on x^4. Ranges, x in [0, 1.032 p) — what reduce_wide_loose and the canonical additions
// of the round constants deliver (exact bounds: tools/poseidon2_bounds.py):
//   x2 = x*x   < 1.499 p      x3 = x2*x  < 1.725 p      x4 = x2*x2 < 2.053 p (< 2^32 = 2.133 p), after the subtraction < 1.053 p
//   x3*x4 < 1.82 p^2: the last product is valid; lazy it is < 1.851 p, reduced it is canonical.
// 14 instructions (sbox7_lazy) / 16 (sbox7) instead of 16 / 18 with two fully reduced squarings.
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
// All in 64-bit accumulators: every output is
// 2x_i + 3x_{i+1} + x_{i+2} + x_{i+3} (four multiply-adds seeded with the folded constant) plus the column sum
// (64-bit adds), reduced once — 140 instructions instead of 72 modular additions + 16 constant additions. Inputs may
// be lazy S-box outputs in [0, 2p): an output is < 5 * (7 * 2p + p) = 75 p, inside reduce_wide's 128 p.
// LOOSE: the outputs only feed sbox7_lazy, so the last conditional subtraction is skipped ([0, 1.03 p)).
template <bool FOLD, bool LOOSE>
PW_HD void external_layer_fold(uint32_t* s, const uint64_t* fold) {
    uint64_t y[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x0 = s[4 * b + i], x1 = s[4 * b + ((i + 1) & 3)], x2 = s[4 * b + ((i + 2) & 3)], x3 = s[4 * b + ((i + 3) & 3)];
            uint64_t a = FOLD ? bb::wide_fma_uniform(fold[4 * b + i], x0, 2) : bb::wide_mul(x0, 2);
            a = bb::wide_fma(a, x1, 3);
            a = bb::wide_add(a, x2);
            y[4 * b + i] = bb::wide_add(a, x3);
        }
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
// product sum * (R mod p) is shared, every element adds its own product to it with a single multiply-add and pays
// one reduction — 5 instructions per element where a Montgomery product plus a modular addition took 8.
PW_HD void internal_layer(uint32_t* s, const uint32_t* diag) {
    uint64_t wide = s[0];  // 16 terms < p: one reduction at the end instead of 15
#pragma unroll
    for (int i = 1; i < 16; ++i) wide = bb::wide_add(wide, s[i]);
    const uint32_t sum = bb::reduce_sum(wide);
    const uint64_t sum_r = (uint64_t)sum * bb::R_MOD_P;  // < p^2; + mu_i * s_i < 2 p^2 stays reducible
    s[0] = bb::monty_reduce(bb::wide_mad_uniform(sum_r, s[0], diag[0]));
    s[1] = bb::add(sum, s[1]);
#pragma unroll
    for (int i = 2; i < 16; ++i) s[i] = bb::monty_reduce(bb::wide_mad_uniform(sum_r, s[i], diag[i]));
}

// The round loops are deliberately NOT unrolled: one full round + one partial round is
// ~1 K instructions (8 KB) and stays resident in the instruction cache shared by a CU pair;
// the fully unrolled permutation (~50 KB of code) would thrash it. Round constants are
// indexed by the (wave-uniform) round counter and arrive through scalar loads.
// Round constants of the external rounds are added by the linear layer that PRECEDES the round (external_layer_fold),
// except for round 4, which follows an internal layer.
PW_HD void permute(uint32_t* s, const Params& P) {
    external_layer_fold<true, true>(s, P.ext_fold[0]);
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sbox7_lazy(s[i]);
        external_layer_fold<true, true>(s, P.ext_fold[r + 1]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = sbox7_lazy(s[i]);
    external_layer_fold<false, false>(s, nullptr);
#pragma unroll 1
    for (int r = 0; r < 13; ++r) {
        s[0] = sbox7(bb::add(s[0], P.int_rc[r]));
        internal_layer(s, P.diag);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = bb::add(s[i], P.ext_rc[4][i]);
#pragma unroll 1
    for (int r = 4; r < 7; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sbox7_lazy(s[i]);
        external_layer_fold<true, true>(s, P.ext_fold[r + 1]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = sbox7_lazy(s[i]);
    external_layer_fold<false, false>(s, nullptr);
}

}  // namespace p2
