// Poseidon2 over BabyBear, width 16, x^7, 8 external + 13 internal rounds, on
// Montgomery words. Used by the Merkle kernels (device) and by the Fiat-Shamir
// transcript (host). Shape as restated in oracle/stark_oracle.cpp (Plonky3's
// BabyBear instance shape; the round-constant table is EXTERNAL to the reference
// checkout and therefore generated here by the documented splitmix64 stream —
// one swappable table, see DESIGN.md "pw-stark v0").
//
// The state is held as SIGNED representatives (babybear.hpp, "signed representatives"): a signed Montgomery reduction
// needs no conditional subtraction, so x^7 is 4 products x 3 instructions = 12 (the unsigned form needs 14 to 18), the
// fifteen passive words of a partial round cost 3 instructions each, and the linear layers are sums in signed 64-bit
// accumulators (v_mad_i64_i32: one instruction per term or per small-constant multiply-add) with one 3-instruction
// reduction per output. Per permutation about 3 750 VALU instructions (the first version of round 1 took 7 089); every
// range the code relies on is computed exactly by tools/poseidon2_bounds.py and exercised by field_selftest.hpp.
// The MDS layer is not a dense contraction worth an MFMA (measured: DESIGN.md 3.4).
#pragma once
#include "babybear.hpp"

// unroll factors of the round loops (tools/microbench_hash.hip builds variants; 1 = rolled measured fastest, see permute)
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
    // the definition: canonical Montgomery words
    uint32_t ext_rc[8][16];
    uint32_t int_rc[13];
    uint32_t diag[16];
    // derived tables, in the form the rounds consume them (centred representatives; 64-bit ones enter 64-bit accumulators
    // from a scalar register pair):
    // ext_rc[r] folded into the external linear layer that precedes round r (external_layer). The layer adds to every
    // output the sum of its column over the four blocks, so block q, column i must carry f[q][i] = c[q][i] - (sum_q' c[q'][i]) / 5
    // before the column sums. The M4 network shares its partial sums, so a block takes its constants as two seeds
    // (a into x0 + x1, b into x2 + x3: outputs get 2a + b, a + b, a + 2b, a + b) and two corrections:
    // ext_fold[r][4q + {0,1,2,3}] = {a, b, f[q][1] - a - b, f[q][3] - a - b} with a = (2 f0 - f2) / 3, b = (2 f2 - f0) / 3
    int64_t ext_fold[8][16];
    // the constant s_0 meets next, as a raw product c * (R mod p) that joins s_0's multiply-add of partial round r
    // (internal_layer): int_rc[r + 1] for r < 12, ext_rc[4][0] for the last one
    int64_t int_fold[13];
    // ext_rc[4][i] * (R mod p), i >= 1: joins the multiply-adds of the LAST partial round
    int64_t exit_fold[16];
    int32_t sdiag[16];  // centred diag
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
    const uint32_t inv3 = bb::inv(m(3));
    uint32_t f[8][16];
    for (int r = 0; r < 8; ++r)
        for (int i = 0; i < 4; ++i) {
            uint32_t col = 0;
            for (int q = 0; q < 4; ++q) col = bb::add(col, p.ext_rc[r][4 * q + i]);
            const uint32_t t = bb::mul(col, inv5);
            for (int q = 0; q < 4; ++q) f[r][4 * q + i] = bb::sub(p.ext_rc[r][4 * q + i], t);
        }
    for (int r = 0; r < 8; ++r)
        for (int q = 0; q < 4; ++q) {
            const uint32_t f0 = f[r][4 * q], f1 = f[r][4 * q + 1], f2 = f[r][4 * q + 2], f3 = f[r][4 * q + 3];
            const uint32_t a = bb::mul(bb::sub(bb::double_(f0), f2), inv3), b = bb::mul(bb::sub(bb::double_(f2), f0), inv3);
            const uint32_t ab = bb::add(a, b);
            p.ext_fold[r][4 * q] = bb::centred(a);
            p.ext_fold[r][4 * q + 1] = bb::centred(b);
            p.ext_fold[r][4 * q + 2] = bb::centred(bb::sub(f1, ab));
            p.ext_fold[r][4 * q + 3] = bb::centred(bb::sub(f3, ab));
        }
    for (int r = 0; r < 13; ++r) p.int_fold[r] = (int64_t)bb::centred(r < 12 ? p.int_rc[r + 1] : p.ext_rc[4][0]) * (int64_t)bb::R_MOD_P;
    for (int i = 0; i < 16; ++i) {
        p.exit_fold[i] = i ? (int64_t)bb::centred(p.ext_rc[4][i]) * (int64_t)bb::R_MOD_P : 0;
        p.sdiag[i] = bb::centred(p.diag[i]);
    }
}

// x^7 on a signed representative, |x| < 1.09 p (the rounds deliver |x| < 1.011 p): x2 = x x, x3 = x2 x, x4 = x2 x2, x7 = x3 x4, each a signed Montgomery
// product (3 instructions) whose result is below p in magnitude — no conditional subtraction anywhere; |x^7| < 0.93 p.
PW_HD int32_t sbox7(int32_t x) {
    const int32_t x2 = bb::smont(bb::smul(x, x));
    const int32_t x3 = bb::smont(bb::smul(x2, x));
    const int32_t x4 = bb::smont(bb::smul(x2, x2));
    return bb::smont(bb::smul(x3, x4));
}

// External linear layer: M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on each block of four words, then every word
// gets the sum of its column over the four blocks added — followed by the addition of the next round's constants.
// All in signed 64-bit accumulators (one instruction per 32+64-bit or 64+64-bit addition, per small-constant
// multiply-add), with M4's shared partial sums: t01 = x0 + x1, t23 = x2 + x3, t = t01 + t23, ta = t + x1, tb = t + x3,
// y0 = ta + t01, y1 = ta + 2 x2, y2 = tb + t23, y3 = tb + 2 x0 — 11 instructions per block, 13 with the constants
// (Params::ext_fold) — then the column sums (12 + 16) and one 3-instruction reduction per output: 128 in all.
// Inputs |x| < 1.011 p: an output is below 5 * (7 * 1.011 + 1.5) p = 43 p in magnitude, inside sreduce_wide_loose's 64 p; the
// results, in (-0.011 p, 1.011 p), are S-box inputs.
// BIAS (the last layer of the permutation, no constants): the seeds are 4 p, which puts 40 p or 60 p on every output —
// more than the 35 * 0.93 p the sum can be negative — so the results are in [0, 1.011 p) and one conditional subtraction
// makes them canonical.
template <bool FOLD, bool BIAS>
PW_HD void external_layer(int32_t* s, const int64_t* fold) {
    int64_t y[16];
    constexpr int64_t kBias = 4 * (int64_t)bb::P;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int32_t x0 = s[4 * b], x1 = s[4 * b + 1], x2 = s[4 * b + 2], x3 = s[4 * b + 3];
        const int64_t t01 = bb::swide_add(FOLD ? bb::swide_fma_uniform(fold[4 * b], x0, 1) : BIAS ? bb::swide_fma_uniform(kBias, x0, 1) : bb::swide_mul(x0, 1), x1);
        const int64_t t23 = bb::swide_add(FOLD ? bb::swide_fma_uniform(fold[4 * b + 1], x2, 1) : BIAS ? bb::swide_fma_uniform(kBias, x2, 1) : bb::swide_mul(x2, 1), x3);
        const int64_t t = t01 + t23;
        const int64_t ta = bb::swide_add(t, x1), tb = bb::swide_add(t, x3);
        y[4 * b] = ta + t01;
        y[4 * b + 1] = bb::swide_fma(ta, x2, 2) + (FOLD ? fold[4 * b + 2] : (int64_t)0);
        y[4 * b + 2] = tb + t23;
        y[4 * b + 3] = bb::swide_fma(tb, x0, 2) + (FOLD ? fold[4 * b + 3] : (int64_t)0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t col = (y[i] + y[4 + i]) + (y[8 + i] + y[12 + i]);
#pragma unroll
        for (int b = 0; b < 4; ++b) s[4 * b + i] = bb::sreduce_wide_loose(y[4 * b + i] + col);
    }
}

// s_i <- sum + mu_i * s_i with mu = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27, -2^-8, -1/16, -2^-27]
// (sdiag[i] = the centred Montgomery word of mu_i). Each output is ONE signed Montgomery reduction of
// sum * R + mu_i * s_i: the raw product sum * (R mod p) is shared and every element adds its own product to it with a
// single multiply-add — 3 instructions per word, s_1 (mu = 1) included. s_0 arrives as an S-box output and leaves with the
// constant it meets next already added (`next_c` = that constant times R mod p), an S-box input again.
// Ranges (tools/poseidon2_bounds.py): |sum| < 1.004 p, |sum * R| < 0.134 p^2; with |mu_i| < p / 2 a word below B p in
// magnitude gives (0.134 + B / 2) p^2 / 2^32 + p / 2: 1.008 p on entry, 0.80 p after one round, fixed point 0.735 p;
// s_0 leaves below 0.82 p; every product stays below 0.67 p^2 (the reduction takes 1.209 p^2).
// LAST: the last partial round also adds the constants of the external round that follows (`exit_c`, Params::exit_fold).
template <bool LAST>
PW_HD void internal_layer(int32_t* s, const int32_t* sdiag, int64_t next_c, const int64_t* exit_c) {
    int64_t wide = bb::swide_mul(s[0], 1);
#pragma unroll
    for (int i = 1; i < 16; ++i) wide = bb::swide_add(wide, s[i]);
    const int32_t sum = bb::sreduce_wide_loose(wide);
    const int64_t sum_r = bb::smul_uniform(sum, (int32_t)bb::R_MOD_P);
    s[0] = bb::smont(bb::swide_mad_uniform(sum_r + next_c, s[0], sdiag[0]));
#pragma unroll
    for (int i = 1; i < 16; ++i) s[i] = bb::smont(bb::swide_mad_uniform(LAST ? sum_r + exit_c[i] : sum_r, s[i], sdiag[i]));
}

// The round loops are deliberately NOT unrolled: one full round + one partial round is
// ~1 K instructions (8 KB) and stays resident in the instruction cache shared by a CU pair;
// the fully unrolled permutation (~50 KB of code) would thrash it. Round constants are
// indexed by the (wave-uniform) round counter and arrive through scalar loads.
// Round constants of the external rounds are added by the linear layer that PRECEDES the round (external_layer) or, for
// round 4, by the last partial round.
// Input: any representatives with -0.011 p < x < 1.011 p as int32 (canonical words qualify).
// SPONGE = false: canonical output. SPONGE = true: the permutation of an absorbing sponge whose output is not read — the
// rate words are overwritten and the capacity words only enter the next permutation — leaves [0, 1.011 p).
template <bool SPONGE = false>
PW_HD void permute(uint32_t* words, const Params& P) {
    int32_t* s = reinterpret_cast<int32_t*>(words);
    external_layer<true, false>(s, P.ext_fold[0]);
PW_P2_ROUND_LOOP
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sbox7(s[i]);
        external_layer<true, false>(s, P.ext_fold[r + 1]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = sbox7(s[i]);
    external_layer<false, false>(s, nullptr);
    // s_0 + int_rc[0] as an S-box input: unsigned, the sum is below 2.011 p < 2^32 or just below zero; min(x, x - p)
    // maps [0, 2.011 p) to [0, 1.011 p) and the small negatives to (-1.011 p, -p)
    {
        const uint32_t x = (uint32_t)s[0] + P.int_rc[0];
        s[0] = (int32_t)bb::umin(x, x - bb::P);
    }
PW_P2_PARTIAL_LOOP
    for (int r = 0; r < 12; ++r) {
        s[0] = sbox7(s[0]);
        internal_layer<false>(s, P.sdiag, P.int_fold[r], nullptr);
    }
    s[0] = sbox7(s[0]);
    internal_layer<true>(s, P.sdiag, P.int_fold[12], P.exit_fold);
PW_P2_ROUND_LOOP
    for (int r = 4; r < 7; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sbox7(s[i]);
        external_layer<true, false>(s, P.ext_fold[r + 1]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = sbox7(s[i]);
    external_layer<false, true>(s, nullptr);
    if (!SPONGE) {
#pragma unroll
        for (int i = 0; i < 16; ++i) words[i] = bb::reduce_2p(words[i]);
    }
}

}  // namespace p2
