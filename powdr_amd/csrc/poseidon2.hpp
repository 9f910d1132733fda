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
// accumulators (v_mad_i64_i32: one instruction per term or per small-constant multiply-add) with one 2-instruction
// Montgomery reduction per output (per-stage scale factors ride in the constant tables, see Params). Per permutation 3 697 VALU instructions (PMC; the first version of round 1 took 7 089); every
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
    // Derived tables, in the form the rounds consume them (centred representatives; 64-bit ones enter 64-bit accumulators
    // from a scalar register pair). SCALES: a register holds lambda * (Montgomery form of its value) for a per-stage field
    // constant lambda. x -> x^7 by Montgomery products maps lambda to lambda^7 (lambda = 1, the Montgomery form itself, is the
    // fixed point the classic implementation sits on); a linear layer keeps lambda if its constants are scaled by it; reducing
    // a layer's 64-bit outputs with a MONTGOMERY reduction (2 instructions) instead of a Barrett one (3) divides lambda by R.
    // Every layer but the last uses the Montgomery reduction; the tables below carry the scales, the partial rounds — which
    // multiply every word by a constant anyway — move from the scale the first half ends in to the one from which the second
    // half ends at lambda = 1 (7th roots exist: gcd(7, p - 1) = 1), and the last layer reduces Barrett-style at lambda = 1.
    //
    // ext_rc[r] folded into the external linear layer that precedes round r (external_layer). The layer adds to every
    // output the sum of its column over the four blocks, so block q, column i must carry f[q][i] = c[q][i] - (sum_q' c[q'][i]) / 5
    // before the column sums. The M4 network shares its partial sums, so a block takes its constants as two seeds
    // (a into x0 + x1, b into x2 + x3: outputs get 2a + b, a + b, a + 2b, a + b) and two corrections:
    // ext_fold[r][4q + {0,1,2,3}] = {a, b, f[q][1] - a - b, f[q][3] - a - b} with a = (2 f0 - f2) / 3, b = (2 f2 - f0) / 3,
    // all times the scale of the layer's inputs
    int64_t ext_fold[8][16];
    int32_t entry_c;         // int_rc[0] at the scale the first half ends in: added to s_0 before the first partial round
    // partial rounds (internal_layer): s_0 leaves its S-box at scale lambda_in^7, the other words are at lambda_in, all leave
    // at lambda_out.
    // [0]: the first partial round (first-half scale -> second-half scale), [1]: the other twelve (lambda_in = lambda_out) — two
    // sets, so that the twelve rounds of the loop keep theirs in scalar registers
    int32_t part_kappa[2];     // lambda_in / lambda_in^7: brings s_0 to the others' scale inside the 16-term sum
    int32_t part_rho[2];       // (lambda_out / lambda_in) R^2: the reduced sum re-enters the accumulators' domain
    int32_t part_diag[2][16];  // mu_i R lambda_out / lambda_in (i >= 1), mu_0 R lambda_out / lambda_in^7
    // the constant s_0 meets next, as a raw product c * (R mod p) that joins s_0's multiply-add of partial round r:
    // int_rc[r + 1] for r < 12, ext_rc[4][0] for the last one, at scale lambda_out
    int64_t int_fold[13];
    // ext_rc[4][i] * (R mod p), i >= 1, at the second-half scale: joins the multiply-adds of the LAST partial round
    int64_t exit_fold[16];
};

// The placeholder round constants (Montgomery form): a documented splitmix64 stream. The real table of the reference's
// permutation is EXTERNAL to its checkout (p3-baby-bear 0.5.2, /root/reference/number/Cargo.toml:16-19) and is installed at
// run time with pw_set_poseidon2_constants (include/powdr_prover.h).
inline void default_round_constants(Params& p) {
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
}

// Everything else from ext_rc / int_rc (Montgomery words, already in `p`): the internal diagonal and the derived tables.
// Nothing here depends on the VALUES of the round constants beyond their being field elements: they enter as centred
// representatives (|c| <= p / 2), which is what tools/poseidon2_bounds.py assumes for every additive constant.
inline void derive_params(Params& p) {
    auto m = [](uint32_t c) { return bb::to_monty(c); };
    auto inv2k = [&](int k) { return bb::inv(bb::pow_u32(m(2), (uint32_t)k)); };
    const uint32_t d[16] = {bb::neg(m(2)), m(1), m(2), inv2k(1), m(3), m(4), bb::neg(inv2k(1)), bb::neg(m(3)),
                            bb::neg(m(4)), inv2k(8), inv2k(2), inv2k(3), inv2k(27), bb::neg(inv2k(8)),
                            bb::neg(inv2k(4)), bb::neg(inv2k(27))};
    for (int i = 0; i < 16; ++i) p.diag[i] = d[i];
    // ---- derived tables. Scales are handled as plain residues mod p (64-bit arithmetic): a register at scale lam holds
    // lam * w for the Montgomery word w of its value.
    const uint64_t Pm = bb::P;
    auto mulm = [&](uint64_t x, uint64_t y) { return x * y % Pm; };
    auto powm = [&](uint64_t b, uint64_t e) { uint64_t r = 1; b %= Pm; while (e) { if (e & 1) r = r * b % Pm; b = b * b % Pm; e >>= 1; } return r; };
    auto invm = [&](uint64_t x) { return powm(x, Pm - 2); };
    uint64_t d7 = 0;  // 7^-1 mod (p - 1)
    for (uint64_t k = 0; k < 7; ++k) if ((k * (Pm - 1) + 1) % 7 == 0) d7 = (k * (Pm - 1) + 1) / 7;
    const uint64_t Rm = bb::R_MOD_P, Rinv = invm(Rm);
    // a[r] = scale at the input of the S-box of external round r
    uint64_t a[8];
    a[0] = Rinv;                                         // permutation input at scale 1, layer 0 divides by R
    for (int r = 1; r < 4; ++r) a[r] = mulm(powm(a[r - 1], 7), Rinv);
    const uint64_t lam_first = mulm(powm(a[3], 7), Rinv);  // after the layer that follows round 3: entering the partial rounds
    a[7] = 1;                                             // a[7]^7 = 1: the last layer keeps the scale and must leave 1
    for (int r = 6; r >= 4; --r) a[r] = powm(mulm(Rm, a[r + 1]), d7);  // a[r+1] = a[r]^7 / R
    const uint64_t lam_second = a[4];                     // leaving the partial rounds
    auto layer_in_scale = [&](int r) -> uint64_t { return r == 0 ? 1 : powm(a[r - 1], 7); };  // inputs of the layer preceding round r

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
    for (int r = 0; r < 8; ++r) {
        if (r == 4) { for (int i = 0; i < 16; ++i) p.ext_fold[r][i] = 0; continue; }  // round 4's constants ride on the last partial round
        const uint64_t lam = layer_in_scale(r);
        for (int q = 0; q < 4; ++q) {
            const uint32_t f0 = f[r][4 * q], f1 = f[r][4 * q + 1], f2 = f[r][4 * q + 2], f3 = f[r][4 * q + 3];
            const uint32_t a_ = bb::mul(bb::sub(bb::double_(f0), f2), inv3), b_ = bb::mul(bb::sub(bb::double_(f2), f0), inv3);
            const uint32_t ab = bb::add(a_, b_);
            p.ext_fold[r][4 * q] = bb::centred((uint32_t)mulm(a_, lam));
            p.ext_fold[r][4 * q + 1] = bb::centred((uint32_t)mulm(b_, lam));
            p.ext_fold[r][4 * q + 2] = bb::centred((uint32_t)mulm(bb::sub(f1, ab), lam));
            p.ext_fold[r][4 * q + 3] = bb::centred((uint32_t)mulm(bb::sub(f3, ab), lam));
        }
    }
    p.entry_c = bb::centred((uint32_t)mulm(p.int_rc[0], lam_first));
    for (int k = 0; k < 2; ++k) {
        const uint64_t lin = k == 0 ? lam_first : lam_second, lout = lam_second, l0 = powm(lin, 7);
        const uint64_t ratio = mulm(lout, invm(lin)), ratio0 = mulm(lout, invm(l0));
        p.part_kappa[k] = bb::centred((uint32_t)mulm(lin, invm(l0)));
        p.part_rho[k] = bb::centred((uint32_t)mulm(ratio, mulm(Rm, Rm)));
        for (int i = 0; i < 16; ++i) p.part_diag[k][i] = bb::centred((uint32_t)mulm(p.diag[i], i ? ratio : ratio0));
    }
    for (int r = 0; r < 13; ++r) {
        const uint32_t c = r < 12 ? p.int_rc[r + 1] : p.ext_rc[4][0];
        p.int_fold[r] = (int64_t)bb::centred((uint32_t)mulm(c, lam_second)) * (int64_t)bb::R_MOD_P;
    }
    for (int i = 0; i < 16; ++i) p.exit_fold[i] = i ? (int64_t)bb::centred((uint32_t)mulm(p.ext_rc[4][i], lam_second)) * (int64_t)bb::R_MOD_P : 0;
}

// host-side generation with the placeholder constants
inline void generate_params(Params& p) {
    default_round_constants(p);
    derive_params(p);
}

// x^7 on a signed representative, |x| < 1.09 p (the rounds deliver |x| < 1.024 p): x2 = x x, x3 = x2 x, x4 = x2 x2, x7 = x3 x4, each a signed Montgomery
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
// (Params::ext_fold) — then the column sums (12 + 16) and one reduction per output: a signed MONTGOMERY reduction (2
// instructions; it divides the scale by R, see Params) in every layer but the last: 112 in all. Inputs |x| < 1.024 p: an output
// is below 5 * (7 * 1.024 + 1.5) p = 43.3 p in magnitude and its reduction below 43 p / 2^32 + p / 2: S-box inputs.
// BIAS (the last layer of the permutation, no constants, scale 1): the seeds are 4 p, which puts 40 p or 60 p on every
// output — more than the sum can be negative — and the reduction is the Barrett-style sreduce_wide_loose (3 instructions,
// keeps the scale): results in [0, 1.021 p), one conditional subtraction makes them canonical.
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
        for (int b = 0; b < 4; ++b) s[4 * b + i] = BIAS ? bb::sreduce_wide_loose(y[4 * b + i] + col) : bb::smont(y[4 * b + i] + col);
    }
}

// s_i <- sum + mu_i * s_i with mu = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27, -2^-8, -1/16, -2^-27].
// Each output is ONE signed Montgomery reduction of sum * rho + m_i * s_i: the raw product sum * rho is shared and every
// element adds its own product to it with a single multiply-add — 3 instructions per word, s_1 (mu = 1) included. The
// constants (Params::part_*) carry mu_i in Montgomery form AND the scales: s_0 arrives as an S-box output (scale lambda^7),
// enters the 16-term sum through kappa = lambda / lambda^7, the sum is Montgomery-reduced (2 instructions) and re-enters
// through rho; s_0 leaves with the constant it meets next already added (`next_c`), an S-box input again.
// Ranges (tools/poseidon2_bounds.py): |kappa|, |rho|, |m_i| <= p / 2; |sum| < 0.74 p, |sum * rho| < 0.37 p^2; a word below
// B p in magnitude gives (0.37 + B / 2) p^2 / 2^32 + p / 2, at most 0.91 p; every product stays below 0.9 p^2 (the
// reduction takes 1.209 p^2).
// LAST: the last partial round also adds the constants of the external round that follows (`exit_c`, Params::exit_fold).
template <bool LAST>
PW_HD void internal_layer(int32_t* s, const int32_t* m, int32_t kappa, int32_t rho, int64_t next_c, const int64_t* exit_c) {
    int64_t wide = bb::smul_uniform(s[0], kappa);
#pragma unroll
    for (int i = 1; i < 16; ++i) wide = bb::swide_add(wide, s[i]);
    const int32_t sum = bb::smont(wide);
    const int64_t sum_r = bb::smul_uniform(sum, rho);
    s[0] = bb::smont(bb::swide_mad_uniform(sum_r + next_c, s[0], m[0]));
#pragma unroll
    for (int i = 1; i < 16; ++i) s[i] = bb::smont(bb::swide_mad_uniform(LAST ? sum_r + exit_c[i] : sum_r, s[i], m[i]));
}

// The round loops are deliberately NOT unrolled: one full round + one partial round is
// ~1 K instructions (8 KB) and stays resident in the instruction cache shared by a CU pair;
// the fully unrolled permutation (~50 KB of code) would thrash it. Round constants are
// indexed by the (wave-uniform) round counter and arrive through scalar loads.
// Round constants of the external rounds are added by the linear layer that PRECEDES the round (external_layer) or, for
// round 4, by the last partial round.
// Input: any representatives with -0.02 p < x < 1.021 p as int32 (canonical words qualify).
// SPONGE = false: canonical output. SPONGE = true: the permutation of an absorbing sponge whose output is not read — the
// rate words are overwritten and the capacity words only enter the next permutation — leaves [0, 1.021 p).
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
    s[0] += P.entry_c;  // int_rc[0] at the current scale; |s_0| < 0.51 p + p / 2: an S-box input
    s[0] = sbox7(s[0]);
    internal_layer<false>(s, P.part_diag[0], P.part_kappa[0], P.part_rho[0], P.int_fold[0], nullptr);
PW_P2_PARTIAL_LOOP
    for (int r = 1; r < 12; ++r) {
        s[0] = sbox7(s[0]);
        internal_layer<false>(s, P.part_diag[1], P.part_kappa[1], P.part_rho[1], P.int_fold[r], nullptr);
    }
    s[0] = sbox7(s[0]);
    internal_layer<true>(s, P.part_diag[1], P.part_kappa[1], P.part_rho[1], P.int_fold[12], P.exit_fold);
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
