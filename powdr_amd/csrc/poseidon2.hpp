// Poseidon2 over BabyBear, width 16, x^7, 8 external + 13 internal rounds, on
// Montgomery words. Used by the Merkle kernels (device) and by the Fiat-Shamir
// transcript (host). Shape as restated in oracle/stark_oracle.cpp (Plonky3's
// BabyBear instance shape; the round-constant table is EXTERNAL to the reference
// checkout and therefore generated here by the documented splitmix64 stream —
// one swappable table, see DESIGN.md "pw-stark v0").
//
// Cost per permutation: S-boxes 8*16*4 + 13*4 = 564 Montgomery products, the
// internal diagonal adds ~8 products per internal round; the linear layers are
// additions only (M4 = circ-like [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] needs no
// multiplier). The MDS layer is 16x16 with entries in {1,2,3,4,6}: it is not a dense
// contraction worth an MFMA ("MFMA for the MDS").
#pragma once
#include "babybear.hpp"

namespace p2 {

struct Params {
    uint32_t ext_rc[8][16];
    uint32_t int_rc[13];
    uint32_t diag[16];
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
}

PW_HD uint32_t sbox7(uint32_t x) {
    uint32_t x2 = bb::sqr(x);
    uint32_t x3 = bb::mul_lazy(x2, x);  // in [0, 2p): only ever the lazy operand of the last product
    uint32_t x4 = bb::sqr(x2);
    return bb::mul(x3, x4);
}

// [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] * (a,b,c,d)
PW_HD void m4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    // 11 modular additions
    uint32_t ab = bb::add(a, b), cd = bb::add(c, d);
    uint32_t t = bb::add(ab, cd);
    uint32_t tb = bb::add(t, b);                            // a + 2b + c + d
    uint32_t td = bb::add(t, d);                            // a + b + c + 2d
    uint32_t o3 = bb::add(td, bb::double_(a));              // 3a + b + c + 2d
    uint32_t o1 = bb::add(tb, bb::double_(c));              // a + 2b + 3c + d
    uint32_t o0 = bb::add(tb, ab);                          // 2a + 3b + c + d
    uint32_t o2 = bb::add(td, cd);                          // a + b + 2c + 3d
    a = o0; b = o1; c = o2; d = o3;
}

PW_HD void external_layer(uint32_t* s) {
#pragma unroll
    for (int b = 0; b < 4; ++b) m4(s[4 * b], s[4 * b + 1], s[4 * b + 2], s[4 * b + 3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t col = bb::add(bb::add(s[i], s[4 + i]), bb::add(s[8 + i], s[12 + i]));
#pragma unroll
        for (int b = 0; b < 4; ++b) s[4 * b + i] = bb::add(s[4 * b + i], col);
    }
}

// s_i <- sum + mu_i * s_i with mu = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27, -2^-8, -1/16, -2^-27]
PW_HD void internal_layer(uint32_t* s, const uint32_t* diag) {
    uint32_t sum = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) sum = bb::add(sum, s[i]);
    uint32_t d2, h;
    s[0] = bb::sub(sum, bb::double_(s[0]));
    s[1] = bb::add(sum, s[1]);
    s[2] = bb::add(sum, bb::double_(s[2]));
    h = bb::halve(s[3]);
    s[3] = bb::add(sum, h);
    d2 = bb::double_(s[4]);
    s[4] = bb::add(sum, bb::add(d2, s[4]));
    s[5] = bb::add(sum, bb::double_(bb::double_(s[5])));
    h = bb::halve(s[6]);
    s[6] = bb::sub(sum, h);
    d2 = bb::double_(s[7]);
    s[7] = bb::sub(sum, bb::add(d2, s[7]));
    s[8] = bb::sub(sum, bb::double_(bb::double_(s[8])));
#pragma unroll
    for (int i = 9; i < 16; ++i) s[i] = bb::add(sum, bb::mul(diag[i], s[i]));
}

// The round loops are deliberately NOT unrolled: one full round + one partial round is
// ~1 K instructions (8 KB) and stays resident in the instruction cache shared by a CU pair;
// the fully unrolled permutation (~50 KB of code) would thrash it. Round constants are
// indexed by the (wave-uniform) round counter and arrive through scalar loads.
PW_HD void permute(uint32_t* s, const Params& P) {
    external_layer(s);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sbox7(bb::add(s[i], P.ext_rc[r][i]));
        external_layer(s);
    }
#pragma unroll 1
    for (int r = 0; r < 13; ++r) {
        s[0] = sbox7(bb::add(s[0], P.int_rc[r]));
        internal_layer(s, P.diag);
    }
#pragma unroll 1
    for (int r = 4; r < 8; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sbox7(bb::add(s[i], P.ext_rc[r][i]));
        external_layer(s);
    }
}

}  // namespace p2
