// Self-test of the range-sensitive field helpers (babybear.hpp, ext.hpp) against plain 64-bit modular arithmetic, on
// random values and on the edges of the ranges their comments promise. One function for host and device: on the
// device the inline multiply-add instructions are exercised, on the host their portable fall-backs.
#pragma once
#include "ext.hpp"
#include "poseidon2.hpp"

namespace pw {

// returns 0 or the number of the first failing check
PW_HD int field_selftest_checks(uint64_t seed, uint32_t iterations) {
    using namespace bb;
    uint64_t s = seed ? seed : 1;
    auto rnd = [&]() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    auto rp = [&]() { return (uint32_t)(rnd() % P); };
    const uint64_t edges64[] = {0, 1, P - 1, P, P + 1, 2ull * P - 1, 2ull * P, 16ull * P - 1, 75ull * P, 128ull * P - 1};
    for (uint64_t x : edges64) {
        if (x < 16ull * P && reduce_sum(x) != x % P) return 1;
        const uint32_t l = reduce_wide_loose(x);
        if (l % P != x % P || l >= (uint64_t)P + P / 32) return 2;
        if (reduce_wide(x) != x % P) return 3;
    }
    const uint32_t edges[] = {0, 1, P - 1, P, P + 1, 2 * P - 1};
    for (uint32_t a : edges)
        for (uint32_t b : edges) {
            const uint32_t r = add_2p(a, b), d = sub_2p(a, b);
            if (r >= 2 * P || r % P != (uint32_t)(((uint64_t)a + b) % P)) return 4;
            if (d >= 2 * P || d % P != (uint32_t)(((uint64_t)a + 2ull * P - b) % P)) return 5;
        }
    // signed representatives: reduction of wide sums, the S-box and the internal layer at the edges of their ranges
    auto smodp = [](int64_t v) { const int64_t r = v % (int64_t)P; return (uint32_t)(r < 0 ? r + P : r); };
    {
        const int64_t ys[] = {0, 1, -1, (int64_t)P, -(int64_t)P, 64ll * P - 1, -64ll * P + 1, 43ll * P, -43ll * P, 12345678901ll, -12345678901ll};
        for (int64_t y : ys) {
            const int32_t r = sreduce_wide_loose(y);
            if (smodp(r) != smodp(y) || r <= -(int32_t)(P / 50) || r >= (int32_t)(P + P / 50)) return 6;
            if (r > -(int32_t)P && canonical_of(r) != smodp(y)) return 6;
        }
        const int32_t hi = (int32_t)((uint64_t)P + (uint64_t)P * 45 / 1000);  // 1.045 p
        const int32_t xs[] = {0, 1, -1, (int32_t)P - 1, (int32_t)P, -(int32_t)P, hi - 1, -(hi - 1), (int32_t)(P / 2), -(int32_t)(P / 2)};
        for (int32_t x : xs) {
            uint32_t want = R_MOD_P;
            for (int k = 0; k < 7; ++k) want = mul(want, smodp(x));
            const int32_t l = p2::sbox7(x);
            if (smodp(l) != want || l <= -(int32_t)P || l >= (int32_t)P) return 7;
        }
    }
    for (uint32_t it = 0; it < 4 + iterations / 16; ++it) {
        // internal_layer with arbitrary (centred) constants: out_i = (kappa s_0 + sum_{i>=1} s_i) / R * rho / R + s_i m_i / R (+ constants)
        const int32_t top0 = (int32_t)((uint64_t)P * 97 / 100), top = (int32_t)((uint64_t)P * 92 / 100);  // |s_0| < 0.97 p, others < 0.92 p
        int32_t st[16], m[16];
        uint32_t want[16];
        for (int i = 0; i < 16; ++i) {
            const int32_t hi = i ? top : top0;
            const int32_t mag = it < 3 ? hi - 1 : (int32_t)(rnd() % (uint64_t)hi);
            st[i] = it == 0 ? mag : it == 1 ? -mag : it == 2 ? ((i & 1) ? mag : -mag) : ((rnd() & 1) ? mag : -mag);
            m[i] = it < 3 ? ((i & 2) ? (int32_t)(P / 2) : -(int32_t)(P / 2)) : centred(rp());
        }
        const int32_t kappa = it < 3 ? ((it & 1) ? (int32_t)(P / 2) : -(int32_t)(P / 2)) : centred(rp());
        const int32_t rho = it < 3 ? ((it & 2) ? (int32_t)(P / 2) : -(int32_t)(P / 2)) : centred(rp());
        const uint32_t next = it < 3 ? P - 1 : rp(), ex = it < 3 ? (P + 1) / 2 : rp();
        int64_t exit_c[16];
        for (int i = 0; i < 16; ++i) exit_c[i] = (int64_t)centred(ex) * (int64_t)R_MOD_P;
        const bool last = it & 1;
        // reference with canonical Montgomery arithmetic: x / R = mul(x, 1)
        uint32_t wide = mul(mul(smodp(st[0]), smodp(kappa)), R2_MOD_P);  // s_0 kappa (exact product, as a residue)
        for (int i = 1; i < 16; ++i) wide = add(wide, smodp(st[i]));
        const uint32_t sum = mul(wide, 1u);
        const uint32_t sum_r = mul(mul(sum, smodp(rho)), R2_MOD_P);      // sum * rho as a residue
        for (int i = 0; i < 16; ++i) {
            const uint32_t prod = mul(mul(smodp(st[i]), smodp(m[i])), R2_MOD_P);
            const uint32_t c = i ? (last ? mul(mul(smodp(centred(ex)), R_MOD_P), R2_MOD_P) : 0u) : mul(mul(smodp(centred(next)), R_MOD_P), R2_MOD_P);
            want[i] = mul(add(add(sum_r, prod), c), 1u);
        }
        if (last) p2::internal_layer<true>(st, m, kappa, rho, (int64_t)centred(next) * (int64_t)R_MOD_P, exit_c);
        else p2::internal_layer<false>(st, m, kappa, rho, (int64_t)centred(next) * (int64_t)R_MOD_P, exit_c);
        for (int i = 0; i < 16; ++i)
            if (smodp(st[i]) != want[i]) return 9;
        for (int i = 0; i < 16; ++i)
            if (st[i] <= -(int32_t)P || st[i] >= (int32_t)P) return 9;
    }
    for (uint32_t it = 0; it < iterations; ++it) {
        const uint32_t a = rp(), b = rp(), c = rp(), d = rp();
        {
            const int32_t x = (int32_t)((int64_t)(rnd() % (2ull * P + P / 11)) - (int64_t)(P + P / 22));  // (-1.045 p, 1.045 p); in 64 bits: UBSan, round 6
            uint32_t want = R_MOD_P;
            for (int k = 0; k < 7; ++k) want = mul(want, smodp(x));
            if (smodp(p2::sbox7(x)) != want) return 8;
            const int64_t y = (int64_t)(rnd() % (128ull * P)) - 64ll * P;
            if (smodp(sreduce_wide_loose(y)) != smodp(y)) return 8;
        }
        // Montgomery products against the definition a*b*R^-1
        const uint32_t ab = mul(a, b);
        if (ab >= P || (uint64_t)ab * R_MOD_P % P != (uint64_t)a * b % P) return 10;
        const uint32_t m2 = mul2(a, b, c, d);
        if (m2 != add(ab, mul(c, d))) return 11;
        const uint32_t la = (uint32_t)(rnd() % (2ull * P));  // a lazy operand
        const uint32_t ml = mul_lazy(la, b);
        if (ml >= 2 * P || ml % P != mul(la % P, b)) return 12;
        if (add_2p(la, (uint32_t)(rnd() % (2ull * P))) >= 2 * P) return 13;
        const uint64_t w = rnd() % (128ull * P);
        if (reduce_wide(w) != w % P) return 14;
        const uint32_t loose = reduce_wide_loose(w);
        if (loose % P != w % P || loose >= (uint64_t)P + P / 32) return 15;
        const uint64_t w16 = rnd() % (16ull * P);
        if (reduce_sum(w16) != w16 % P) return 16;
        if (wide_add(w16, a) != w16 + a || wide_fma(w16, a, 3) != w16 + 3ull * a || wide_mul(a, 4) != 4ull * a) return 17;
        // extension field: (x*y)*z == x*(y*z), x * x^-1 == 1, wide accumulation == sum of scalings
        const Ext x{{rp(), rp(), rp(), rp()}}, y{{rp(), rp(), rp(), rp()}}, z{{rp(), rp(), rp(), rp()}};
        if (!ext_eq(ext_mul(ext_mul(x, y), z), ext_mul(x, ext_mul(y, z)))) return 20;
        if (!ext_eq(ext_mul(x, ext_add(y, z)), ext_add(ext_mul(x, y), ext_mul(x, z)))) return 21;
        if ((x.c[0] | x.c[1] | x.c[2] | x.c[3]) && !ext_eq(ext_mul(x, ext_inv(x)), ext_one())) return 22;
        ExtWideAcc acc;
        Ext want = ext_zero();
        for (int k = 0; k < 37; ++k) {
            const Ext e{{rp(), rp(), rp(), rp()}};
            const uint32_t v = rp();
            acc.fma(e, v);
            want = ext_add(want, ext_scale(e, v));
        }
        if (!ext_eq(acc.result(), want)) return 23;
        ExtProductAcc pacc;
        Ext pwant = ext_zero();
        for (int k = 0; k < 29; ++k) {
            const bool edge = k < 2;  // all coordinates p - 1: the largest raw sums
            const Ext e{{edge ? P - 1 : rp(), edge ? P - 1 : rp(), edge ? P - 1 : rp(), edge ? P - 1 : rp()}};
            const Ext f{{edge ? P - 1 : rp(), edge ? P - 1 : rp(), edge ? P - 1 : rp(), edge ? P - 1 : rp()}};
            pacc.fma(e, f);
            pwant = ext_add(pwant, ext_mul(e, f));
            const uint32_t v = k == 2 ? P - 1 : rp();
            pacc.fma_base(f, v);
            pwant = ext_add(pwant, ext_scale(f, v));
        }
        if (!ext_eq(pacc.result(), pwant)) return 24;
        // centred signed accumulation (ExtCentredAcc): four terms between folds, coordinates at the edges of the centred range
        // (both neighbours of p / 2, 0, p - 1) in the first rounds — the largest magnitudes the fold's domain has to take
        ExtCentredAcc cacc, cacc_u;
        Ext cwant = ext_zero();
        const uint32_t half_lo = (P - 1) / 2, half_hi = (P + 1) / 2;
        for (int k = 0; k < 43; ++k) {
            const uint32_t ev = k < 8 ? ((k & 1) ? half_lo : half_hi) : rp();
            const Ext e{{k < 8 ? ev : rp(), k < 8 ? ev : rp(), k < 8 ? (P - 1) : rp(), k < 8 ? 0u : rp()}};
            const uint32_t v = k < 8 ? ((k & 2) ? half_hi : half_lo) : rp();
            int32_t ec[4];
            ext_centred(e, ec);
            cacc.fma(ec, centred(v));
            cacc_u.fma_uniform(ec, centred(v));
            cwant = ext_add(cwant, ext_scale(e, v));
            if ((k & 3) == 3) { cacc.fold(); cacc_u.fold(); }
        }
        if (!ext_eq(cacc.result(), cwant) || !ext_eq(cacc_u.result(), cwant)) return 25;
    }
    return 0;
}

}  // namespace pw
