// Degree-4 extension E = F[X]/(X^4 - 11) of BabyBear on Montgomery words, for
// device kernels and the host transcript. Challenges (alpha, zeta, gamma, beta)
// and everything derived from them live in E; trace data stays in F.
#pragma once
#include "babybear.hpp"

namespace bb {

struct Ext {
    uint32_t c[4];
};

// 11 in Montgomery form: 11 * 2^32 mod p
PW_HD uint32_t w11() { return 939524073u; }  // 0x37ffffe9

PW_HD Ext ext_zero() { return {{0u, 0u, 0u, 0u}}; }
PW_HD Ext ext_one() { return {{R_MOD_P, 0u, 0u, 0u}}; }
PW_HD Ext ext_from_base(uint32_t a) { return {{a, 0u, 0u, 0u}}; }
PW_HD Ext ext_add(const Ext& a, const Ext& b) {
    return {{add(a.c[0], b.c[0]), add(a.c[1], b.c[1]), add(a.c[2], b.c[2]), add(a.c[3], b.c[3])}};
}
PW_HD Ext ext_sub(const Ext& a, const Ext& b) {
    return {{sub(a.c[0], b.c[0]), sub(a.c[1], b.c[1]), sub(a.c[2], b.c[2]), sub(a.c[3], b.c[3])}};
}
PW_HD Ext ext_neg(const Ext& a) { return {{neg(a.c[0]), neg(a.c[1]), neg(a.c[2]), neg(a.c[3])}}; }
PW_HD Ext ext_scale(const Ext& a, uint32_t k) {
    return {{mul(a.c[0], k), mul(a.c[1], k), mul(a.c[2], k), mul(a.c[3], k)}};
}
PW_HD bool ext_eq(const Ext& a, const Ext& b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}

// Products are reduced in pairs (bb::mul2: a*b + c*d with one Montgomery reduction), which halves the reductions and
// saves a modular addition per pair: 75 instructions instead of 131 for the schoolbook form.
PW_HD Ext ext_mul(const Ext& a, const Ext& b) {
    const uint32_t W = w11();
    // the parts that get multiplied by 11
    const uint32_t h0 = add(mul2(a.c[1], b.c[3], a.c[2], b.c[2]), mul(a.c[3], b.c[1]));
    const uint32_t h1 = mul2(a.c[2], b.c[3], a.c[3], b.c[2]);
    const uint32_t h2 = mul(a.c[3], b.c[3]);
    Ext r;
    r.c[0] = mul2(a.c[0], b.c[0], W, h0);
    r.c[1] = add(mul2(a.c[0], b.c[1], a.c[1], b.c[0]), mul(W, h1));
    r.c[2] = add(mul2(a.c[0], b.c[2], a.c[1], b.c[1]), mul2(a.c[2], b.c[0], W, h2));
    r.c[3] = add(mul2(a.c[0], b.c[3], a.c[1], b.c[2]), mul2(a.c[2], b.c[1], a.c[3], b.c[0]));
    return r;
}
PW_HD Ext ext_sqr(const Ext& a) { return ext_mul(a, a); }

// sum_k e_k * x_k with e_k in E, x_k in F, reduced once at the end: per term and coordinate one 32x32+64-bit
// multiply-add into a 96-bit accumulator (plus the carry) instead of a Montgomery product and a modular addition
// (8 instructions). The raw products of Montgomery words sum to (sum e x) R^2, so one division by R at the end
// returns the Montgomery form of the sum. Exact for up to 2^32 terms.
struct ExtWideAcc {
    uint64_t lo[4];
    uint32_t hi[4];
    PW_HD ExtWideAcc() : lo{0, 0, 0, 0}, hi{0, 0, 0, 0} {}
    PW_HD void fma(const Ext& e, uint32_t x) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t s = lo[k] + (uint64_t)e.c[k] * x;
            hi[k] += s < lo[k] ? 1u : 0u;
            lo[k] = s;
        }
    }
    PW_HD Ext result() const {
        Ext r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // hi 2^64 + lo (mod p), 2^64 = R^2
            const uint64_t t = (uint64_t)(hi[k] % P) * R2_MOD_P + lo[k] % P;
            r.c[k] = mul((uint32_t)(t % P), 1u);  // * R^-1
        }
        return r;
    }
};

// The same sums with CENTRED operands in signed 64-bit accumulators (round 3): e_k given as centred representatives (|e| <= p/2,
// the wave-uniform ones straight from a table the host centred), x_k centred on the fly; a product is below p^2 / 4, so FOUR terms
// (p^2) fit the signed Montgomery reduction's domain (1.209 p^2) on top of what an earlier fold left (<= 0.14 p^2), and every
// fourth term `fold` brings the accumulators back: acc <- smont(acc) * (R mod p), the same residue, |.| <= 0.14 p^2. Per term and
// coordinate 1 multiply-add + 3/4 instruction of folding + a share of the centring, against a multiply-add and a carry chain (3-4).
struct ExtCentredAcc {
    int64_t a[4];
    PW_HD ExtCentredAcc() : a{0, 0, 0, 0} {}
    // e: centred coordinates, x: centred value; at most four calls between two folds
    PW_HD void fma(const int32_t (&e)[4], int32_t x) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
            int64_t out;
            asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(out) : "v"(e[k]), "v"(x), "v"(a[k]) : "vcc");
            a[k] = out;
#else
            a[k] += (int64_t)e[k] * x;
#endif
        }
    }
    PW_HD void fma_uniform(const int32_t (&e)[4], int32_t x) {  // e wave-uniform (scalar registers)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = swide_mad_uniform(a[k], x, e[k]);
    }
    PW_HD void fold() {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = smul_uniform(smont(a[k]), (int32_t)R_MOD_P);
    }
    // the raw products of Montgomery words carry R^2: one signed reduction returns the Montgomery form of the sum
    PW_HD Ext result() {
        fold();  // |a| <= 0.14 p^2: the reduction below lands in (-0.57 p, 0.57 p)
        Ext r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.c[k] = canonical_of(smont(a[k]));
        return r;
    }
};
PW_HD void ext_centred(const Ext& e, int32_t (&out)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = centred(e.c[k]);
}

// sum_k a_k * c_k with a_k, c_k in E, reduced once at the end: the seven coefficients of the product polynomial
// (before X^4 = 11) are sums of at most four raw 64-bit products (4 p^2 < 2^64) and go into 96-bit accumulators — 16
// multiply-adds + 7 carries per term where ext_mul + ext_add take 87 instructions. Exact for up to 2^32 terms.
struct ExtProductAcc {
    uint64_t lo[7];
    uint32_t hi[7];
    PW_HD ExtProductAcc() : lo{0, 0, 0, 0, 0, 0, 0}, hi{0, 0, 0, 0, 0, 0, 0} {}
    PW_HD void add_raw(int k, uint64_t e) {
        const uint64_t s = lo[k] + e;
        hi[k] += s < e ? 1u : 0u;
        lo[k] = s;
    }
    PW_HD void fma(const Ext& a, const Ext& c) {
        add_raw(0, (uint64_t)a.c[0] * c.c[0]);
        add_raw(1, (uint64_t)a.c[0] * c.c[1] + (uint64_t)a.c[1] * c.c[0]);
        add_raw(2, (uint64_t)a.c[0] * c.c[2] + (uint64_t)a.c[1] * c.c[1] + (uint64_t)a.c[2] * c.c[0]);
        add_raw(3, ((uint64_t)a.c[0] * c.c[3] + (uint64_t)a.c[1] * c.c[2]) + ((uint64_t)a.c[2] * c.c[1] + (uint64_t)a.c[3] * c.c[0]));
        add_raw(4, (uint64_t)a.c[1] * c.c[3] + (uint64_t)a.c[2] * c.c[2] + (uint64_t)a.c[3] * c.c[1]);
        add_raw(5, (uint64_t)a.c[2] * c.c[3] + (uint64_t)a.c[3] * c.c[2]);
        add_raw(6, (uint64_t)a.c[3] * c.c[3]);
    }
    PW_HD void fma_base(const Ext& a, uint32_t x) {  // a * (x, 0, 0, 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) add_raw(k, (uint64_t)a.c[k] * x);
    }
    PW_HD Ext result() const {
        uint32_t r[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            // hi 2^64 + lo (mod p), 2^64 = R^2; the raw products carry R^2, one division by R returns Montgomery form
            const uint64_t t = (uint64_t)(hi[k] % P) * R2_MOD_P + lo[k] % P;
            r[k] = mul((uint32_t)(t % P), 1u);
        }
        const uint32_t W = w11();
        return {{add(r[0], mul(W, r[4])), add(r[1], mul(W, r[5])), add(r[2], mul(W, r[6])), r[3]}};
    }
};

// Inverse via the norm to the quadratic subfield K = F[Y]/(Y^2 - 11), Y = X^2:
// a = A0 + A1 X with A0 = (a0, a2), A1 = (a1, a3) in K; a^-1 = (A0 - A1 X) / (A0^2 - Y A1^2).
PW_HD Ext ext_inv(const Ext& a) {
    const uint32_t W = w11();
    // A0^2 = (a0^2 + 11 a2^2, 2 a0 a2), A1^2 = (a1^2 + 11 a3^2, 2 a1 a3)
    uint32_t s0 = add(sqr(a.c[0]), mul(W, sqr(a.c[2])));
    uint32_t s1 = double_(mul(a.c[0], a.c[2]));
    uint32_t t0 = add(sqr(a.c[1]), mul(W, sqr(a.c[3])));
    uint32_t t1 = double_(mul(a.c[1], a.c[3]));
    // D = A0^2 - Y*A1^2,  Y*(t0 + t1 Y) = 11 t1 + t0 Y
    uint32_t d0 = sub(s0, mul(W, t1));
    uint32_t d1 = sub(s1, t0);
    // D^-1 = (d0 - d1 Y) / (d0^2 - 11 d1^2)
    uint32_t n = sub(sqr(d0), mul(W, sqr(d1)));
    uint32_t ni = inv(n);
    uint32_t e0 = mul(d0, ni);
    uint32_t e1 = neg(mul(d1, ni));
    // R0 = A0 * E, R1 = -A1 * E  (K multiplication)
    uint32_t r00 = add(mul(a.c[0], e0), mul(W, mul(a.c[2], e1)));
    uint32_t r01 = add(mul(a.c[0], e1), mul(a.c[2], e0));
    uint32_t r10 = neg(add(mul(a.c[1], e0), mul(W, mul(a.c[3], e1))));
    uint32_t r11 = neg(add(mul(a.c[1], e1), mul(a.c[3], e0)));
    return {{r00, r10, r01, r11}};
}

PW_HD Ext ext_pow(Ext a, uint64_t e) {
    Ext r = ext_one();
    while (e) {
        if (e & 1) r = ext_mul(r, a);
        a = ext_sqr(a);
        e >>= 1;
    }
    return r;
}

}  // namespace bb
