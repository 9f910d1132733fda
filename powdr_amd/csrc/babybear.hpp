// BabyBear field arithmetic for gfx950 device code and for the host-side
// transcript. p = 2^31 - 2^27 + 1 (pinned by
// /root/reference/number/src/baby_bear.rs:46-55). Elements live in memory in
// Montgomery form with R = 2^32, the representation of the reference's device
// type `Fp` (used at openvm/cuda/src/expr_eval.cuh:36-89) and of
// p3_baby_bear::BabyBear behind `BabyBearField`
// (number/src/plonky3_macros.rs:42) — assumption A1 of SURVEY.md.
//
// All values are kept fully reduced in [0, p). The Montgomery product is the
// additive form: t = a*b; m = lo32(t) * (-p^-1); (t + m*p) >> 32 < 2p, one
// conditional subtract. On CDNA4 this is two v_mad_u64_u32 + one v_mul_lo_u32.
#pragma once
#if defined(__HIPCC_RTC__)  // hiprtc (csrc/jit.hpp: run-time specialised kernels): no system headers, the runtime is built in
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint8_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
#else
#include <stdint.h>
#endif

#if defined(__HIPCC__)
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif
#define PW_HD __host__ __device__ __forceinline__
#else
#define PW_HD inline
#endif

namespace bb {

constexpr uint32_t P = 0x78000001u;
constexpr uint32_t NEG_PINV = 0x77ffffffu;  // -p^-1 mod 2^32
constexpr uint32_t R_MOD_P = 0x0ffffffeu;   // 2^32 mod p  == monty(1)
constexpr uint32_t R2_MOD_P = 1172168163u;  // 2^64 mod p

PW_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// x in [0, 2p) -> [0, p)
PW_HD uint32_t reduce_2p(uint32_t x) { return umin(x, x - P); }

PW_HD uint32_t add(uint32_t a, uint32_t b) { return reduce_2p(a + b); }
PW_HD uint32_t sub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;
    return umin(d, d + P);
}
PW_HD uint32_t neg(uint32_t a) { return a == 0 ? 0u : P - a; }

// Montgomery reduction of t < p * 2^32: returns t * 2^-32 mod p.
PW_HD uint32_t monty_reduce(uint64_t t) {
    uint32_t m = (uint32_t)t * NEG_PINV;
    uint64_t u = t + (uint64_t)m * P;
    return reduce_2p((uint32_t)(u >> 32));
}
// the same without the conditional subtraction: t < 2^64 - 2^32 p (= 2.418 p^2)  ->  a representative in [0, t / 2^32 + p)
PW_HD uint32_t monty_reduce_lazy(uint64_t t) {
    uint32_t m = (uint32_t)t * NEG_PINV;
    uint64_t u = t + (uint64_t)m * P;
    return (uint32_t)(u >> 32);
}
// a in [0, 1.032 p) (reduce_wide_loose's range), b in [0, p)  ->  a representative of a + b in [0, 1.032 p)
PW_HD uint32_t add_loose(uint32_t a, uint32_t b) { return reduce_2p(a + b); }
PW_HD uint32_t mul(uint32_t a, uint32_t b) { return monty_reduce((uint64_t)a * b); }
// a*b + c*d in one Montgomery reduction: 2 p^2 + p 2^32 < 2^64, and the result is < 2 * 0.47 p + p < 2p.
// (Three raw products do not fit: the reduction itself adds up to p 2^32.)
PW_HD uint32_t mul2(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return monty_reduce((uint64_t)a * b + (uint64_t)c * d); }
PW_HD uint32_t sqr(uint32_t a) { return mul(a, a); }

// canonical u32 (< p) <-> Montgomery
PW_HD uint32_t to_monty(uint32_t canonical) { return mul(canonical, R2_MOD_P); }
PW_HD uint32_t from_monty(uint32_t x) { return monty_reduce((uint64_t)x); }

// Lazy product: a in [0, 2p), b in [0, p) -> result in [0, 2p) (a*b < 2p^2 < p * 2^32).
// Note for the record: lazy ADDITION is not available for this prime in 32-bit lanes — sums of
// two [0, 2p) values reach 4p > 2^32 — so the NTT butterflies stay fully reduced.
PW_HD uint32_t mul_lazy(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b;
    uint32_t m = (uint32_t)t * NEG_PINV;
    uint64_t u = t + (uint64_t)m * P;
    return (uint32_t)(u >> 32);
}

// Sums of many field elements without intermediate reductions: a 64-bit accumulator takes one instruction per term
// (v_mad_u64_u32 with the multiplier 1 is a 32+64-bit add; a modular addition is 3), reduced once by reduce_sum.
PW_HD uint64_t wide_add(uint64_t acc, uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t out;
    asm("v_mad_u64_u32 %0, vcc, %1, 1, %2" : "=v"(out) : "v"(x), "v"(acc) : "vcc");
    return out;
#else
    return acc + x;
#endif
}
// acc + k * x in one instruction (k a small constant)
PW_HD uint64_t wide_fma(uint64_t acc, uint32_t x, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t out;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(out) : "v"(x), "n"(k), "v"(acc) : "vcc");
    return out;
#else
    return acc + (uint64_t)x * k;
#endif
}
// c + k * x for a wave-uniform 64-bit c (the compiler keeps it in an SGPR pair: no copy into vector registers)
PW_HD uint64_t wide_fma_uniform(uint64_t c, uint32_t x, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t out;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(out) : "v"(x), "n"(k), "s"(c) : "vcc");
    return out;
#else
    return c + (uint64_t)x * k;
#endif
}
// acc + x * u for a wave-uniform u (a scalar register). Written as an instruction so that the compiler keeps `acc`
// as the addend of the multiply-add (it otherwise computes x * u early and spends a second multiply-add on the sum).
PW_HD uint64_t wide_mad_uniform(uint64_t acc, uint32_t x, uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t out;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(out) : "v"(x), "s"(u), "v"(acc) : "vcc");
    return out;
#else
    return acc + (uint64_t)x * u;
#endif
}
// k * x as a 64-bit value
PW_HD uint64_t wide_mul(uint32_t x, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t out;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(out) : "v"(x), "n"(k) : "vcc");
    return out;
#else
    return (uint64_t)x * k;
#endif
}
// x < 128 p  ->  a representative of x mod p in [0, 1.03 p).  q = floor((x >> 7) * 273 / 2^32) with
// 273 = floor(2^39 / p) = floor(273.07): q <= x / p and x - q p < (1 - 273 / 273.07) x + p + 273 * 128 < 1.03 p.
// Three instructions: a funnel shift, a mul_hi and ONE multiply-add — x + q (2^32 - p), of which only the low word is
// kept (the ISA has no 32-bit multiply-subtract; the 64-bit multiply-add does it when its high half is ignored).
PW_HD uint32_t reduce_wide_loose(uint64_t x) {
    const uint32_t q = (uint32_t)(((x >> 7) & 0xffffffffull) * 273u >> 32);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_REDUCE_MAD)
    uint64_t out;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(out) : "v"(q), "s"(0u - P), "v"(x) : "vcc");
    return (uint32_t)out;
#else
    return (uint32_t)x - q * P;
#endif
}
// x < 128 p  ->  x mod p
PW_HD uint32_t reduce_wide(uint64_t x) { return reduce_2p(reduce_wide_loose(x)); }
// x < 16 p  ->  x mod p.  q = floor(x / 2^31) <= x / p, and x - q p < x (2^27 - 1) / 2^31 + p < 2p for x < 16 p.
PW_HD uint32_t reduce_sum(uint64_t x) {
    const uint32_t q = (uint32_t)(x >> 31);
    return reduce_2p((uint32_t)x - q * P);
}

// ---- signed representatives --------------------------------------------------------------------------------------------
// The Poseidon2 permutation (poseidon2.hpp) keeps its state as SIGNED 32-bit representatives: with a signed Montgomery
// factor m in [-2^31, 2^31) the reduction (t + m p) / 2^32 lands in (t / 2^32 - p / 2, t / 2^32 + p / 2), so a product of
// two words below 1.04 p is again below p in magnitude WITHOUT any conditional subtraction — x^7 is four products of three
// instructions each — and sums go through signed 64-bit multiply-adds (v_mad_i64_i32) exactly like the unsigned ones.
// t must satisfy |t| + 2^31 p < 2^63, i.e. |t| < 1.209 p^2.
PW_HD int64_t smul(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t out;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(out) : "v"(a), "v"(b) : "vcc");
    return out;
#else
    return (int64_t)a * b;
#endif
}
PW_HD int64_t smul_uniform(int32_t a, int32_t u) {  // u wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t out;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(out) : "v"(a), "s"(u) : "vcc");
    return out;
#else
    return (int64_t)a * u;
#endif
}
PW_HD int32_t smont(int64_t t) {
    const int32_t m = (int32_t)((uint32_t)t * NEG_PINV);
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t u;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(u) : "v"(m), "s"((int32_t)P), "v"(t) : "vcc");
    return (int32_t)(u >> 32);
#else
    return (int32_t)((t + (int64_t)m * (int64_t)P) >> 32);
#endif
}
// acc + x, acc + k x (k a small constant), c + k x for a wave-uniform c, acc + x u for a wave-uniform u, k x
PW_HD int64_t swide_add(int64_t acc, int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t out;
    asm("v_mad_i64_i32 %0, vcc, %1, 1, %2" : "=v"(out) : "v"(x), "v"(acc) : "vcc");
    return out;
#else
    return acc + x;
#endif
}
PW_HD int64_t swide_fma(int64_t acc, int32_t x, int32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t out;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(out) : "v"(x), "n"(k), "v"(acc) : "vcc");
    return out;
#else
    return acc + (int64_t)x * k;
#endif
}
PW_HD int64_t swide_fma_uniform(int64_t c, int32_t x, int32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t out;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(out) : "v"(x), "n"(k), "s"(c) : "vcc");
    return out;
#else
    return c + (int64_t)x * k;
#endif
}
PW_HD int64_t swide_mad_uniform(int64_t acc, int32_t x, int32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t out;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(out) : "v"(x), "s"(u), "v"(acc) : "vcc");
    return out;
#else
    return acc + (int64_t)x * u;
#endif
}
PW_HD int64_t swide_mul(int32_t x, int32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t out;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(out) : "v"(x), "n"(k) : "vcc");
    return out;
#else
    return (int64_t)x * k;
#endif
}
// |y| < 64 p  ->  a representative of y mod p in (-0.017 p, 1.017 p) (for the |y| < 43 p of Poseidon2: (-0.011 p, 1.011 p)):  q = floor(floor(y / 2^7) * 273 / 2^32) with
// 273 = floor(2^39 / p); y - q p = y (1 - 273 p / 2^39) + (rounding of the two floors, in [0, p + 273 p / 2^32)), and
// 1 - 273 p / 2^39 = 0.000256. Three instructions: a funnel shift, a signed mul_hi and one multiply-add y + q (-p) of
// which only the low word is kept.
PW_HD int32_t sreduce_wide_loose(int64_t y) {
    const int32_t a = (int32_t)(y >> 7);
    const int32_t q = (int32_t)(((int64_t)a * 273) >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t out;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(out) : "v"(q), "s"(-(int32_t)P), "v"(y) : "vcc");
    return (int32_t)out;
#else
    return (int32_t)((uint32_t)y - (uint32_t)q * P);
#endif
}
// the centred representative (-p/2, p/2) of a canonical word
PW_HD int32_t centred(uint32_t canonical) { return canonical > P / 2 ? (int32_t)(canonical - P) : (int32_t)canonical; }
// a signed representative in (-p, 1.13 p) -> canonical (u + p must not wrap for the non-negative ones)
PW_HD uint32_t canonical_of(int32_t x) {
    uint32_t u = (uint32_t)x;
    u = umin(u, u + P);      // negative (huge as unsigned) -> + p
    return umin(u, u - P);   // [p, 2p) -> - p
}

// Addition and subtraction on representatives in [0, 2p) (2p < 2^32 < 4p, so the sum needs its carry). Used by the
// forward NTT, whose butterflies then take a lazy product as they come: 3 + 4 + 3 instructions instead of 5 + 3 + 3.
constexpr uint32_t TWO_P = 2u * P;
PW_HD uint32_t add_2p(uint32_t a, uint32_t b) {
    const uint32_t s = a + b;
    const uint32_t t = s - TWO_P;
    return s < a ? t : umin(s, t);  // carry: the true sum is s + 2^32 >= 2p, and s - 2p (mod 2^32) is it minus 2p
}
PW_HD uint32_t sub_2p(uint32_t a, uint32_t b) {
    const uint32_t d = a - b;
    return a < b ? d + TWO_P : d;
}

PW_HD uint32_t double_(uint32_t a) { return add(a, a); }
PW_HD uint32_t halve(uint32_t a) {
    // a/2 mod p: if odd add p (p odd) then shift. a + p < 2^32.
    uint32_t t = (a & 1u) ? a + P : a;
    return t >> 1;
}

PW_HD uint32_t pow_u32(uint32_t a, uint32_t e) {
    uint32_t r = R_MOD_P;
    while (e) {
        if (e & 1u) r = mul(r, a);
        a = sqr(a);
        e >>= 1;
    }
    return r;
}

// a^(p-2); p-2 = 0b1110111111111111111111111111111 (31 bits).
// Chain: x^7 -> x^(2^3-1); then p-2 = (2^4-1)<<27 - ... handled generically:
// p-2 = 0x77ffffff = 7 * 2^28 + (2^27 - 1).
PW_HD uint32_t inv(uint32_t a) {
    uint32_t x2 = sqr(a);
    uint32_t x3 = mul(x2, a);          // a^3      (2 bits)
    uint32_t x6 = sqr(x3);
    uint32_t x7 = mul(x6, a);          // a^7      (3 bits)
    // a^(2^27-1): build by doubling chain of all-ones exponents
    uint32_t o3 = x7;                  // 2^3-1
    uint32_t o6 = o3;
    for (int i = 0; i < 3; ++i) o6 = sqr(o6);
    o6 = mul(o6, o3);                  // 2^6-1
    uint32_t o12 = o6;
    for (int i = 0; i < 6; ++i) o12 = sqr(o12);
    o12 = mul(o12, o6);                // 2^12-1
    uint32_t o24 = o12;
    for (int i = 0; i < 12; ++i) o24 = sqr(o24);
    o24 = mul(o24, o12);               // 2^24-1
    uint32_t o27 = o24;
    for (int i = 0; i < 3; ++i) o27 = sqr(o27);
    o27 = mul(o27, o3);                // 2^27-1
    // result = a^(7*2^28) * a^(2^27-1): (a^7)^(2^28)
    uint32_t hi = x7;
    for (int i = 0; i < 28; ++i) hi = sqr(hi);
    return mul(hi, o27);
}
PW_HD uint32_t inv_or_zero(uint32_t a) { return a == 0 ? 0u : inv(a); }

}  // namespace bb
