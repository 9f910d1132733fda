// Device-side helpers of the RUN-TIME SPECIALISED expression kernels (csrc/jit.hpp): this header, babybear.hpp and ext.hpp
// are embedded in libpowdr_gpu as text and handed to hiprtc together with the per-AIR source csrc/jit_codegen.cpp emits.
// The same functions are used by the ahead-of-time kernels of logup_kernels.hip, so the interpreter ("parity twin") and the
// specialised code share their arithmetic.
#pragma once
#include "ext.hpp"

namespace pwj {

using bb::Ext;

// trace cell (column base + byte offset of the row): `base` is wave-uniform (a scalar register pair), `off4` a 32-bit lane
// offset — the global_load form with a scalar base and a 32-bit vector offset, no 64-bit lane arithmetic per access
__device__ __forceinline__ uint32_t ld(const uint32_t* __restrict__ base, uint32_t off4) {
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + off4);
}
__device__ __forceinline__ void st(uint32_t* __restrict__ base, uint32_t off4, uint32_t v) {
    *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + off4) = v;
}

// d_i = al + bus_i + sum_j bl^(j+1) a_ij. Every coordinate is a sum of products accumulated RAW in a signed 64-bit register
// (one v_mad_i64_i32 per term) and reduced by signed Montgomery reductions (bb::smont, no conditional subtraction): the
// challenge powers are wave-uniform, so their centred representatives (|b| <= p/2) come from the scalar unit for free, an
// argument a is a canonical word, a product is below p^2 / 2 and two of them fit the reduction's domain (1.209 p^2) on top of
// what is already there (<= 0.134 p^2); after two products the accumulator is reduced and re-enters as r * (R mod p).
struct DenominatorSeeds {
    int64_t s[4];  // centred(al_k) * (R mod p): the value al_k in the accumulators' domain (k = 0: without the bus)
    uint32_t al0;
};
__device__ __forceinline__ DenominatorSeeds denominator_seeds(const Ext& al) {
    DenominatorSeeds sd;
#pragma unroll
    for (int k = 0; k < 4; ++k) sd.s[k] = (int64_t)bb::centred(al.c[k]) * (int64_t)bb::R_MOD_P;
    sd.al0 = al.c[0];
    return sd;
}
struct DenominatorAcc {
    int64_t T[4];
    uint32_t pending;
    __device__ __forceinline__ DenominatorAcc(const DenominatorSeeds& sd, uint32_t bus_monty)
        : T{(int64_t)bb::centred(bb::add(sd.al0, bus_monty)) * (int64_t)bb::R_MOD_P, sd.s[1], sd.s[2], sd.s[3]}, pending(0) {}
    __device__ __forceinline__ void reduce() {
#pragma unroll
        for (int k = 0; k < 4; ++k) T[k] = bb::smul_uniform(bb::smont(T[k]), (int32_t)bb::R_MOD_P);
        pending = 0;
    }
    // argument j (canonical word a), b = bl^(j+1) (wave-uniform)
    __device__ __forceinline__ void add(uint32_t a, const Ext& b) {
        if (pending == 2) reduce();
#pragma unroll
        for (int k = 0; k < 4; ++k) T[k] = bb::swide_mad_uniform(T[k], (int32_t)a, bb::centred(b.c[k]));
        ++pending;
    }
    __device__ __forceinline__ Ext result() {
        if (pending == 2) reduce();  // keep the last reduction's result inside (-p, p)
        Ext d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x = (uint32_t)bb::smont(T[k]);  // in (-0.8 p, 0.8 p)
            d.c[k] = bb::umin(x, x + bb::P);
        }
        return d;
    }
};
template <int NA>
__device__ __forceinline__ Ext denominator(const DenominatorSeeds& sd, uint32_t bus_monty, const Ext* __restrict__ blpow, const uint32_t (&a)[NA]) {
    DenominatorAcc acc(sd, bus_monty);
#pragma unroll
    for (int j = 0; j < NA; ++j) acc.add(a[j], blpow[j + 1]);
    return acc.result();
}
__device__ __forceinline__ Ext denominator0(const DenominatorSeeds& sd, uint32_t bus_monty) {  // an interaction without arguments
    DenominatorAcc acc(sd, bus_monty);
    return acc.result();
}

// d1 m0 + d0 m1 (the numerator of a two-member group): the two products of a coordinate share one Montgomery reduction
__device__ __forceinline__ Ext scale2(const Ext& a, uint32_t x, const Ext& b, uint32_t y) {
    return {{bb::mul2(a.c[0], x, b.c[0], y), bb::mul2(a.c[1], x, b.c[1], y), bb::mul2(a.c[2], x, b.c[2], y), bb::mul2(a.c[3], x, b.c[3], y)}};
}

// Extension-field inversion split around its one base-field inversion (bb::ext_inv: norm to the quadratic subfield, then to
// the base field), so that several elements can share the ~40-multiplication exponentiation (Montgomery's trick on the norms).
struct ExtInvPrep { uint32_t d0, d1, n; };
__device__ __forceinline__ ExtInvPrep ext_inv_prepare(const Ext& a) {
    const uint32_t W = bb::w11();
    const uint32_t s0 = bb::add(bb::sqr(a.c[0]), bb::mul(W, bb::sqr(a.c[2])));
    const uint32_t s1 = bb::double_(bb::mul(a.c[0], a.c[2]));
    const uint32_t t0 = bb::add(bb::sqr(a.c[1]), bb::mul(W, bb::sqr(a.c[3])));
    const uint32_t t1 = bb::double_(bb::mul(a.c[1], a.c[3]));
    ExtInvPrep p;
    p.d0 = bb::sub(s0, bb::mul(W, t1));
    p.d1 = bb::sub(s1, t0);
    p.n = bb::sub(bb::sqr(p.d0), bb::mul(W, bb::sqr(p.d1)));
    return p;
}
__device__ __forceinline__ Ext ext_inv_finish(const Ext& a, const ExtInvPrep& p, uint32_t n_inv) {
    const uint32_t W = bb::w11();
    const uint32_t e0 = bb::mul(p.d0, n_inv);
    const uint32_t e1 = bb::neg(bb::mul(p.d1, n_inv));
    const uint32_t r00 = bb::add(bb::mul(a.c[0], e0), bb::mul(W, bb::mul(a.c[2], e1)));
    const uint32_t r01 = bb::add(bb::mul(a.c[0], e1), bb::mul(a.c[2], e0));
    const uint32_t r10 = bb::neg(bb::add(bb::mul(a.c[1], e0), bb::mul(W, bb::mul(a.c[3], e1))));
    const uint32_t r11 = bb::neg(bb::add(bb::mul(a.c[1], e1), bb::mul(a.c[3], e0)));
    return {{r00, r10, r01, r11}};
}
// 1 / n[i] for K base-field elements with ONE inversion; a zero stays a zero's partner: the norm of an extension element is
// zero only for the zero element, whose inverse ext_inv_finish returns as zero whatever factor it is handed (bb::inv(0) = 0
// in the one-by-one form), so a zero norm is replaced by one here.
template <int K>
__device__ __forceinline__ void batch_inverse(const uint32_t (&n)[K], uint32_t (&out)[K]) {
    uint32_t x[K], pre[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        x[i] = n[i] == 0u ? bb::R_MOD_P : n[i];
        pre[i] = i ? bb::mul(pre[i - 1], x[i]) : x[i];
    }
    uint32_t inv = bb::inv(pre[K - 1]);
#pragma unroll
    for (int i = K - 1; i > 0; --i) {
        out[i] = bb::mul(inv, pre[i - 1]);
        inv = bb::mul(inv, x[i]);
    }
    out[0] = inv;
}

}  // namespace pwj
