/*
 * TEST INFRASTRUCTURE — CPU oracle. Not part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * BabyBear on canonical representatives, plain 64-bit `%` arithmetic (no
 * Montgomery form on purpose, so that it shares nothing with the device code).
 * Modulus pinned by /root/reference/number/src/baby_bear.rs:46-55
 * ((p-1)/2 == 0x3c000000 => p == 0x78000001); canonical-u32 I/O as in
 * number/src/plonky3_macros.rs:38-68.
 */
#ifndef ORACLE_BABYBEAR_H
#define ORACLE_BABYBEAR_H
#include <stdint.h>

#define OR_P 0x78000001u

static inline uint32_t or_add(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % OR_P); }
static inline uint32_t or_sub(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + OR_P - b) % OR_P); }
static inline uint32_t or_mul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % OR_P); }
static inline uint32_t or_neg(uint32_t a) { return a ? OR_P - a : 0u; }
static inline uint32_t or_pow(uint32_t a, uint64_t e) {
    uint32_t r = 1;
    while (e) {
        if (e & 1) r = or_mul(r, a);
        a = or_mul(a, a);
        e >>= 1;
    }
    return r;
}
/* Fermat inverse; callers handle 0 (reference: `if divisor.is_zero() {0} else {inverse}`,
 * openvm/src/powdr_extension/trace_generator/cpu/mod.rs:192-199) */
static inline uint32_t or_inv(uint32_t a) { return or_pow(a, OR_P - 2u); }

#endif
