/* numpy's `Generator(PCG64).integers(0, bound, size=n, dtype=uint32)` restated in C, bit for bit: the synthetic inputs of the tests and
 * of the bench's CPU baseline (powdr_amd/synth.py fill_dummy_traces_numpy) are drawn from that stream — 3.6 G values for the 2^17-row
 * CPU sample — and numpy spends 7 ns on each (one core of the GPU box). Same generator (PCG64: 128-bit LCG, XSL-RR output, numpy's
 * buffered 32-bit reads), same bounded method (Lemire's multiply-and-reject, numpy/random/src/distributions/distributions.c
 * buffered_bounded_lemire_uint32), so the golden digests that depend on these inputs do not move; tests/test_synth_rng.py compares it
 * with numpy draw by draw. Two steps per block of outputs: (1) the RAW 64-bit outputs, in parallel — an LCG jumps ahead in O(log n)
 * (pcg_advance_lcg_128), so every thread starts its share of the stream where the serial generator would be; (2) Lemire's rejection
 * over the raw 32-bit halves, sequentially (it decides how many raw values an output consumes; one multiply and a rarely taken branch
 * per value). The caller passes numpy's bit-generator state in and writes the returned state back, so numpy draws that follow
 * continue the stream. Input generation only: not linked into libpowdr_gpu.so. */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

typedef unsigned __int128 u128;

typedef struct {
    uint64_t state_hi, state_lo, inc_hi, inc_lo;
    uint32_t has_uint32, uinteger;
} SynthPcg64;

static inline uint64_t rotr64(uint64_t v, unsigned r) { return (v >> r) | (v << ((-r) & 63)); }
static inline uint64_t output64(u128 state) {
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    return rotr64(hi ^ lo, (unsigned)(hi >> 58));
}
#define PCG_MULT ((((u128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL)

/* state after `delta` steps of state <- state * mult + inc */
static u128 advance(u128 state, u128 delta, u128 inc) {
    u128 acc_mult = 1, acc_plus = 0, cur_mult = PCG_MULT, cur_plus = inc;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    return acc_mult * state + acc_plus;
}

int synth_pcg64_bounded_u32(SynthPcg64* s, uint32_t bound_excl, uint32_t* out, size_t n) {
    if (!s || !out || bound_excl < 2) return 1; /* (bound 1 and the full range take other branches in numpy: not needed here) */
    u128 state = ((u128)s->state_hi << 64) | s->state_lo;
    const u128 inc = ((u128)s->inc_hi << 64) | s->inc_lo;
    uint32_t has = s->has_uint32, held = s->uinteger;
    const uint32_t rng = bound_excl - 1u;
    const uint32_t threshold = (uint32_t)((0xFFFFFFFFu - rng) % bound_excl);
    /* outputs per block: 2^21 -> 9 MB of raw values, written by the team and read back by one thread out of the caches; the buffer is
     * kept between calls (a fresh 146 MB block per call spent its time in page faults, 256 threads queueing on the mm lock) */
    const size_t kBlock = (size_t)1 << 21;
    static __thread uint64_t* raw = NULL;
    static __thread size_t raw_cap = 0;
    size_t done = 0;
    while (done < n) {
        const size_t want = n - done < kBlock ? n - done : kBlock;
        /* raw 64-bit outputs for `want` results: 2 halves each, rejection rate threshold / 2^32 (< 1/2), plus slack; a block that runs
         * out of raw values simply ends early and the next one continues from the state reached */
        const double rej = (double)threshold / 4294967296.0;
        size_t m = (size_t)((double)want * (1.0 + 1.05 * rej / (1.0 - rej)) / 2.0) + 4096;
        if (m > raw_cap) { free(raw); raw = (uint64_t*)malloc(m * sizeof(uint64_t)); raw_cap = raw ? m : 0; if (!raw) return 2; }
        uint64_t* const rawp = raw; /* (`raw` is per thread: the team writes through the calling thread's pointer) */
#pragma omp parallel num_threads(16)
        {
#ifdef _OPENMP
            extern int omp_get_thread_num(void);
            extern int omp_get_num_threads(void);
            const size_t t = (size_t)omp_get_thread_num(), nt = (size_t)omp_get_num_threads();
#else
            const size_t t = 0, nt = 1;
#endif
            const size_t per = (m + nt - 1) / nt, lo = t * per, hi = lo + per < m ? lo + per : m;
            if (lo < hi) {
                u128 st = advance(state, (u128)lo, inc);
                for (size_t i = lo; i < hi; ++i) { st = st * PCG_MULT + inc; rawp[i] = output64(st); }
            }
        }
        /* Lemire over the halves: the held half of an earlier 64-bit output first, then low, high, low, high ... */
        size_t k = 0;      /* 64-bit outputs whose LOW half has been consumed */
        size_t produced = 0;
        while (produced < want) {
            uint32_t r;
            if (has) { r = held; has = 0; }
            else { if (k == m) break; r = (uint32_t)rawp[k]; held = (uint32_t)(rawp[k] >> 32); has = 1; ++k; }
            const uint64_t mm = (uint64_t)r * bound_excl;
            /* numpy tests `leftover < bound` first to put off computing the threshold; threshold < bound, so this test alone decides
             * the same way */
            if (__builtin_expect((uint32_t)mm < threshold, 0)) continue;
            out[done + produced++] = (uint32_t)(mm >> 32);
        }
        state = advance(state, (u128)k, inc);
        done += produced;
    }
    s->state_hi = (uint64_t)(state >> 64); s->state_lo = (uint64_t)state;
    s->has_uint32 = has; s->uinteger = held;
    return 0;
}
