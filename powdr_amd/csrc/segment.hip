// Segment-level entry points: prove / verify all AIRs of one segment.
//
// The counterpart of the engine call the reference makes once per segment with the traces of all chips,
// `engine.prove(pk, ProvingContext{per_trace})` (/root/reference/openvm/src/trace_generation.rs:97-139,
// openvm-riscv/src/lib.rs:327-341; `per_trace: [(air_id, AirProvingContext)]`, openvm/src/empirical_constraints.rs:131).
// Real segments hold tens of AIRs of very different heights; a 2^10-row proof is latency bound
// (~3.5 ms of launches and Fiat-Shamir round trips, the GPU mostly idle), so the AIRs are proven concurrently by a
// few host threads, each driving its own HIP stream (the library's launch stream is per host thread).
#include "common.hpp"
#include "prover_internal.hpp"
#include "../../include/powdr_prover.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

namespace {

struct Digest { uint32_t w[8]; };

Digest compress_monty(const Digest& l, const Digest& r) {
    uint32_t st[16];
    memcpy(st, l.w, 32);
    memcpy(st + 8, r.w, 32);
    p2::permute(st, pw::poseidon2_params_host());
    Digest d;
    memcpy(d.w, st, 32);
    return d;
}

// run fn(i) for i in `order` on n_workers host threads, each with its own stream on the caller's device
template <class F>
int for_each_air(const std::vector<size_t>& order, unsigned n_workers, F fn) {
    std::atomic<size_t> next{0};
    std::atomic<int> first_error{0};
    auto body = [&] {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= order.size() || first_error.load()) return;
            const int rc = fn(order[k]);
            int expected = 0;
            if (rc) first_error.compare_exchange_strong(expected, rc);
        }
    };
    if (n_workers <= 1) {
        body();
        return first_error.load();
    }
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return (int)hipGetLastError();
    // Ordering contract (powdr_prover.h): everything the caller enqueued on ITS launch stream before this call — the
    // trace generation kernels of _apc_tracegen / _apc_apply_* in particular — happens before any worker touches a trace.
    // The worker streams are non-blocking (they do not synchronise with the null stream implicitly), so the dependency is
    // an event recorded on the caller's stream that every worker stream waits for.
    hipEvent_t ready = nullptr;
    if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
    if (hipEventRecord(ready, pw::stream()) != hipSuccess) { (void)hipEventDestroy(ready); return (int)hipGetLastError(); }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n_workers; ++t)
        th.emplace_back([&, device] {
            hipStream_t s = nullptr;
            if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
                hipStreamWaitEvent(s, ready, 0) != hipSuccess) {
                int expected = 0;
                first_error.compare_exchange_strong(expected, (int)hipErrorInvalidDevice);
                if (s) (void)hipStreamDestroy(s);
                return;
            }
            pw::set_stream(s);
            body();
            (void)hipStreamSynchronize(s);
            pw::set_stream(nullptr);
            (void)hipStreamDestroy(s);
        });
    for (auto& t : th) t.join();
    (void)hipEventDestroy(ready);
    return first_error.load();
}

}  // namespace

// One digest over an ordered list of 8-word commitments: binary Poseidon2 tree, an odd node is paired with zeros;
// the digest of a single commitment is the commitment, of none zeros. Canonical words in and out.
extern "C" void pw_commitment_digest(const uint32_t* roots8, size_t n, uint32_t* digest8) {
    std::vector<Digest> level(n);
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 8; ++k) level[i].w[k] = bb::to_monty(roots8[8 * i + k] % bb::P);
    if (level.empty()) {
        memset(digest8, 0, 32);
        return;
    }
    const Digest zero{};
    while (level.size() > 1) {
        std::vector<Digest> up;
        for (size_t i = 0; i < level.size(); i += 2) up.push_back(compress_monty(level[i], i + 1 < level.size() ? level[i + 1] : zero));
        level.swap(up);
    }
    for (int k = 0; k < 8; ++k) digest8[k] = bb::from_monty(level[0].w[k]);
}

extern "C" int pw_prove_airs(const PwSegmentAir* airs, size_t n_airs, int shared_bus_seed, unsigned n_workers,
                                const uint32_t** proofs, size_t* n_words, uint32_t* bus_seed8) {
    if (!airs || !proofs || !n_words) return (int)hipErrorInvalidValue;
    for (size_t i = 0; i < n_airs; ++i)
        if (!airs[i].prover || !airs[i].d_trace) return (int)hipErrorInvalidValue;
    if (n_workers == 0) n_workers = 4;
    if (n_workers > n_airs) n_workers = (unsigned)n_airs;
    // largest (by cells) first: the tail of the schedule is filled with the small proofs
    std::vector<size_t> order(n_airs);
    std::iota(order.begin(), order.end(), (size_t)0);
    auto cells = [&](size_t i) { return (uint64_t)pw_prover_width(airs[i].prover) << airs[i].log_height; };
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cells(a) > cells(b); });
    uint32_t seed[8] = {0};
    if (shared_bus_seed) {
        // phase 1: all trace commitments (each prover keeps its LDE and tree for phase 2), then the seed binds them all
        std::vector<uint32_t> roots(8 * n_airs);
        int rc = for_each_air(order, n_workers, [&](size_t i) {
            return pw_prover_trace_root(airs[i].prover, airs[i].d_trace, airs[i].log_height, &roots[8 * i]);
        });
        if (rc) return rc;
        pw_commitment_digest(roots.data(), n_airs, seed);
    }
    if (bus_seed8) memcpy(bus_seed8, seed, 32);
    return for_each_air(order, n_workers, [&](size_t i) {
        if (shared_bus_seed) {
            const int rc = pw_prover_set_bus_seed(airs[i].prover, seed);
            if (rc) return rc;
        }
        return pw_prover_prove(airs[i].prover, airs[i].d_trace, airs[i].log_height, &proofs[i], &n_words[i]);
    });
}

extern "C" int pw_verify_airs(const PwStarkConfig* cfg, const PwAirDescription* airs, size_t n_airs, const uint32_t* const* proofs,
                                 const size_t* n_words, int shared_bus_seed, int check_balance, uint32_t* total_sum4) {
    if (!cfg || !airs || !proofs || !n_words) return 10;
    uint32_t seed[8];
    if (shared_bus_seed) {
        // the seed every proof must have used: the digest of the trace roots the proofs themselves carry
        std::vector<uint32_t> roots(8 * n_airs);
        for (size_t i = 0; i < n_airs; ++i) {
            if (n_words[i] < 15) return (int)((i + 1) << 8) | 10;
            memcpy(&roots[8 * i], proofs[i] + 7, 32);  // "PWS2" header is 7 words, then the trace root
        }
        pw_commitment_digest(roots.data(), n_airs, seed);
    }
    uint64_t total[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n_airs; ++i) {
        const PwAirDescription& a = airs[i];
        uint32_t S[4] = {0, 0, 0, 0};
        int rc;
        if (shared_bus_seed || a.logup)
            rc = pw_verify_logup(cfg, a.width, a.log_height, a.cons_bytecode, a.bytecode_len, a.cons_spans, a.n_constraints,
                                 a.interactions, a.n_interactions, a.inter_spans, a.n_inter_spans, a.inter_bytecode,
                                 a.inter_bytecode_len, shared_bus_seed ? seed : nullptr, proofs[i], n_words[i], S, nullptr);
        else
            rc = pw_verify(cfg, a.width, a.log_height, a.cons_bytecode, a.bytecode_len, a.cons_spans, a.n_constraints, proofs[i],
                           n_words[i]);
        if (rc) return (int)((i + 1) << 8) | rc;
        for (int k = 0; k < 4; ++k) total[k] = (total[k] + S[k]) % bb::P;
    }
    if (total_sum4) for (int k = 0; k < 4; ++k) total_sum4[k] = (uint32_t)total[k];
    if (check_balance && (total[0] | total[1] | total[2] | total[3])) return 14;
    return 0;
}
