// Segment-level sharding over the GPUs of one node BEHIND THE C ABI (SURVEY.md §8e level 1).
//
// The reference proves the segments of an execution one after the other on one device
// (/root/reference/openvm/src/trace_generation.rs:111-141: `for (seg_idx, segment) in segments { ... vm.generate_proving_ctx;
// callback(seg_idx, vm, pk, ctx) }`). Once metered execution has fixed the boundaries (:107-109) the per-segment proofs are
// independent, so a host with N GPUs runs N of these loops side by side: one host thread per worker (hipSetDevice, a launch
// stream of its own — the library's state is per device / per host thread), segments placed on the workers by cells, largest
// first, NO data-path collective. The only exchange is north_star's "final commitment merge": the 8-word main commitment of
// every segment, all-gathered over RCCL (xGMI) so that every device — and the host — holds the full, segment-ordered list (one communicator
// per device set, made at first use and kept).
//
// RCCL is loaded with dlopen (the library keeps working without it: the merge then happens on the host and
// pw_multi_last_merge() says so). Two workers may share a device (tests on a one-GPU box; two streams per GPU): the
// communicator has one rank per DISTINCT device and a device's workers pool their commitments before the collective.
#include "../common.hpp"
#include "../../../include/powdr_prover.h"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef struct ncclComm* ncclComm_t;
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int /*ncclDataType_t*/, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    bool ok = false;
};
constexpr int kNcclUint32 = 3;  // ncclUint32 (nccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3)

const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char* e = getenv("POWDR_MULTI_NO_RCCL")) if (atoi(e) != 0) return;
        std::vector<std::string> names;
        if (const char* e = getenv("POWDR_RCCL_LIB")) names.push_back(e);
        for (const char* n : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) names.push_back(n);
        for (auto& n : names) {
            r.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        auto sym = [&](const char* s) { return dlsym(r.handle, s); };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.ok = r.CommInitAll && r.AllGather && r.GroupStart && r.GroupEnd && r.CommDestroy;
    });
    return r;
}

struct MergeContext {
    std::vector<int> devices;  // the distinct devices, in rank order
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> sts;
    std::vector<uint32_t*> d_send, d_recv;
    size_t slot_words = 0;     // capacity of d_send[r]; d_recv[r] holds ranks x that
};
std::mutex g_merge_mu;
std::vector<std::unique_ptr<MergeContext>> g_merge;  // never torn down: communicators live until the process exits

thread_local int g_last_merge = 0;  // 0 none yet, 1 RCCL all-gather, 2 host merge (RCCL not available / disabled / failed)

// one record per segment a rank proved: {segment index, 8 commitment words}; fixed-size contribution per rank
constexpr size_t kRecordWords = 9;

}  // namespace

extern "C" size_t pw_assign_units(const uint64_t* cells, size_t n_units, size_t n_workers, uint32_t* worker_of_unit) {
    if (!n_workers || (n_units && (!cells || !worker_of_unit))) return 0;
    std::vector<size_t> order(n_units);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cells[a] > cells[b]; });  // ties: lower index first
    std::vector<uint64_t> load(n_workers, 0);
    for (size_t u : order) {
        size_t w = 0;
        for (size_t k = 1; k < n_workers; ++k) if (load[k] < load[w]) w = k;  // ties: lower worker first
        worker_of_unit[u] = (uint32_t)w;
        load[w] += cells[u];
    }
    return n_units;
}

extern "C" int pw_multi_last_merge(void) { return g_last_merge; }

extern "C" int pw_prove_segments_multi(const int* devices, size_t n_workers, const uint64_t* segment_cells, size_t n_segments,
                                       PwSegmentProveFn prove, void* user, uint32_t* commitments, uint32_t* worker_of_segment) {
    if (!devices || !n_workers || !prove || (n_segments && (!segment_cells || !commitments))) return (int)hipErrorInvalidValue;
    int n_dev_present = 0;
    if (hipGetDeviceCount(&n_dev_present) != hipSuccess) return (int)hipGetLastError();
    for (size_t w = 0; w < n_workers; ++w) if (devices[w] < 0 || devices[w] >= n_dev_present) return (int)hipErrorInvalidDevice;
    std::vector<uint32_t> owner(n_segments);
    pw_assign_units(segment_cells, n_segments, n_workers, owner.data());
    if (n_segments) memset(commitments, 0, n_segments * 8 * 4);
    // The placement by cells is the PLAN: every worker's queue, largest segment first. A worker that runs dry STEALS — the smallest
    // unstarted segment of the worker with the most cells still queued — because proving time is not proportional to cells (a segment's
    // short tail proves at half the rate of a full one, a segment that has to stream an AIR takes 6-20 % longer: profiles/r06_*), and a
    // node whose devices differ by a few percent should not wait for its slowest. POWDR_MULTI_STEAL=0: the plan is kept as it is.
    const char* steal_env = getenv("POWDR_MULTI_STEAL");
    const bool steal = !(steal_env && atoi(steal_env) == 0);
    std::vector<std::vector<uint32_t>> queue(n_workers);
    for (size_t s = 0; s < n_segments; ++s) queue[owner[s]].push_back((uint32_t)s);
    for (auto& q : queue) std::stable_sort(q.begin(), q.end(), [&](uint32_t a, uint32_t b) { return segment_cells[a] > segment_cells[b]; });
    std::vector<size_t> next(n_workers, 0);  // queue[w][next[w] ..] are unstarted
    std::mutex queue_mu;
    auto take = [&](size_t w, uint32_t* seg) -> bool {
        std::lock_guard<std::mutex> lk(queue_mu);
        if (next[w] < queue[w].size()) { *seg = queue[w][next[w]++]; return true; }
        if (!steal) return false;
        size_t victim = n_workers;
        uint64_t most = 0;
        for (size_t v = 0; v < n_workers; ++v) {
            uint64_t left = 0;
            for (size_t k = next[v]; k < queue[v].size(); ++k) left += segment_cells[queue[v][k]];
            if (left > most) { most = left; victim = v; }
        }
        if (victim == n_workers) return false;
        *seg = queue[victim].back();
        queue[victim].pop_back();
        owner[*seg] = (uint32_t)w;
        return true;
    };
    int caller_device = 0;
    (void)hipGetDevice(&caller_device);

    // ---- the N segment loops ---------------------------------------------------------------------------------
    std::vector<std::vector<uint32_t>> records(n_workers);  // per worker: kRecordWords per proved segment
    std::atomic<int> first_error{0};
    auto worker = [&](size_t w) {
        int rc = (int)hipSetDevice(devices[w]);
        hipStream_t st = nullptr;
        if (!rc) rc = (int)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (!rc) {
            const hipStream_t callers = pw::stream();  // worker 0 runs on the caller's thread: its launch stream comes back afterwards
            pw::set_stream(st);
            uint32_t s = 0;
            while (!first_error.load() && take(w, &s)) {
                uint32_t c[8] = {0};
                rc = prove(user, s, w, devices[w], c);
                if (rc) break;
                records[w].push_back(s);
                records[w].insert(records[w].end(), c, c + 8);
            }
            const int rs = (int)hipStreamSynchronize(st);
            if (!rc) rc = rs;
            pw::set_stream(callers);
            (void)hipStreamDestroy(st);
        }
        int expected = 0;
        if (rc) first_error.compare_exchange_strong(expected, rc);
    };
    {
        std::vector<std::thread> th;
        for (size_t w = 1; w < n_workers; ++w) th.emplace_back(worker, w);
        worker(0);
        for (auto& t : th) t.join();
    }
    (void)hipSetDevice(caller_device);
    if (worker_of_segment && n_segments) memcpy(worker_of_segment, owner.data(), n_segments * 4);  // who proved it in the end (stolen segments included)
    if (first_error.load()) return first_error.load();

    // ---- the final commitment merge: all-gather over the distinct devices ------------------------------------------
    std::vector<int> ranks;  // distinct devices in order of first appearance
    std::vector<size_t> rank_of_worker(n_workers);
    for (size_t w = 0; w < n_workers; ++w) {
        auto it = std::find(ranks.begin(), ranks.end(), devices[w]);
        rank_of_worker[w] = (size_t)(it - ranks.begin());
        if (it == ranks.end()) ranks.push_back(devices[w]);
    }
    const size_t R = ranks.size();
    std::vector<std::vector<uint32_t>> send(R);
    for (size_t w = 0; w < n_workers; ++w) send[rank_of_worker[w]].insert(send[rank_of_worker[w]].end(), records[w].begin(), records[w].end());
    size_t slot = 1;  // words per rank: [number of records, records...], padded to the longest
    for (auto& v : send) slot = std::max(slot, 1 + v.size());
    for (auto& v : send) { v.insert(v.begin(), (uint32_t)(v.size() / kRecordWords)); v.resize(slot, 0u); }
    std::vector<uint32_t> gathered(R * slot, 0u);
    bool merged = false;
    const Rccl& nc = rccl();
    if (nc.ok) {
        // the communicator of a device set, its streams and staging buffers are made once and kept for the life of the process
        // (ncclCommInitAll costs ~0.5 s: a per-call communicator made the in-process C4 step 12 % slower than its merge-free twin)
        std::lock_guard<std::mutex> lk(g_merge_mu);
        MergeContext* mc = nullptr;
        for (auto& c : g_merge) if (c->devices == ranks) mc = c.get();
        bool ok = true;
        if (!mc) {
            std::unique_ptr<MergeContext> c(new MergeContext());
            c->devices = ranks;
            c->comms.assign(R, nullptr); c->sts.assign(R, nullptr); c->d_send.assign(R, nullptr); c->d_recv.assign(R, nullptr);
            ok = nc.CommInitAll(c->comms.data(), (int)R, ranks.data()) == 0;
            for (size_t r = 0; r < R && ok; ++r)
                ok = hipSetDevice(ranks[r]) == hipSuccess && hipStreamCreateWithFlags(&c->sts[r], hipStreamNonBlocking) == hipSuccess;
            if (ok) { g_merge.push_back(std::move(c)); mc = g_merge.back().get(); }
            // (a half-built context is dropped without ncclCommDestroy: destroying communicators after a failed init can hang)
        }
        if (ok && mc->slot_words < slot) {
            for (size_t r = 0; r < R && ok; ++r) {
                ok = hipSetDevice(ranks[r]) == hipSuccess;
                if (ok && mc->d_send[r]) (void)hipFree(mc->d_send[r]);
                if (ok && mc->d_recv[r]) (void)hipFree(mc->d_recv[r]);
                mc->d_send[r] = mc->d_recv[r] = nullptr;
                ok = ok && hipMalloc(&mc->d_send[r], slot * 4) == hipSuccess && hipMalloc(&mc->d_recv[r], R * slot * 4) == hipSuccess;
            }
            mc->slot_words = ok ? slot : 0;
        }
        for (size_t r = 0; r < R && ok; ++r)
            ok = hipSetDevice(ranks[r]) == hipSuccess &&
                 hipMemcpyAsync(mc->d_send[r], send[r].data(), slot * 4, hipMemcpyHostToDevice, mc->sts[r]) == hipSuccess;
        if (ok) {
            ok = nc.GroupStart() == 0;
            for (size_t r = 0; r < R && ok; ++r) ok = nc.AllGather(mc->d_send[r], mc->d_recv[r], slot, kNcclUint32, mc->comms[r], mc->sts[r]) == 0;
            ok = nc.GroupEnd() == 0 && ok;
        }
        for (size_t r = 0; r < R && ok; ++r) ok = hipSetDevice(ranks[r]) == hipSuccess && hipStreamSynchronize(mc->sts[r]) == hipSuccess;
        // every rank holds the same table; the host reads rank 0's copy (and checks the last rank's against it)
        if (ok) ok = hipSetDevice(ranks[0]) == hipSuccess && hipMemcpy(gathered.data(), mc->d_recv[0], R * slot * 4, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok && R > 1) {
            std::vector<uint32_t> other(R * slot);
            ok = hipSetDevice(ranks[R - 1]) == hipSuccess && hipMemcpy(other.data(), mc->d_recv[R - 1], R * slot * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                 other == gathered;
        }
        (void)hipGetLastError();
        merged = ok;
    }
    (void)hipSetDevice(caller_device);
    if (!merged) for (size_t r = 0; r < R; ++r) memcpy(gathered.data() + r * slot, send[r].data(), slot * 4);  // host merge
    g_last_merge = merged ? 1 : 2;
    size_t seen = 0;
    for (size_t r = 0; r < R; ++r) {
        const uint32_t* g = gathered.data() + r * slot;
        const uint32_t n = g[0];
        if (1 + (size_t)n * kRecordWords > slot) return (int)hipErrorUnknown;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t* rec = g + 1 + (size_t)k * kRecordWords;
            if (rec[0] >= n_segments) return (int)hipErrorUnknown;
            memcpy(commitments + (size_t)rec[0] * 8, rec + 1, 32);
            ++seen;
        }
    }
    return seen == n_segments ? 0 : (int)hipErrorUnknown;
}
