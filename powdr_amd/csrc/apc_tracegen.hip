// _apc_tracegen for gfx950: the APC-trace gather, redesigned around the memory
// system of MI355X instead of the reference's one-thread-per-row loop
// (/root/reference/openvm/cuda/src/apc_tracegen.cu:35-66).
//
// Semantics (identical to the reference, apc_tracegen.cu:43-64):
//   for every Subst s and every r < H:
//     out[s.apc_col*H + r] = r < num_calls
//         ? air[s.air_index].buffer[s.col*air.height + s.row + r*air.row_block_size]
//         : 0
// Duplicate apc_col targets resolve like the reference's sequential loop: the
// substitution with the highest index wins.
//
// Why a redesign: in the reference a wave's 64 lanes (= 64 consecutive APC
// calls) read one source cell each at a stride of row_block_size*4 bytes
// (1 272 B for the keccak BaseAlu block), i.e. one 128-B line per 4 useful
// bytes, and every thread re-reads the 16-B Subst + 24-B OriginalAir records
// for each cell. Here the substitutions are regrouped by SOURCE COLUMN
// (air, col). All cells of one source column that feed APC calls
// r0..r0+R are the contiguous range  col*height + [r0*b, (r0+R)*b)  of the
// column-major dummy trace, so a workgroup streams that range with 16-byte
// coalesced loads into an LDS tile (row pitch padded to an odd number of words
// -> conflict-free transposed reads) and then writes each substituted APC
// column as full 128/256-byte coalesced segments. Sparse source columns are cut
// into row chunks so that unused stretches of a block are never fetched.
//
// The regrouping ("plan") is computed on the host from a D2H copy of the
// Subst/OriginalAir tables (a few hundred KB at most) and memoised by content
// hash: the same APC is replayed for every segment.
#include "babybear.hpp"
#include "common.hpp"
#include "../../include/powdr_gpu.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

// One unit of work: rows [j0, j0+J) of the row block of source column (air, col),
// for a tile of R consecutive APC calls.
struct GatherJob {
    int32_t air;        // index into d_original_airs
    int32_t col;        // source column
    int32_t b;          // row_block_size of the air
    int32_t j0;         // first block row covered
    int32_t J;          // number of block rows covered
    int32_t pitch;      // LDS row pitch in words (odd)
    uint32_t magicJ;    // ceil(2^32 / J) for e / J
    uint32_t sub_begin; // into plan subs
    uint32_t sub_count;
    int32_t sparse;     // 1: only the used rows are fetched, cell by cell (J = their number, tile row c = [row s of call c])
    int32_t pad;
};

struct PlanSub {
    int32_t j;        // block row (absolute, s.row)
    int32_t apc_col;
};

struct RClass {
    int R;              // calls per tile
    uint32_t job_begin; // into jobs array
    uint32_t job_count;
    size_t lds_bytes;   // max over jobs
};

struct Plan {
    GatherJob* d_jobs = nullptr;
    PlanSub* d_subs = nullptr;
    std::vector<RClass> classes;
    size_t n_jobs = 0, n_subs = 0;
    uint32_t n_sparse = 0, n_whole = 0, n_chunk = 0;  // jobs by form (powdr_gpu_call_stats)
    // what the plan was built from: a cache hit is confirmed by comparing contents, never by the hash alone
    std::vector<Subst> key_subs;
    std::vector<int32_t> key_bsize;
    int device = 0;           // the device d_jobs / d_subs live on
    uint64_t last_use = 0;    // for eviction
    ~Plan() {
        if (d_jobs) (void)hipFree(d_jobs);  // hipFree waits for kernels that may still read the tables
        if (d_subs) (void)hipFree(d_subs);
    }
    Plan() = default;
    Plan(const Plan&) = delete;
    Plan& operator=(const Plan&) = delete;
};

constexpr int kBlock = 256;
constexpr int kMaxR = 1024;
// The planner's tuning knobs, read from the environment on every call (unset = default) and part of the plan-cache key, so a
// plan built under other knobs is never reused.
struct Knobs {
    int max_tile_words = 5 * 1024;  // <= 20 KB LDS tiles -> 8 workgroups / CU: 24.9 ms vs 35.5 ms with 48 KB tiles at C2 (POWDR_GATHER_TILE_WORDS)
    int min_r = 16;                 // POWDR_GATHER_MIN_R
    int sparse = 1;                 // POWDR_GATHER_SPARSE=0: never use the cell-by-cell form
    int sparse_pct = 100;           // ... which must move at most this share of the streaming forms' bytes (POWDR_GATHER_SPARSE_PCT)
    int max_chunk_j() const { return max_tile_words / min_r - 1; }  // 319
    bool operator==(const Knobs& o) const { return max_tile_words == o.max_tile_words && min_r == o.min_r && sparse == o.sparse && sparse_pct == o.sparse_pct; }
    static Knobs from_env() {
        Knobs k;
        if (const char* e = getenv("POWDR_GATHER_TILE_WORDS")) { const int v = atoi(e); if (v >= 2048 && v <= 16384) k.max_tile_words = v; }
        if (const char* e = getenv("POWDR_GATHER_SPARSE")) k.sparse = atoi(e) != 0;
        if (const char* e = getenv("POWDR_GATHER_SPARSE_PCT")) { const int v = atoi(e); if (v >= 10 && v <= 200) k.sparse_pct = v; }
        if (const char* e = getenv("POWDR_GATHER_MIN_R")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) k.min_r = v; }
        return k;
    }
};

// OriginalAir records handed to the kernels BY VALUE (kernel arguments) when the caller has them on the host and there are at
// most this many: no device table to allocate, upload or keep alive across the launch.
constexpr int kInlineAirs = 16;
struct InlineAirs { OriginalAir a[kInlineAirs]; };

__device__ __forceinline__ uint32_t fast_div(uint32_t e, uint32_t magic, uint32_t /*J*/) {
    // floor(e / J) with magic = ceil(2^32 / J): exact for e < 2^16, J < 2^16; J == 1 is magic == 0
    return magic ? __umulhi(e, magic) : e;
}

template <int R>
__global__ __launch_bounds__(kBlock) void apc_gather_tile_kernel(
    uint32_t* __restrict__ out, size_t H, const OriginalAir* __restrict__ airs,
    const GatherJob* __restrict__ jobs, const PlanSub* __restrict__ subs, int num_calls, InlineAirs inl, int use_inl) {
    extern __shared__ uint32_t tile[];
    const GatherJob job = jobs[blockIdx.y];
    const size_t r0 = (size_t)blockIdx.x * R;
    const int tid = threadIdx.x;

    // number of calls of this tile that carry data
    int valid = 0;
    if (r0 < (size_t)num_calls) {
        size_t v = (size_t)num_calls - r0;
        valid = v < (size_t)R ? (int)v : R;
    }

    if (valid > 0) {
        const OriginalAir air = use_inl ? inl.a[job.air] : airs[job.air];
        const uint32_t* __restrict__ src =
            air.buffer + (size_t)job.col * (size_t)(uint32_t)air.height;
        const int J = job.J, b = job.b, pitch = job.pitch;
        if (job.sparse) {
            // Scattered survivors: fetch ONLY the used cells. Lane e -> (call e / J, used row e % J): the lanes of one call run
            // through its used rows in ascending order, so cells that share a 64-byte sector are fetched by one request and
            // the sectors that hold no used cell are never read (at 7 % density ~30 % of them). The rows come from the job's
            // substitution list, staged in LDS behind the tile.
            uint32_t* rows_lds = tile + (size_t)R * pitch;
            const PlanSub* __restrict__ js0 = subs + job.sub_begin;
            for (uint32_t s = tid; s < (uint32_t)J; s += kBlock) rows_lds[s] = (uint32_t)js0[s].j;
            __syncthreads();
            const uint32_t* base = src + r0 * (size_t)b;
            const uint32_t n = (uint32_t)valid * (uint32_t)J;
            for (uint32_t e = tid; e < n; e += kBlock) {
                const uint32_t i = fast_div(e, job.magicJ, J);
                const uint32_t sidx = e - i * (uint32_t)J;
                tile[i * pitch + sidx] = __builtin_nontemporal_load(base + (size_t)i * b + rows_lds[sidx]);
            }
        } else if (J == b) {
            // J == b: the per-call segments [j0, j0+b) abut, so the tile is one contiguous range of the column
            // (j0 != 0 happens when substitutions reach past their own block into the next call's rows)
            const uint32_t* base = src + r0 * (size_t)b + job.j0;
            const uint32_t n = (uint32_t)valid * (uint32_t)b;
            if (pitch == J) {
                // linear copy: tile[e] = base[e]
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                if ((((uintptr_t)base) & 15u) == 0) {
                    const u32x4* base4 = reinterpret_cast<const u32x4*>(base);
                    u32x4* tile4 = reinterpret_cast<u32x4*>(tile);
                    const uint32_t n4 = n >> 2;
                    uint32_t q = tid;
                    for (; q + 3 * kBlock < n4; q += 4 * kBlock) {  // four independent 16-byte loads in flight per lane
                        u32x4 v0 = __builtin_nontemporal_load(base4 + q);
                        u32x4 v1 = __builtin_nontemporal_load(base4 + q + kBlock);
                        u32x4 v2 = __builtin_nontemporal_load(base4 + q + 2 * kBlock);
                        u32x4 v3 = __builtin_nontemporal_load(base4 + q + 3 * kBlock);
                        tile4[q] = v0; tile4[q + kBlock] = v1; tile4[q + 2 * kBlock] = v2; tile4[q + 3 * kBlock] = v3;
                    }
                    for (; q < n4; q += kBlock) tile4[q] = __builtin_nontemporal_load(base4 + q);
                    for (uint32_t e = (n4 << 2) + tid; e < n; e += kBlock) tile[e] = base[e];
                } else {
                    for (uint32_t e = tid; e < n; e += kBlock) tile[e] = __builtin_nontemporal_load(base + e);
                }
            } else if ((((uintptr_t)base) & 15u) == 0) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4* base4 = reinterpret_cast<const u32x4*>(base);
                const uint32_t n4 = n >> 2;
                // four independent 16-byte loads in flight per lane before any LDS store
                auto scatter = [&](uint32_t q, u32x4 v) {
                    uint32_t e = q << 2;
                    uint32_t i = fast_div(e, job.magicJ, J);
                    uint32_t j = e - i * (uint32_t)J;
                    uint32_t vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        tile[i * pitch + j] = vals[k];
                        if (++j == (uint32_t)J) { j = 0; ++i; }
                    }
                };
                uint32_t q = tid;
                for (; q + 3 * kBlock < n4; q += 4 * kBlock) {
                    u32x4 v0 = __builtin_nontemporal_load(base4 + q);
                    u32x4 v1 = __builtin_nontemporal_load(base4 + q + kBlock);
                    u32x4 v2 = __builtin_nontemporal_load(base4 + q + 2 * kBlock);
                    u32x4 v3 = __builtin_nontemporal_load(base4 + q + 3 * kBlock);
                    scatter(q, v0); scatter(q + kBlock, v1); scatter(q + 2 * kBlock, v2); scatter(q + 3 * kBlock, v3);
                }
                for (; q < n4; q += kBlock) scatter(q, __builtin_nontemporal_load(base4 + q));
                for (uint32_t e = (n4 << 2) + tid; e < n; e += kBlock) {
                    uint32_t i = fast_div(e, job.magicJ, J);
                    uint32_t j = e - i * (uint32_t)J;
                    tile[i * pitch + j] = base[e];
                }
            } else {
                for (uint32_t e = tid; e < n; e += kBlock) {
                    uint32_t i = fast_div(e, job.magicJ, J);
                    uint32_t j = e - i * (uint32_t)J;
                    tile[i * pitch + j] = __builtin_nontemporal_load(base + e);
                }
            }
        } else {
            // a chunk of each block: `valid` segments of J words, b words apart
            const uint32_t* base = src + r0 * (size_t)b + job.j0;
            const uint32_t n = (uint32_t)valid * (uint32_t)J;
            for (uint32_t e = tid; e < n; e += kBlock) {
                uint32_t i = fast_div(e, job.magicJ, J);
                uint32_t j = e - i * (uint32_t)J;
                tile[i * pitch + j] = __builtin_nontemporal_load(base + (size_t)i * b + j);
            }
        }
    }
    __syncthreads();

    // Transposed write-out: lanes run over calls, (sub-)waves over substitutions.
    const size_t rows_left = H - r0;  // H may be smaller than the tile
    const int rows = rows_left < (size_t)R ? (int)rows_left : R;
    constexpr int kLanes = R < 64 ? R : 64;       // lanes per substitution per pass
    constexpr int kSubsPerPass = kBlock / kLanes; // substitutions handled at once
    const int lane_i = tid % kLanes;
    const int slot = tid / kLanes;
    const PlanSub* __restrict__ js = subs + job.sub_begin;
    for (uint32_t s = slot; s < job.sub_count; s += kSubsPerPass) {
        const PlanSub ps = js[s];
        uint32_t* __restrict__ dst = out + (size_t)ps.apc_col * H + r0;
        const int j = job.sparse ? (int)s : ps.j - job.j0;
#pragma unroll
        for (int i = lane_i; i < R; i += kLanes) {
            if (i < rows) {
                uint32_t v = (i < valid) ? tile[i * job.pitch + j] : 0u;
                __builtin_nontemporal_store(v, dst + i);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host: plan construction + cache
// ---------------------------------------------------------------------------------------------

uint64_t fnv1a(const void* p, size_t n, uint64_t h) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}

struct PlanKey {
    uint64_t hash;
    size_t n_subs;
    int device;
    Knobs knobs;
    bool operator==(const PlanKey& o) const { return hash == o.hash && n_subs == o.n_subs && device == o.device && knobs == o.knobs; }
};
struct PlanKeyHash {
    size_t operator()(const PlanKey& k) const {
        return (size_t)(k.hash ^ ((uint64_t)k.device << 56) ^ ((uint64_t)k.knobs.max_tile_words << 20) ^ ((uint64_t)k.knobs.min_r << 8) ^
                        ((uint64_t)k.knobs.sparse << 40) ^ ((uint64_t)k.knobs.sparse_pct << 44));
    }
};

// Plans are shared_ptrs: a launch keeps its plan alive while another host thread evicts it from the (bounded) cache.
constexpr size_t kMaxCachedPlans = 64;
std::mutex g_plan_mu;
std::unordered_map<PlanKey, std::shared_ptr<Plan>, PlanKeyHash> g_plans;
uint64_t g_plan_clock = 0;

int pick_R(int J, const Knobs& kn) {
    int pitch = J | 1;
    int R = kMaxR;
    while (R > kn.min_r && (size_t)R * pitch > (size_t)kn.max_tile_words) R >>= 1;
    return R;
}

int build_plan(const std::vector<Subst>& subs_in, const std::vector<int32_t>& bsize, const Knobs& kn, Plan& plan) {
    const int max_chunk_j = kn.max_chunk_j();
    const bool sparse_jobs = kn.sparse != 0;
    const int sparse_pct = kn.sparse_pct;
    const size_t n = subs_in.size();
    // 1. resolve duplicate destinations like the sequential reference loop: last wins
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    std::vector<char> keep(n, 1);
    {
        std::unordered_map<int32_t, uint32_t> last;
        last.reserve(n * 2);
        for (size_t i = 0; i < n; ++i) {
            auto it = last.find(subs_in[i].apc_col);
            if (it != last.end()) keep[it->second] = 0;
            last[subs_in[i].apc_col] = (uint32_t)i;
        }
    }
    // 2. sort by (air, col, row)
    std::vector<uint32_t> idx;
    idx.reserve(n);
    for (size_t i = 0; i < n; ++i) if (keep[i]) idx.push_back((uint32_t)i);
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t c) {
        const Subst& x = subs_in[a]; const Subst& y = subs_in[c];
        if (x.air_index != y.air_index) return x.air_index < y.air_index;
        if (x.col != y.col) return x.col < y.col;
        if (x.row != y.row) return x.row < y.row;
        return a < c;
    });
    // 3. cut every (air, col) group into jobs
    std::vector<GatherJob> jobs;
    std::vector<PlanSub> psubs;
    psubs.reserve(idx.size());
    std::vector<int> jobR;
    size_t g = 0;
    while (g < idx.size()) {
        size_t ge = g;
        const Subst& s0 = subs_in[idx[g]];
        while (ge < idx.size() && subs_in[idx[ge]].air_index == s0.air_index &&
               subs_in[idx[ge]].col == s0.col) ++ge;
        const int b = bsize[s0.air_index];
        // candidate A: one job covering whole blocks (contiguous stream, no edge waste)
        // candidate B: chunks around the used rows. Cost model in bytes per call.
        std::vector<std::pair<size_t, size_t>> chunks;  // [begin,end) into idx
        {
            size_t c = g;
            while (c < ge) {
                size_t ce = c + 1;
                int jstart = subs_in[idx[c]].row;
                while (ce < ge && subs_in[idx[ce]].row - jstart < max_chunk_j &&
                       // do not bridge gaps that cost more than a fresh segment (128 B edge)
                       subs_in[idx[ce]].row - subs_in[idx[ce - 1]].row <= 48)
                    ++ce;
                chunks.push_back({c, ce});
                c = ce;
            }
        }
        size_t cost_chunks = 0;
        for (auto& ch : chunks) {
            int J = subs_in[idx[ch.second - 1]].row - subs_in[idx[ch.first]].row + 1;
            cost_chunks += (size_t)J * 4 + 128;
        }
        const int max_row = subs_in[idx[ge - 1]].row;
        const bool whole_ok = b >= 1 && b <= max_chunk_j && max_row < b;
        const size_t cost_whole = (size_t)b * 4;
        // candidate C: fetch the used cells one by one — bytes per call = 64-byte sectors that hold a used row (+ one for the
        // alignment of a call's block, which shifts from call to call)
        const size_t U = ge - g;
        size_t cost_sparse = 64;
        for (size_t k = g; k < ge; ++k)
            if (k == g || subs_in[idx[k]].row / 16 != subs_in[idx[k - 1]].row / 16) cost_sparse += 64;
        const bool sparse_ok = sparse_jobs && U <= (size_t)max_chunk_j && U >= 1;
        auto emit = [&](size_t cb, size_t ce, int j0, int J, bool sparse = false) {
            GatherJob job;
            job.sparse = sparse ? 1 : 0; job.pad = 0;
            job.air = s0.air_index; job.col = s0.col; job.b = b; job.j0 = j0; job.J = J;
            // whole-block jobs (J == b) copy their contiguous range into the tile as it is (pitch = J: 16-byte LDS stores,
            // no per-element index arithmetic — the copy, not the transposed read-out of the few cells that are used,
            // is where the time goes); chunked jobs keep an odd pitch for conflict-free transposed reads
            job.pitch = J == b ? J : (J | 1);
            job.magicJ = J == 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)J - 1) / (uint64_t)J);
            job.sub_begin = (uint32_t)psubs.size();
            job.sub_count = (uint32_t)(ce - cb);
            for (size_t k = cb; k < ce; ++k)
                psubs.push_back({subs_in[idx[k]].row, subs_in[idx[k]].apc_col});
            jobs.push_back(job);
            jobR.push_back(pick_R(J, kn));
        };
        // the sparse form pays a request per cell instead of 16-byte streaming loads: it has to save a share of the bytes
        if (sparse_ok && cost_sparse * 100 <= std::min(whole_ok ? cost_whole : (size_t)-1, cost_chunks) * (size_t)sparse_pct) {
            emit(g, ge, 0, (int)U, true);
        } else if (whole_ok && cost_whole <= cost_chunks) {
            emit(g, ge, 0, b);
        } else {
            for (auto& ch : chunks) {
                int j0 = subs_in[idx[ch.first]].row;
                int J = subs_in[idx[ch.second - 1]].row - j0 + 1;
                emit(ch.first, ch.second, j0, J);
            }
        }
        g = ge;
    }
    // 4. order jobs by tile class R
    std::vector<uint32_t> jorder(jobs.size());
    for (size_t i = 0; i < jobs.size(); ++i) jorder[i] = (uint32_t)i;
    std::stable_sort(jorder.begin(), jorder.end(), [&](uint32_t a, uint32_t c) { return jobR[a] < jobR[c]; });
    std::vector<GatherJob> sorted(jobs.size());
    plan.classes.clear();
    for (size_t i = 0; i < jorder.size(); ++i) {
        sorted[i] = jobs[jorder[i]];
        if (sorted[i].sparse) ++plan.n_sparse;
        else if (sorted[i].J == sorted[i].b) ++plan.n_whole;
        else ++plan.n_chunk;
        int R = jobR[jorder[i]];
        size_t lds = (size_t)R * sorted[i].pitch * 4 + (sorted[i].sparse ? (size_t)sorted[i].J * 4 : 0);
        if (plan.classes.empty() || plan.classes.back().R != R)
            plan.classes.push_back({R, (uint32_t)i, 0, 0});
        plan.classes.back().job_count++;
        plan.classes.back().lds_bytes = std::max(plan.classes.back().lds_bytes, lds);
    }
    plan.n_jobs = sorted.size();
    plan.n_subs = psubs.size();
    if (plan.n_jobs) {
        PW_HIP_TRY(hipMalloc(&plan.d_jobs, sorted.size() * sizeof(GatherJob)));
        PW_HIP_TRY(hipMalloc(&plan.d_subs, psubs.size() * sizeof(PlanSub)));
        PW_HIP_TRY(hipMemcpy(plan.d_jobs, sorted.data(), sorted.size() * sizeof(GatherJob), hipMemcpyHostToDevice));
        PW_HIP_TRY(hipMemcpy(plan.d_subs, psubs.data(), psubs.size() * sizeof(PlanSub), hipMemcpyHostToDevice));
    }
    return 0;
}

template <int R>
void launch_class(const RClass& c, uint32_t* out, size_t H, const OriginalAir* airs,
                  const Plan& plan, int num_calls, const InlineAirs& inl, int use_inl) {
    unsigned tiles = pw::div_up(H, R);
    // gridDim.y is limited to 65535 jobs per launch
    for (uint32_t j = 0; j < c.job_count; j += 65535u) {
        uint32_t cnt = std::min<uint32_t>(65535u, c.job_count - j);
        dim3 grid(tiles, cnt, 1);
        hipLaunchKernelGGL(apc_gather_tile_kernel<R>, grid, dim3(kBlock), c.lds_bytes, pw::stream(),
                           out, H, airs, plan.d_jobs + c.job_begin + j, plan.d_subs, num_calls, inl, use_inl);
    }
}

}  // namespace

// The gather proper: `subs` / `bsize` are host copies of the Subst table and of the AIRs' row_block_size; the kernels read
// the OriginalAir records (buffer pointers, heights) from the device table.
static int tracegen_with_host_tables(PowdrFp* d_output, size_t H, const OriginalAir* d_original_airs,
                                     const std::vector<Subst>& subs, const std::vector<int32_t>& bsize, int num_apc_calls,
                                     const OriginalAir* h_airs = nullptr, size_t n_h_airs = 0) {
    const size_t n_subs = subs.size();
    InlineAirs inl{};
    const int use_inl = h_airs && n_h_airs <= (size_t)kInlineAirs ? 1 : 0;
    if (use_inl) memcpy(inl.a, h_airs, n_h_airs * sizeof(OriginalAir));
    if (!use_inl && !d_original_airs) return (int)hipErrorInvalidValue;
    const Knobs knobs = Knobs::from_env();
    uint64_t h = fnv1a(subs.data(), n_subs * sizeof(Subst), 1469598103934665603ull);
    h = fnv1a(bsize.data(), bsize.size() * sizeof(int32_t), h);
    int device = 0;
    PW_HIP_TRY(hipGetDevice(&device));
    const PlanKey key{h, n_subs, device, knobs};

    std::shared_ptr<Plan> plan;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto it = g_plans.find(key);
        if (it != g_plans.end()) {
            const Plan& c = *it->second;
            const bool same = c.key_subs.size() == n_subs && c.key_bsize == bsize &&
                              (n_subs == 0 || memcmp(c.key_subs.data(), subs.data(), n_subs * sizeof(Subst)) == 0);
            if (!same) { g_plans.erase(it); it = g_plans.end(); }  // 64-bit hash collision: the newer table takes the slot
        }
        if (it == g_plans.end()) {
            if (g_plans.size() >= kMaxCachedPlans) {  // evict the least recently used plan
                auto victim = g_plans.begin();
                for (auto j = g_plans.begin(); j != g_plans.end(); ++j) if (j->second->last_use < victim->second->last_use) victim = j;
                g_plans.erase(victim);
            }
            auto p = std::make_shared<Plan>();
            int rc = build_plan(subs, bsize, knobs, *p);
            if (rc) return rc;
            p->key_subs = subs;
            p->key_bsize = bsize;
            p->device = device;
            it = g_plans.emplace(key, std::move(p)).first;
        }
        plan = it->second;
        plan->last_use = ++g_plan_clock;
    }

    {
        uint64_t* st = pw::call_stats();
        st[pw::kStatGatherSparseJobs] += plan->n_sparse;
        st[pw::kStatGatherWholeJobs] += plan->n_whole;
        st[pw::kStatGatherChunkJobs] += plan->n_chunk;
        st[pw::kStatGatherCalls] += 1;
    }
    pw::ScopedKernelTimer t("apc_gather_tile_kernel");
    uint32_t* out = d_output;
    for (const RClass& c : plan->classes) {
        switch (c.R) {
            case 16: launch_class<16>(c, out, H, d_original_airs, *plan, num_apc_calls, inl, use_inl); break;
            case 32: launch_class<32>(c, out, H, d_original_airs, *plan, num_apc_calls, inl, use_inl); break;
            case 64: launch_class<64>(c, out, H, d_original_airs, *plan, num_apc_calls, inl, use_inl); break;
            case 128: launch_class<128>(c, out, H, d_original_airs, *plan, num_apc_calls, inl, use_inl); break;
            case 256: launch_class<256>(c, out, H, d_original_airs, *plan, num_apc_calls, inl, use_inl); break;
            case 512: launch_class<512>(c, out, H, d_original_airs, *plan, num_apc_calls, inl, use_inl); break;
            case 1024: launch_class<1024>(c, out, H, d_original_airs, *plan, num_apc_calls, inl, use_inl); break;
            default: return (int)hipErrorInvalidValue;
        }
    }
    return (int)hipGetLastError();
}

static int check_tables(const std::vector<Subst>& subs, size_t n_airs_known, int& max_air) {
    max_air = -1;
    for (auto& s : subs) {
        if (s.air_index < 0 || s.col < 0 || s.row < 0 || s.apc_col < 0) return (int)hipErrorInvalidValue;
        max_air = std::max(max_air, s.air_index);
    }
    if (n_airs_known && (size_t)max_air >= n_airs_known) return (int)hipErrorInvalidValue;
    return 0;
}
constexpr size_t kAirCountUnknown = 0;  // _apc_tracegen: the table's length is not part of the reference ABI

extern "C" int _apc_tracegen(PowdrFp* d_output, size_t output_height,
                             const OriginalAir* d_original_airs, const Subst* d_subs,
                             size_t n_subs, int num_apc_calls) {
    (void)hipGetLastError();  // do not report a stale error of an unrelated earlier call
    const size_t H = output_height;
    if ((H & (H - 1)) != 0) return (int)hipErrorInvalidValue;  // reference: assert, apc_tracegen.cu:134
    if (H == 0 || n_subs == 0) return (int)hipGetLastError();
    if (num_apc_calls < 0) num_apc_calls = 0;
    if ((size_t)num_apc_calls > H) num_apc_calls = (int)H;  // rows r >= H do not exist

    // The reference ABI hands over device tables only: D2H the (small) tables to look the plan up. Hosts that still hold
    // the tables (powdr_apc_generate_witness_gpu does) call powdr_apc_tracegen_host_tables and skip both round trips.
    std::vector<Subst> subs(n_subs);
    PW_HIP_TRY(hipMemcpyAsync(subs.data(), d_subs, n_subs * sizeof(Subst), hipMemcpyDeviceToHost, pw::stream()));
    PW_HIP_TRY(hipStreamSynchronize(pw::stream()));
    int max_air = -1;
    if (int rc = check_tables(subs, kAirCountUnknown, max_air)) return rc;
    std::vector<OriginalAir> airs((size_t)max_air + 1);
    PW_HIP_TRY(hipMemcpyAsync(airs.data(), d_original_airs, airs.size() * sizeof(OriginalAir), hipMemcpyDeviceToHost, pw::stream()));
    PW_HIP_TRY(hipStreamSynchronize(pw::stream()));
    std::vector<int32_t> bsize(airs.size());
    for (size_t i = 0; i < airs.size(); ++i) {
        bsize[i] = airs[i].row_block_size;
        if (bsize[i] < 0) return (int)hipErrorInvalidValue;
    }
    return tracegen_with_host_tables(d_output, H, d_original_airs, subs, bsize, num_apc_calls);
}

// Extension (not in the reference ABI): the same gather for a caller that still has the tables on the host — the
// reference's own host code does, it builds them right before the upload (cuda/mod.rs:272-332). No device-to-host copy,
// no stream synchronisation: the call only enqueues kernels.
extern "C" int powdr_apc_tracegen_host_tables(PowdrFp* d_output, size_t output_height, const OriginalAir* d_original_airs,
                                              const OriginalAir* h_original_airs, size_t n_airs, const Subst* h_subs,
                                              size_t n_subs, int num_apc_calls) {
    (void)hipGetLastError();
    const size_t H = output_height;
    if ((H & (H - 1)) != 0) return (int)hipErrorInvalidValue;
    if (H == 0 || n_subs == 0) return (int)hipGetLastError();
    if (!h_original_airs || !h_subs) return (int)hipErrorInvalidValue;
    if (num_apc_calls < 0) num_apc_calls = 0;
    if ((size_t)num_apc_calls > H) num_apc_calls = (int)H;
    std::vector<Subst> subs(h_subs, h_subs + n_subs);
    int max_air = -1;
    if (int rc = check_tables(subs, kAirCountUnknown, max_air)) return rc;
    if ((size_t)max_air >= n_airs) return (int)hipErrorInvalidValue;  // also n_airs == 0 with substitutions present
    std::vector<int32_t> bsize(n_airs);
    for (size_t i = 0; i < n_airs; ++i) {
        bsize[i] = h_original_airs[i].row_block_size;
        if (bsize[i] < 0) return (int)hipErrorInvalidValue;
    }
    return tracegen_with_host_tables(d_output, H, d_original_airs, subs, bsize, num_apc_calls, h_original_airs, n_airs);
}

// ---------------------------------------------------------------------------------------------------------------------
// Row (f)-1 of SURVEY.md §8, as far as it can go without the original chips: the gather when the sources arrive CALL-MAJOR
// AND COMPACTED — for every APC call the cells the APC actually uses, contiguous (`buffer[r * cells_per_call + slot]`) —
// instead of as full column-major dummy traces of which the optimised APC keeps ~7 % (C2: 27 521 source cells per call,
// 2 021 used). That layout is what an original chip's trace generation would write if it were handed the (row, column) ->
// slot map of the APC (it computes every cell of its rows from one record anyway, /root/reference/openvm/src/
// powdr_extension/trace_generator/cuda/mod.rs:228-253 calls `chip.generate_proving_ctx(record_arena)` per original AIR).
// The gather then is a plain tiled transpose [calls x slots] -> [columns x rows]: 4 bytes read + 4 written per APC cell,
// which is SURVEY 8d's algorithmic figure for this stage.
namespace {

constexpr int kCmMaxAirs = 16;
struct CMAirs {
    const uint32_t* buf[kCmMaxAirs];
    int32_t cells[kCmMaxAirs];
};
struct CMJob { int32_t air, slot0, col_off, n_slots; };

__global__ __launch_bounds__(256) void apc_gather_callmajor_kernel(uint32_t* __restrict__ out, size_t H, CMAirs airs,
                                                                    const CMJob* __restrict__ jobs, const int32_t* __restrict__ col_of,
                                                                    int num_calls) {
    __shared__ uint32_t tile[64][65];
    const CMJob job = jobs[blockIdx.y];
    const size_t r0 = (size_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const uint32_t* __restrict__ src = airs.buf[job.air];
    const size_t U = (size_t)airs.cells[job.air];
    const bool slot_ok = tx < job.n_slots;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ty + 4 * i;
        uint32_t v = 0u;
        if (slot_ok && r0 + r < (size_t)num_calls) v = __builtin_nontemporal_load(src + (r0 + r) * U + job.slot0 + tx);
        tile[r][tx] = v;
    }
    __syncthreads();
    if (r0 + tx >= H) return;
    for (int s = ty; s < job.n_slots; s += 4) {
        const int32_t col = col_of[job.col_off + job.slot0 + s];
        if (col >= 0) __builtin_nontemporal_store(tile[tx][s], out + (size_t)col * H + r0 + tx);
    }
}

struct CMPlan {
    CMJob* d_jobs = nullptr;
    int32_t* d_col_of = nullptr;
    size_t n_jobs = 0;
    std::vector<PowdrSubstCM> key_subs;
    std::vector<int32_t> key_cells;
    int device = 0;
    uint64_t last_use = 0;
    ~CMPlan() { if (d_jobs) (void)hipFree(d_jobs); if (d_col_of) (void)hipFree(d_col_of); }
};
std::mutex g_cm_mu;
std::unordered_map<uint64_t, std::shared_ptr<CMPlan>> g_cm_plans;
uint64_t g_cm_clock = 0;

}  // namespace

extern "C" int powdr_apc_tracegen_callmajor(PowdrFp* d_output, size_t output_height, const PowdrCallMajorAir* h_airs, size_t n_airs,
                                            const PowdrSubstCM* h_subs, size_t n_subs, int num_apc_calls) {
    (void)hipGetLastError();
    const size_t H = output_height;
    if ((H & (H - 1)) != 0 || n_airs > (size_t)kCmMaxAirs || (n_subs && (!h_airs || !h_subs))) return (int)hipErrorInvalidValue;
    if (H == 0 || n_subs == 0) return (int)hipGetLastError();
    if (num_apc_calls < 0) num_apc_calls = 0;
    if ((size_t)num_apc_calls > H) num_apc_calls = (int)H;
    std::vector<int32_t> cells(n_airs);
    CMAirs airs{};
    for (size_t a = 0; a < n_airs; ++a) {
        if (h_airs[a].cells_per_call < 0) return (int)hipErrorInvalidValue;
        cells[a] = h_airs[a].cells_per_call;
        airs.buf[a] = h_airs[a].buffer;
        airs.cells[a] = cells[a];
    }
    for (size_t i = 0; i < n_subs; ++i)
        if (h_subs[i].air_index < 0 || (size_t)h_subs[i].air_index >= n_airs || h_subs[i].slot < 0 ||
            h_subs[i].slot >= cells[h_subs[i].air_index] || h_subs[i].apc_col < 0)
            return (int)hipErrorInvalidValue;
    uint64_t key = fnv1a(h_subs, n_subs * sizeof(PowdrSubstCM), 1469598103934665603ull);
    key = fnv1a(cells.data(), cells.size() * 4, key);
    int device = 0;
    PW_HIP_TRY(hipGetDevice(&device));
    key = fnv1a(&device, sizeof device, key);
    std::shared_ptr<CMPlan> plan;
    {
        std::lock_guard<std::mutex> lk(g_cm_mu);
        auto it = g_cm_plans.find(key);
        if (it != g_cm_plans.end()) {
            const CMPlan& c = *it->second;
            if (c.device != device || c.key_cells != cells || c.key_subs.size() != n_subs ||
                memcmp(c.key_subs.data(), h_subs, n_subs * sizeof(PowdrSubstCM)) != 0) { g_cm_plans.erase(it); it = g_cm_plans.end(); }
        }
        if (it == g_cm_plans.end()) {
            if (g_cm_plans.size() >= kMaxCachedPlans) {
                auto victim = g_cm_plans.begin();
                for (auto j = g_cm_plans.begin(); j != g_cm_plans.end(); ++j) if (j->second->last_use < victim->second->last_use) victim = j;
                g_cm_plans.erase(victim);
            }
            auto p = std::make_shared<CMPlan>();
            // slot -> APC column per AIR (duplicate destinations resolve like the sequential reference loop: the last Subst wins)
            std::vector<size_t> off(n_airs + 1, 0);
            for (size_t a = 0; a < n_airs; ++a) off[a + 1] = off[a] + (size_t)cells[a];
            std::vector<int32_t> col_of(off[n_airs] ? off[n_airs] : 1, -1);
            std::unordered_map<int32_t, size_t> last;
            for (size_t i = 0; i < n_subs; ++i) last[h_subs[i].apc_col] = i;
            for (size_t i = 0; i < n_subs; ++i)
                if (last[h_subs[i].apc_col] == i) {
                    int32_t& dst = col_of[off[h_subs[i].air_index] + h_subs[i].slot];
                    // one slot, one APC column: a producer that feeds two columns from one cell writes it into two slots
                    if (dst >= 0 && dst != h_subs[i].apc_col) return (int)hipErrorInvalidValue;
                    dst = h_subs[i].apc_col;
                }
            std::vector<CMJob> jobs;
            for (size_t a = 0; a < n_airs; ++a)
                for (int32_t s0 = 0; s0 < cells[a]; s0 += 64)
                    jobs.push_back({(int32_t)a, s0, (int32_t)off[a], std::min<int32_t>(64, cells[a] - s0)});
            p->n_jobs = jobs.size();
            PW_HIP_TRY(hipMalloc(&p->d_jobs, (jobs.size() + 1) * sizeof(CMJob)));
            PW_HIP_TRY(hipMalloc(&p->d_col_of, col_of.size() * 4));
            if (!jobs.empty()) PW_HIP_TRY(hipMemcpy(p->d_jobs, jobs.data(), jobs.size() * sizeof(CMJob), hipMemcpyHostToDevice));
            PW_HIP_TRY(hipMemcpy(p->d_col_of, col_of.data(), col_of.size() * 4, hipMemcpyHostToDevice));
            p->key_subs.assign(h_subs, h_subs + n_subs);
            p->key_cells = cells;
            p->device = device;
            it = g_cm_plans.emplace(key, std::move(p)).first;
        }
        plan = it->second;
        plan->last_use = ++g_cm_clock;
    }
    pw::ScopedKernelTimer t("apc_gather_callmajor_kernel");
    const unsigned row_tiles = pw::div_up(H, 64);
    for (size_t j = 0; j < plan->n_jobs; j += 65535) {
        const unsigned cnt = (unsigned)std::min<size_t>(65535, plan->n_jobs - j);
        hipLaunchKernelGGL(apc_gather_callmajor_kernel, dim3(row_tiles, cnt), dim3(256), 0, pw::stream(), d_output, H, airs, plan->d_jobs + j,
                           plan->d_col_of, num_apc_calls);
    }
    return (int)hipGetLastError();
}
