// Poseidon2 Merkle commitment kernels for gfx950.
//
//   leaf j   = overwrite-mode sponge (rate 8, width 16) over row j of a column-major
//              matrix; one lane per row, so every column access of a wave is one
//              contiguous 256-byte segment and the sponge state lives in 16 VGPRs.
//   node     = first 8 words of Poseidon2(left || right); one lane per parent,
//              children are 64 contiguous bytes.
// This stage is integer-VALU bound (~670 Montgomery products + ~1 500 modular additions
// per permutation, 253 permutations per 2 022-column row), not HBM bound and not MFMA
// work — see DESIGN.md §kernels.
#include "prover_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace pw {

namespace {

constexpr int kBlock = 256;
__constant__ p2::Params c_params;
p2::Params g_host_params;
bool g_params_ready = false;
uint64_t g_params_uploaded = 0;  // bit d: the __constant__ table of device d holds the parameters (one process may drive several GPUs)

// MINW = minimum waves per SIMD the compiler must leave room for: 8 caps the kernel at 64 VGPRs (less interleaving of the
// independent S-box chains), 6 allows 80 (what the unconstrained kernel wants, give or take two). POWDR_HASH_WAVES picks at run time.
template <int MINW>
__global__ __launch_bounds__(kBlock, MINW) void leaf_hash_kernel(const uint32_t* __restrict__ m, size_t height,
                                                            uint32_t width, size_t col_stride,
                                                            uint32_t* __restrict__ digests, size_t dstride, size_t doff) {
    // row j's digest goes to slot j * dstride + doff (1, 0: the matrix is the whole committed matrix; 2^b, r: it holds the rows
    // r + 2^b i of it — one sub-coset of the streamed prover)
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= height) return;
    uint32_t st[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0u;
    const uint32_t* col = m + j;
    // one copy of the permutation in the kernel: the short last chunk keeps the words it does not overwrite (scalar tests);
    // between permutations nothing reads the state, so they skip the final conditional subtractions (permute<true>) and
    // the digest is reduced once
#pragma unroll 1
    for (uint32_t c0 = 0; c0 < width; c0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c0 + k < width) st[k] = col[(size_t)(c0 + k) * col_stride];
        p2::permute<true>(st, c_params);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) st[k] = bb::reduce_2p(st[k]);
    uint4* out = reinterpret_cast<uint4*>(digests + (j * dstride + doff) * 8);
    out[0] = make_uint4(st[0], st[1], st[2], st[3]);
    out[1] = make_uint4(st[4], st[5], st[6], st[7]);
}

// The leaf hash of a mixed-height commitment runs over COLUMN-POINTER tables: the hashed row is the concatenation of row j of
// several matrices of one height (all AIRs of that height in a segment, AIR order); the pointers are wave-uniform scalar loads.
// All heights of a mixed-height commitment in ONE launch: a workgroup belongs to one level (first_block table, scalar scan);
// the levels are ordered widest first, so the threads that run the most permutations start first. A segment with eleven
// different heights otherwise pays eleven launches of which the short ones are latency bound — 64 waves hashing 2 400
// columns take as long as 300 dependent permutations take, however few rows there are.
struct LeafLevel {
    const uint32_t* const* cols;
    uint32_t n_cols;
    uint32_t first_block;  // blocks [first_block, next level's first_block) hash this level's rows
    uint64_t height;
    uint32_t* out;
};
struct LeafLevels {  // passed by value in the kernel arguments (28 x 32 bytes): no table upload, nothing to keep alive
    LeafLevel lv[28];
    int n;
};
template <int MINW>
__global__ __launch_bounds__(kBlock, MINW) void leaf_hash_levels_kernel(const LeafLevels levels) {
    int k = 0;
    while (k + 1 < levels.n && blockIdx.x >= levels.lv[k + 1].first_block) ++k;
    const LeafLevel lv = levels.lv[k];
    const size_t j = (size_t)(blockIdx.x - lv.first_block) * kBlock + threadIdx.x;
    if (j >= lv.height) return;
    uint32_t st[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0u;
#pragma unroll 1
    for (uint32_t c0 = 0; c0 < lv.n_cols; c0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (c0 + q < lv.n_cols) st[q] = lv.cols[c0 + q][j];
        p2::permute<true>(st, c_params);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) st[q] = bb::reduce_2p(st[q]);
    uint4* out = reinterpret_cast<uint4*>(lv.out + j * 8);
    out[0] = make_uint4(st[0], st[1], st[2], st[3]);
    out[1] = make_uint4(st[4], st[5], st[6], st[7]);
}

// The general form of a level's row digest: the hashed row is the CONCATENATION of the rows of several matrices, of which some exist
// only one sub-coset at a time (streamed AIRs, prover_stream.hpp). The concatenation is absorbed run by run — a run = consecutive
// resident matrices (column-pointer table) or one streamed matrix (one launch per sub-coset) — with the rows' sponge states parked
// in a buffer between the runs: a run continues the rate block the previous one left open at position pos0, and only the last run
// closes a partial block and writes digests. Same words as one sponge over the whole row (leaf_hash_levels_kernel).
struct AbsorbArgs {
    const uint32_t* m;            // a matrix (column stride col_stride) when cols == nullptr
    size_t col_stride;
    const uint32_t* const* cols;  // or a column-pointer table
    uint32_t n_cols, pos0;
    size_t height;                // rows of this launch; its row j is row j * rstride + roff of the level
    size_t rstride, roff;
    uint32_t* state;              // 16 words per row of the level: read unless `first`, written unless `last`
    uint32_t* digests;            // 8 words per row of the level: written when `last`
    int first, last;
};
template <int MINW>
__global__ __launch_bounds__(kBlock, MINW) void leaf_absorb_kernel(const AbsorbArgs a) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= a.height) return;
    const size_t row = j * a.rstride + a.roff;
    uint32_t st[16];
    if (a.first) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0u;
    } else {
        const uint4* in = reinterpret_cast<const uint4*>(a.state + row * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) { const uint4 x = in[v]; st[4 * v] = x.x; st[4 * v + 1] = x.y; st[4 * v + 2] = x.z; st[4 * v + 3] = x.w; }
    }
    auto cell = [&](uint32_t c) -> uint32_t { return a.cols ? a.cols[c][j] : a.m[(size_t)c * a.col_stride + j]; };
    uint32_t c = 0, pos = a.pos0;  // wave-uniform
    if (pos) {  // complete the rate block the previous run left open
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((uint32_t)k >= pos && (uint32_t)k - pos < a.n_cols) st[k] = cell((uint32_t)k - pos);
        const uint32_t take = 8u - pos < a.n_cols ? 8u - pos : a.n_cols;
        c = take;
        pos += take;
        if (pos == 8u) { p2::permute<true>(st, c_params); pos = 0; }
    }
#pragma unroll 1
    for (; c < a.n_cols; c += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c + k < a.n_cols) st[k] = cell(c + k);
        if (c + 8 <= a.n_cols) p2::permute<true>(st, c_params);
        else pos = a.n_cols - c;  // an open block: the next run continues it, the last run closes it below
    }
    if (a.last) {
        if (pos) p2::permute<true>(st, c_params);
        uint4* out = reinterpret_cast<uint4*>(a.digests + row * 8);
        out[0] = make_uint4(bb::reduce_2p(st[0]), bb::reduce_2p(st[1]), bb::reduce_2p(st[2]), bb::reduce_2p(st[3]));
        out[1] = make_uint4(bb::reduce_2p(st[4]), bb::reduce_2p(st[5]), bb::reduce_2p(st[6]), bb::reduce_2p(st[7]));
    } else {
        uint4* out = reinterpret_cast<uint4*>(a.state + row * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) out[v] = make_uint4(st[4 * v], st[4 * v + 1], st[4 * v + 2], st[4 * v + 3]);
    }
}

// One level of the mixed-height tree: parent j = compress(child j, child j + n) — the sibling pairing of the FRI fold, so
// that the ancestor of leaf q on the level of n nodes is q mod n — and, where matrices of height n exist, the digest of
// their rows is injected: parent = compress(parent, inject[j]).
__global__ __launch_bounds__(kBlock) void compress_strided_kernel(const uint32_t* __restrict__ children, size_t n,
                                                                   const uint32_t* __restrict__ inject, uint32_t* __restrict__ parents) {
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const uint4* l = reinterpret_cast<const uint4*>(children + j * 8);
    const uint4* r = reinterpret_cast<const uint4*>(children + (j + n) * 8);
    uint4 a = l[0], b = l[1], c = r[0], d = r[1];
    uint32_t st[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    p2::permute(st, c_params);
    if (inject) {
        const uint4* q = reinterpret_cast<const uint4*>(inject + j * 8);
        c = q[0]; d = q[1];
        st[8] = c.x; st[9] = c.y; st[10] = c.z; st[11] = c.w; st[12] = d.x; st[13] = d.y; st[14] = d.z; st[15] = d.w;
        p2::permute(st, c_params);
    }
    uint4* out = reinterpret_cast<uint4*>(parents + j * 8);
    out[0] = make_uint4(st[0], st[1], st[2], st[3]);
    out[1] = make_uint4(st[4], st[5], st[6], st[7]);
}

__global__ __launch_bounds__(kBlock) void ext_pair_leaf_kernel(const bb::Ext* __restrict__ v, size_t half,
                                                                uint32_t* __restrict__ digests) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= half) return;
    uint32_t st[16];
    const bb::Ext a = v[i], b = v[i + half];
#pragma unroll
    for (int k = 0; k < 4; ++k) { st[k] = a.c[k]; st[4 + k] = b.c[k]; st[8 + k] = 0u; st[12 + k] = 0u; }
    p2::permute(st, c_params);
    uint4* out = reinterpret_cast<uint4*>(digests + i * 8);
    out[0] = make_uint4(st[0], st[1], st[2], st[3]);
    out[1] = make_uint4(st[4], st[5], st[6], st[7]);
}

__global__ __launch_bounds__(kBlock) void compress_kernel(const uint32_t* __restrict__ children, size_t n_parents,
                                                           uint32_t* __restrict__ parents) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_parents) return;
    const uint4* in = reinterpret_cast<const uint4*>(children + i * 16);
    uint4 a = in[0], b = in[1], c = in[2], d = in[3];
    uint32_t st[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    p2::permute(st, c_params);
    uint4* out = reinterpret_cast<uint4*>(parents + i * 8);
    out[0] = make_uint4(st[0], st[1], st[2], st[3]);
    out[1] = make_uint4(st[4], st[5], st[6], st[7]);
}

// Proof-of-work: thread t tries witness base + t; the transcript absorbs the pending words plus
// the witness (overwrite mode), permutes, and samples the LAST rate word (out.pop_back()).
__global__ __launch_bounds__(kBlock) void pow_kernel(const uint32_t* __restrict__ state16, const uint32_t* __restrict__ pending,
                                                      uint32_t in_len, uint32_t bits, uint32_t base, uint32_t* best) {
    const uint32_t w = base + blockIdx.x * kBlock + threadIdx.x;
    if (w >= bb::P) return;
    uint32_t st[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = state16[i];
    // absorb pending words then the witness
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if ((uint32_t)i < in_len) st[i] = pending[i];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if ((uint32_t)i == in_len) st[i] = bb::to_monty(w);
    p2::permute(st, c_params);
    // whether the witness filled the rate (duplex inside observe) or not (duplex inside sample), exactly one
    // permutation follows and sample() pops the last rate word
    const uint32_t sample = st[7];
    if ((bb::from_monty(sample) & ((1u << bits) - 1u)) == 0u) atomicMin(best, w);
}

// The top of a tree (<= 2048 nodes on a level) in ONE launch: a single 1024-thread workgroup walks
// the levels; the levels live in global memory, the workgroup's own writes are made visible to
// its other waves by __threadfence_block() + the barrier. Replaces ~11 tiny launches per tree
// (22 trees per proof).
constexpr size_t kTailNodes = 2048;
// `root_out` (optional, host-mapped pinned memory): the root also goes straight to the host, which reads it after its next
// stream synchronisation — a 32-byte hipMemcpyAsync D2H is a blit dispatch that starts ~40 us after the kernel it follows
// (profiles/r02_segment_gaps.txt), 27 times per proof.
__global__ __launch_bounds__(1024) void compress_tail_kernel(uint32_t* __restrict__ digests, size_t n_first, uint32_t* __restrict__ root_out) {
    size_t off = 0;
    for (size_t n = n_first; n > 1; n >>= 1) {
        const size_t parents = n >> 1;
        for (size_t i = threadIdx.x; i < parents; i += 1024) {
            const uint32_t* in = digests + off + i * 16;
            uint32_t st[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) st[k] = in[k];
            p2::permute(st, c_params);
            uint32_t* out = digests + off + n * 8 + i * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) out[k] = st[k];
        }
        off += n * 8;
        __threadfence_block();
        __syncthreads();
    }
    if (root_out && threadIdx.x < 8) root_out[threadIdx.x] = digests[off + threadIdx.x];
}

int hash_min_waves() {
    const char* e = getenv("POWDR_HASH_WAVES");
    return e ? atoi(e) : 6;
}

// per host thread: 8 words of host-mapped pinned memory for the root of the tree being built
struct RootMailbox {
    uint32_t* host = nullptr;
    uint32_t* dev = nullptr;
    int device = -1;
    ~RootMailbox() { if (host) (void)hipHostFree(host); }  // worker threads (pw_prove_airs, a caller's pool) come and go
};
thread_local RootMailbox g_mailbox;

int build_levels(uint32_t* digests, size_t n_leaves, uint32_t* root_out = nullptr) {
    size_t off = 0;
    size_t n = n_leaves;
    for (; n > kTailNodes; n >>= 1) {
        size_t parents = n >> 1;
        ScopedKernelTimer t("compress_kernel");
        hipLaunchKernelGGL(compress_kernel, dim3(div_up(parents, kBlock)), dim3(kBlock), 0, stream(), digests + off,
                           parents, digests + off + n * 8);
        off += n * 8;
    }
    if (n > 1 || root_out) {  // a single leaf is its own root: the kernel only forwards it
        ScopedKernelTimer t("compress_tail_kernel");
        hipLaunchKernelGGL(compress_tail_kernel, dim3(1), dim3(1024), 0, stream(), digests + off, n, root_out);
    }
    return (int)hipGetLastError();
}

}  // namespace

static std::mutex g_params_mu;

const p2::Params& poseidon2_params_host() {
    static std::once_flag once;
    std::call_once(once, [] {
        std::lock_guard<std::mutex> lk(g_params_mu);
        if (!g_params_ready) { p2::generate_params(g_host_params); g_params_ready = true; }
    });
    return g_host_params;
}

// Install another set of round constants (canonical words; nullptr = the placeholder stream): host table now, every
// device's __constant__ copy at its next use.
int poseidon2_set_constants(const uint32_t* ext_rc128, const uint32_t* int_rc13) {
    if ((ext_rc128 == nullptr) != (int_rc13 == nullptr)) return -1;
    if (ext_rc128) {
        for (int i = 0; i < 128; ++i) if (ext_rc128[i] >= bb::P) return -1;
        for (int i = 0; i < 13; ++i) if (int_rc13[i] >= bb::P) return -1;
    }
    (void)poseidon2_params_host();
    std::lock_guard<std::mutex> lk(g_params_mu);
    p2::Params p{};
    if (ext_rc128) {
        for (int r = 0; r < 8; ++r) for (int i = 0; i < 16; ++i) p.ext_rc[r][i] = bb::to_monty(ext_rc128[16 * r + i]);
        for (int r = 0; r < 13; ++r) p.int_rc[r] = bb::to_monty(int_rc13[r]);
    } else {
        p2::default_round_constants(p);
    }
    p2::derive_params(p);
    g_host_params = p;
    g_params_uploaded = 0;
    return 0;
}

int poseidon2_upload_params() {
    std::lock_guard<std::mutex> lk(g_params_mu);
    int device = 0;
    PW_HIP_TRY(hipGetDevice(&device));
    if (device < 64 && (g_params_uploaded >> device) & 1) return 0;
    if (!g_params_ready) { p2::generate_params(g_host_params); g_params_ready = true; }
    const p2::Params& p = g_host_params;
    PW_HIP_TRY(hipDeviceSynchronize());  // kernels in flight (any stream) still read the table being replaced
    PW_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_params), &p, sizeof(p2::Params)));  // synchronous; the current device's copy
    if (device < 64) g_params_uploaded |= 1ull << device;
    return 0;
}

// Host-mapped landing place of the calling thread for a tree's root (valid after the next synchronisation of the stream the
// tree was built on); nullptr if pinned memory cannot be had — callers then copy the root with hipMemcpyAsync.
uint32_t* merkle_root_mailbox(uint32_t** device_ptr) {
    RootMailbox& mb = g_mailbox;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    if (mb.host && mb.device != device) { (void)hipHostFree(mb.host); mb.host = nullptr; }
    if (!mb.host) {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return nullptr; }
        mb.host = (uint32_t*)h; mb.dev = (uint32_t*)d; mb.device = device;
    }
    *device_ptr = mb.dev;
    return mb.host;
}

int merkle_leaf_hash(const uint32_t* m, size_t height, uint32_t width, size_t col_stride, uint32_t* digests, size_t digest_stride,
                     size_t digest_offset) {
    int rc = poseidon2_upload_params();
    if (rc) return rc;
    ScopedKernelTimer t("leaf_hash_kernel");
    if (hash_min_waves() >= 8)
        hipLaunchKernelGGL(leaf_hash_kernel<8>, dim3(div_up(height, kBlock)), dim3(kBlock), 0, stream(), m, height, width, col_stride, digests,
                           digest_stride, digest_offset);
    else
        hipLaunchKernelGGL(leaf_hash_kernel<6>, dim3(div_up(height, kBlock)), dim3(kBlock), 0, stream(), m, height, width, col_stride, digests,
                           digest_stride, digest_offset);
    return (int)hipGetLastError();
}

int merkle_build_levels(uint32_t* digests, size_t n_leaves, uint32_t* root_out) { return build_levels(digests, n_leaves, root_out); }

int merkle_leaf_absorb(const uint32_t* m, size_t col_stride, const uint32_t* const* d_cols, uint32_t n_cols, uint32_t pos0, size_t height, size_t rstride,
                       size_t roff, uint32_t* state, uint32_t* digests, bool first, bool last) {
    int rc = poseidon2_upload_params();
    if (rc) return rc;
    if (!height) return 0;
    const AbsorbArgs a{m, col_stride, d_cols, n_cols, pos0, height, rstride, roff, state, digests, first ? 1 : 0, last ? 1 : 0};
    ScopedKernelTimer t("leaf_hash_kernel");
    if (hash_min_waves() >= 8) hipLaunchKernelGGL(leaf_absorb_kernel<8>, dim3(div_up(height, kBlock)), dim3(kBlock), 0, stream(), a);
    else hipLaunchKernelGGL(leaf_absorb_kernel<6>, dim3(div_up(height, kBlock)), dim3(kBlock), 0, stream(), a);
    return (int)hipGetLastError();
}

int merkle_commit_matrix(const uint32_t* m, size_t height, uint32_t width, size_t col_stride, uint32_t* digests, uint32_t* root_out) {
    const int rc = merkle_leaf_hash(m, height, width, col_stride, digests, 1, 0);
    return rc ? rc : build_levels(digests, height, root_out);
}

int merkle_commit_mixed(const MixedLevelCols* by_log, int L, uint32_t* digests, uint32_t* d_inject) {
    int rc = poseidon2_upload_params();
    if (rc) return rc;
    if (L < 0 || L > 27 || (!by_log[L].n_cols && !by_log[L].external)) return (int)hipErrorInvalidValue;
    const size_t N = (size_t)1 << L;
    // row digests of every height in one launch: level L into the tree's leaves, smaller heights into their slice of
    // d_inject (offset 2^lg * 8 words: the slices of all heights below L fit in 2^L * 8 words)
    LeafLevels levels{};
    LeafLevel* h_levels = levels.lv;
    int n_levels = 0;
    for (int lg = L; lg >= 0; --lg)
        if (by_log[lg].n_cols && !by_log[lg].external)
            h_levels[n_levels++] = LeafLevel{by_log[lg].d_cols, by_log[lg].n_cols, 0u, (uint64_t)1 << lg,
                                             lg == L ? digests : d_inject + ((size_t)1 << lg) * 8};
    std::stable_sort(h_levels, h_levels + n_levels, [](const LeafLevel& a, const LeafLevel& b) { return a.n_cols > b.n_cols; });
    uint32_t blocks = 0;
    for (int k = 0; k < n_levels; ++k) { h_levels[k].first_block = blocks; blocks += div_up((size_t)h_levels[k].height, kBlock); }
    levels.n = n_levels;
    if (n_levels) {
        ScopedKernelTimer t("leaf_hash_kernel");
        if (hash_min_waves() >= 8) hipLaunchKernelGGL(leaf_hash_levels_kernel<8>, dim3(blocks), dim3(kBlock), 0, stream(), levels);
        else hipLaunchKernelGGL(leaf_hash_levels_kernel<6>, dim3(blocks), dim3(kBlock), 0, stream(), levels);
    }
    size_t off = 0;
    for (int lg = L - 1; lg >= 0; --lg) {
        const size_t n = (size_t)1 << lg;
        const uint32_t* inj = (by_log[lg].n_cols || by_log[lg].external) ? d_inject + n * 8 : nullptr;
        ScopedKernelTimer t("compress_kernel");
        hipLaunchKernelGGL(compress_strided_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0, stream(), digests + off, n, inj, digests + off + 2 * n * 8);
        off += 2 * n * 8;
    }
    (void)N;
    return (int)hipGetLastError();
}

int merkle_commit_ext_pairs(const bb::Ext* v, size_t half, uint32_t* digests, uint32_t* root_out) {
    int rc = poseidon2_upload_params();
    if (rc) return rc;
    {
        ScopedKernelTimer t("ext_pair_leaf_kernel");
        hipLaunchKernelGGL(ext_pair_leaf_kernel, dim3(div_up(half, kBlock)), dim3(kBlock), 0, stream(), v, half, digests);
    }
    return build_levels(digests, half, root_out);
}

int pow_grind(const uint32_t* d_state16, const uint32_t* d_pending, uint32_t in_len, uint32_t bits, uint32_t* d_best,
              uint32_t* witness_out) {
    int rc = poseidon2_upload_params();
    if (rc) return rc;
    const uint32_t batch = 1u << 20;
    uint32_t best = 0xffffffffu;
    for (uint64_t base = 0; base < bb::P; base += batch) {
        PW_HIP_TRY(hipMemcpyAsync(d_best, &best, 4, hipMemcpyHostToDevice, stream()));
        hipLaunchKernelGGL(pow_kernel, dim3(batch / kBlock), dim3(kBlock), 0, stream(), d_state16, d_pending, in_len, bits,
                           (uint32_t)base, d_best);
        PW_HIP_TRY(hipMemcpyAsync(&best, d_best, 4, hipMemcpyDeviceToHost, stream()));
        PW_HIP_TRY(hipStreamSynchronize(stream()));
        if (best != 0xffffffffu) break;
    }
    *witness_out = best;
    return best == 0xffffffffu ? (int)hipErrorUnknown : 0;
}

}  // namespace pw
