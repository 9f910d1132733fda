// pw-stark v1 on the device: ONE proof for all AIRs of a segment (proof magic "PWS3").
//
// The reference makes one engine call per segment with the traces of all chips, `engine.prove(pk, ProvingContext{
// per_trace})` (/root/reference/openvm/src/trace_generation.rs:136-139, openvm-riscv/src/lib.rs:327-332). Protocol
// definition: oracle/stark_segment.inc (header comment); the words produced here must equal the oracle's.
//
//   phase 1   LDE of every main trace (per-AIR kernels of prover.hip), ONE mixed-height Poseidon2 tree over all of them:
//             the leaf hash runs over a column-pointer table, so all AIRs of one height are hashed by one launch — a
//             2^12-row AIR no longer gets a launch (and a quarter-empty GPU) of its own
//   phase 2   LogUp permutation matrices (shared challenges straight from the segment transcript), one mixed tree
//   phase 3   quotients (shared alpha), one mixed tree over the 8-column chunk matrices
//   phase 4   openings at zeta / g_a zeta, all AIRs' values fetched with one copy
//   phase 5   reduced openings per height; FRI over the tallest domain with the smaller heights rolled in; one query phase
// Host round trips per segment: 3 roots + 1 (sums) + 1 (openings) + L FRI roots + queries, instead of that per AIR.
//
// STREAMED AIRs (round 4): an AIR whose LDE does not fit beside the others — BASELINE configs[2]'s 3 731 + 4 632 committed
// columns x 2^23 rows — is proven from coefficient arrays inside the same segment proof: its row digests are hashed sub-coset by
// sub-coset into its level of the mixed trees (merkle.hip `external` levels), its quotient, DEEP numerator and query rows come from
// prover_stream.hpp. A level's row digest is a sponge over the CONCATENATED rows of all matrices of that height: a level that holds a
// streamed matrix is absorbed run by run with the rows' sponge states parked in between (merkle.hip leaf_absorb_kernel).
#include "prover_state.hpp"
#include "prover_stream.hpp"
#include "logup_groups.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace pw;

#define TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

namespace {

constexpr int kMaxSide = 8;
struct SegCtx {
    DeviceBuf dig;     // three mixed trees | FRI trees
    DeviceBuf inject;  // row digests of the smaller heights (one level at a time)
    DeviceBuf ext;     // FRI layers (2 N) | reduced openings of the smaller heights (N) | scratch vector (N)
    DeviceBuf misc;    // column-pointer tables, opened values, gamma powers, query indices / answers
    DeviceBuf state;   // streamed AIRs: the rows' sponge states of the level being hashed run by run (16 words per row)
    std::vector<uint32_t> proof;
    size_t last_plan[3] = {0, 0, 0};   // bytes of the last call's plan: all AIRs resident | as chosen | what the policy had to work with
    std::vector<uint32_t> last_modes;  // per AIR of the last call: log2(sub-cosets) | 0x100 if the trace was eaten (pw_segment_last_modes)
    // side streams for the per-AIR stages of a segment with many AIRs (fork from / join into the caller's stream by events)
    hipStream_t side[kMaxSide] = {};
    hipEvent_t fork_ev = nullptr, done_ev[kMaxSide] = {};
    int n_side = 0, side_device = -1;
    int ensure_side(int want) {
        int dev = 0;
        PW_HIP_TRY(hipGetDevice(&dev));
        if (dev != side_device) { n_side = 0; fork_ev = nullptr; side_device = dev; }  // handles of another device are simply dropped
        if (!fork_ev) PW_HIP_TRY(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
        for (; n_side < want; ++n_side) {
            PW_HIP_TRY(hipStreamCreateWithFlags(&side[n_side], hipStreamNonBlocking));
            PW_HIP_TRY(hipEventCreateWithFlags(&done_ev[n_side], hipEventDisableTiming));
        }
        return 0;
    }
    // a host thread that proved segments exits: its streams, events and (through ~DeviceBuf) its device buffers go with it
    ~SegCtx() {
        int dev = -1;
        if (n_side == 0 && !fork_ev) return;
        if (hipGetDevice(&dev) != hipSuccess || dev != side_device) { (void)hipGetLastError(); return; }
        for (int k = 0; k < n_side; ++k) { (void)hipStreamDestroy(side[k]); (void)hipEventDestroy(done_ev[k]); }
        if (fork_ev) (void)hipEventDestroy(fork_ev);
    }
};
thread_local SegCtx g_ctx;  // one per host thread (= per launch stream)

struct Shape {
    uint32_t W, nc, n_int, n_g, Wp, M, K, log_h;
    size_t H, N, koff;
    int logN;
};

// device buffers of one RESIDENT AIR in the segment flow (the per-AIR prover object owns them; nothing here is freed per segment):
// the sizes first — the streaming decision below adds up exactly what ensure_air would allocate — then the allocation
struct AirPlan {
    size_t panel_cols = 0, coef = 0, lde = 0, perm = 0, plde = 0, q = 0, qpart = 0, qcoef = 0, qlde = 0, ext_arena = 0, misc = 0, gbuf = 0;
    size_t total() const { return coef + lde + perm + plde + q + qpart + qcoef + qlde + ext_arena + misc + gbuf; }
};
AirPlan plan_air(const PwProver* p, const Shape& s, bool logup) {
    AirPlan B;
    B.panel_cols = lde_panel_cols(s.H, logup ? std::max<size_t>(s.W, s.Wp) : s.W);
    B.coef = B.panel_cols * s.H * 4;
    B.lde = (size_t)s.W * s.N * 4;
    if (logup) {  // + the uncommitted per-row-sum columns of the specialised path
        B.perm = (size_t)(s.Wp + kJitExtraPermCols) * s.H * 4;
        B.plde = (size_t)(s.Wp + kJitExtraPermCols) * s.N * 4;
    }
    B.q = 4 * s.N * 4;
    if (!logup) {
        const uint32_t chunks = quotient_chunks(s.N, s.nc);
        if (chunks > 1) B.qpart = (size_t)chunks * 4 * s.N * 4;
    }
    // the specialised kernels' partial sums (not lazily in the middle of the proof, while other AIRs' side streams run)
    if (specialised(p)) B.qpart = std::max(B.qpart, jit_part_bytes(p, s.H, s.N));
    B.qcoef = 8 * s.H * 4;
    B.qlde = 8 * s.N * 4;
    B.ext_arena = (3 * s.H + s.H / 4096 + 32) * sizeof(bb::Ext);  // weights | weights at g zeta | row sums + block totals
    const uint32_t n_chunks = div_up(s.H, 8192);
    const uint32_t dot_cols = std::max({s.W, s.Wp, 8u});
    B.misc = (2 * (size_t)dot_cols * n_chunks + s.M + p->max_args + 64) * sizeof(bb::Ext) + 4096;  // ext_dot_columns2: two sets of partial sums
    if (s.log_h >= kDeepComboMinLogHeight) B.gbuf = (size_t)24 * s.H * 4;  // the DEEP combinations and their LDE
    return B;
}
int ensure_air(PwProver* p, const Shape& s, bool logup, CommitLayout& Lc) {
    const AirPlan B = plan_air(p, s, logup);
    Lc.H = s.H; Lc.N = s.N; Lc.tree_words = 0; Lc.fri_words = 0; Lc.n_trees = 0;
    Lc.panel_cols = B.panel_cols;
    TRY(p->coef.ensure(B.coef));
    TRY(p->lde.ensure(B.lde));
    if (B.perm) { TRY(p->perm.ensure(B.perm)); TRY(p->plde.ensure(B.plde)); }
    TRY(p->q.ensure(B.q));
    if (B.qpart > p->qpart.bytes) TRY(p->qpart.ensure(B.qpart));
    TRY(p->qcoef.ensure(B.qcoef));
    TRY(p->qlde.ensure(B.qlde));
    TRY(p->ext_arena.ensure(B.ext_arena));
    TRY(p->misc.ensure(B.misc));
    if (B.gbuf) TRY(p->gbuf.ensure(B.gbuf));
    return 0;
}

}  // namespace

// honour_flags: pw_prove_segment_consuming — PwSegmentAir::flags is read (PW_AIR_HAND_OVER); pw_prove_segment ignores the word
// (callers of the older struct left it uninitialised)
static int prove_segment_impl(const PwSegmentAir* airs, size_t n_airs, int logup_flag, bool honour_flags, const uint32_t** proof_words,
                              size_t* n_words) {
    if (!airs || !n_airs || !proof_words || !n_words) return (int)hipErrorInvalidValue;
    const bool lg = logup_flag != 0;
    const size_t A = n_airs;
    // handed[a]: the caller gives the trace away (the reference moves `common_main` into the engine for EVERY AIR of a segment,
    // cuda/mod.rs:415-419). Only a STREAMED AIR uses that (eat[a], below): its coefficient arrays then live in the caller's buffer.
    std::vector<char> handed(A, 0);
    for (size_t a = 0; a < A; ++a) {
        handed[a] = honour_flags && (airs[a].flags & PW_AIR_HAND_OVER) ? 1 : 0;
        // a handed-over trace becomes a coefficient array that is read 2 / 4 words at a time (fold loads, the DEEP combination)
        if (handed[a] && ((uintptr_t)airs[a].d_trace & 15)) return (int)hipErrorInvalidValue;
        if (!airs[a].prover || !airs[a].d_trace || airs[a].log_height < 1 || airs[a].log_height > 26) return (int)hipErrorInvalidValue;
        if (lg && !airs[a].prover->logup) return (int)hipErrorInvalidValue;  // needs the interaction tables (pw_prover_create_logup)
        // the per-AIR device buffers live in the prover object and the AIRs of a segment run concurrently on side streams: one
        // prover cannot serve two AIRs of the same segment; one FRI / query phase means one configuration for all of them
        if (airs[a].prover->cfg.num_queries != airs[0].prover->cfg.num_queries || airs[a].prover->cfg.pow_bits != airs[0].prover->cfg.pow_bits)
            return (int)hipErrorInvalidValue;
    }
    {
        std::vector<const PwProver*> seen(A);
        for (size_t a = 0; a < A; ++a) seen[a] = airs[a].prover;
        std::sort(seen.begin(), seen.end());
        if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return (int)hipErrorInvalidValue;
    }
    (void)hipGetLastError();
    hipStream_t st = stream();
    TRY(poseidon2_upload_params());
    {   // run-time specialised expression kernels of all AIRs that qualify, compiled in one concurrent batch (prover_jit.hip)
        std::vector<PwProver*> ps(A);
        std::vector<uint32_t> lhs(A);
        for (size_t a = 0; a < A; ++a) { ps[a] = airs[a].prover; lhs[a] = airs[a].log_height; }
        (void)specialise_provers(ps.data(), A, lhs.data(), false);
    }
    SegCtx& cx = g_ctx;
    // The per-AIR stages (LDE, permutation trace, quotient, openings, query rows) of different AIRs are independent chains of
    // small launches; on one stream a segment of 60 AIRs pays ~10 us of dispatch latency between 1 400 dependent kernels
    // (profiles/r02_segment_gaps.txt: 15 of 176 ms). With K side streams the chains of different AIRs overlap:
    // fork = the side streams wait for what the caller's stream has done, AIR a runs on side[a % K], join = the caller's stream
    // waits for all of them. POWDR_SEGMENT_STREAMS=0 keeps everything on the caller's stream.
    int K = 4;
    if (const char* e = getenv("POWDR_SEGMENT_STREAMS")) K = atoi(e);
    if (K > kMaxSide) K = kMaxSide;
    if (K < 2 || A < 4) K = 0;
    if (K) TRY(cx.ensure_side(K));
    struct StreamRestore { hipStream_t s; ~StreamRestore() { set_stream(s); } } restore{st};
    auto fork = [&]() -> int {
        if (!K) return 0;
        PW_HIP_TRY(hipEventRecord(cx.fork_ev, st));
        for (int k = 0; k < K; ++k) PW_HIP_TRY(hipStreamWaitEvent(cx.side[k], cx.fork_ev, 0));
        return 0;
    };
    auto on_air = [&](size_t a) { if (K) set_stream(cx.side[a % (size_t)K]); };
    auto join = [&]() -> int {
        if (!K) return 0;
        set_stream(st);
        for (int k = 0; k < K; ++k) {
            PW_HIP_TRY(hipEventRecord(cx.done_ev[k], cx.side[k]));
            PW_HIP_TRY(hipStreamWaitEvent(st, cx.done_ev[k], 0));
        }
        return 0;
    };
    const PwStarkConfig cfg = airs[0].prover->cfg;
    const uint32_t nq = cfg.num_queries;

    // ---- shapes -----------------------------------------------------------------------------------------------
    std::vector<Shape> sh(A);
    size_t K_total = 0;
    int L = 0;
    for (size_t a = 0; a < A; ++a) {
        const PwProver* p = airs[a].prover;
        Shape& s = sh[a];
        s.W = p->width; s.nc = p->n_constraints; s.log_h = airs[a].log_height;
        s.n_int = lg ? p->n_inter : 0; s.n_g = lg ? p->n_groups : 0; s.Wp = lg ? 4 * (s.n_g + 1) : 0;
        s.M = s.nc + (lg ? s.n_g + 3 : 0);
        s.K = s.W + 2 * s.Wp + 8;
        s.H = (size_t)1 << s.log_h; s.N = 2 * s.H; s.logN = (int)s.log_h + 1;
        s.koff = K_total;
        K_total += s.K;
        if (s.logN > L) L = s.logN;
    }
    const size_t Nmax = (size_t)1 << L;
    const size_t tree_words = merkle_words(Nmax);
    const int n_trees = lg ? 3 : 2;
    const int rounds = L - 1;
    std::vector<size_t> layer_off(rounds + 1), ftree_off(rounds);
    size_t fri_words = 0;
    { size_t o = 0; for (int l = 0; l <= rounds; ++l) { layer_off[l] = o; o += Nmax >> l; } }
    for (int l = 0; l < rounds; ++l) { ftree_off[l] = fri_words; fri_words += merkle_words((Nmax >> l) / 2); }
    // reduced-opening vectors of the smaller heights, in the ext arena after the FRI layers
    std::vector<size_t> ro_off(L + 1, 0);
    std::vector<char> has_height(L + 1, 0);
    for (size_t a = 0; a < A; ++a) has_height[sh[a].logN] = 1;
    size_t ext_words = 2 * Nmax;
    for (int k = L - 1; k >= 2; --k) if (has_height[k]) { ro_off[k] = ext_words; ext_words += (size_t)1 << k; }
    const size_t tmp_off = ext_words;
    ext_words += Nmax;

    // ---- buffers ----------------------------------------------------------------------------------------------
    TRY(cx.dig.ensure((n_trees * tree_words + fri_words) * 4));
    TRY(cx.inject.ensure((Nmax + 1) * 8 * 4));  // row digests of the heights below N: the slice of height n starts at n * 8 words
    TRY(cx.ext.ensure(ext_words * sizeof(bb::Ext)));
    size_t cols_total = 0;
    for (size_t a = 0; a < A; ++a) cols_total += std::max<size_t>({sh[a].W, sh[a].Wp, 8});
    size_t row_words = 0;  // query answers: rows of every tree
    for (size_t a = 0; a < A; ++a) row_words += (size_t)nq * (sh[a].W + sh[a].Wp + 8);
    const size_t n_dig = (size_t)nq * ((size_t)n_trees * L + (size_t)rounds * L) + 16;  // upper bound of digest records
    const size_t misc_bytes = cols_total * 8 + 4 * A * 8 + 2 * K_total * sizeof(bb::Ext) + (size_t)A * nq * 4 + row_words * 4 + n_dig * (8 + 32) +
                              (size_t)nq * rounds * (8 + 16) + 4 * A * 4 + 3 * A * sizeof(GatherRowsJob) + 8192;
    TRY(cx.misc.ensure(misc_bytes));
    uint32_t* d_dig = cx.dig.as<uint32_t>();
    uint32_t* d_fdig = d_dig + n_trees * tree_words;
    bb::Ext* d_v = cx.ext.as<bb::Ext>();
    // misc layout (8-byte aligned pieces first)
    uint8_t* mp = cx.misc.as<uint8_t>();
    const uint32_t** d_cols = reinterpret_cast<const uint32_t**>(mp); mp += cols_total * 8;
    const uint32_t** d_sptrs = reinterpret_cast<const uint32_t**>(mp); mp += 4 * A * 8;
    GatherRowsJob* d_jobs = reinterpret_cast<GatherRowsJob*>(mp); mp += 3 * A * sizeof(GatherRowsJob);  // query phase: one per AIR and tree
    uint64_t* d_offs = reinterpret_cast<uint64_t*>(mp); mp += n_dig * 8 + (size_t)nq * rounds * 8;
    bb::Ext* d_opened = reinterpret_cast<bb::Ext*>(mp); mp += K_total * sizeof(bb::Ext);
    bb::Ext* d_gpow = reinterpret_cast<bb::Ext*>(mp); mp += K_total * sizeof(bb::Ext);
    uint32_t* d_idx = reinterpret_cast<uint32_t*>(mp); mp += (size_t)A * nq * 4;
    uint32_t* d_rows = reinterpret_cast<uint32_t*>(mp); mp += row_words * 4;
    uint32_t* d_dig_out = reinterpret_cast<uint32_t*>(mp); mp += n_dig * 32;
    uint32_t* d_ext_out = reinterpret_cast<uint32_t*>(mp); mp += (size_t)nq * rounds * 16;
    uint32_t* d_small = reinterpret_cast<uint32_t*>(mp);  // 4 A words (cumulative sums), PoW scratch

    // ---- which AIRs are streamed (sbv[a] = log2 of the number of sub-cosets, 0 = LDE resident) ------------------------------
    std::vector<int> sbv(A, 0);
    cx.last_plan[0] = cx.last_plan[1] = cx.last_plan[2] = 0;
    {
        auto may_stream = [&](size_t a) { return sh[a].log_h >= 3; };  // (AIRs that share a height: their level is hashed run by run)
        auto b_max = [&](size_t a) { return std::min((int)sh[a].log_h - 1, 5); };
        if (const char* e = getenv("POWDR_STREAM_LOG_BLOCKS")) {  // forced (tests): every AIR that may be streamed is
            const int v = atoi(e);
            for (size_t a = 0; a < A && v > 0; ++a) if (may_stream(a)) sbv[a] = std::min(v, b_max(a));
        } else {
            size_t held = cx.dig.bytes + cx.inject.bytes + cx.ext.bytes + cx.misc.bytes + cx.state.bytes, avail = 0;
            for (size_t a = 0; a < A; ++a) held += pw_prover_device_bytes(airs[a].prover);
            if (device_room(held, &avail)) {
                // what this call allocates: the segment's own arenas + per AIR exactly what ensure_air (resident) or ensure_proof_buffers
                // (streamed; a handed-over trace needs no tcoef) would (not the one-AIR plan, which counts trees and FRI layers
                // per AIR that a segment shares)
                const size_t own = (n_trees * tree_words + fri_words) * 4 + (Nmax + 1) * 32 + ext_words * sizeof(bb::Ext) + misc_bytes;
                size_t need = own;
                std::vector<size_t> bytes(A);
                for (size_t a = 0; a < A; ++a) { bytes[a] = plan_air(airs[a].prover, sh[a], lg).total(); need += bytes[a]; }
                cx.last_plan[0] = need;  // everything resident
                cx.last_plan[2] = avail;
                // Round by round: first the largest AIRs with TWO sub-cosets each until the segment fits; only when every AIR that may be
                // streamed is, four sub-cosets (largest first again), and so on. (Round 5's order — one AIR taken to 32 sub-cosets before
                // the next is touched — cost a C4 segment under a 0.8 budget 905 ms instead of 452: a sub-coset pass has a fixed cost per
                // column, and 2 sub-cosets of three AIRs re-read far less than 32 of one. profiles/r06_c4_budget_sweep.txt)
                std::vector<size_t> order(A);
                for (size_t a = 0; a < A; ++a) order[a] = a;
                std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return bytes[x] > bytes[y]; });
                for (int b = 1; b <= 5 && need > avail; ++b) {
                    for (size_t k = 0; k < A && need > avail; ++k) {
                        const size_t a = order[k];
                        if (!may_stream(a) || sh[a].log_h < stream_min_log_height() || b > b_max(a)) continue;
                        size_t nb = proof_plan_bytes(airs[a].prover, sh[a].log_h, b, handed[a] != 0);
                        nb += (size_t)16 * 4 * ((size_t)1 << sh[a].logN);  // + the parked sponge states of its level (cx.state)
                        if (nb >= bytes[a]) continue;
                        need = need - bytes[a] + nb;
                        bytes[a] = nb;
                        sbv[a] = b;
                    }
                }
                cx.last_plan[1] = need;  // as planned
            }
        }
    }
    std::vector<char> eat(A, 0);  // the trace is overwritten by its coefficient arrays
    for (size_t a = 0; a < A; ++a) eat[a] = handed[a] && sbv[a] > 0;
    cx.last_modes.assign(A, 0);
    for (size_t a = 0; a < A; ++a) cx.last_modes[a] = (uint32_t)sbv[a] | (eat[a] ? 0x100u : 0u);
    std::vector<CommitLayout> Lc(A);
    for (size_t a = 0; a < A; ++a) {
        if (sbv[a]) TRY(ensure_proof_buffers(airs[a].prover, sh[a].log_h, sbv[a], Lc[a], eat[a] != 0));
        else TRY(ensure_air(airs[a].prover, sh[a], lg, Lc[a]));
    }
    auto sctx = [&](size_t a) {
        return streamed::Ctx{airs[a].prover, sh[a].log_h, sbv[a], Lc[a].perm_panels, sh[a].H, sh[a].N, Lc[a].m, sh[a].W, sh[a].Wp};
    };

    std::vector<uint32_t>& pf = cx.proof;
    pf.clear();
    auto put = [&](uint32_t canonical) { pf.push_back(canonical); };
    auto put_monty = [&](const uint32_t* w, size_t n) { for (size_t i = 0; i < n; ++i) pf.push_back(bb::from_monty(w[i])); };
    Challenger ch;
    for (uint32_t x : {kMagic3, (uint32_t)A, lg ? 1u : 0u, cfg.num_queries, cfg.pow_bits}) { ch.observe_canonical(x % bb::P); put(x); }
    for (size_t a = 0; a < A; ++a)
        for (uint32_t x : {sh[a].log_h, sh[a].W, sh[a].nc, sh[a].n_int}) { ch.observe_canonical(x % bb::P); put(x); }

    // mixed commitment of one matrix per AIR: matrix(a) = (device pointer, width); heights are the AIRs' LDE heights
    std::vector<const uint32_t*> h_cols;
    auto commit_mixed = [&](auto matrix_of, int tree, uint32_t* root_monty) -> int {
        h_cols.clear();
        uint32_t* dg = d_dig + (size_t)tree * tree_words;
        std::vector<MixedLevelCols> by_log(L + 1, MixedLevelCols{nullptr, 0});
        // a level that holds a streamed matrix is hashed RUN BY RUN (merkle.hip leaf_absorb_kernel): consecutive resident matrices
        // through their slice of the column-pointer table, a streamed matrix sub-coset by sub-coset, the rows' sponge states parked
        // in cx.state in between; the level is then `external` for merkle_commit_mixed (which only injects / compresses)
        struct Run { size_t first_col; uint32_t n_cols; size_t air; bool streamed; };
        std::vector<std::vector<Run>> runs(L + 1);
        for (int k = L; k >= 0; --k) {
            const size_t first = h_cols.size();
            bool any_streamed = false;
            for (size_t a = 0; a < A; ++a) any_streamed = any_streamed || (sh[a].logN == k && sbv[a] && tree != 1);
            for (size_t a = 0; a < A; ++a) {
                if (sh[a].logN != k) continue;
                const uint32_t* m; uint32_t w;
                matrix_of(a, m, w);  // a streamed AIR hands over the matrix's COEFFICIENT arrays (column stride H)
                if (sbv[a] && tree != 1) { runs[k].push_back(Run{0, w, a, true}); continue; }
                if (any_streamed) {
                    if (runs[k].empty() || runs[k].back().streamed) runs[k].push_back(Run{h_cols.size(), 0, a, false});
                    runs[k].back().n_cols += w;
                }
                for (uint32_t c = 0; c < w; ++c) h_cols.push_back(m + (size_t)c * sh[a].N);
            }
            by_log[k] = MixedLevelCols{d_cols + first, (uint32_t)(h_cols.size() - first)};
            by_log[k].external = any_streamed ? 1u : 0u;
        }
        PW_HIP_TRY(hipMemcpyAsync(d_cols, h_cols.data(), h_cols.size() * 8, hipMemcpyHostToDevice, st));
        for (int k = L; k >= 0; --k) {
            if (runs[k].empty()) continue;
            const size_t rows = (size_t)1 << k;
            TRY(cx.state.ensure(rows * 16 * 4));
            uint32_t* level_digests = k == L ? dg : cx.inject.as<uint32_t>() + rows * 8;
            uint32_t pos = 0;
            for (size_t i = 0; i < runs[k].size(); ++i) {
                const Run& r = runs[k][i];
                const bool first_run = i == 0, last_run = i + 1 == runs[k].size();
                if (!r.streamed) {
                    TRY(merkle_leaf_absorb(nullptr, 0, d_cols + r.first_col, r.n_cols, pos, rows, 1, 0, cx.state.as<uint32_t>(), level_digests, first_run, last_run));
                } else {
                    const uint32_t* coef; uint32_t w;
                    matrix_of(r.air, coef, w);
                    const streamed::Ctx c = sctx(r.air);
                    TRY(streamed::for_each_subcoset(c, {streamed::CoefMatrix{coef, w}}, nullptr, [&](uint32_t sub) {
                        return merkle_leaf_absorb(c.p->lde.as<uint32_t>(), c.m, nullptr, w, pos, c.m, (size_t)1 << c.b, sub, cx.state.as<uint32_t>(), level_digests,
                                                  first_run, last_run);
                    }));
                }
                pos = (pos + r.n_cols) & 7u;
            }
        }
        TRY(merkle_commit_mixed(by_log.data(), L, dg, cx.inject.as<uint32_t>()));
        PW_HIP_TRY(hipMemcpyAsync(root_monty, dg + tree_words - 8, 32, hipMemcpyDeviceToHost, st));
        return 0;
    };

    // ---- 1. main traces ---------------------------------------------------------------------------------------
    // A streamed AIR's trace coefficients: in its prover's tcoef — or, handed over (eat), in the caller's own buffer from the moment
    // nothing reads the trace's VALUES any more. With LogUp the permutation columns are computed from the values after this
    // commitment, so the coefficients wait in the (still empty) permutation buffer until then (prover.hip prove_impl does the same).
    auto tcoef_of = [&](size_t a) { return eat[a] ? const_cast<uint32_t*>(airs[a].d_trace) : airs[a].prover->tcoef.as<uint32_t>(); };
    auto coef_first = [&](size_t a) { return eat[a] && lg ? airs[a].prover->perm.as<uint32_t>() : tcoef_of(a); };
    uint32_t root[8];
    TRY(fork());
    for (size_t a = 0; a < A; ++a) {
        PwProver* p = airs[a].prover;
        p->committed_trace = nullptr;
        on_air(a);
        if (sbv[a]) {  // streamed: the trace's coefficient arrays; its LDE rows exist one sub-coset at a time from here on
            TRY(intt_dif(airs[a].d_trace, coef_first(a), sh[a].H, sh[a].H, sh[a].W, (int)sh[a].log_h));
        } else {
            TRY(lde_matrix(p, Lc[a], sh[a].log_h, airs[a].d_trace, sh[a].W, p->lde.as<uint32_t>()));
        }
    }
    TRY(join());
    TRY(commit_mixed([&](size_t a, const uint32_t*& m, uint32_t& w) {
        m = sbv[a] ? coef_first(a) : airs[a].prover->lde.as<uint32_t>();
        w = sh[a].W;
    }, 0, root));
    PW_HIP_TRY(hipStreamSynchronize(st));
    put_monty(root, 8);
    ch.observe_words(root, 8);

    // per-AIR scratch pointers
    auto weights_of = [&](size_t a) { return airs[a].prover->ext_arena.as<bb::Ext>(); };
    auto scratch_of = [&](size_t a) { return airs[a].prover->misc.as<bb::Ext>(); };
    auto apow_of = [&](size_t a) { return scratch_of(a) + 2 * (size_t)std::max({sh[a].W, sh[a].Wp, 8u}) * div_up(sh[a].H, 8192); };
    auto blpow_of = [&](size_t a) { return apow_of(a) + sh[a].M + 4; };
    auto logup_program = [&](size_t a) {
        const PwProver* p = airs[a].prover;
        return LogupProgram{p->d_inter, sh[a].n_int, p->d_ixspans, p->d_icode, p->d_gstarts, sh[a].n_g, p->d_iforms};
    };

    // ---- 2. LogUp ---------------------------------------------------------------------------------------------
    bb::Ext al = bb::ext_zero(), bl = bb::ext_zero();
    std::vector<bb::Ext> S(A, bb::ext_zero());
    if (lg) {
        al = ch.sample_ext();
        bl = ch.sample_ext();
        std::vector<const uint32_t*> sp;
        TRY(fork());
        for (size_t a = 0; a < A; ++a) {
            PwProver* p = airs[a].prover;
            on_air(a);
            TRY(ext_powers(bl, p->max_args + 2, false, false, blpow_of(a)));  // beta^0 .. (computed where they are used: no upload per AIR)
            bb::Ext* rowsum = weights_of(a) + 2 * sh[a].H;
            // eat: the matrix's VALUES live in the sub-coset buffer (idle between two passes) until its coefficient arrays exist
            uint32_t* d_pval = eat[a] ? p->lde.as<uint32_t>() : p->perm.as<uint32_t>();
            if (specialised(p)) TRY(logup_perm_trace_jit(p, airs[a].d_trace, sh[a].H, al, blpow_of(a), d_pval, rowsum, rowsum + sh[a].H));
            else TRY(logup_perm_trace(airs[a].d_trace, sh[a].H, logup_program(a), al, blpow_of(a), d_pval, rowsum, rowsum + sh[a].H));
            if (sbv[a]) {
                // streamed: only phi and the per-row sums are extended for good (the boundary terms read them at rows j and j + 2); S is
                // saved before the matrix becomes its coefficient arrays in place
                uint32_t* d_perm_a = p->perm.as<uint32_t>();
                if (!specialised(p)) TRY(ext_to_cols(rowsum, sh[a].H, d_pval + (size_t)(4 * sh[a].n_g + 4) * sh[a].H));
                TRY(lde_matrix(p, Lc[a], sh[a].log_h, d_pval + (size_t)(4 * sh[a].n_g) * sh[a].H, 8, p->plde.as<uint32_t>()));
                uint32_t* d_save = p->gbuf.as<uint32_t>();
                for (int k = 0; k < 4; ++k) {
                    PW_HIP_TRY(hipMemcpyAsync(d_save + k, d_pval + ((size_t)(4 * sh[a].n_g + k) * sh[a].H + (sh[a].H - 1)), 4, hipMemcpyDeviceToDevice, stream()));
                    sp.push_back(d_save + k);
                }
                if (eat[a]) {
                    // the trace's values are dead now: its coefficient arrays move into its place, the permutation buffer takes the matrix's
                    PW_HIP_TRY(hipMemcpyAsync(tcoef_of(a), d_perm_a, (size_t)sh[a].W * sh[a].H * 4, hipMemcpyDeviceToDevice, stream()));
                    TRY(intt_dif(d_pval, d_perm_a, sh[a].H, sh[a].H, sh[a].Wp, (int)sh[a].log_h));
                } else TRY(intt_dif(d_perm_a, d_perm_a, sh[a].H, sh[a].H, sh[a].Wp, (int)sh[a].log_h));
                continue;
            }
            TRY(lde_matrix(p, Lc[a], sh[a].log_h, p->perm.as<uint32_t>(), sh[a].Wp + (specialised(p) ? kJitExtraPermCols : 0u), p->plde.as<uint32_t>()));
            for (int k = 0; k < 4; ++k) sp.push_back(p->perm.as<uint32_t>() + ((size_t)(4 * sh[a].n_g + k) * sh[a].H + (sh[a].H - 1)));  // S = phi(last row)
        }
        TRY(join());
        TRY(commit_mixed([&](size_t a, const uint32_t*& m, uint32_t& w) {
            m = sbv[a] ? airs[a].prover->perm.as<uint32_t>() : airs[a].prover->plde.as<uint32_t>();  // (streamed: the coefficient arrays)
            w = sh[a].Wp;
        }, 2, root));
        PW_HIP_TRY(hipMemcpyAsync(d_sptrs, sp.data(), sp.size() * 8, hipMemcpyHostToDevice, st));
        TRY(gather_words(d_sptrs, (uint32_t)sp.size(), d_small));
        std::vector<uint32_t> sw(4 * A);
        PW_HIP_TRY(hipMemcpyAsync(sw.data(), d_small, sw.size() * 4, hipMemcpyDeviceToHost, st));
        PW_HIP_TRY(hipStreamSynchronize(st));
        put_monty(root, 8);
        ch.observe_words(root, 8);
        for (size_t a = 0; a < A; ++a) {
            for (int k = 0; k < 4; ++k) S[a].c[k] = sw[4 * a + k];
            put_monty(S[a].c, 4);
            ch.observe_ext(S[a]);
        }
    }

    // ---- 3. quotients -----------------------------------------------------------------------------------------
    const bb::Ext alpha = ch.sample_ext();
    {
        const uint32_t s_m = bb::to_monty(field::kCosetShift), one = bb::R_MOD_P;
        TRY(fork());
        for (size_t a = 0; a < A; ++a) {
            PwProver* p = airs[a].prover;
            const Shape& s = sh[a];
            on_air(a);
            TRY(ext_powers(alpha, s.M, true, false, apow_of(a)));  // alpha^(M - 1) .. alpha^0
            uint32_t sH = s_m;
            for (uint32_t i = 0; i < s.log_h; ++i) sH = bb::sqr(sH);
            const uint32_t zv_even = bb::sub(sH, one), zv_odd = bb::sub(bb::neg(sH), one);
            ConstraintProgram prog{p->d_bytecode, p->d_spans, s.nc, p->is_xbc};
            uint32_t* d_q = p->q.as<uint32_t>();
            if (sbv[a]) {
                // streamed: the current-row terms sub-coset by sub-coset, then the boundary terms / the division by Z_H over all rows
                TRY(streamed::quotient_sums(sctx(a), specialised(p), lg, s.nc, prog, lg ? logup_program(a) : LogupProgram{}, tcoef_of(a),
                                            p->perm.as<uint32_t>(), apow_of(a), al, blpow_of(a), S[a], s.logN, d_q));
                if (lg)
                    TRY(quotient_logup_tail(d_q, 1, p->plde.as<uint32_t>(), p->plde.as<uint32_t>() + 4 * s.N, s.N, s.logN, apow_of(a) + s.nc + s.n_g, S[a],
                                            zv_even, zv_odd, d_q));
                else
                    TRY(quotient_combine(d_q, 1, s.N, bb::inv(zv_even), bb::inv(zv_odd), d_q));
            } else if (lg && specialised(p))
                TRY(quotient_eval_logup_jit(p, p->lde.as<uint32_t>(), p->plde.as<uint32_t>(), s.N, s.logN, apow_of(a), al, blpow_of(a), S[a], zv_even,
                                            zv_odd, d_q));
            else if (lg)
                TRY(quotient_eval_logup(p->lde.as<uint32_t>(), p->plde.as<uint32_t>(), s.N, s.logN, prog, logup_program(a), apow_of(a), al,
                                        blpow_of(a), S[a], zv_even, zv_odd, d_q));
            else if (specialised(p) && s.nc)
                TRY(quotient_eval_jit(p, p->lde.as<uint32_t>(), s.N, apow_of(a), bb::inv(zv_even), bb::inv(zv_odd), d_q));
            else
                TRY(quotient_eval(p->lde.as<uint32_t>(), s.N, prog, apow_of(a), bb::inv(zv_even), bb::inv(zv_odd), d_q, p->qpart.as<uint32_t>(),
                                  quotient_chunks(s.N, s.nc)));
            TRY(intt_dif(d_q, d_q, s.N, s.N, 4, s.logN));
            TRY(quotient_split(d_q, s.H, (int)s.log_h, p->qcoef.as<uint32_t>()));
            TRY(coset_lde_from_coeffs(p->qcoef.as<uint32_t>(), p->qlde.as<uint32_t>(), s.H, s.N, 8, (int)s.log_h));
        }
        TRY(join());
        TRY(commit_mixed([&](size_t a, const uint32_t*& m, uint32_t& w) { m = airs[a].prover->qlde.as<uint32_t>(); w = 8; }, 1, root));
        PW_HIP_TRY(hipStreamSynchronize(st));
    }
    put_monty(root, 8);
    ch.observe_words(root, 8);

    // ---- 4. openings: per AIR main | perm at zeta | quotient | perm at g zeta ----------------------------------
    const bb::Ext zeta = ch.sample_ext();
    std::vector<bb::Ext> gzeta(A);
    TRY(fork());
    for (size_t a = 0; a < A; ++a) {
        PwProver* p = airs[a].prover;
        const Shape& s = sh[a];
        on_air(a);
        bb::Ext* o = d_opened + s.koff;
        bb::Ext* w1 = weights_of(a);
        bb::Ext* w2 = w1 + s.H;
        gzeta[a] = bb::ext_scale(zeta, field::root_of_unity((int)s.log_h));
        // trace columns: barycentric evaluation straight from the caller's trace; quotient chunks from their coefficients
        if (!eat[a]) {
            TRY(barycentric_weights(zeta, (int)s.log_h, w1));
            TRY(ext_dot_columns(airs[a].d_trace, s.H, s.W, s.H, w1, o, scratch_of(a)));
        }
        if (lg && !sbv[a]) {  // the permutation matrix at zeta and at g zeta: one pass over its columns
            TRY(barycentric_weights(gzeta[a], (int)s.log_h, w2));
            TRY(ext_dot_columns2(p->perm.as<uint32_t>(), s.H, s.Wp, s.H, w1, w2, o + s.W, o + s.W + s.Wp + 8, scratch_of(a)));
        }
        TRY(zeta_weights(zeta, (int)s.log_h, w1));
        if (eat[a]) TRY(ext_dot_columns(tcoef_of(a), s.H, s.W, s.H, w1, o, scratch_of(a)));  // the trace is its coefficient arrays by now
        if (lg && sbv[a]) {  // streamed: p->perm holds the matrix's coefficient arrays
            TRY(zeta_weights(gzeta[a], (int)s.log_h, w2));
            TRY(ext_dot_columns2(p->perm.as<uint32_t>(), s.H, s.Wp, s.H, w1, w2, o + s.W, o + s.W + s.Wp + 8, scratch_of(a)));
        }
        TRY(ext_dot_columns(p->qcoef.as<uint32_t>(), s.H, 8, s.H, w1, o + s.W + s.Wp, scratch_of(a)));
    }
    TRY(join());
    std::vector<bb::Ext> opened(K_total);
    PW_HIP_TRY(hipMemcpyAsync(opened.data(), d_opened, K_total * sizeof(bb::Ext), hipMemcpyDeviceToHost, st));
    PW_HIP_TRY(hipStreamSynchronize(st));
    for (const bb::Ext& e : opened) { put_monty(e.c, 4); ch.observe_ext(e); }

    // ---- 5. reduced openings per height ------------------------------------------------------------------------
    const bb::Ext gamma = ch.sample_ext();
    std::vector<bb::Ext> gpow(K_total);
    { bb::Ext g = bb::ext_one(); for (auto& x : gpow) { x = g; g = bb::ext_mul(g, gamma); } }
    // the DEEP kernels take the powers as CENTRED representatives (signed 64-bit accumulation, bb::ExtCentredAcc); the host sums
    // below use the canonical ones
    std::vector<bb::Ext> gpow_c(gpow);
    for (auto& e : gpow_c) for (auto& c : e.c) c = (uint32_t)bb::centred(c);
    PW_HIP_TRY(hipMemcpyAsync(d_gpow, gpow_c.data(), K_total * sizeof(bb::Ext), hipMemcpyHostToDevice, st));
    {
        std::vector<char> started(L + 1, 0);
        for (size_t a = 0; a < A; ++a) {
            PwProver* p = airs[a].prover;
            const Shape& s = sh[a];
            const size_t K1 = (size_t)s.W + s.Wp + 8;
            bb::Ext sum1 = bb::ext_zero(), sum2 = bb::ext_zero();
            for (size_t k = 0; k < K1; ++k) sum1 = bb::ext_add(sum1, bb::ext_mul(gpow[s.koff + k], opened[s.koff + k]));
            for (size_t k = K1; k < s.K; ++k) sum2 = bb::ext_add(sum2, bb::ext_mul(gpow[s.koff + k], opened[s.koff + k]));
            bb::Ext* target = s.logN == L ? d_v : d_v + ro_off[s.logN];
            bb::Ext* out = started[s.logN] ? d_v + tmp_off : target;
            if (sbv[a])
                TRY(streamed::deep_from_coefficients(sctx(a), lg, tcoef_of(a), p->perm.as<uint32_t>(), p->qlde.as<uint32_t>(), s.logN,
                                                     d_gpow + s.koff, [] {}, sum1, sum2, zeta, gzeta[a], out));
            else if (s.log_h >= kDeepComboMinLogHeight && !getenv("POWDR_DEEP_DIRECT") && !((uintptr_t)airs[a].d_trace & 7)) {
                // resident and tall: the numerator is combined on the evaluations over <g_n> — the caller's trace, the permutation
                // matrix: half the bytes their LDE holds — and extended as 4 (+ 4) columns, like the one-AIR prover does
                uint32_t* d_gev = p->gbuf.as<uint32_t>();
                uint32_t* d_glde = d_gev + 8 * s.H;
                TRY(ext_lincomb(airs[a].d_trace, s.W, p->perm.as<uint32_t>(), s.Wp, s.H, d_gpow + s.koff, lg ? (uint32_t)K1 : 0u, d_gev));
                TRY(lde_matrix(p, Lc[a], s.log_h, d_gev, lg ? 8u : 4u, d_glde));
                TRY(deep_from_combo(d_glde, p->qlde.as<uint32_t>(), s.N, s.logN, d_gpow + s.koff + s.W + s.Wp, sum1, sum2, zeta, gzeta[a], lg ? 1 : 0, out));
            } else if (lg)
                TRY(deep_quotient_logup(p->lde.as<uint32_t>(), s.W, p->plde.as<uint32_t>(), s.Wp, p->qlde.as<uint32_t>(), s.N, s.logN,
                                        d_gpow + s.koff, sum1, sum2, zeta, gzeta[a], out));
            else
                TRY(deep_quotient(p->lde.as<uint32_t>(), s.W, p->qlde.as<uint32_t>(), 8, s.N, s.logN, d_gpow + s.koff, sum1, zeta, out));
            if (started[s.logN]) TRY(ext_axpy(target, nullptr, out, s.N));
            started[s.logN] = 1;
        }
    }
    PW_HIP_TRY(hipStreamSynchronize(st));  // gpow is a local

    // ---- 6. FRI over the unshifted subgroups, smaller heights rolled in ------------------------------------------
    for (int l = 0; l < rounds; ++l) {
        const size_t half = (Nmax >> l) / 2;
        bb::Ext* v = d_v + layer_off[l];
        uint32_t* dg = d_fdig + ftree_off[l];
        // the tree's tail kernel writes the root into host-mapped memory: no copy dispatch between the layers
        uint32_t* d_mail = nullptr;
        uint32_t* h_mail = merkle_root_mailbox(&d_mail);
        TRY(merkle_commit_ext_pairs(v, half, dg, h_mail ? d_mail : nullptr));
        if (!h_mail) PW_HIP_TRY(hipMemcpyAsync(root, dg + merkle_words(half) - 8, 32, hipMemcpyDeviceToHost, st));
        PW_HIP_TRY(hipStreamSynchronize(st));
        if (h_mail) memcpy(root, h_mail, 32);
        put_monty(root, 8);
        ch.observe_words(root, 8);
        const bb::Ext beta = ch.sample_ext();
        bb::Ext* nv = d_v + layer_off[l + 1];
        TRY(fri_fold(v, half, L - l, bb::R_MOD_P, beta, nv));
        const int lk = L - l - 1;
        if (lk >= 2 && has_height[lk]) {
            const bb::Ext beta2 = bb::ext_mul(beta, beta);
            TRY(ext_axpy(nv, &beta2, d_v + ro_off[lk], half));
        }
    }
    bb::Ext final_poly;
    PW_HIP_TRY(hipMemcpyAsync(&final_poly, d_v + layer_off[rounds], sizeof(bb::Ext), hipMemcpyDeviceToHost, st));
    PW_HIP_TRY(hipStreamSynchronize(st));
    put_monty(final_poly.c, 4);
    ch.observe_ext(final_poly);

    // ---- 7. proof of work ----------------------------------------------------------------------------------------
    uint32_t witness = 0;
    if (cfg.pow_bits) {
        uint32_t* d_state = d_small;
        uint32_t pending[8] = {0};
        for (size_t i = 0; i < ch.in.size(); ++i) pending[i] = ch.in[i];
        PW_HIP_TRY(hipMemcpyAsync(d_state, ch.st, 64, hipMemcpyHostToDevice, st));
        PW_HIP_TRY(hipMemcpyAsync(d_state + 16, pending, 32, hipMemcpyHostToDevice, st));
        TRY(pow_grind(d_state, d_state + 16, (uint32_t)ch.in.size(), cfg.pow_bits, d_state + 24, &witness));
    }
    put(witness);
    ch.observe_canonical(witness);
    if (cfg.pow_bits) (void)ch.sample_bits((int)cfg.pow_bits);

    // ---- 8. queries ------------------------------------------------------------------------------------------------
    if (nq) {
        std::vector<uint32_t> qs(nq), idx((size_t)A * nq);
        for (auto& q : qs) q = ch.sample_bits(L);
        for (size_t a = 0; a < A; ++a)
            for (uint32_t i = 0; i < nq; ++i) idx[a * nq + i] = qs[i] & (uint32_t)(sh[a].N - 1);
        PW_HIP_TRY(hipMemcpyAsync(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, st));
        // rows: tree by tree (proof order main, perm, quotient), AIR by AIR, nq rows each
        const int tree_of_phase[3] = {0, 2, 1};  // digest arena order: main | quotient | perm
        std::vector<size_t> row_off[3];
        std::vector<GatherRowsJob> jobs;  // (alive until the synchronisation below)
        uint32_t max_w = 0;
        size_t ro = 0;
        TRY(fork());  // after the index upload
        for (int ph = 0; ph < 3; ++ph) {
            if (ph == 1 && !lg) continue;
            row_off[ph].resize(A);
            for (size_t a = 0; a < A; ++a) {
                PwProver* p = airs[a].prover;
                const uint32_t* m = ph == 0 ? p->lde.as<uint32_t>() : ph == 1 ? p->plde.as<uint32_t>() : p->qlde.as<uint32_t>();
                const uint32_t w = ph == 0 ? sh[a].W : ph == 1 ? sh[a].Wp : 8u;
                row_off[ph][a] = ro;
                on_air(a);
                if (sbv[a] && ph != 2) {
                    // streamed: the rows are rebuilt from the coefficient arrays (index scratch: the AIR's own gbuf, free by now)
                    uint32_t* d_scr = p->gbuf.as<uint32_t>();
                    TRY(streamed::query_rows(sctx(a), ph == 0 ? tcoef_of(a) : p->perm.as<uint32_t>(), w, idx.data() + a * nq, nq, d_scr,
                                             d_scr + nq, d_rows + ro));
                } else {
                    jobs.push_back(GatherRowsJob{m, (uint64_t)sh[a].N, w, (uint32_t)(a * nq), (uint64_t)ro});
                    max_w = std::max(max_w, w);
                }
                ro += (size_t)nq * w;
            }
        }
        TRY(join());
        if (!jobs.empty()) {  // the resident matrices' rows: one launch for all AIRs and trees
            PW_HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(GatherRowsJob), hipMemcpyHostToDevice, st));
            TRY(gather_rows_multi(d_jobs, (uint32_t)jobs.size(), max_w, d_idx, nq, d_rows));
        }
        std::vector<uint64_t> dig_offs, ext_offs;
        for (uint32_t qi = 0; qi < nq; ++qi) {
            const size_t q = qs[qi];
            for (int ph = 0; ph < 3; ++ph) {
                if (ph == 1 && !lg) continue;
                const size_t base = (size_t)tree_of_phase[ph] * tree_words;
                for (int l = 0; l < L; ++l) {
                    const size_t size = Nmax >> l, p = q & (size - 1);
                    dig_offs.push_back(base + merkle_level_offset(Nmax, l) + (((p + size / 2) & (size - 1)) * 8));
                }
            }
            for (int l = 0; l < rounds; ++l) {
                const size_t Nl = Nmax >> l, half = Nl / 2, pp = q & (Nl - 1);
                ext_offs.push_back((layer_off[l] + (pp ^ half)) * 4);
                const size_t leaf = pp & (half - 1);
                for (int lv = 0; lv < L - 1 - l; ++lv)
                    dig_offs.push_back((size_t)n_trees * tree_words + ftree_off[l] + merkle_level_offset(half, lv) + (((leaf >> lv) ^ 1) * 8));
            }
        }
        const size_t nd = dig_offs.size(), ne = ext_offs.size();
        if (nd > n_dig) return (int)hipErrorInvalidValue;
        uint64_t* d_dig_offs = d_offs;
        uint64_t* d_ext_offs = d_offs + nd;
        PW_HIP_TRY(hipMemcpyAsync(d_dig_offs, dig_offs.data(), nd * 8, hipMemcpyHostToDevice, st));
        if (ne) PW_HIP_TRY(hipMemcpyAsync(d_ext_offs, ext_offs.data(), ne * 8, hipMemcpyHostToDevice, st));
        TRY(gather_records(d_dig, d_dig_offs, 8u, (uint32_t)nd, d_dig_out));
        TRY(gather_records(reinterpret_cast<const uint32_t*>(d_v), d_ext_offs, 4u, (uint32_t)ne, d_ext_out));
        // the answers leave the device as canonical words
        TRY(canonicalize_words(d_rows, ro));
        TRY(canonicalize_words(d_dig_out, nd * 8));
        TRY(canonicalize_words(d_ext_out, ne * 4));
        std::vector<uint32_t> rows(ro + 1), dig(nd * 8 + 1), ext(ne * 4 + 1);
        PW_HIP_TRY(hipMemcpyAsync(rows.data(), d_rows, ro * 4, hipMemcpyDeviceToHost, st));
        PW_HIP_TRY(hipMemcpyAsync(dig.data(), d_dig_out, nd * 32, hipMemcpyDeviceToHost, st));
        if (ne) PW_HIP_TRY(hipMemcpyAsync(ext.data(), d_ext_out, ne * 16, hipMemcpyDeviceToHost, st));
        PW_HIP_TRY(hipStreamSynchronize(st));
        auto put_raw = [&](const uint32_t* w, size_t n) { pf.insert(pf.end(), w, w + n); };
        pf.reserve(pf.size() + nq + ro + nd * 8 + ne * 4);
        size_t dpos = 0, epos = 0;
        for (uint32_t qi = 0; qi < nq; ++qi) {
            put(qs[qi]);
            for (int ph = 0; ph < 3; ++ph) {
                if (ph == 1 && !lg) continue;
                for (size_t a = 0; a < A; ++a) {
                    const uint32_t w = ph == 0 ? sh[a].W : ph == 1 ? sh[a].Wp : 8u;
                    put_raw(&rows[row_off[ph][a] + (size_t)qi * w], w);
                }
                put_raw(&dig[dpos * 8], (size_t)L * 8);
                dpos += L;
            }
            for (int l = 0; l < rounds; ++l) {
                put_raw(&ext[epos * 4], 4);
                epos += 1;
                const size_t depth = (size_t)L - 1 - l;
                put_raw(&dig[dpos * 8], depth * 8);
                dpos += depth;
            }
        }
    }
    *proof_words = pf.data();
    *n_words = pf.size();
    return (int)hipGetLastError();
}

extern "C" int pw_prove_segment(const PwSegmentAir* airs, size_t n_airs, int logup_flag, const uint32_t** proof_words, size_t* n_words) {
    return prove_segment_impl(airs, n_airs, logup_flag, false, proof_words, n_words);
}

extern "C" int pw_prove_segment_consuming(const PwSegmentAir* airs, size_t n_airs, int logup_flag, const uint32_t** proof_words, size_t* n_words) {
    return prove_segment_impl(airs, n_airs, logup_flag, true, proof_words, n_words);
}

// the mode the AIRs of the calling thread's LAST segment proof ran in: out[a] = log2 of the number of sub-cosets (0 = resident),
// + 0x100 when the trace was overwritten by its coefficient arrays (pw_trace_from_coefficients restores it)
extern "C" void pw_segment_last_plan(size_t* resident_bytes, size_t* planned_bytes, size_t* available_bytes) {
    if (resident_bytes) *resident_bytes = g_ctx.last_plan[0];
    if (planned_bytes) *planned_bytes = g_ctx.last_plan[1];
    if (available_bytes) *available_bytes = g_ctx.last_plan[2];
}

extern "C" size_t pw_segment_last_modes(uint32_t* out, size_t cap) {
    const std::vector<uint32_t>& m = g_ctx.last_modes;
    for (size_t a = 0; a < m.size() && a < cap; ++a) out[a] = m[a];
    return m.size();
}
