// State shared by the prover's translation units (prover.hip: one AIR = one proof; segment_prover.hip: one proof per
// segment): the per-AIR prover object, the host transcript, a growable device buffer. Not part of the C ABI.
#pragma once
#include "prover_internal.hpp"
#include "jit_codegen.hpp"
#include "jit.hpp"
#include "../../include/powdr_prover.h"

#include <cstring>
#include <vector>

namespace pw {

constexpr uint32_t kMagic = 0x31535750u;   // "PWS1"
constexpr uint32_t kMagic2 = 0x32535750u;  // "PWS2": with the LogUp extension
constexpr uint32_t kMagic3 = 0x33535750u;  // "PWS3": one proof per segment (segment_prover.hip)

// ---- duplex-sponge challenger on Montgomery words (spec: oracle/stark_oracle.cpp Challenger) ----
struct Challenger {
    uint32_t st[16];
    std::vector<uint32_t> in, out;
    Challenger() { memset(st, 0, sizeof st); }
    void duplex() {
        for (size_t i = 0; i < in.size(); ++i) st[i] = in[i];
        in.clear();
        p2::permute(st, poseidon2_params_host());
        out.assign(st, st + 8);
    }
    void observe(uint32_t m) { out.clear(); in.push_back(m); if (in.size() == 8) duplex(); }
    void observe_canonical(uint32_t c) { observe(bb::to_monty(c)); }
    void observe_words(const uint32_t* w, size_t n) { for (size_t i = 0; i < n; ++i) observe(w[i]); }
    void observe_ext(const bb::Ext& e) { observe_words(e.c, 4); }
    uint32_t sample() { if (!in.empty() || out.empty()) duplex(); uint32_t v = out.back(); out.pop_back(); return v; }
    bb::Ext sample_ext() { bb::Ext e; for (int i = 0; i < 4; ++i) e.c[i] = sample(); return e; }
    uint32_t sample_bits(int b) { return bb::from_monty(sample()) & ((1u << b) - 1u); }
};

struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int device = -1;  // the device the buffer lives on: a host thread (thread_local contexts) may have moved to another GPU
    int ensure(size_t need) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (p && dev == device && need <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        e = hipMalloc(&p, need);
        if (e != hipSuccess) return (int)e;
        bytes = need;
        device = dev;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
    DeviceBuf() = default;
    DeviceBuf(const DeviceBuf&) = delete;
    DeviceBuf& operator=(const DeviceBuf&) = delete;
    ~DeviceBuf() { release(); }  // thread_local owners (per-thread segment contexts) give their memory back when the thread exits
};

// out[i] = arena[offsets[r] + w] for record r, word w (query answers: digests, FRI siblings)
int gather_records(const uint32_t* arena, const uint64_t* d_offsets, uint32_t words_per_record, uint32_t n, uint32_t* out);

}  // namespace pw

struct PwProver {
    PwStarkConfig cfg;
    uint32_t width;
    std::vector<uint32_t> h_spans;
    uint32_t n_constraints;
    uint32_t* d_bytecode = nullptr;
    uint32_t* d_spans = nullptr;
    bool is_xbc = false;  // d_bytecode/d_spans hold plan-compiled xbc code (xbc.hpp) instead of post-fix code
    int max_degree = 0;   // highest degree among the constraint programs (99 = a malformed / non-polynomial one)
    // LogUp extension (pw_prover_create_logup): the AIR's bus interactions as xbc programs
    bool logup = false;
    uint32_t n_inter = 0, n_groups = 0, max_args = 0;
    uint32_t* d_gstarts = nullptr;  // group boundaries (logup_groups.hpp)
    pw::LogupInteraction* d_inter = nullptr;
    uint32_t* d_ixspans = nullptr;
    uint32_t* d_icode = nullptr;
    pw::SmallForm* d_iforms = nullptr;  // all spans as small forms, when every one of them is one (else nullptr)
    pw::DeviceBuf perm, plde;
    bool has_bus_seed = false;
    uint32_t bus_seed[8] = {0};  // Montgomery
    // pw_prover_trace_root leaves the trace's LDE and Merkle tree in `lde` / `digests`; a pw_prover_prove of the
    // same (pointer, height) right after it starts from them instead of recomputing (one-shot)
    const uint32_t* committed_trace = nullptr;
    uint32_t committed_log_h = 0;
    uint32_t committed_root[8] = {0};  // Montgomery
    int committed_b = 0;               // the mode that commitment was made in (0 resident, b >= 1 streamed over 2^b sub-cosets)
    // device buffers, grown on demand
    pw::DeviceBuf coef, lde, digests, q, qcoef, qlde, ext_arena, misc;
    pw::DeviceBuf qpart;  // partial quotient sums when the constraint list is split over workgroup rows (short traces)
    // streamed mode (prover.hip "streamed proofs"): the trace's coefficient arrays, the scale table of the sub-coset being evaluated,
    // the DEEP combinations (8 columns of coefficients + their LDE); `lde` then holds ONE sub-coset of all committed columns
    pw::DeviceBuf tcoef, fscale, gbuf;
    // pw_prover_prove step 3: host-mapped landing place of the opened values — the dot-product kernels write them straight into
    // host memory, slice by slice with an event after each, and the host absorbs a slice into the transcript while the device
    // computes the next (4 487 sequential transcript permutations at BASELINE configs[1]); host == nullptr: plain copy afterwards
    struct OpenedMailbox {
        static constexpr int kEvents = 8;
        bb::Ext* host = nullptr;
        bb::Ext* dev = nullptr;
        size_t cap = 0;
        int device = -1;
        hipEvent_t ev[kEvents] = {};
        int n_events = 0;
    } opened_mb;
    std::vector<uint32_t> proof;
    // host copies of the plan-compiled (xbc) programs: the source of the run-time specialised kernels (jit_codegen.hpp)
    std::vector<uint32_t> h_xcode, h_xspans, h_icode, h_ixspans, h_gstarts;
    std::vector<pw::LogupInteraction> h_inter;
    struct Specialised {
        int state = 0;  // 0: not tried, 1: ready, -1: not available (no hiprtc, POWDR_JIT=0, post-fix fallback programs, compile error)
        pw::jit::Generated quotient, perm;
        std::vector<pw::jit::ProgramPtr> quotient_prog, perm_prog;  // one per unit
        std::string error;
    } jit;
};

namespace pw {
struct CommitLayout {
    size_t H, N, tree_words, fri_words, n_trees, panel_cols;
    int b = 0;     // streamed mode: the extended domain is walked as 2^b sub-cosets (0: the LDE is resident)
    size_t m = 0;  // rows of a sub-coset, N >> b
    bool perm_panels = false;  // streamed quotient: the permutation columns come in unit by unit (specialised LogUp kernels) instead of all at once
};
// traces of at least 2^this rows: the DEEP numerator is combined on the un-extended matrices and extended as 4 (+ 4) columns through
// PwProver::gbuf (24 words per row); shorter ones accumulate it over the LDE (prover.hip "streamed proofs")
constexpr uint32_t kDeepComboMinLogHeight = 16;
// LDE of a column-major matrix (cols x 2^log_h) through the prover's coefficient panel buffer into `out` (cols x 2^(log_h+1))
int lde_matrix(PwProver* p, const CommitLayout& L, uint32_t log_h, const uint32_t* m, uint32_t cols, uint32_t* out);
// columns per LDE panel for a 2^log_h-row matrix of `widest` columns (POWDR_PANEL_LOG_WORDS, read per call)
size_t lde_panel_cols(size_t H, size_t widest);

// ---- streamed proofs: what the segment prover needs from prover.hip ("streamed proofs") ---------------------------------
// device bytes a proof of a 2^log_h-row trace of this AIR needs with the LDE resident (b = 0) or streamed over 2^b sub-cosets
// (consume: the caller hands the trace over — a streamed proof keeps the trace's coefficient arrays in the caller's buffer, no tcoef)
size_t proof_plan_bytes(const PwProver* p, uint32_t log_h, int b, bool consume = false);
// every buffer of such a proof (grown on demand), the layout in L (b, m, perm_panels, panel_cols)
int ensure_proof_buffers(PwProver* p, uint32_t log_h, int b, CommitLayout& L, bool consume = false);
// traces shorter than 2^this are never streamed by the automatic policy (a few MB either way); POWDR_STREAM_MIN_LOG_HEIGHT: tests
inline uint32_t stream_min_log_height() {
    const char* e = getenv("POWDR_STREAM_MIN_LOG_HEIGHT");
    return e ? (uint32_t)atoi(e) : 16u;
}
// bytes a proof may plan for on the current device when the caller's provers hold `held` now (prover.hip; pw_set_device_budget)
bool device_room(size_t held, size_t* avail);
size_t device_budget();

// ---- run-time specialised expression kernels (prover_jit.hip) ---------------------------------------------------------
// Compile the specialised kernels of every prover in `ps` that qualifies and has none yet (all translation units of all of
// them in ONE concurrent hiprtc batch). force: regardless of the trace height. Returns 0; failures are not errors — the
// prover keeps using the interpreter and records why (jit.error).
int specialise_provers(PwProver* const* ps, size_t n, const uint32_t* log_heights, bool force);
pw::jit::Generated generate_sources(const PwProver* p, int which, uint32_t chunk_cost, uint32_t chunks_per_unit);  // which: 0 quotient, 1 LogUp permutation
inline bool specialised(const PwProver* p) { return p->jit.state == 1; }
// the three stages with the specialised kernels; same contracts as quotient_eval / quotient_eval_logup / logup_perm_trace.
// `perm` / `plde` must have room for 4 extra columns after the committed ones (the per-row sums and their LDE).
int quotient_eval_jit(PwProver* p, const uint32_t* lde, size_t N, const bb::Ext* d_apow, uint32_t zinv_even, uint32_t zinv_odd, uint32_t* q);
// the chunk kernels of the specialised quotient alone, on `rows` rows of (T | Pm) with column stride `rows`: partial sums
// part[(c * 4 + k) * rows + j], c < *n_chunks (the streamed path runs them per sub-coset; Pm / d_blpow: nullptr without LogUp)
int quotient_parts_jit(PwProver* p, const uint32_t* T, const uint32_t* Pm, size_t rows, const bb::Ext* d_apow, bb::Ext al, const bb::Ext* d_blpow,
                       uint32_t* part, uint32_t* n_chunks);
// The same unit by unit (the streamed prover extends only the permutation columns a unit reads): quotient_units_jit = number of units;
// quotient_unit_groups: the LogUp groups [*g0, *g1) unit u covers (equal: none); quotient_unit_jit launches unit u — Pm is the address
// permutation column 0 WOULD have (the caller's panel holds columns 4 g0 .. 4 g1 - 1 at Pm + 4 g0 * rows), every chunk of the unit
// writes part[(chunk * 4 + k) * rows + j] at its own chunk index, as quotient_parts_jit does.
uint32_t quotient_units_jit(const PwProver* p);
void quotient_unit_groups(const PwProver* p, uint32_t u, uint32_t* g0, uint32_t* g1);
int quotient_unit_jit(PwProver* p, uint32_t u, const uint32_t* T, const uint32_t* Pm, size_t rows, const bb::Ext* d_apow, bb::Ext al,
                      const bb::Ext* d_blpow, uint32_t* part);
// widest permutation-column range (in columns) any unit of the specialised quotient reads
uint32_t quotient_max_unit_perm_cols(const PwProver* p);
// bytes of `qpart` the specialised kernels need for a trace of H rows whose quotient is evaluated `q_rows` rows at a time
size_t jit_part_bytes(const PwProver* p, size_t H, size_t q_rows);
int quotient_eval_logup_jit(PwProver* p, const uint32_t* lde, const uint32_t* plde, size_t N, int logN, const bb::Ext* d_apow, bb::Ext al,
                            const bb::Ext* d_blpow, bb::Ext S, uint32_t zval_even, uint32_t zval_odd, uint32_t* q);
int logup_perm_trace_jit(PwProver* p, const uint32_t* trace, size_t H, bb::Ext al, const bb::Ext* d_blpow, uint32_t* perm, bb::Ext* d_rowsum,
                         bb::Ext* d_block_totals);
// extra (uncommitted) columns the specialised path keeps after the 4 (groups + 1) permutation columns
constexpr uint32_t kJitExtraPermCols = 4;
}  // namespace pw
