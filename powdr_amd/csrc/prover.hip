// pw-stark v0 prover: host orchestration of the HIP stages, Fiat-Shamir transcript, proof
// assembly. Protocol definition: oracle/stark_oracle.cpp (header comment) and DESIGN.md;
// the byte string produced here must equal the oracle's for the same trace.
//
// Data stays in HBM from the caller's trace to the last FRI layer; the host only sees
// 32-byte roots, the W+8 opened values and the query answers (a few hundred KB), which is
// what it needs to run the transcript.
#include "prover_state.hpp"
#include "prover_stream.hpp"
#include "logup_groups.hpp"
#include "xbc_compile.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <initializer_list>
#include <vector>

namespace pw {

namespace {

__global__ void gather_records_kernel(const uint32_t* __restrict__ arena, const uint64_t* __restrict__ offsets,
                                      uint32_t words_per_record, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * words_per_record) return;
    const uint32_t r = i / words_per_record, w = i - r * words_per_record;
    out[i] = arena[offsets[r] + w];
}

}  // namespace

int gather_records(const uint32_t* arena, const uint64_t* d_offsets, uint32_t words_per_record, uint32_t n, uint32_t* out) {
    if (!n) return 0;
    hipLaunchKernelGGL(gather_records_kernel, dim3(div_up((size_t)n * words_per_record, 256)), dim3(256), 0, stream(), arena, d_offsets,
                       words_per_record, n, out);
    return (int)hipGetLastError();
}

}  // namespace pw

using namespace pw;

extern "C" void pw_prover_destroy(PwProver* p);
extern "C" size_t pw_prover_device_bytes(const PwProver* p);
extern "C" int pw_prover_specialise(PwProver* p);
extern "C" int pw_prover_specialised(const PwProver* p, size_t* n_kernels, size_t* code_bytes, size_t* n_chunks);

// `device` = false: host tables only (pw_jit_compile_check: code generation + hiprtc need no GPU)
static PwProver* create_prover(const PwStarkConfig* cfg, uint32_t width, const uint32_t* bc, size_t bc_len,
                               const uint32_t* spans, size_t n_constraints, bool device) {
    if (!cfg || !width) return nullptr;
    PwProver* p = new PwProver();
    p->cfg = *cfg;
    p->width = width;
    p->n_constraints = (uint32_t)n_constraints;
    p->h_spans.assign(spans, spans + 2 * n_constraints);
    for (size_t k = 0; k < n_constraints; ++k) {
        const uint32_t off = spans[2 * k], len = spans[2 * k + 1];
        const int d = (size_t)off + len <= bc_len ? pw::postfix_degree(bc + off, len) : pw::kBadDegree;
        // a span past the bytecode, an unknown opcode, an unbalanced or too deep stack, a column the trace does not have: no prover — the
        // post-fix fallback below is for well-formed programs the xbc compiler declines, not for these (they would index past the
        // bytecode or the trace on the device)
        if (d == pw::kBadDegree || !pw::postfix_columns_below(bc + off, len, width)) { delete p; return nullptr; }
        if (d > p->max_degree) p->max_degree = d;
    }
    // compile the post-fix constraint programs to xbc (xbc.hpp); fall back to the post-fix interpreter if
    // any program is malformed or too deep
    std::vector<uint32_t> code, xspans;
    bool ok = getenv("POWDR_QUOTIENT_XBC") ? atoi(getenv("POWDR_QUOTIENT_XBC")) != 0 : true;
    {
        xbc::Compiler cc;
        for (size_t k = 0; k < n_constraints && ok; ++k) {
            const uint32_t off = spans[2 * k], len = spans[2 * k + 1];
            if ((size_t)off + len > bc_len) { ok = false; break; }
            const uint32_t o = (uint32_t)(code.size() / 2);
            if (!cc.compile(bc + off, len, code)) { ok = false; break; }
            xspans.push_back(o);
            xspans.push_back((uint32_t)(code.size() / 2) - o);
        }
    }
    const uint32_t* up_bc = ok ? code.data() : bc;
    const uint32_t* up_sp = ok ? xspans.data() : spans;
    const size_t up_bc_len = ok ? code.size() : bc_len;
    p->is_xbc = ok;
    if (ok) { p->h_xcode = code; p->h_xspans = xspans; }
    if (!device) return p;
    size_t bl = up_bc_len ? up_bc_len : 1, sl = n_constraints ? 2 * n_constraints : 1;
    if (hipMalloc(&p->d_bytecode, (bl + 2) * 4) != hipSuccess || hipMalloc(&p->d_spans, sl * 4) != hipSuccess) {
        delete p;
        return nullptr;
    }
    if (up_bc_len) (void)hipMemcpy(p->d_bytecode, up_bc, up_bc_len * 4, hipMemcpyHostToDevice);
    if (n_constraints) (void)hipMemcpy(p->d_spans, up_sp, 2 * n_constraints * 4, hipMemcpyHostToDevice);
    return p;
}

extern "C" PwProver* pw_prover_create(const PwStarkConfig* cfg, uint32_t width, const uint32_t* bc, size_t bc_len,
                                      const uint32_t* spans, size_t n_constraints) {
    return create_prover(cfg, width, bc, bc_len, spans, n_constraints, true);
}

static PwProver* create_prover_logup(const PwStarkConfig* cfg, uint32_t width, const uint32_t* bc, size_t bc_len,
                                     const uint32_t* spans, size_t n_constraints, const uint32_t* inter, size_t n_inter,
                                     const uint32_t* ispans, size_t n_ispans, const uint32_t* ibc, size_t ibc_len, bool device) {
    PwProver* p = create_prover(cfg, width, bc, bc_len, spans, n_constraints, device);
    if (!p) return nullptr;
    // interactions: {bus, n_args, first span}; spans [mult, arg0, ...] into ibc (post-fix, column operands)
    std::vector<pw::LogupInteraction> li(n_inter);
    std::vector<uint32_t> xspans, code;
    std::vector<pw::SmallForm> forms;  // one per span; the few that are no small forms (sums of many flags, ...) stay with the interpreter
    size_t not_small = 0;
    xbc::Compiler cc;
    bool ok = true;
    for (size_t i = 0; i < n_inter && ok; ++i) {
        const uint32_t bus = inter[3 * i], na = inter[3 * i + 1], first = inter[3 * i + 2];
        if ((size_t)first + 1 + na > n_ispans) { ok = false; break; }
        li[i] = {bb::to_monty(bus % bb::P), na, (uint32_t)(xspans.size() / 2)};
        if (na > p->max_args) p->max_args = na;
        for (uint32_t k = 0; k <= na && ok; ++k) {
            const uint32_t off = ispans[2 * (first + k)], len = ispans[2 * (first + k) + 1];
            if ((size_t)off + len > ibc_len) { ok = false; break; }
            if (pw::postfix_degree(ibc + off, len) == pw::kBadDegree || !pw::postfix_columns_below(ibc + off, len, width)) { ok = false; break; }
            const uint32_t o = (uint32_t)(code.size() / 2);
            if (!cc.compile(ibc + off, len, code)) { ok = false; break; }
            pw::SmallForm f{};
            if (!pw::analyze_small_form(ibc + off, len, f)) { f = pw::SmallForm{}; f.flags = pw::SmallForm::NOT_SMALL; ++not_small; }
            forms.push_back(f);
            xspans.push_back(o);
            xspans.push_back((uint32_t)(code.size() / 2) - o);
        }
    }
    auto up = [&](void** d, const void* h, size_t bytes) {
        if (!device) return true;
        if (hipMalloc(d, bytes ? bytes : 4) != hipSuccess) return false;
        return !bytes || hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    std::vector<uint32_t> gstarts{0};
    if (ok) gstarts = pw::logup_group_starts(inter, n_inter, ispans, ibc);
    if (!ok || !up((void**)&p->d_gstarts, gstarts.data(), gstarts.size() * 4) ||
        !up((void**)&p->d_inter, li.data(), li.size() * sizeof(pw::LogupInteraction)) ||
        !up((void**)&p->d_ixspans, xspans.data(), xspans.size() * 4) || !up((void**)&p->d_icode, code.data(), code.size() * 4) ||
        (2 * not_small <= forms.size() && !getenv("POWDR_LOGUP_INTERPRET") &&
         !up((void**)&p->d_iforms, forms.data(), forms.size() * sizeof(pw::SmallForm)))) {
        pw_prover_destroy(p);
        return nullptr;
    }
    p->logup = true;
    p->n_inter = (uint32_t)n_inter;
    p->n_groups = (uint32_t)gstarts.size() - 1;
    p->h_icode = std::move(code);
    p->h_ixspans = std::move(xspans);
    p->h_inter = std::move(li);
    p->h_gstarts = std::move(gstarts);
    return p;
}

extern "C" PwProver* pw_prover_create_logup(const PwStarkConfig* cfg, uint32_t width, const uint32_t* bc, size_t bc_len,
                                            const uint32_t* spans, size_t n_constraints, const uint32_t* inter, size_t n_inter,
                                            const uint32_t* ispans, size_t n_ispans, const uint32_t* ibc, size_t ibc_len) {
    return create_prover_logup(cfg, width, bc, bc_len, spans, n_constraints, inter, n_inter, ispans, n_ispans, ibc, ibc_len, true);
}

// Code generation + hiprtc compilation of an AIR's specialised kernels WITHOUT a GPU (hiprtc cross-compiles): what
// pw_prover_specialise would build for a prover created from the same tables. interactions == NULL: constraints only.
extern "C" int pw_jit_compile_check(uint32_t width, const uint32_t* bc, size_t bc_len, const uint32_t* spans, size_t n_constraints,
                                    const uint32_t* inter, size_t n_inter, const uint32_t* ispans, size_t n_ispans, const uint32_t* ibc,
                                    size_t ibc_len, size_t* n_kernels, size_t* code_bytes, size_t* n_chunks, char* err, size_t err_cap) {
    const PwStarkConfig cfg{1, 0};
    PwProver* p = inter ? create_prover_logup(&cfg, width, bc, bc_len, spans, n_constraints, inter, n_inter, ispans, n_ispans, ibc, ibc_len, false)
                        : create_prover(&cfg, width, bc, bc_len, spans, n_constraints, false);
    if (!p) return -2;
    const int rc = pw_prover_specialise(p);
    (void)pw_prover_specialised(p, n_kernels, code_bytes, n_chunks);
    if (err && err_cap) { strncpy(err, p->jit.error.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
    pw_prover_destroy(p);
    return rc;
}

// Test hook: the HIP source the code generator emits for translation unit `unit` of an AIR's specialised kernels (which: 0 quotient
// numerator, 1 LogUp permutation columns) under explicit chunking parameters, no compilation. Returns the source's length (0: no such
// unit / the programs did not compile to xbc); buf receives at most cap - 1 characters.
extern "C" size_t pw_jit_generated_source(uint32_t width, const uint32_t* bc, size_t bc_len, const uint32_t* spans, size_t n_constraints,
                                          const uint32_t* inter, size_t n_inter, const uint32_t* ispans, size_t n_ispans, const uint32_t* ibc,
                                          size_t ibc_len, int which, uint32_t chunk_cost, uint32_t chunks_per_unit, size_t unit, char* buf, size_t cap,
                                          char* kernel_name, size_t name_cap, uint32_t* first_chunk, uint32_t* n_chunks, uint32_t* total_chunks) {
    const PwStarkConfig cfg{1, 0};
    PwProver* p = inter ? create_prover_logup(&cfg, width, bc, bc_len, spans, n_constraints, inter, n_inter, ispans, n_ispans, ibc, ibc_len, false)
                        : create_prover(&cfg, width, bc, bc_len, spans, n_constraints, false);
    if (!p) return 0;
    const pw::jit::Generated g = pw::generate_sources(p, which, chunk_cost ? chunk_cost : 8000, chunks_per_unit ? chunks_per_unit : 8);
    pw_prover_destroy(p);
    if (total_chunks) *total_chunks = g.n_chunks;
    if (unit >= g.units.size()) return 0;
    const pw::jit::Unit& u = g.units[unit];
    if (buf && cap) { strncpy(buf, u.source.c_str(), cap - 1); buf[cap - 1] = 0; }
    if (kernel_name && name_cap) { strncpy(kernel_name, u.kernel.c_str(), name_cap - 1); kernel_name[name_cap - 1] = 0; }
    if (first_chunk) *first_chunk = u.first_chunk;
    if (n_chunks) *n_chunks = u.n_chunks;
    return u.source.size();
}

#define TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

extern "C" int pw_prover_logup_path(const PwProver* p) { return !p || !p->logup ? 0 : p->d_iforms ? 2 : 1; }

extern "C" int pw_prover_set_bus_seed(PwProver* p, const uint32_t* seed8) {
    if (!p || !p->logup) return (int)hipErrorInvalidValue;
    p->has_bus_seed = seed8 != nullptr;
    if (seed8) for (int i = 0; i < 8; ++i) p->bus_seed[i] = bb::to_monty(seed8[i] % bb::P);
    return 0;
}

namespace pw {
// (read per call, so that tests can force many small panels: floor 2^12 words, and never fewer than 8 columns)
size_t lde_panel_cols(size_t H, size_t widest) {
    int panel_log_words = 28;
    if (const char* e = getenv("POWDR_PANEL_LOG_WORDS")) { const int v = atoi(e); if (v >= 12 && v <= 32) panel_log_words = v; }
    size_t cols = ((size_t)1 << panel_log_words) / H;
    if (cols < 8) cols = 8;
    if (cols > widest) cols = widest;
    return cols ? cols : 1;
}
int lde_matrix(PwProver* p, const CommitLayout& L, uint32_t log_h, const uint32_t* m, uint32_t cols, uint32_t* out) {
    uint32_t* d_coef = p->coef.as<uint32_t>();
    for (size_t c0 = 0; c0 < cols; c0 += L.panel_cols) {
        const uint32_t pc = (uint32_t)(cols - c0 < L.panel_cols ? cols - c0 : L.panel_cols);
        // Fused schedule (three passes, the coefficient array never touches HBM) where it measured faster than the four-pass
        // one; both read the twiddles of their contiguous stages from tables (profiles/r02_bench_ntt.txt: 1.33x at 2^12 rows —
        // one launch per LDE —, 1.03-1.19x at 2^15..2^20; 0.85-0.98x at 2^13, 2^14, 2^21 and 2^22, where the split of the strided
        // stages is worse); POWDR_LDE_FUSED=0/1 forces one of them.
        const char* ef = getenv("POWDR_LDE_FUSED");
        const bool fused = ef ? atoi(ef) != 0 : (log_h <= 12 || (log_h >= 15 && log_h <= 20));
        if (!fused) {
            TRY(intt_dif(m + c0 * L.H, d_coef, L.H, L.H, pc, (int)log_h));
            TRY(coset_lde_from_coeffs(d_coef, out + c0 * L.N, L.H, L.N, pc, (int)log_h));
        } else {
            TRY(lde_fused(m + c0 * L.H, d_coef, out + c0 * L.N, L.H, L.H, L.N, pc, (int)log_h));
        }
    }
    return 0;
}
}  // namespace pw

namespace {

// ---- streamed proofs ---------------------------------------------------------------------------------------------------
// A proof needs the LDE of every committed column three times: to hash its rows, to evaluate the quotient on them and to answer
// the queries (the DEEP numerator is a polynomial and is extended on its own, see step 4). Resident, that is 8 bytes per committed
// cell next to the caller's trace: BASELINE configs[2] with its bus interactions (3 731 main + 4 632 permutation columns x 2^22
// rows) would need 280 GB. In STREAMED mode (b >= 1) the prover keeps the COEFFICIENTS instead — the trace's in `tcoef`, the
// permutation matrix's in place of the matrix — and walks the extended domain as 2^b sub-cosets (rows r + 2^b i): for each one the
// LDE rows of all columns are rebuilt from the coefficients (ntt.hip subcoset_lde) into `lde`, consumed (leaf hash / quotient
// terms / query rows) and dropped. Same field elements everywhere, hence the same proof words as the resident path
// (tests/test_streamed_prover.py compares both with the oracle).
// Mode: POWDR_STREAM_LOG_BLOCKS = 0 never, b >= 1 always with 2^b sub-cosets (tests); unset: resident when its buffers fit into
// what the device has free (plus what the prover already holds), else the smallest b whose buffers do.
// The DEEP numerator sum_k gamma^k P_k is a polynomial: from 2^16 rows on it is combined on the UN-extended matrices (half the bytes
// the LDE holds) and extended as 4 + 4 columns instead of being accumulated over the LDE of every column (deep_kernel); shorter traces
// keep the one-kernel form (the extra launches cost more than the bytes they save).
// (kDeepComboMinLogHeight: prover_state.hpp — the segment prover follows the same rule)

struct BufferPlan {
    size_t coef = 0, lde = 0, digests = 0, perm = 0, plde = 0, q = 0, qpart = 0, qcoef = 0, qlde = 0, ext_arena = 0, misc = 0,
           tcoef = 0, fscale = 0, gbuf = 0;
    size_t commit_total() const { return coef + lde + digests + tcoef + fscale; }
    size_t total() const { return commit_total() + perm + plde + q + qpart + qcoef + qlde + ext_arena + misc + gbuf; }
};

// consume: the caller hands its trace over (pw_prover_prove_consuming) — a streamed proof then keeps the trace's coefficient arrays IN
// the caller's buffer (no tcoef); with LogUp the permutation matrix is computed into `lde` first (its evaluations are only needed until
// its coefficients exist), while `perm` lends its room to the trace's coefficients until the trace itself is dead.
void plan_buffers(const PwProver* p, uint32_t log_h, int b, CommitLayout& L, BufferPlan& B, bool consume = false) {
    L.H = (size_t)1 << log_h;
    L.N = 2 * L.H;
    L.b = b;
    L.m = L.N >> b;
    L.tree_words = merkle_words(L.N);
    L.fri_words = 0;
    for (uint32_t l = 0; l < log_h; ++l) L.fri_words += merkle_words((L.N >> l) / 2);
    L.n_trees = p->logup ? 3 : 2;  // trace | quotient | (perm) | FRI
    const size_t H = L.H, N = L.N;
    const int logN = (int)log_h + 1;
    const bool lg = p->logup;
    const uint32_t W = p->width, nc = p->n_constraints;
    const uint32_t n_g = lg ? p->n_groups : 0;
    const uint32_t Wp = lg ? 4 * (n_g + 1) : 0;
    const uint32_t K = W + 2 * Wp + 8;
    const uint32_t M = nc + (lg ? n_g + 3 : 0);
    // coefficients exist only per column panel (1 GB by default; larger panels = fewer, larger launches): iNTT -> panel ->
    // coset NTT into the resident LDE. Streamed: the panel serves the eight phi / row-sum columns only.
    const size_t widest = b ? 8 : (lg ? std::max<size_t>(W, Wp) : W);
    L.panel_cols = lde_panel_cols(H, widest);
    B.coef = L.panel_cols * H * 4;
    // streamed: `lde` holds one sub-coset. The commitments and the query rows take the two matrices one after the other (max(W, Wp)
    // columns); the quotient needs main AND permutation columns of the same rows — all of them for the interpreter, but a unit of the
    // specialised LogUp kernels reads only the columns of its own groups: the main block + one unit's panel. That is what lets
    // configs[2] run with 4 sub-cosets instead of 8 (half the coefficient re-reads, the transforms' first stage group is bound by them).
    L.perm_panels = b && lg && p->jit.state == 1 && !getenv("POWDR_STREAM_NO_PANELS");
    if (!b) B.lde = (size_t)W * N * 4;
    else if (L.perm_panels) B.lde = std::max<size_t>(std::max(W, Wp), (size_t)W + quotient_max_unit_perm_cols(p)) * L.m * 4;
    else B.lde = (size_t)(W + Wp) * L.m * 4;
    B.digests = (L.n_trees * L.tree_words + L.fri_words) * 4;
    if (b) { B.tcoef = (size_t)W * H * 4; B.fscale = (size_t)1 << 15; }
    if (b || log_h >= kDeepComboMinLogHeight) B.gbuf = (size_t)24 * H * 4;
    if (lg) {  // + the uncommitted per-row-sum columns (kJitExtraPermCols; the streamed path keeps them on every path)
        B.perm = (size_t)(Wp + kJitExtraPermCols) * H * 4;
        B.plde = b ? (size_t)8 * N * 4 : (size_t)(Wp + kJitExtraPermCols) * N * 4;
    }
    B.q = 4 * N * 4;
    const size_t q_rows = b ? L.m : N;
    if (!lg) {
        const uint32_t chunks = quotient_chunks(q_rows, nc);
        if (chunks > 1) B.qpart = (size_t)chunks * 4 * q_rows * 4;
    }
    if (b) B.qpart += 4 * L.m * 4;  // the interpreter kernels' unscaled sums of one sub-coset
    B.qpart = std::max(B.qpart, jit_part_bytes(p, H, q_rows));
    if (b && consume) {
        B.tcoef = 0;
        if (lg) {
            B.perm = std::max(B.perm, (size_t)W * H * 4);
            B.lde = std::max(B.lde, (size_t)(Wp + kJitExtraPermCols) * H * 4);
        }
    }
    B.qcoef = 8 * H * 4;
    B.qlde = 8 * N * 4;
    // ext arena: FRI layer vectors v_0 (N) .. v_log_h (2): 2N ext; weights (H); LogUp: second weights, row sums
    B.ext_arena = (2 * N + (lg ? 3 : 1) * H + H / 4096 + 32) * sizeof(bb::Ext);
    const uint32_t n_chunks = div_up(H, 8192);
    const uint32_t dot_cols = std::max({W, Wp, 8u});
    const size_t misc_ext = 2 * (size_t)dot_cols * n_chunks + K + K + M + p->max_args + 64;  // ext_dot_columns2 keeps two sets of partial sums
    const uint32_t nq = p->cfg.num_queries;
    const size_t path_records = (size_t)nq * (L.n_trees * (size_t)logN + (size_t)log_h * logN) + 16;
    B.misc = misc_ext * sizeof(bb::Ext) + (size_t)nq * 8 + (size_t)nq * (W + Wp + 8) * 4 + path_records * (8 + 32) +
             (size_t)nq * log_h * (8 + 16) + 4096;
}

}  // namespace

namespace pw {
static std::atomic<size_t> g_device_budget{(size_t)-1};  // (size_t)-1: not set yet (POWDR_DEVICE_BUDGET_BYTES is read once); 0: none
size_t device_budget() {
    size_t b = g_device_budget.load();
    if (b == (size_t)-1) {
        const char* e = getenv("POWDR_DEVICE_BUDGET_BYTES");
        b = e ? (size_t)strtoull(e, nullptr, 10) : 0;
        g_device_budget.store(b);
    }
    return b;
}
// What a proof (one AIR's, or a whole segment's) may plan for: buffers are grown by free + malloc, so what the caller's provers hold
// now (`held`) counts as available; 8 % of head room for the allocator's granularity, the twiddle tables and the caller's own small
// allocations during the proof; the device as a whole stays below 90 % (288 GB: 259 GB) — a streamed proof trades a few sub-cosets
// more for room the caller may need between two proofs; and never more than the budget the embedder set (pw_set_device_budget:
// a server that shares the device between several engines). false: the runtime could not say (the caller stays resident).
bool device_room(size_t held, size_t* avail) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
    const size_t others = total_b > free_b + held ? total_b - free_b - held : 0;  // the caller's traces, other provers, the runtime
    size_t a = (size_t)((double)(free_b + held) * 0.92);
    const size_t cap = (size_t)((double)total_b * 0.90);
    a = std::min(a, cap > others ? cap - others : (size_t)0);
    if (const size_t budget = device_budget()) a = std::min(a, budget);
    *avail = a;
    return true;
}
}  // namespace pw

extern "C" void pw_set_device_budget(size_t bytes) { pw::g_device_budget.store(bytes); }
extern "C" size_t pw_get_device_budget(void) { return pw::device_budget(); }

namespace {
// 0 = resident, b >= 1 = streamed over 2^b sub-cosets; < 0: nothing fits
int stream_log_blocks(const PwProver* p, uint32_t log_h, bool consume = false) {
    const int b_max = std::min((int)log_h - 1, 5);  // sub-cosets of at least 4 rows; at most 32 of them (subcoset_lde)
    if (const char* e = getenv("POWDR_STREAM_LOG_BLOCKS")) {
        const int v = atoi(e);
        if (v <= 0 || b_max < 1) return 0;
        return v < b_max ? v : b_max;
    }
    if (log_h < stream_min_log_height()) return 0;  // (a few MB either way)
    size_t avail = 0;
    if (!device_room(pw_prover_device_bytes(p), &avail)) return 0;
    for (int b = 0; b <= b_max; ++b) {
        CommitLayout L;
        BufferPlan B;
        plan_buffers(p, log_h, b, L, B, consume);
        if (B.total() <= avail) return b;
    }
    return -1;
}

int apply_commit_buffers(PwProver* p, const BufferPlan& B) {
    // release before growing: stream_log_blocks counted a held tcoef as available (a consuming proof after a plain streamed one)
    if (!B.tcoef) p->tcoef.release();
    TRY(p->coef.ensure(B.coef));
    TRY(p->lde.ensure(B.lde));
    TRY(p->digests.ensure(B.digests));
    if (B.tcoef) TRY(p->tcoef.ensure(B.tcoef));
    if (B.fscale) TRY(p->fscale.ensure(B.fscale));
    return 0;
}

// Buffers of the trace commitment, sized as pw_prover_prove needs them (so that a later prove does not reallocate)
int ensure_commit_buffers(PwProver* p, uint32_t log_h, int b, CommitLayout& L) {
    BufferPlan B;
    plan_buffers(p, log_h, b, L, B);
    return apply_commit_buffers(p, B);
}

streamed::Ctx stream_ctx(PwProver* p, const CommitLayout& L, uint32_t log_h) {
    const uint32_t Wp = p->logup ? 4 * (p->n_groups + 1) : 0;
    return streamed::Ctx{p, log_h, L.b, L.perm_panels, L.H, L.N, L.m, p->width, Wp};
}

// streamed commitment of a matrix given by its coefficient arrays: every sub-coset's rows are hashed into their leaves
int commit_coefficients(PwProver* p, const CommitLayout& L, uint32_t log_h, const uint32_t* coef, uint32_t cols, uint32_t* d_tree) {
    TRY(streamed::leaf_hashes(stream_ctx(p, L, log_h), coef, cols, d_tree));
    return merkle_build_levels(d_tree, L.N);
}

// LDE + Merkle tree of the trace into p->lde / the first tree of p->digests; root (Montgomery) to the host.
// Streamed: the trace's coefficients into p->tcoef, the tree from the sub-cosets.
int commit_trace(PwProver* p, const CommitLayout& L, const uint32_t* d_trace, uint32_t log_h, uint32_t* root) {
    hipStream_t st = stream();
    uint32_t* d_tdig = p->digests.as<uint32_t>();
    if (L.b) {
        TRY(intt_dif(d_trace, p->tcoef.as<uint32_t>(), L.H, L.H, p->width, (int)log_h));
        TRY(commit_coefficients(p, L, log_h, p->tcoef.as<uint32_t>(), p->width, d_tdig));
    } else {
        TRY(lde_matrix(p, L, log_h, d_trace, p->width, p->lde.as<uint32_t>()));
        TRY(merkle_commit_matrix(p->lde.as<uint32_t>(), L.N, p->width, L.N, d_tdig));
    }
    PW_HIP_TRY(hipMemcpyAsync(root, d_tdig + L.tree_words - 8, 32, hipMemcpyDeviceToHost, st));
    PW_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}


// Every device buffer a proof of a 2^log_h-row trace needs (grown on demand; pw_prover_reserve calls this at set-up
// time so that the first proof does not pay for tens of gigabytes of hipMalloc).
int ensure_prove_buffers(PwProver* p, uint32_t log_h, int b, CommitLayout& L, bool consume = false) {
    BufferPlan B;
    plan_buffers(p, log_h, b, L, B, consume);
    TRY(apply_commit_buffers(p, B));
    if (B.perm) { TRY(p->perm.ensure(B.perm)); TRY(p->plde.ensure(B.plde)); }
    TRY(p->q.ensure(B.q));
    if (B.qpart) TRY(p->qpart.ensure(B.qpart));
    TRY(p->qcoef.ensure(B.qcoef));
    TRY(p->qlde.ensure(B.qlde));
    TRY(p->ext_arena.ensure(B.ext_arena));
    TRY(p->misc.ensure(B.misc));
    if (B.gbuf) TRY(p->gbuf.ensure(B.gbuf));
    return 0;
}
}  // namespace

namespace pw {
size_t proof_plan_bytes(const PwProver* p, uint32_t log_h, int b, bool consume) {
    CommitLayout L;
    BufferPlan B;
    plan_buffers(p, log_h, b, L, B, consume && b > 0);
    return B.total();
}
int ensure_proof_buffers(PwProver* p, uint32_t log_h, int b, CommitLayout& L, bool consume) {
    return ensure_prove_buffers(p, log_h, b, L, consume && b > 0);
}
}  // namespace pw

// Trace commitment only (LDE + Merkle root): what a segment's AIRs exchange before the bus seed can be formed.
extern "C" int pw_prover_trace_root(PwProver* p, const uint32_t* d_trace, uint32_t log_h, uint32_t* root8) {
    if (!p || !d_trace || !root8 || log_h < 1 || log_h > 26) return (int)hipErrorInvalidValue;
    (void)hipGetLastError();
    TRY(poseidon2_upload_params());
    CommitLayout L;
    p->committed_trace = nullptr;
    (void)specialise_provers(&p, 1, &log_h, false);  // before the buffers are sized (the specialised kernels' partial sums)
    const int sb = stream_log_blocks(p, log_h);
    if (sb < 0) return (int)hipErrorOutOfMemory;
    TRY(ensure_commit_buffers(p, log_h, sb, L));
    TRY(commit_trace(p, L, d_trace, log_h, p->committed_root));
    for (int i = 0; i < 8; ++i) root8[i] = bb::from_monty(p->committed_root[i]);
    p->committed_trace = d_trace;
    p->committed_log_h = log_h;
    p->committed_b = sb;
    return (int)hipGetLastError();
}

// The prover's host-mapped landing place for K opened values (PwProver::OpenedMailbox) on the current device, or nullptr (no pinned
// memory to be had, POWDR_OPENINGS_OVERLAP=0): the caller then copies the values after the fact.
static void release_opened_mailbox(PwProver* p) {
    PwProver::OpenedMailbox& mb = p->opened_mb;
    for (int i = 0; i < mb.n_events; ++i) (void)hipEventDestroy(mb.ev[i]);
    mb.n_events = 0;
    if (mb.host) (void)hipHostFree(mb.host);
    mb.host = mb.dev = nullptr;
    mb.cap = 0;
    mb.device = -1;
}

static PwProver::OpenedMailbox* opened_mailbox(PwProver* p, size_t K) {
    const char* env = getenv("POWDR_OPENINGS_OVERLAP");
    if (env && atoi(env) == 0) return nullptr;
    PwProver::OpenedMailbox& mb = p->opened_mb;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (mb.host && (mb.device != device || mb.cap < K)) release_opened_mailbox(p);
    if (!mb.host) {
        void *h = nullptr, *d = nullptr;
        const size_t cap = std::max<size_t>(K, 64);
        if (hipHostMalloc(&h, cap * sizeof(bb::Ext), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return nullptr; }
        mb.host = (bb::Ext*)h; mb.dev = (bb::Ext*)d; mb.cap = cap; mb.device = device;
        while (mb.n_events < PwProver::OpenedMailbox::kEvents) {
            if (hipEventCreateWithFlags(&mb.ev[mb.n_events], hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                release_opened_mailbox(p);
                return nullptr;
            }
            ++mb.n_events;
        }
    }
    return &mb;
}

extern "C" void pw_prover_destroy(PwProver* p) {
    if (!p) return;
    for (DeviceBuf* b : {&p->coef, &p->lde, &p->digests, &p->q, &p->qcoef, &p->qlde, &p->ext_arena, &p->misc, &p->perm, &p->plde, &p->qpart, &p->tcoef,
                         &p->fscale, &p->gbuf})
        b->release();
    for (void* q : {(void*)p->d_inter, (void*)p->d_ixspans, (void*)p->d_icode, (void*)p->d_gstarts, (void*)p->d_iforms}) if (q) (void)hipFree(q);
    if (p->d_bytecode) (void)hipFree(p->d_bytecode);
    if (p->d_spans) (void)hipFree(p->d_spans);
    release_opened_mailbox(p);
    delete p;
}

extern "C" uint32_t pw_prover_width(const PwProver* p) { return p ? p->width : 0; }
extern "C" int pw_prover_max_constraint_degree(const PwProver* p) { return p ? p->max_degree : -1; }

extern "C" int pw_prover_reserve(PwProver* p, uint32_t log_h) {
    if (!p || log_h < 1 || log_h > 26) return (int)hipErrorInvalidValue;
    (void)hipGetLastError();
    CommitLayout L;
    // the kernels are specialised NOW when the height policy will specialise them at the first proof: their partial-sum buffer is
    // part of the reservation, and the seconds of compilation belong to set-up time as well
    (void)specialise_provers(&p, 1, &log_h, false);
    const int sb = stream_log_blocks(p, log_h);
    if (sb < 0) return (int)hipErrorOutOfMemory;
    return ensure_prove_buffers(p, log_h, sb, L);
}

// 0: the last reservation / proof of a 2^log_height-row trace would keep the LDE resident; b >= 1: it is streamed over 2^b sub-cosets
// (the policy of stream_log_blocks for the CURRENT free memory); -1: not even the streamed buffers fit.
extern "C" int pw_prover_stream_log_blocks(const PwProver* p, uint32_t log_h) {
    if (!p || log_h < 1 || log_h > 26) return -1;
    return stream_log_blocks(p, log_h);
}

extern "C" size_t pw_prover_device_bytes(const PwProver* p) {
    return p->coef.bytes + p->lde.bytes + p->digests.bytes + p->q.bytes + p->qcoef.bytes + p->qlde.bytes +
           p->ext_arena.bytes + p->misc.bytes + p->perm.bytes + p->plde.bytes + p->qpart.bytes + p->tcoef.bytes + p->fscale.bytes + p->gbuf.bytes;
}


namespace {
// consume: pw_prover_prove_consuming — d_trace is the caller's to give away. Only a STREAMED proof uses that: the coefficient arrays
// of the trace end up in d_trace itself (no tcoef buffer: 62.6 GB at configs[2], which is what lets it run on 2 sub-cosets instead of 4).
int prove_impl(PwProver* p, const uint32_t* d_trace, uint32_t log_h, const uint32_t** proof_words, size_t* n_words, bool consume) {
    if (!p || !d_trace || log_h < 1 || log_h > 26) return -1;
    // a handed-over trace becomes a coefficient array that is read 2 / 4 words at a time (fold loads, the DEEP combination)
    if (consume && ((uintptr_t)d_trace & 15)) return (int)hipErrorInvalidValue;
    (void)hipGetLastError();
    const uint32_t W = p->width, nc = p->n_constraints;
    const size_t H = (size_t)1 << log_h, N = 2 * H;
    const int logN = (int)log_h + 1;
    const bool lg = p->logup;
    const uint32_t n_int = lg ? p->n_inter : 0;
    const uint32_t n_g = lg ? p->n_groups : 0;
    const uint32_t Wp = lg ? 4 * (n_g + 1) : 0;     // permutation matrix: q_g coordinates per group, then phi
    const uint32_t K1 = W + Wp + 8;                  // polynomials opened at zeta: main | perm | quotient
    const uint32_t K = K1 + Wp;                      // + perm opened at g*zeta
    const uint32_t M = nc + (lg ? n_g + 3 : 0);      // folded constraints
    hipStream_t st = stream();
    TRY(poseidon2_upload_params());
    (void)specialise_provers(&p, 1, &log_h, false);  // run-time specialised expression kernels, compiled once per prover (prover_jit.hip)
    const bool jit = specialised(p);

    // ---- buffers --------------------------------------------------------------------------
    // digest arena: trace tree | quotient tree | (perm tree) | FRI trees
    // (a consuming proof commits in its own way: a pw_prover_trace_root before it is not reused)
    bool have_commitment = !consume && p->committed_trace == d_trace && p->committed_log_h == log_h;
    p->committed_trace = nullptr;  // one-shot
    // resident LDE, or streamed over 2^sb sub-cosets of the extended domain ("streamed proofs" above)
    int sb = have_commitment ? p->committed_b : stream_log_blocks(p, log_h, consume);
    if (sb < 0) return (int)hipErrorOutOfMemory;
    CommitLayout L;
    int rc_buf = ensure_prove_buffers(p, log_h, sb, L, consume && sb > 0);
    if (rc_buf == (int)hipErrorOutOfMemory && have_commitment && sb == 0) {
        // pw_prove_airs commits every AIR first: several of them may each have chosen "resident" against the same free memory
        // (ADVICE r4). The commitment is dropped and made again in the streamed mode that fits NOW — same root, same words.
        (void)hipGetLastError();
        p->lde.release();
        have_commitment = false;
        sb = stream_log_blocks(p, log_h, false);
        if (sb <= 0) return (int)hipErrorOutOfMemory;
        rc_buf = ensure_prove_buffers(p, log_h, sb, L, false);
    }
    if (rc_buf) return rc_buf;
    const bool eat = consume && sb > 0;  // the trace is overwritten by its coefficient arrays
    const size_t tree_words = L.tree_words, n_trees = L.n_trees;
    // streamed: the trace's coefficient arrays — in tcoef, or (eat) in the caller's buffer, from the moment nothing reads the trace any more
    uint32_t* d_tcoef = eat ? const_cast<uint32_t*>(d_trace) : p->tcoef.as<uint32_t>();
    const uint32_t n_chunks = div_up(H, 8192);
    const uint32_t dot_cols = std::max({W, Wp, 8u});  // widest matrix ext_dot_columns sees (the quotient has 8 columns)
    const uint32_t nq = p->cfg.num_queries;

    uint32_t* d_lde = p->lde.as<uint32_t>();
    uint32_t* d_dig = p->digests.as<uint32_t>();
    uint32_t* d_qdig = d_dig + tree_words;
    uint32_t* d_pdig = d_dig + 2 * tree_words;          // LogUp only
    uint32_t* d_fdig = d_dig + n_trees * tree_words;
    uint32_t* d_perm = p->perm.as<uint32_t>();
    uint32_t* d_plde = p->plde.as<uint32_t>();
    uint32_t* d_q = p->q.as<uint32_t>();
    uint32_t* d_qcoef = p->qcoef.as<uint32_t>();
    uint32_t* d_qlde = p->qlde.as<uint32_t>();
    bb::Ext* d_v = p->ext_arena.as<bb::Ext>();          // FRI layers, consecutive
    bb::Ext* d_weights = d_v + 2 * N;
    bb::Ext* d_weights2 = d_weights + H;                   // LogUp: weights at g*zeta
    bb::Ext* d_rowsum = d_weights2 + H;                    // LogUp: per-row sums, then block totals
    bb::Ext* d_scratch = p->misc.as<bb::Ext>();            // dot_cols * n_chunks
    bb::Ext* d_opened = d_scratch + 2 * (size_t)dot_cols * n_chunks;  // K
    bb::Ext* d_gpow = d_opened + K;                        // K
    bb::Ext* d_apow = d_gpow + K;                          // M
    bb::Ext* d_blpow = d_apow + M + 4;                     // max_args + 2
    uint8_t* d_tail = reinterpret_cast<uint8_t*>(d_blpow + p->max_args + 8);

    std::vector<uint32_t>& pf = p->proof;
    pf.clear();
    auto put = [&](uint32_t canonical) { pf.push_back(canonical); };
    auto put_monty = [&](const uint32_t* w, size_t n) { for (size_t i = 0; i < n; ++i) pf.push_back(bb::from_monty(w[i])); };

    Challenger ch;
    if (lg) {
        for (uint32_t x : {kMagic2 % bb::P, log_h, W, nc, n_int, p->cfg.num_queries, p->cfg.pow_bits}) ch.observe_canonical(x);
        for (uint32_t x : {kMagic2, log_h, W, nc, n_int, p->cfg.num_queries, p->cfg.pow_bits}) put(x);
    } else {
        for (uint32_t x : {kMagic % bb::P, log_h, W, nc, p->cfg.num_queries, p->cfg.pow_bits}) ch.observe_canonical(x);
        for (uint32_t x : {kMagic, log_h, W, nc, p->cfg.num_queries, p->cfg.pow_bits}) put(x);
    }

    // ---- 1. trace: coefficients, LDE, commitment ---------------------------------------------
    uint32_t root[8];
    // eat + LogUp: the permutation columns are computed from the trace's VALUES after this commitment, so the coefficients go to the
    // (still empty) permutation buffer first; eat without LogUp: nothing reads the values again — in place.
    uint32_t* d_coef_tmp = eat ? (lg ? p->perm.as<uint32_t>() : d_tcoef) : nullptr;
    if (have_commitment) memcpy(root, p->committed_root, 32);  // pw_prover_trace_root already did this step
    else if (eat) {
        TRY(intt_dif(d_trace, d_coef_tmp, H, H, W, (int)log_h));
        TRY(commit_coefficients(p, L, log_h, d_coef_tmp, W, p->digests.as<uint32_t>()));
        PW_HIP_TRY(hipMemcpyAsync(root, p->digests.as<uint32_t>() + tree_words - 8, 32, hipMemcpyDeviceToHost, st));
        PW_HIP_TRY(hipStreamSynchronize(st));
    } else TRY(commit_trace(p, L, d_trace, log_h, root));
    put_monty(root, 8);
    ch.observe_words(root, 8);

    // ---- 1b. LogUp: permutation trace, its LDE and commitment ---------------------------------------
    bb::Ext al = bb::ext_zero(), S = bb::ext_zero();
    LogupProgram lp{p->d_inter, n_int, p->d_ixspans, p->d_icode, p->d_gstarts, n_g, p->d_iforms};
    if (lg) {
        // the bus challenges come from a transcript that saw only the bus seed (shared by all AIRs of a segment;
        // a lone AIR uses its own trace root), see oracle/stark_oracle.cpp bus_challenges
        uint32_t seed[8];
        memcpy(seed, p->has_bus_seed ? p->bus_seed : root, 32);
        put_monty(seed, 8);
        ch.observe_words(seed, 8);
        Challenger cb;
        cb.observe_canonical(kMagic2 % bb::P);
        cb.observe_words(seed, 8);
        al = cb.sample_ext();
        const bb::Ext bl = cb.sample_ext();
        std::vector<bb::Ext> blpow(p->max_args + 2);
        { bb::Ext b = bb::ext_one(); for (auto& x : blpow) { x = b; b = bb::ext_mul(b, bl); } }
        PW_HIP_TRY(hipMemcpyAsync(d_blpow, blpow.data(), blpow.size() * sizeof(bb::Ext), hipMemcpyHostToDevice, st));
        PW_HIP_TRY(hipStreamSynchronize(st));
        // eat: the matrix's VALUES live in the sub-coset buffer (idle between two passes) until its coefficient arrays exist
        uint32_t* d_pval = eat ? d_lde : d_perm;
        if (jit) TRY(logup_perm_trace_jit(p, d_trace, H, al, d_blpow, d_pval, d_rowsum, d_rowsum + H));
        else TRY(logup_perm_trace(d_trace, H, lp, al, d_blpow, d_pval, d_rowsum, d_rowsum + H));
        if (!sb) {
            TRY(lde_matrix(p, L, log_h, d_perm, Wp + (jit ? kJitExtraPermCols : 0u), d_plde));
            TRY(merkle_commit_matrix(d_plde, N, Wp, N, d_pdig));
        } else {
            // streamed: only phi and the per-row sums (the boundary terms read them at rows j and j + 2) are extended for good
            if (!jit) TRY(ext_to_cols(d_rowsum, H, d_pval + (size_t)(4 * n_g + 4) * H));
            TRY(lde_matrix(p, L, log_h, d_pval + (size_t)(4 * n_g) * H, 8, d_plde));
        }
        uint32_t sw[4];
        PW_HIP_TRY(hipMemcpyAsync(root, d_pdig + tree_words - 8, 32, hipMemcpyDeviceToHost, st));
        for (int k = 0; k < 4; ++k)  // S = phi(last row)
            PW_HIP_TRY(hipMemcpyAsync(&sw[k], d_pval + ((size_t)(4 * n_g + k) * H + (H - 1)), 4, hipMemcpyDeviceToHost, st));
        if (sb) {
            // (S is read from the matrix first:) the permutation matrix becomes its coefficient arrays in place, committed sub-coset by sub-coset
            PW_HIP_TRY(hipStreamSynchronize(st));
            if (eat) {
                // the trace's values are dead now: its coefficient arrays move into its place, the permutation buffer takes the matrix's
                PW_HIP_TRY(hipMemcpyAsync(d_tcoef, d_coef_tmp, (size_t)W * H * 4, hipMemcpyDeviceToDevice, st));
                TRY(intt_dif(d_pval, d_perm, H, H, Wp, (int)log_h));
            } else TRY(intt_dif(d_perm, d_perm, H, H, Wp, (int)log_h));
            TRY(commit_coefficients(p, L, log_h, d_perm, Wp, d_pdig));
            PW_HIP_TRY(hipMemcpyAsync(root, d_pdig + tree_words - 8, 32, hipMemcpyDeviceToHost, st));
        }
        PW_HIP_TRY(hipStreamSynchronize(st));
        put_monty(root, 8);
        ch.observe_words(root, 8);
        for (int k = 0; k < 4; ++k) S.c[k] = sw[k];
        put_monty(S.c, 4);
        ch.observe_ext(S);
    }

    // ---- 2. quotient ---------------------------------------------------------------------------
    const bb::Ext alpha = ch.sample_ext();
    {
        std::vector<bb::Ext> apow(M ? M : 1);
        bb::Ext a = bb::ext_one();
        for (size_t j = M; j-- > 0;) { apow[j] = a; a = bb::ext_mul(a, alpha); }
        if (M) PW_HIP_TRY(hipMemcpyAsync(d_apow, apow.data(), M * sizeof(bb::Ext), hipMemcpyHostToDevice, st));
        PW_HIP_TRY(hipStreamSynchronize(st));  // apow is a stack-local vector
    }
    const uint32_t s_m = bb::to_monty(field::kCosetShift);
    uint32_t sH = s_m;
    for (uint32_t i = 0; i < log_h; ++i) sH = bb::sqr(sH);
    const uint32_t one = bb::R_MOD_P;
    const uint32_t zinv_even = bb::inv(bb::sub(sH, one));
    const uint32_t zinv_odd = bb::inv(bb::sub(bb::neg(sH), one));
    ConstraintProgram prog{p->d_bytecode, p->d_spans, nc, p->is_xbc};
    if (sb) {
        // streamed: the terms that read the current row only, sub-coset by sub-coset (unscaled sums scattered to their rows of d_q), then
        // the boundary terms / the division by Z_H over all rows
        TRY(streamed::quotient_sums(stream_ctx(p, L, log_h), jit, lg, nc, prog, lp, d_tcoef, d_perm, d_apow, al, d_blpow, S, logN, d_q));
        if (lg)
            TRY(quotient_logup_tail(d_q, 1, d_plde, d_plde + 4 * N, N, logN, d_apow + nc + n_g, S, bb::sub(sH, one), bb::sub(bb::neg(sH), one), d_q));
        else
            TRY(quotient_combine(d_q, 1, N, zinv_even, zinv_odd, d_q));
    } else if (lg && jit)
        TRY(quotient_eval_logup_jit(p, d_lde, d_plde, N, logN, d_apow, al, d_blpow, S, bb::sub(sH, one), bb::sub(bb::neg(sH), one), d_q));
    else if (lg)
        TRY(quotient_eval_logup(d_lde, d_plde, N, logN, prog, lp, d_apow, al, d_blpow, S, bb::sub(sH, one), bb::sub(bb::neg(sH), one), d_q));
    else if (jit && nc)
        TRY(quotient_eval_jit(p, d_lde, N, d_apow, zinv_even, zinv_odd, d_q));
    else
        TRY(quotient_eval(d_lde, N, prog, d_apow, zinv_even, zinv_odd, d_q, p->qpart.as<uint32_t>(), quotient_chunks(N, nc)));
    TRY(intt_dif(d_q, d_q, N, N, 4, logN));
    TRY(quotient_split(d_q, H, (int)log_h, d_qcoef));
    TRY(coset_lde_from_coeffs(d_qcoef, d_qlde, H, N, 8, (int)log_h));
    TRY(merkle_commit_matrix(d_qlde, N, 8, N, d_qdig));
    PW_HIP_TRY(hipMemcpyAsync(root, d_qdig + tree_words - 8, 32, hipMemcpyDeviceToHost, st));
    PW_HIP_TRY(hipStreamSynchronize(st));
    put_monty(root, 8);
    ch.observe_words(root, 8);

    // ---- 3. openings at zeta -------------------------------------------------------------------
    const bb::Ext zeta = ch.sample_ext();
    // trace columns: barycentric evaluation straight from the caller's trace (natural order on <g_n>);
    // quotient chunks: from their (small) coefficient arrays
    // internal order of `opened` (= order of the gamma powers): main | perm@zeta | quotient | perm@g*zeta
    const bb::Ext gzeta = bb::ext_scale(zeta, field::root_of_unity((int)log_h));
    // Every matrix's values go straight into host-mapped memory when the prover has it (opened_mailbox), the permutation matrix's in
    // four column slices, with an event after each piece: the host absorbs a piece into the transcript — a sequential sponge, two
    // opened values per permutation — while the device computes the next.
    PwProver::OpenedMailbox* mb = opened_mailbox(p, K);
    bb::Ext* d_open = mb ? mb->dev : d_opened;
    int n_marks = 0;
    auto mark = [&]() -> int {
        if (mb) PW_HIP_TRY(hipEventRecord(mb->ev[n_marks], st));
        ++n_marks;
        return 0;
    };
    const uint32_t n_slices = (mb && Wp >= 1024) ? 4u : 1u;
    uint32_t slice_at[5];
    for (uint32_t i = 0; i <= n_slices; ++i) slice_at[i] = i == n_slices ? Wp : (uint32_t)((uint64_t)Wp * i / n_slices) & ~7u;
    auto open_perm = [&]() -> int {  // at zeta and at g zeta: one pass over the columns of a slice
        for (uint32_t i = 0; i < n_slices; ++i) {
            const uint32_t c0 = slice_at[i], c1 = slice_at[i + 1];
            if (c1 > c0) TRY(ext_dot_columns2(d_perm + (size_t)c0 * H, H, c1 - c0, H, d_weights, d_weights2, d_open + W + c0, d_open + K1 + c0, d_scratch));
            TRY(mark());
        }
        return 0;
    };
    if (!eat) {
        TRY(barycentric_weights(zeta, (int)log_h, d_weights));
        TRY(ext_dot_columns(d_trace, H, W, H, d_weights, d_open, d_scratch));
        TRY(mark());
    }
    if (lg && !sb) {
        TRY(barycentric_weights(gzeta, (int)log_h, d_weights2));
        TRY(open_perm());
    }
    TRY(zeta_weights(zeta, (int)log_h, d_weights));
    if (eat) {  // the trace is its coefficient arrays by now: opened like the permutation matrix below
        TRY(ext_dot_columns(d_tcoef, H, W, H, d_weights, d_open, d_scratch));
        TRY(mark());
    }
    if (lg && sb) {  // streamed: d_perm holds the matrix's coefficient arrays
        TRY(zeta_weights(gzeta, (int)log_h, d_weights2));
        TRY(open_perm());
    }
    TRY(ext_dot_columns(d_qcoef, H, 8, H, d_weights, d_open + W + Wp, d_scratch));
    TRY(mark());
    std::vector<bb::Ext> opened_copy;
    const bb::Ext* opened = mb ? mb->host : nullptr;
    if (!mb) {
        opened_copy.resize(K);
        PW_HIP_TRY(hipMemcpyAsync(opened_copy.data(), d_opened, K * sizeof(bb::Ext), hipMemcpyDeviceToHost, st));
        PW_HIP_TRY(hipStreamSynchronize(st));
        opened = opened_copy.data();
    }
    {
        // proof / transcript order: main, perm@zeta, perm@g*zeta, quotient
        int waited = 0;
        auto wait = [&]() -> int {
            if (mb) PW_HIP_TRY(hipEventSynchronize(mb->ev[waited]));
            ++waited;
            return 0;
        };
        auto emit = [&](size_t a, size_t b) { for (size_t k = a; k < b; ++k) { put_monty(opened[k].c, 4); ch.observe_ext(opened[k]); } };
        TRY(wait());
        emit(0, W);
        if (lg)
            for (uint32_t i = 0; i < n_slices; ++i) {
                TRY(wait());
                emit((size_t)W + slice_at[i], (size_t)W + slice_at[i + 1]);
            }
        TRY(wait());  // the quotient's values: everything is there
        emit(K1, K);
        emit((size_t)W + Wp, K1);
    }

    // ---- 4. reduced-opening vector -------------------------------------------------------------
    const bb::Ext gamma = ch.sample_ext();
    // the powers of gamma are computed where they are used (CENTRED, the form the DEEP kernels take); the host only needs
    // sum_k gamma^k opened[k] — by Horner's rule, and where the DEEP numerator is combined first (below) while those kernels run
    TRY(gamma_powers(gamma, K, d_gpow));
    bb::Ext opened_sum = bb::ext_zero(), opened_sum2 = bb::ext_zero();
    auto opened_sums = [&]() {
        for (uint32_t k = K1; k-- > 0;) opened_sum = bb::ext_add(bb::ext_mul(opened_sum, gamma), opened[k]);
        for (uint32_t k = K; k-- > K1;) opened_sum2 = bb::ext_add(bb::ext_mul(opened_sum2, gamma), opened[k]);
        opened_sum2 = bb::ext_mul(opened_sum2, bb::ext_pow(gamma, K1));
    };
    if (sb) {
        // streamed: sum_k gamma^k P_k is a POLYNOMIAL — combined on the coefficient arrays (one pass over them) and extended as 4 (+ 4 for
        // the second opening point) columns; the eight quotient columns join from their resident LDE
        TRY(streamed::deep_from_coefficients(stream_ctx(p, L, log_h), lg, d_tcoef, d_perm, d_qlde, logN, d_gpow, opened_sums, opened_sum, opened_sum2, zeta,
                                             gzeta, d_v));
    } else if (log_h >= kDeepComboMinLogHeight && !getenv("POWDR_DEEP_DIRECT") && !((uintptr_t)d_trace & 7)) {  // (the combination reads 8-byte pairs)
        // resident: the same combination on the evaluations over <g_n> (the caller's trace, the permutation matrix), extended like any column
        uint32_t* d_gev = p->gbuf.as<uint32_t>();
        uint32_t* d_glde = d_gev + 8 * H;
        const uint32_t gc = lg ? 8u : 4u;
        TRY(ext_lincomb(d_trace, W, d_perm, Wp, H, d_gpow, lg ? K1 : 0u, d_gev));
        TRY(lde_matrix(p, L, log_h, d_gev, gc, d_glde));
        opened_sums();  // (the device is busy with the two launches above)
        TRY(deep_from_combo(d_glde, d_qlde, N, logN, d_gpow + W + Wp, opened_sum, opened_sum2, zeta, gzeta, lg ? 1 : 0, d_v));
    } else if (lg) {
        opened_sums();
        TRY(deep_quotient_logup(d_lde, W, d_plde, Wp, d_qlde, N, logN, d_gpow, opened_sum, opened_sum2, zeta, gzeta, d_v));
    } else {
        opened_sums();
        TRY(deep_quotient(d_lde, W, d_qlde, 8, N, logN, d_gpow, opened_sum, zeta, d_v));
    }

    // ---- 5. FRI commit phase --------------------------------------------------------------------
    std::vector<size_t> layer_off(log_h + 1), tree_off(log_h);  // offsets in Ext / in words
    {
        size_t o = 0, t = 0;
        for (uint32_t l = 0; l <= log_h; ++l) { layer_off[l] = o; o += N >> l; }
        for (uint32_t l = 0; l < log_h; ++l) { tree_off[l] = t; t += merkle_words((N >> l) / 2); }
    }
    uint32_t shift = s_m;
    for (uint32_t l = 0; l < log_h; ++l) {
        const size_t half = (N >> l) / 2;
        bb::Ext* v = d_v + layer_off[l];
        uint32_t* dg = d_fdig + tree_off[l];
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
        TRY(fri_fold(v, half, logN - (int)l, shift, beta, d_v + layer_off[l + 1]));
        shift = bb::sqr(shift);
    }
    bb::Ext final_poly;
    PW_HIP_TRY(hipMemcpyAsync(&final_poly, d_v + layer_off[log_h], sizeof(bb::Ext), hipMemcpyDeviceToHost, st));
    PW_HIP_TRY(hipStreamSynchronize(st));
    put_monty(final_poly.c, 4);
    ch.observe_ext(final_poly);

    // ---- 6. proof of work ------------------------------------------------------------------------
    uint32_t witness = 0;
    if (p->cfg.pow_bits) {
        uint32_t* d_state = reinterpret_cast<uint32_t*>(d_tail);
        uint32_t pending[8] = {0};
        for (size_t i = 0; i < ch.in.size(); ++i) pending[i] = ch.in[i];
        PW_HIP_TRY(hipMemcpyAsync(d_state, ch.st, 64, hipMemcpyHostToDevice, st));
        PW_HIP_TRY(hipMemcpyAsync(d_state + 16, pending, 32, hipMemcpyHostToDevice, st));
        TRY(pow_grind(d_state, d_state + 16, (uint32_t)ch.in.size(), p->cfg.pow_bits, d_state + 24, &witness));
    }
    put(witness);
    ch.observe_canonical(witness);
    if (p->cfg.pow_bits) (void)ch.sample_bits((int)p->cfg.pow_bits);

    // ---- 7. queries --------------------------------------------------------------------------------
    if (nq) {
        std::vector<uint32_t> idx(nq);
        for (auto& i : idx) i = ch.sample_bits(logN);
        // layout of the tail buffer
        uint32_t* d_idx = reinterpret_cast<uint32_t*>(d_tail);
        uint32_t* d_loc = d_idx + nq;  // streamed: the queries' rows inside their sub-cosets, sorted by sub-coset
        uint32_t* d_trows = d_loc + nq;
        uint32_t* d_prows = d_trows + (size_t)nq * W;
        uint32_t* d_qrows = d_prows + (size_t)nq * Wp;
        uint64_t* d_offs = reinterpret_cast<uint64_t*>(d_qrows + (size_t)nq * 8 + (((size_t)nq * (W + Wp + 10)) & 1));
        // digest records (8 words) then FRI sibling records (4 words)
        std::vector<uint64_t> dig_offs, ext_offs;
        for (uint32_t qi = 0; qi < nq; ++qi) {
            const size_t i = idx[qi];
            // proof order: trace path, (perm path), quotient path; arena order: trace | quotient | perm
            for (int tree : {0, 2, 1}) {
                if (tree == 2 && !lg) continue;
                const size_t base = (size_t)tree * tree_words;
                for (int l = 0; l < logN; ++l)
                    dig_offs.push_back(base + merkle_level_offset(N, l) + (((i >> l) ^ 1) * 8));
            }
            for (uint32_t l = 0; l < log_h; ++l) {
                const size_t Nl = N >> l, half = Nl / 2, pp = i & (Nl - 1);
                ext_offs.push_back((layer_off[l] + (pp ^ half)) * 4);
                const size_t leaf = pp & (half - 1);
                for (int lv = 0; lv < logN - 1 - (int)l; ++lv)
                    dig_offs.push_back(n_trees * tree_words + tree_off[l] + merkle_level_offset(half, lv) + (((leaf >> lv) ^ 1) * 8));
            }
        }
        const size_t n_dig = dig_offs.size(), n_ext = ext_offs.size();
        uint64_t* d_dig_offs = d_offs;
        uint64_t* d_ext_offs = d_offs + n_dig;
        uint32_t* d_dig_out = reinterpret_cast<uint32_t*>(d_ext_offs + n_ext);
        uint32_t* d_ext_out = d_dig_out + n_dig * 8;
        PW_HIP_TRY(hipMemcpyAsync(d_idx, idx.data(), nq * 4, hipMemcpyHostToDevice, st));
        PW_HIP_TRY(hipMemcpyAsync(d_dig_offs, dig_offs.data(), n_dig * 8, hipMemcpyHostToDevice, st));
        if (n_ext) PW_HIP_TRY(hipMemcpyAsync(d_ext_offs, ext_offs.data(), n_ext * 8, hipMemcpyHostToDevice, st));
        if (!sb) {
            TRY(gather_rows(d_lde, N, W, d_idx, nq, d_trows));
            if (lg) TRY(gather_rows(d_plde, N, Wp, d_idx, nq, d_prows));
        } else {
            // streamed: one more pass over the sub-cosets that hold a queried row (prover_stream.hpp query_rows); d_loc and the first nq
            // words of d_qrows (filled afterwards) are its index scratch
            const streamed::Ctx sc = stream_ctx(p, L, log_h);
            TRY(streamed::query_rows(sc, d_tcoef, W, idx.data(), nq, d_loc, d_qrows, d_trows));
            if (lg) TRY(streamed::query_rows(sc, d_perm, Wp, idx.data(), nq, d_loc, d_qrows, d_prows));
        }
        TRY(gather_rows(d_qlde, N, 8, d_idx, nq, d_qrows));
        TRY(gather_records(d_dig, d_dig_offs, 8u, (uint32_t)n_dig, d_dig_out));
        TRY(gather_records(reinterpret_cast<const uint32_t*>(d_v), d_ext_offs, 4u, (uint32_t)n_ext, d_ext_out));
        // the answers leave the device as canonical words (d_trows .. d_qrows are contiguous)
        TRY(canonicalize_words(d_trows, (size_t)nq * (W + Wp + 8)));
        TRY(canonicalize_words(d_dig_out, n_dig * 8));
        TRY(canonicalize_words(d_ext_out, n_ext * 4));
        std::vector<uint32_t> trows((size_t)nq * W), prows((size_t)nq * Wp + 1), qrows((size_t)nq * 8), dig(n_dig * 8), ext(n_ext * 4 + 1);
        PW_HIP_TRY(hipMemcpyAsync(trows.data(), d_trows, trows.size() * 4, hipMemcpyDeviceToHost, st));
        if (lg) PW_HIP_TRY(hipMemcpyAsync(prows.data(), d_prows, (size_t)nq * Wp * 4, hipMemcpyDeviceToHost, st));
        PW_HIP_TRY(hipMemcpyAsync(qrows.data(), d_qrows, qrows.size() * 4, hipMemcpyDeviceToHost, st));
        PW_HIP_TRY(hipMemcpyAsync(dig.data(), d_dig_out, n_dig * 32, hipMemcpyDeviceToHost, st));
        if (n_ext) PW_HIP_TRY(hipMemcpyAsync(ext.data(), d_ext_out, n_ext * 16, hipMemcpyDeviceToHost, st));
        PW_HIP_TRY(hipStreamSynchronize(st));
        auto put_raw = [&](const uint32_t* w, size_t n) { pf.insert(pf.end(), w, w + n); };
        pf.reserve(pf.size() + nq + trows.size() + prows.size() + qrows.size() + dig.size() + ext.size());
        size_t dpos = 0, epos = 0;
        for (uint32_t qi = 0; qi < nq; ++qi) {
            put(idx[qi]);
            put_raw(&trows[(size_t)qi * W], W);
            put_raw(&dig[dpos * 8], (size_t)logN * 8); dpos += logN;
            if (lg) {
                put_raw(&prows[(size_t)qi * Wp], Wp);
                put_raw(&dig[dpos * 8], (size_t)logN * 8); dpos += logN;
            }
            put_raw(&qrows[(size_t)qi * 8], 8);
            put_raw(&dig[dpos * 8], (size_t)logN * 8); dpos += logN;
            for (uint32_t l = 0; l < log_h; ++l) {
                put_raw(&ext[epos * 4], 4); epos += 1;
                const size_t depth = (size_t)logN - 1 - l;
                put_raw(&dig[dpos * 8], depth * 8); dpos += depth;
            }
        }
    }
    *proof_words = pf.data();
    *n_words = pf.size();
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int pw_prover_prove(PwProver* p, const uint32_t* d_trace, uint32_t log_h, const uint32_t** proof_words, size_t* n_words) {
    return prove_impl(p, d_trace, log_h, proof_words, n_words, false);
}

// The caller hands the trace over (the reference moves `common_main` into the engine: openvm/src/powdr_extension/trace_generator/
// cuda/mod.rs:415-419). Same proof words. A resident proof leaves the trace as it was; a streamed one leaves its H-scaled coefficient
// arrays in bit-reversed order there (pw_trace_from_coefficients turns them back into the trace).
extern "C" int pw_prover_prove_consuming(PwProver* p, uint32_t* d_trace, uint32_t log_h, const uint32_t** proof_words, size_t* n_words) {
    return prove_impl(p, d_trace, log_h, proof_words, n_words, true);
}

extern "C" int pw_prover_stream_log_blocks_consuming(const PwProver* p, uint32_t log_h) {
    if (!p || log_h < 1 || log_h > 26) return -1;
    return stream_log_blocks(p, log_h, true);
}

// the inverse of what a streamed consuming proof did to the caller's buffer: coefficient arrays (H-scaled, bit-reversed) -> values on <g_n>
extern "C" int pw_trace_from_coefficients(uint32_t* d_coeffs, uint32_t width, uint32_t log_h, uint32_t* d_scratch8k) {
    (void)hipGetLastError();
    if (!d_coeffs || !d_scratch8k || !width || log_h < 1 || log_h > 26) return -1;
    if ((uintptr_t)d_coeffs & 15) return (int)hipErrorInvalidValue;  // the first stage group stages whole tiles with 16-byte loads
    TRY(values_from_coefficients(d_coeffs, d_coeffs, (size_t)1 << log_h, (size_t)1 << log_h, width, (int)log_h, d_scratch8k));
    return (int)hipGetLastError();
}

// ---- mock prover -------------------------------------------------------------------------------------

extern "C" int pw_prover_check_constraints(PwProver* p, const uint32_t* d_trace, uint32_t log_h, uint64_t* n_violations,
                                           uint64_t* first_row, uint32_t* first_constraint) {
    if (!p || !d_trace || log_h > 40) return -1;
    (void)hipGetLastError();
    const size_t H = (size_t)1 << log_h;
    TRY(p->misc.ensure(4096));
    unsigned long long* d = p->misc.as<unsigned long long>();
    unsigned long long init[2] = {~0ull, 0ull}, res[2];
    PW_HIP_TRY(hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, stream()));
    ConstraintProgram prog{p->d_bytecode, p->d_spans, p->n_constraints, p->is_xbc};
    if (p->n_constraints) TRY(check_constraints(d_trace, H, prog, d));
    PW_HIP_TRY(hipMemcpyAsync(res, d, sizeof res, hipMemcpyDeviceToHost, stream()));
    PW_HIP_TRY(hipStreamSynchronize(stream()));
    if (n_violations) *n_violations = res[1];
    if (res[1] && p->n_constraints) {
        if (first_row) *first_row = res[0] / p->n_constraints;
        if (first_constraint) *first_constraint = (uint32_t)(res[0] % p->n_constraints);
    }
    return (int)hipGetLastError();
}

// ---- single stages -----------------------------------------------------------------------------------

extern "C" int pw_lde_batch(const uint32_t* d_trace, uint32_t width, uint32_t log_h, uint32_t* d_coeffs, uint32_t* d_lde) {
    (void)hipGetLastError();
    const size_t H = (size_t)1 << log_h;
    TRY(intt_dif(d_trace, d_coeffs, H, H, width, (int)log_h));
    TRY(coset_lde_from_coeffs(d_coeffs, d_lde, H, 2 * H, width, (int)log_h));
    return (int)hipGetLastError();
}

extern "C" int pw_lde_fused(const uint32_t* d_trace, uint32_t width, uint32_t log_h, uint32_t* d_tmp, uint32_t* d_lde) {
    (void)hipGetLastError();
    const size_t H = (size_t)1 << log_h;
    TRY(lde_fused(d_trace, d_tmp, d_lde, H, H, 2 * H, width, (int)log_h));
    return (int)hipGetLastError();
}

extern "C" int pw_lde_subcoset(const uint32_t* d_coeffs, uint32_t width, uint32_t log_h, uint32_t log_blocks, uint32_t r, uint32_t* d_scale,
                               uint32_t* d_out) {
    (void)hipGetLastError();
    if (!d_coeffs || !d_scale || !d_out || !width || log_blocks < 1 || log_blocks > log_h || log_h > 26 || (r >> log_blocks)) return -1;
    if ((uintptr_t)d_coeffs & 15) return (int)hipErrorInvalidValue;  // (16-byte staged loads, as above)
    const size_t H = (size_t)1 << log_h, m = (2 * H) >> log_blocks;
    TRY(subcoset_lde(d_coeffs, d_out, H, m, width, (int)log_h, (int)log_blocks, r, d_scale));
    return (int)hipGetLastError();
}

extern "C" int pw_merkle_commit(const uint32_t* d_matrix, size_t height, uint32_t width, uint32_t* d_digests) {
    (void)hipGetLastError();
    if (!height || (height & (height - 1))) return -1;
    return merkle_commit_matrix(d_matrix, height, width, height, d_digests);
}

extern "C" int pw_set_poseidon2_constants(const uint32_t* ext_rc, const uint32_t* int_rc) {
    return poseidon2_set_constants(ext_rc, int_rc);
}

extern "C" void pw_get_poseidon2_constants(uint32_t* ext_rc, uint32_t* int_rc, uint32_t* diag) {
    const p2::Params& p = poseidon2_params_host();
    if (ext_rc) for (int r = 0; r < 8; ++r) for (int i = 0; i < 16; ++i) ext_rc[16 * r + i] = bb::from_monty(p.ext_rc[r][i]);
    if (int_rc) for (int r = 0; r < 13; ++r) int_rc[r] = bb::from_monty(p.int_rc[r]);
    if (diag) for (int i = 0; i < 16; ++i) diag[i] = bb::from_monty(p.diag[i]);
}

extern "C" void pw_poseidon2_permute_host(uint32_t* s) {
    uint32_t m[16];
    for (int i = 0; i < 16; ++i) m[i] = bb::to_monty(s[i] % bb::P);
    p2::permute(m, poseidon2_params_host());
    for (int i = 0; i < 16; ++i) s[i] = bb::from_monty(m[i]);
}
