// The prover's use of run-time specialised kernels (jit.hpp, jit_codegen.hpp): which provers get them, compilation of all
// translation units in one concurrent batch, and the three stages that run them — quotient (constraints only), quotient with
// the LogUp terms, LogUp permutation columns. The interpreter kernels (stark_kernels.hip, logup_kernels.hip) compute the same
// canonical words; POWDR_JIT=0 / 1 forces either path, the tests compare proofs made with both.
#include "prover_state.hpp"

#include <algorithm>
#include <cstdlib>
#include <thread>

#define PW_TRY_INT(x) do { const int _rc = (x); if (_rc) return _rc; } while (0)

namespace pw {

namespace {

uint32_t env_u32(const char* name, uint32_t dflt, uint32_t lo, uint32_t hi) {
    if (const char* e = getenv(name)) { const long v = atol(e); if (v >= (long)lo && v <= (long)hi) return (uint32_t)v; }
    return dflt;
}

jit::LogupView logup_view(const PwProver* p) {
    return jit::LogupView{p->h_inter.data(), (uint32_t)p->h_inter.size(), jit::XbcView{p->h_icode.data(), p->h_ixspans.data(), (uint32_t)(p->h_ixspans.size() / 2)},
                          p->h_gstarts.data(), p->n_groups};
}

int launch_unit(const jit::Generated& g, const std::vector<jit::ProgramPtr>& progs, size_t u, size_t rows, void** args) {
    std::string err;
    hipFunction_t f = jit::kernel(*progs[u], g.units[u].kernel.c_str(), &err);
    if (!f) return (int)hipErrorSharedObjectInitFailed;
    const int rc = jit::launch(f, dim3(div_up(rows, 256), g.units[u].n_chunks), dim3(256), args, stream());
    if (rc) return rc;
    call_stats()[kStatJitKernelLaunches] += 1;
    return 0;
}
int launch_units(const jit::Generated& g, const std::vector<jit::ProgramPtr>& progs, size_t rows, void** args) {
    for (size_t u = 0; u < g.units.size(); ++u) {
        const int rc = launch_unit(g, progs, u, rows, args);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

// the source generator alone, with explicit chunking parameters (pw_jit_generated_source: a test hook)
jit::Generated generate_sources(const PwProver* p, int which, uint32_t chunk_cost, uint32_t chunks_per_unit) {
    if (!p || !p->is_xbc) return {};
    const jit::XbcView cons{p->h_xcode.data(), p->h_xspans.data(), p->n_constraints};
    if (which == 0) {
        if (!p->logup) return jit::gen_quotient(cons, nullptr, chunk_cost, chunks_per_unit);
        const jit::LogupView lv = logup_view(p);
        return jit::gen_quotient(cons, &lv, chunk_cost, chunks_per_unit);
    }
    if (which == 1 && p->logup) return jit::gen_logup_perm(logup_view(p), chunk_cost, chunks_per_unit);
    return {};
}

int specialise_provers(PwProver* const* ps, size_t n, const uint32_t* log_heights, bool force) {
    // POWDR_JIT: 0 = never, 1 = always, unset = for traces of at least 2^POWDR_JIT_MIN_LOG_HEIGHT rows (default 18: compiling
    // costs ~0.3 ms of host time per emitted instruction, paid once per prover and amortised over its segments)
    const char* e = getenv("POWDR_JIT");
    const bool always = force || (e && atoi(e) == 1);
    const uint32_t min_log = env_u32("POWDR_JIT_MIN_LOG_HEIGHT", 18, 1, 40);
    const uint32_t chunk_cost = env_u32("POWDR_JIT_CHUNK_COST", 8000, 200, 1000000);
    // chunks per translation unit: few enough that every compiler thread gets a unit, at most POWDR_JIT_UNIT_CHUNKS (8)
    const uint32_t per_unit_max = env_u32("POWDR_JIT_UNIT_CHUNKS", 8, 1, 4096);
    unsigned hw = std::thread::hardware_concurrency();
    hw = hw > 32 ? 32 : hw < 1 ? 1 : hw;
    auto per_unit_for = [&](size_t cost) {
        const size_t chunks = cost / chunk_cost + 1;
        const size_t want = (chunks + hw - 1) / hw;
        return (uint32_t)(want < 1 ? 1 : want > per_unit_max ? per_unit_max : want);
    };
    std::vector<PwProver*> todo;
    for (size_t i = 0; i < n; ++i) {
        PwProver* p = ps[i];
        if (!p || p->jit.state != 0) continue;
        if (!always && (!log_heights || log_heights[i] < min_log)) continue;  // stays untried: a taller trace may come later
        if (!jit::available() || !p->is_xbc || (p->n_constraints == 0 && !p->logup)) { p->jit.state = -1; p->jit.error = "not available"; continue; }
        const jit::XbcView cons{p->h_xcode.data(), p->h_xspans.data(), p->n_constraints};
        const uint32_t per_unit = per_unit_for(6 * (p->h_xcode.size() / 2 + p->h_icode.size() / 2) + 300 * (size_t)p->n_groups);
        if (p->logup) {
            const jit::LogupView lv = logup_view(p);
            p->jit.quotient = jit::gen_quotient(cons, &lv, chunk_cost, per_unit);
            p->jit.perm = jit::gen_logup_perm(lv, chunk_cost, per_unit);
            if (p->n_groups && p->jit.perm.units.empty()) { p->jit.state = -1; p->jit.error = "code generation failed"; continue; }
        } else {
            p->jit.quotient = jit::gen_quotient(cons, nullptr, chunk_cost, per_unit);
        }
        if (p->jit.quotient.units.empty() && (p->n_constraints || p->n_groups)) { p->jit.state = -1; p->jit.error = "code generation failed"; continue; }
        todo.push_back(p);
    }
    if (todo.empty()) return 0;
    std::vector<std::string> sources;
    for (PwProver* p : todo) {
        for (auto& u : p->jit.quotient.units) sources.push_back(u.source);
        for (auto& u : p->jit.perm.units) sources.push_back(u.source);
    }
    std::string err;
    std::vector<jit::ProgramPtr> progs = jit::compile_all(sources, &err);
    if (progs.empty() && todo.size() > 1) {
        // one AIR's unit did not compile: the others must not lose their kernels with it — every prover on its own
        for (PwProver* p : todo) { PwProver* one[1] = {p}; uint32_t lh = 40; p->jit.state = 0; p->jit.quotient = {}; p->jit.perm = {}; (void)specialise_provers(one, 1, &lh, true); }
        return 0;
    }
    int n_dev = 0;
    const bool have_device = hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0;  // (pw_jit_compile_check cross-compiles without a GPU)
    if (!have_device) (void)hipGetLastError();
    size_t k = 0;
    for (PwProver* p : todo) {
        if (progs.empty()) { p->jit.state = -1; p->jit.error = err; p->jit.quotient = {}; p->jit.perm = {}; continue; }
        p->jit.quotient_prog.assign(progs.begin() + (long)k, progs.begin() + (long)(k + p->jit.quotient.units.size()));
        k += p->jit.quotient.units.size();
        p->jit.perm_prog.assign(progs.begin() + (long)k, progs.begin() + (long)(k + p->jit.perm.units.size()));
        k += p->jit.perm.units.size();
        for (auto& u : p->jit.quotient.units) { u.source.clear(); u.source.shrink_to_fit(); }  // the programs keep their text
        for (auto& u : p->jit.perm.units) { u.source.clear(); u.source.shrink_to_fit(); }
        p->jit.state = 1;
        if (have_device) {
            // load the modules and resolve every kernel NOW: a code object this device cannot run (a stale cache entry after a ROCm
            // upgrade, a device that is not gfx950) must leave the prover with the interpreter, not fail its first tall proof
            std::string e2;
            bool ok = true;
            for (size_t u = 0; u < p->jit.quotient.units.size() && ok; ++u) ok = jit::kernel(*p->jit.quotient_prog[u], p->jit.quotient.units[u].kernel.c_str(), &e2) != nullptr;
            for (size_t u = 0; u < p->jit.perm.units.size() && ok; ++u) ok = jit::kernel(*p->jit.perm_prog[u], p->jit.perm.units[u].kernel.c_str(), &e2) != nullptr;
            if (!ok) {
                (void)hipGetLastError();
                p->jit.state = -1;
                p->jit.error = "the compiled kernels do not load on this device: " + e2;
                p->jit.quotient_prog.clear(); p->jit.perm_prog.clear(); p->jit.quotient = {}; p->jit.perm = {};
            }
        }
    }
    return 0;
}

size_t jit_part_bytes(const PwProver* p, size_t H, size_t q_rows) {
    if (p->jit.state != 1) return 0;
    const size_t a = (size_t)(p->jit.perm.n_chunks ? p->jit.perm.n_chunks : (p->logup ? 1 : 0)) * 16 * H;
    const size_t b = (size_t)(p->jit.quotient.n_chunks ? p->jit.quotient.n_chunks : 1) * 16 * q_rows;
    return a > b ? a : b;
}

int quotient_parts_jit(PwProver* p, const uint32_t* T, const uint32_t* Pm, size_t rows, const bb::Ext* d_apow, bb::Ext al, const bb::Ext* d_blpow,
                       uint32_t* part, uint32_t* n_chunks) {
    const jit::Generated& g = p->jit.quotient;
    uint64_t n64 = rows;
    void* args[] = {(void*)&T, (void*)&Pm, (void*)&n64, (void*)&d_apow, (void*)&al, (void*)&d_blpow, (void*)&part};
    ScopedKernelTimer t(Pm ? "quotient_logup_jit_kernel" : "quotient_jit_kernel");
    const int rc = launch_units(g, p->jit.quotient_prog, rows, args);
    if (n_chunks) *n_chunks = g.n_chunks;
    return rc;
}

uint32_t quotient_units_jit(const PwProver* p) { return (uint32_t)p->jit.quotient.units.size(); }
void quotient_unit_groups(const PwProver* p, uint32_t u, uint32_t* g0, uint32_t* g1) {
    *g0 = p->jit.quotient.units[u].g0;
    *g1 = p->jit.quotient.units[u].g1;
}
uint32_t quotient_max_unit_perm_cols(const PwProver* p) {
    uint32_t w = 0;
    for (const auto& u : p->jit.quotient.units) w = std::max(w, 4 * (u.g1 - u.g0));
    return w;
}
int quotient_unit_jit(PwProver* p, uint32_t u, const uint32_t* T, const uint32_t* Pm, size_t rows, const bb::Ext* d_apow, bb::Ext al,
                      const bb::Ext* d_blpow, uint32_t* part) {
    const jit::Generated& g = p->jit.quotient;
    // (a chunk stores to part + <its program-wide chunk id> * 4 * rows — the id is a literal in the generated code — so every unit
    // takes the same base pointer)
    uint64_t n64 = rows;
    void* args[] = {(void*)&T, (void*)&Pm, (void*)&n64, (void*)&d_apow, (void*)&al, (void*)&d_blpow, (void*)&part};
    ScopedKernelTimer t(Pm ? "quotient_logup_jit_kernel" : "quotient_jit_kernel");
    return launch_unit(g, p->jit.quotient_prog, u, rows, args);
}

int quotient_eval_jit(PwProver* p, const uint32_t* lde, size_t N, const bb::Ext* d_apow, uint32_t zinv_even, uint32_t zinv_odd, uint32_t* q) {
    PW_TRY_INT(p->qpart.ensure(jit_part_bytes(p, N / 2, N)));  // (reserved by ensure_prove_buffers; a no-op then)
    uint32_t* part = p->qpart.as<uint32_t>();
    uint32_t n_chunks = 0;
    PW_TRY_INT(quotient_parts_jit(p, lde, nullptr, N, d_apow, bb::ext_zero(), nullptr, part, &n_chunks));
    return quotient_combine(part, n_chunks, N, zinv_even, zinv_odd, q);
}

int quotient_eval_logup_jit(PwProver* p, const uint32_t* lde, const uint32_t* plde, size_t N, int logN, const bb::Ext* d_apow, bb::Ext al,
                            const bb::Ext* d_blpow, bb::Ext S, uint32_t zval_even, uint32_t zval_odd, uint32_t* q) {
    PW_TRY_INT(p->qpart.ensure(jit_part_bytes(p, N / 2, N)));
    uint32_t* part = p->qpart.as<uint32_t>();
    uint32_t n_chunks = 0;
    PW_TRY_INT(quotient_parts_jit(p, lde, plde, N, d_apow, al, d_blpow, part, &n_chunks));
    const uint32_t G = p->n_groups;
    return quotient_logup_tail(part, n_chunks, plde + (size_t)(4 * G) * N, plde + (size_t)(4 * G + 4) * N, N, logN, d_apow + p->n_constraints + G, S,
                               zval_even, zval_odd, q);
}

int logup_perm_trace_jit(PwProver* p, const uint32_t* trace, size_t H, bb::Ext al, const bb::Ext* d_blpow, uint32_t* perm, bb::Ext* d_rowsum,
                         bb::Ext* d_block_totals) {
    const jit::Generated& g = p->jit.perm;
    const uint32_t G = p->n_groups;
    PW_TRY_INT(p->qpart.ensure((size_t)(g.n_chunks ? g.n_chunks : 1) * 4 * H * 4));  // (a no-op after ensure_prove_buffers / ensure_air)
    uint32_t* part = p->qpart.as<uint32_t>();
    uint64_t h64 = H;
    void* args[] = {(void*)&trace, (void*)&h64, (void*)&al, (void*)&d_blpow, (void*)&perm, (void*)&part};
    {
        ScopedKernelTimer t("logup_perm_jit_kernel");
        const int rc = launch_units(g, p->jit.perm_prog, H, args);
        if (rc) return rc;
    }
    int rc = logup_rowsum_combine(part, g.n_chunks, H, d_rowsum, perm + (size_t)(4 * G + 4) * H);
    if (rc) return rc;
    return logup_scan(d_rowsum, H, d_block_totals, perm + (size_t)(4 * G) * H);
}

}  // namespace pw

extern "C" int pw_prover_specialise(PwProver* p) {
    if (!p) return -1;
    PwProver* ps[1] = {p};
    (void)pw::specialise_provers(ps, 1, nullptr, true);
    return p->jit.state == 1 ? 0 : 1;
}

// Every prover of `ps` that has no specialised kernels yet, compiled in ONE concurrent batch (all translation units of all AIRs next to
// each other on the compiler's helper processes) whatever the trace heights: what an embedder does once per AIR set — the reference fixes
// an APC's AIR at key generation and proves it in every segment. Returns how many of the n provers run specialised kernels afterwards.
extern "C" size_t pw_provers_specialise(PwProver* const* ps, size_t n) {
    if (!ps || !n) return 0;
    (void)pw::specialise_provers(ps, n, nullptr, true);
    size_t ok = 0;
    for (size_t i = 0; i < n; ++i) ok += ps[i] && ps[i]->jit.state == 1;
    return ok;
}

extern "C" int pw_prover_specialised(const PwProver* p, size_t* n_kernels, size_t* code_bytes, size_t* n_chunks) {
    if (!p) return -1;
    size_t k = 0, b = 0, c = 0;
    if (p->jit.state == 1) {
        k = p->jit.quotient_prog.size() + p->jit.perm_prog.size();
        for (auto& pr : p->jit.quotient_prog) b += pw::jit::code_bytes(*pr);
        for (auto& pr : p->jit.perm_prog) b += pw::jit::code_bytes(*pr);
        c = p->jit.quotient.n_chunks + p->jit.perm.n_chunks;
    }
    if (n_kernels) *n_kernels = k;
    if (code_bytes) *code_bytes = b;
    if (n_chunks) *n_chunks = c;
    return p->jit.state;
}

extern "C" void pw_jit_cache_stats(uint64_t* units_compiled, uint64_t* units_from_disk) { pw::jit::cache_stats(units_compiled, units_from_disk); }
