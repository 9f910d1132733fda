// hiprtc plumbing of the run-time specialised kernels (see ../jit.hpp).
#include "../jit.hpp"
#include "../common.hpp"

#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace pw { namespace jit {

namespace {

// the few hiprtc entry points used, resolved with dlsym (hiprtc.h is not included: the library must load without hiprtc)
typedef struct _hiprtcProgram* hiprtcProgram;
struct Rtc {
    void* handle = nullptr;
    int (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*CompileProgram)(hiprtcProgram, int, const char* const*) = nullptr;
    int (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
    int (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
    int (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
    int (*GetCode)(hiprtcProgram, char*) = nullptr;
    int (*DestroyProgram)(hiprtcProgram*) = nullptr;
    bool ok = false;
};

const Rtc& rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> names;
        if (const char* e = getenv("POWDR_HIPRTC_LIB")) names.push_back(e);
        for (const char* n : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) names.push_back(n);
        for (auto& n : names) {
            r.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        auto sym = [&](const char* s) { return dlsym(r.handle, s); };
        r.CreateProgram = (decltype(r.CreateProgram))sym("hiprtcCreateProgram");
        r.CompileProgram = (decltype(r.CompileProgram))sym("hiprtcCompileProgram");
        r.GetProgramLogSize = (decltype(r.GetProgramLogSize))sym("hiprtcGetProgramLogSize");
        r.GetProgramLog = (decltype(r.GetProgramLog))sym("hiprtcGetProgramLog");
        r.GetCodeSize = (decltype(r.GetCodeSize))sym("hiprtcGetCodeSize");
        r.GetCode = (decltype(r.GetCode))sym("hiprtcGetCode");
        r.DestroyProgram = (decltype(r.DestroyProgram))sym("hiprtcDestroyProgram");
        r.ok = r.CreateProgram && r.CompileProgram && r.GetProgramLogSize && r.GetProgramLog && r.GetCodeSize && r.GetCode && r.DestroyProgram;
    });
    return r;
}

uint64_t hash64(const std::string& s) {
    uint64_t h = 1469598103934665603ull;
    const uint64_t* w = reinterpret_cast<const uint64_t*>(s.data());
    const size_t k = s.size() / 8;
    for (size_t i = 0; i < k; ++i) { uint64_t v; memcpy(&v, w + i, 8); h ^= v; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    for (size_t i = k * 8; i < s.size(); ++i) { h ^= (unsigned char)s[i]; h *= 1099511628211ull; }
    return h;
}

}  // namespace

struct Program {
    std::string source;      // kept: a cache hit is confirmed by comparing the text, not the hash alone
    std::vector<char> code;  // gfx950 code object
    std::mutex mu;
    std::map<int, hipModule_t> modules;  // device -> loaded module
    ~Program() {
        for (auto& kv : modules) (void)hipModuleUnload(kv.second);
    }
};

namespace {
std::mutex g_cache_mu;
std::unordered_map<uint64_t, ProgramPtr> g_cache;

bool compile_one(const std::string& src, Program& out, std::string* err) {
    const Rtc& r = rtc();
    std::vector<const char*> hdrs, names;
    for (int i = 0; i < kNumEmbeddedHeaders; ++i) { hdrs.push_back(kEmbeddedHeaders[i].text); names.push_back(kEmbeddedHeaders[i].name); }
    hiprtcProgram prog = nullptr;
    if (r.CreateProgram(&prog, src.c_str(), "powdr_jit.hip", (int)hdrs.size(), hdrs.data(), names.data()) != 0) {
        if (err) *err = "hiprtcCreateProgram failed";
        return false;
    }
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    const int rc = r.CompileProgram(prog, 3, opts);
    if (rc != 0) {
        size_t n = 0;
        std::string log;
        if (r.GetProgramLogSize(prog, &n) == 0 && n > 1) { log.resize(n); (void)r.GetProgramLog(prog, &log[0]); }
        if (err) *err = "hiprtcCompileProgram failed (" + std::to_string(rc) + "): " + log.substr(0, 2000);
        (void)r.DestroyProgram(&prog);
        return false;
    }
    size_t n = 0;
    bool ok = r.GetCodeSize(prog, &n) == 0 && n > 0;
    if (ok) { out.code.resize(n); ok = r.GetCode(prog, out.code.data()) == 0; }
    (void)r.DestroyProgram(&prog);
    if (!ok && err) *err = "hiprtcGetCode failed";
    out.source = src;
    return ok;
}
}  // namespace

bool available() {
    if (const char* e = getenv("POWDR_JIT")) if (atoi(e) == 0) return false;
    return rtc().ok;
}

std::vector<ProgramPtr> compile_all(const std::vector<std::string>& sources, std::string* err) {
    std::vector<ProgramPtr> out(sources.size());
    if (!rtc().ok) { if (err) *err = "hiprtc is not available"; return {}; }
    std::vector<size_t> todo;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < sources.size(); ++i) {
            auto it = g_cache.find(hash64(sources[i]));
            if (it != g_cache.end() && it->second->source == sources[i]) out[i] = it->second;
            else todo.push_back(i);
        }
    }
    if (!todo.empty()) {
        unsigned n_threads = std::thread::hardware_concurrency();
        if (n_threads > 32) n_threads = 32;
        if (const char* e = getenv("POWDR_JIT_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) n_threads = (unsigned)v; }
        if (n_threads < 1) n_threads = 1;
        if (n_threads > todo.size()) n_threads = (unsigned)todo.size();
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        std::mutex err_mu;
        auto work = [&] {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= todo.size() || failed.load()) return;
                const size_t i = todo[k];
                auto p = std::make_shared<Program>();
                std::string e;
                if (!compile_one(sources[i], *p, &e)) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (!failed.exchange(true) && err) *err = e;
                    return;
                }
                out[i] = std::move(p);
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < n_threads; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
        if (failed.load()) return {};
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i : todo) {
            auto ins = g_cache.emplace(hash64(sources[i]), out[i]);
            if (!ins.second && ins.first->second->source == sources[i]) out[i] = ins.first->second;  // another thread was faster
        }
    }
    return out;
}

hipFunction_t kernel(Program& p, const char* name, std::string* err) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { if (err) *err = "hipGetDevice failed"; return nullptr; }
    std::lock_guard<std::mutex> lk(p.mu);
    auto it = p.modules.find(device);
    if (it == p.modules.end()) {
        hipModule_t m = nullptr;
        const hipError_t e = hipModuleLoadData(&m, p.code.data());
        if (e != hipSuccess) { if (err) *err = std::string("hipModuleLoadData: ") + hipGetErrorString(e); (void)hipGetLastError(); return nullptr; }
        it = p.modules.emplace(device, m).first;
    }
    hipFunction_t f = nullptr;
    const hipError_t e = hipModuleGetFunction(&f, it->second, name);
    if (e != hipSuccess) { if (err) *err = std::string("hipModuleGetFunction(") + name + "): " + hipGetErrorString(e); (void)hipGetLastError(); return nullptr; }
    return f;
}

int launch(hipFunction_t f, dim3 grid, dim3 block, void** args, hipStream_t st) {
    return (int)hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, 0, st, args, nullptr);
}

size_t code_bytes(const Program& p) { return p.code.size(); }

}}  // namespace pw::jit
