// hiprtc plumbing of the run-time specialised kernels (see ../jit.hpp).
#include "../jit.hpp"
#include "../common.hpp"

#include <dlfcn.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

extern char** environ;

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
        int current = 0;
        const bool have = hipGetDevice(&current) == hipSuccess;
        for (auto& kv : modules)
            if (hipSetDevice(kv.first) == hipSuccess) (void)hipModuleUnload(kv.second);
        if (have) (void)hipSetDevice(current);
        (void)hipGetLastError();
    }
};

namespace {
// Programs are shared between the provers that are alive (two provers of the same AIR, the same chunk in two AIRs) and go away with
// the last of them: the cache holds weak references, so a long-running process that sees many different AIRs does not accumulate
// code objects and loaded modules.
std::mutex g_cache_mu;
std::unordered_map<uint64_t, std::weak_ptr<Program>> g_cache;

bool compile_code(const std::string& src, std::vector<char>& code_out, std::string* err) {
    const Rtc& r = rtc();
    if (!r.ok) { if (err) *err = "hiprtc is not available"; return false; }
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
    if (ok) { code_out.resize(n); ok = r.GetCode(prog, code_out.data()) == 0; }
    (void)r.DestroyProgram(&prog);
    if (!ok && err) *err = "hiprtcGetCode failed";
    return ok;
}

// path of the helper executable: POWDR_JITC, or "powdr_jitc" next to the shared object this code lives in
std::string helper_path() {
    if (const char* e = getenv("POWDR_JITC")) return e;
    Dl_info info;
    if (!dladdr((const void*)&helper_path, &info) || !info.dli_fname) return "";
    std::string p = info.dli_fname;
    const size_t slash = p.rfind('/');
    p = slash == std::string::npos ? "powdr_jitc" : p.substr(0, slash + 1) + "powdr_jitc";
    return access(p.c_str(), X_OK) == 0 ? p : "";
}

bool read_file(const std::string& path, std::vector<char>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n > 0 && fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// Compile sources[todo[k]] in `n_procs` helper processes (round robin). false = the helper could not be used at all (the
// caller falls back to this process); a compiler error is reported through *err with `compiled` left incomplete.
bool compile_with_helpers(const std::string& helper, const std::vector<std::string>& sources, const std::vector<size_t>& todo,
                          unsigned n_procs, std::vector<std::vector<char>>& code, bool& compiled, std::string* err) {
    compiled = false;
    const char* base = getenv("TMPDIR");
    std::string dir = std::string(base && *base ? base : "/tmp") + "/powdr_jit_XXXXXX";
    if (!mkdtemp(&dir[0])) return false;
    std::vector<std::string> src_path(todo.size()), out_path(todo.size());
    bool ok = true;
    for (size_t k = 0; k < todo.size() && ok; ++k) {
        src_path[k] = dir + "/u" + std::to_string(k) + ".hip";
        out_path[k] = dir + "/u" + std::to_string(k) + ".co";
        FILE* f = fopen(src_path[k].c_str(), "wb");
        ok = f && fwrite(sources[todo[k]].data(), 1, sources[todo[k]].size(), f) == sources[todo[k]].size();
        if (f) fclose(f);
    }
    std::vector<pid_t> pids;
    if (ok) {
        for (unsigned p = 0; p < n_procs; ++p) {
            std::vector<std::string> args{helper};
            for (size_t k = p; k < todo.size(); k += n_procs) { args.push_back(src_path[k]); args.push_back(out_path[k]); }
            if (args.size() == 1) continue;
            std::vector<char*> argv;
            for (auto& a : args) argv.push_back(&a[0]);
            argv.push_back(nullptr);
            pid_t pid = 0;
            if (posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv.data(), environ) != 0) { ok = false; break; }
            pids.push_back(pid);
        }
    }
    bool all_zero = ok;
    for (pid_t pid : pids) {
        int status = 0;
        pid_t got;
        do got = waitpid(pid, &status, 0); while (got == -1 && errno == EINTR);
        if (got != pid || !WIFEXITED(status) || WEXITSTATUS(status) != 0) all_zero = false;
    }
    bool usable = ok;
    if (ok) {
        compiled = all_zero;
        for (size_t k = 0; k < todo.size() && compiled; ++k) compiled = read_file(out_path[k], code[k]);
        if (!compiled) {
            std::string msg;
            for (size_t k = 0; k < todo.size() && msg.empty(); ++k) {
                std::vector<char> e;
                if (read_file(out_path[k] + ".err", e)) msg.assign(e.begin(), e.end());
            }
            if (msg.empty()) usable = false;  // the helper itself failed (missing library, killed): let this process try
            else if (err) *err = msg;
        }
    }
    for (size_t k = 0; k < todo.size(); ++k) { (void)unlink(src_path[k].c_str()); (void)unlink(out_path[k].c_str()); (void)unlink((out_path[k] + ".err").c_str()); }
    (void)rmdir(dir.c_str());
    return usable;
}
}  // namespace

bool compile_to_code_object(const std::string& source, std::vector<char>& code, std::string* err) { return compile_code(source, code, err); }

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
            ProgramPtr hit = it != g_cache.end() ? it->second.lock() : nullptr;
            if (hit && hit->source == sources[i]) out[i] = hit;
            else todo.push_back(i);
        }
    }
    if (!todo.empty()) {
        unsigned n_procs = std::thread::hardware_concurrency();
        if (n_procs > 32) n_procs = 32;
        if (const char* e = getenv("POWDR_JIT_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) n_procs = (unsigned)v; }
        if (n_procs < 1) n_procs = 1;
        if (n_procs > todo.size()) n_procs = (unsigned)todo.size();
        std::vector<std::vector<char>> code(todo.size());
        bool compiled = false;
        const std::string helper = n_procs > 1 ? helper_path() : std::string();
        std::string e;
        if (!helper.empty() && compile_with_helpers(helper, sources, todo, n_procs, code, compiled, &e)) {
            if (!compiled) { if (err) *err = e; return {}; }
        } else {
            for (size_t k = 0; k < todo.size(); ++k)
                if (!compile_code(sources[todo[k]], code[k], &e)) { if (err) *err = e; return {}; }
        }
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t k = 0; k < todo.size(); ++k) {
            const size_t i = todo[k];
            auto p = std::make_shared<Program>();
            p->source = sources[i];
            p->code = std::move(code[k]);
            std::weak_ptr<Program>& slot = g_cache[hash64(sources[i])];
            ProgramPtr other = slot.lock();
            if (other && other->source == sources[i]) out[i] = other;  // another thread was faster
            else { slot = p; out[i] = p; }                             // empty / expired slot, or a hash collision: the newer text takes it
        }
        if (g_cache.size() > 4096)  // expired slots of AIRs long gone
            for (auto it = g_cache.begin(); it != g_cache.end();) it = it->second.expired() ? g_cache.erase(it) : std::next(it);
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
