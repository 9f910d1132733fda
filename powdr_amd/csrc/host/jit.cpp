// hiprtc plumbing of the run-time specialised kernels (see ../jit.hpp).
#include "../jit.hpp"
#include "../common.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/file.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
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
    int (*Version)(int*, int*) = nullptr;
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
        r.Version = (decltype(r.Version))sym("hiprtcVersion");
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

// ---- on-disk cache of code objects -------------------------------------------------------------------------------------------
// A code object depends on the unit's source, the embedded headers and the compile options only, so it is kept across processes:
// $POWDR_JIT_CACHE_DIR, else $XDG_CACHE_HOME/powdr_jit, else $HOME/.cache/powdr_jit (POWDR_JIT_CACHE=0: no disk cache). Every rank
// of a node, every test process and every later run of a prover then loads what the first one compiled. An entry holds the
// environment hash and the FULL source next to the code, a hit is confirmed by comparing them (the file name is only a hash);
// entries are written to a private temporary name and renamed into place.
namespace {
constexpr char kDiskMagic[8] = {'P', 'W', 'J', 'C', '0', '0', '0', '1'};
std::atomic<uint64_t> g_units_compiled{0}, g_units_from_disk{0};  // process-wide, for pw_jit_cache_stats

uint64_t environment_hash() {
    static const uint64_t h = [] {
        std::string all = "--offload-arch=gfx950 -O3 -std=c++17";
        // the compiler's own version: a code object built by another ROCm release is not reused after an upgrade
        int major = 0, minor = 0, runtime = 0;
        if (rtc().ok && rtc().Version) (void)rtc().Version(&major, &minor);
        if (hipRuntimeGetVersion(&runtime) != hipSuccess) { runtime = 0; (void)hipGetLastError(); }
        all += " hiprtc=" + std::to_string(major) + "." + std::to_string(minor) + " hip=" + std::to_string(runtime);
        for (int i = 0; i < kNumEmbeddedHeaders; ++i) { all += '\0'; all += kEmbeddedHeaders[i].name; all += '\0'; all += kEmbeddedHeaders[i].text; }
        return hash64(all);
    }();
    return h;
}

std::string disk_cache_dir() {
    if (const char* e = getenv("POWDR_JIT_CACHE")) if (atoi(e) == 0) return "";
    std::string dir;
    if (const char* e = getenv("POWDR_JIT_CACHE_DIR")) dir = e;
    else if (const char* x = getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/powdr_jit";
    else if (const char* h = getenv("HOME")) dir = std::string(h) + "/.cache/powdr_jit";
    if (dir.empty()) return "";
    for (size_t i = 1; i <= dir.size(); ++i)  // mkdir -p; the cache directory itself is private
        if (i == dir.size() || dir[i] == '/') { const std::string part = dir.substr(0, i); if (mkdir(part.c_str(), i == dir.size() ? 0700 : 0755) != 0 && errno != EEXIST) return ""; }
    // code objects are LOADED from here: only a directory that belongs to the caller and that nobody else can write to is trusted
    // (a predictable path under a world-writable parent could otherwise be prepared by another local user)
    struct stat sb;
    const char* why = nullptr;
    if (lstat(dir.c_str(), &sb) != 0) why = "cannot be examined";
    else if (S_ISLNK(sb.st_mode)) why = "is a symbolic link";
    else if (!S_ISDIR(sb.st_mode)) why = "is not a directory";
    else if (sb.st_uid != geteuid()) why = "belongs to another user";
    else if (sb.st_mode & (S_IWGRP | S_IWOTH)) why = "is writable by group or others (chmod go-w)";
    if (why) {
        // said once: every process that gets here recompiles its kernels (seconds per AIR) instead of loading them
        static std::atomic<bool> said{false};
        if (!said.exchange(true)) fprintf(stderr, "powdr jit: the code-object cache %s %s: not used, kernels are recompiled\n", dir.c_str(), why);
        return "";
    }
    return dir;
}

// One compiler at a time per cache directory, ACROSS processes (eight ranks of one node specialise the same AIRs against one cache:
// the first one compiles, the others wait here and then find the code objects on disk). flock on <dir>/.lock; released by close().
struct DirLock {
    int fd = -1;
    explicit DirLock(const std::string& dir) {
        if (dir.empty()) return;
        if (const char* e = getenv("POWDR_JIT_CACHE_LOCK")) if (atoi(e) == 0) return;
        fd = open((dir + "/.lock").c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
        if (fd >= 0 && flock(fd, LOCK_EX) != 0) { close(fd); fd = -1; }
    }
    ~DirLock() { if (fd >= 0) close(fd); }
    DirLock(const DirLock&) = delete;
    DirLock& operator=(const DirLock&) = delete;
};

std::string disk_entry_path(const std::string& dir, const std::string& source) {
    char name[64];
    snprintf(name, sizeof name, "/%016llx-%016llx.pwjc", (unsigned long long)environment_hash(), (unsigned long long)hash64(source));
    return dir + name;
}

bool disk_load(const std::string& dir, const std::string& source, std::vector<char>& code) {
    std::vector<char> f;
    if (dir.empty() || !read_file(disk_entry_path(dir, source), f)) return false;
    const size_t head = 8 + 8 + 8;
    if (f.size() < head + 8 || memcmp(f.data(), kDiskMagic, 8) != 0) return false;
    uint64_t env = 0, src_len = 0, code_len = 0;
    memcpy(&env, f.data() + 8, 8); memcpy(&src_len, f.data() + 16, 8);
    if (env != environment_hash() || src_len != source.size() || f.size() < head + src_len + 8) return false;
    if (memcmp(f.data() + head, source.data(), src_len) != 0) return false;
    memcpy(&code_len, f.data() + head + src_len, 8);
    if (code_len == 0 || f.size() != head + src_len + 8 + code_len) return false;
    code.assign(f.begin() + (long)(head + src_len + 8), f.end());
    return true;
}

void disk_store(const std::string& dir, const std::string& source, const std::vector<char>& code) {
    if (dir.empty() || code.empty()) return;
    static std::atomic<unsigned> serial{0};
    const std::string path = disk_entry_path(dir, source);
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1));
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const uint64_t env = environment_hash(), src_len = source.size(), code_len = code.size();
    bool ok = fwrite(kDiskMagic, 1, 8, f) == 8 && fwrite(&env, 8, 1, f) == 1 && fwrite(&src_len, 8, 1, f) == 1 &&
              fwrite(source.data(), 1, source.size(), f) == source.size() && fwrite(&code_len, 8, 1, f) == 1 &&
              fwrite(code.data(), 1, code.size(), f) == code.size();
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
}
}  // namespace

bool compile_to_code_object(const std::string& source, std::vector<char>& code, std::string* err) { return compile_code(source, code, err); }

void cache_stats(uint64_t* compiled, uint64_t* from_disk) {
    if (compiled) *compiled = g_units_compiled.load();
    if (from_disk) *from_disk = g_units_from_disk.load();
}

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
    // what an earlier process left on disk
    const std::string cache_dir = todo.empty() ? std::string() : disk_cache_dir();
    std::vector<std::pair<size_t, std::vector<char>>> from_disk;
    if (!cache_dir.empty()) {
        std::vector<size_t> still;
        for (size_t i : todo) {
            std::vector<char> c;
            if (disk_load(cache_dir, sources[i], c)) from_disk.emplace_back(i, std::move(c));
            else still.push_back(i);
        }
        todo.swap(still);
        g_units_from_disk += from_disk.size();
    }
    auto publish = [&](size_t i, std::vector<char>&& code_i) {  // under g_cache_mu
        auto p = std::make_shared<Program>();
        p->source = sources[i];
        p->code = std::move(code_i);
        std::weak_ptr<Program>& slot = g_cache[hash64(sources[i])];
        ProgramPtr other = slot.lock();
        if (other && other->source == sources[i]) out[i] = other;  // another thread was faster
        else { slot = p; out[i] = p; }                             // empty / expired slot, or a hash collision: the newer text takes it
    };
    if (!from_disk.empty()) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (auto& e : from_disk) publish(e.first, std::move(e.second));
    }
    if (!todo.empty()) {
        // the compile phase of this batch under the cache directory's lock; what another process compiled while this one waited is loaded
        DirLock lock(cache_dir);
        if (lock.fd >= 0) {
            std::vector<size_t> still;
            std::vector<std::pair<size_t, std::vector<char>>> late;
            for (size_t i : todo) {
                std::vector<char> c;
                if (disk_load(cache_dir, sources[i], c)) late.emplace_back(i, std::move(c));
                else still.push_back(i);
            }
            todo.swap(still);
            g_units_from_disk += late.size();
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (auto& e : late) publish(e.first, std::move(e.second));
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
            g_units_compiled += todo.size();
            for (size_t k = 0; k < todo.size(); ++k) disk_store(cache_dir, sources[todo[k]], code[k]);
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (size_t k = 0; k < todo.size(); ++k) publish(todo[k], std::move(code[k]));
            if (g_cache.size() > 4096)  // expired slots of AIRs long gone
                for (auto it = g_cache.begin(); it != g_cache.end();) it = it->second.expired() ? g_cache.erase(it) : std::next(it);
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
