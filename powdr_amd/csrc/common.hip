// libpowdr_gpu runtime plumbing: launch stream + per-kernel event timing.
#include "common.hpp"
#include "../../include/powdr_gpu.h"

#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>
#include <cstring>

namespace pw {

// Per host thread: a thread that never calls powdr_gpu_set_stream launches on the null stream, like the
// reference (cuda/mod.rs:374-378); worker threads proving different segments use their own streams.
static thread_local hipStream_t g_stream = nullptr;
static bool g_timing = false;
static std::mutex g_mu;
struct TimedLaunch {
    const char* name;
    hipEvent_t e0, e1;
};
static std::vector<TimedLaunch> g_launches;

static thread_local uint64_t g_call_stats[kStatCount] = {0};
uint64_t* call_stats() { return g_call_stats; }

hipStream_t stream() { return g_stream; }
void set_stream(hipStream_t s) { g_stream = s; }
bool timing_enabled() { return g_timing; }

void timing_begin(const char*, hipEvent_t* e0) {
    (void)hipEventCreate(e0);
    (void)hipEventRecord(*e0, g_stream);
}
void timing_end(const char* name, hipEvent_t e0) {
    hipEvent_t e1;
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e1, g_stream);
    std::lock_guard<std::mutex> lk(g_mu);
    g_launches.push_back({name, e0, e1});
}

static void clear_launches() {
    for (auto& l : g_launches) {
        (void)hipEventDestroy(l.e0);
        (void)hipEventDestroy(l.e1);
    }
    g_launches.clear();
}

}  // namespace pw

extern "C" {

void powdr_gpu_set_stream(void* s) { pw::set_stream((hipStream_t)s); }
void* powdr_gpu_get_stream(void) { return (void*)pw::stream(); }

void powdr_gpu_timing_enable(int enable) {
    std::lock_guard<std::mutex> lk(pw::g_mu);
    pw::clear_launches();
    pw::g_timing = enable != 0;
}

size_t powdr_gpu_timing_report(char* buf, size_t cap) {
    (void)hipStreamSynchronize(pw::g_stream);
    std::lock_guard<std::mutex> lk(pw::g_mu);
    std::map<std::string, std::pair<int, double>> agg;
    std::vector<std::string> order;
    for (auto& l : pw::g_launches) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, l.e0, l.e1);
        auto it = agg.find(l.name);
        if (it == agg.end()) {
            order.push_back(l.name);
            agg[l.name] = {1, (double)ms};
        } else {
            it->second.first += 1;
            it->second.second += ms;
        }
    }
    std::string out;
    char line[256];
    for (auto& n : order) {
        snprintf(line, sizeof line, "%s %d %.6f\n", n.c_str(), agg[n].first, agg[n].second);
        out += line;
    }
    if (buf && cap) {
        size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return out.size() + 1;
}

void powdr_gpu_call_stats(uint64_t* out16, int reset) {
    uint64_t* s = pw::call_stats();
    if (out16) memcpy(out16, s, pw::kStatCount * sizeof(uint64_t));
    if (reset) memset(s, 0, pw::kStatCount * sizeof(uint64_t));
}

const char* powdr_gpu_version(void) { return "powdr_gpu-mi355x 0.1 (gfx950; Fp=BabyBear Montgomery R=2^32)"; }

}  // extern "C"

// ---- device run of the field self-test (field_selftest.hpp) ---------------------------------------------------
#include "field_selftest.hpp"

namespace {
__global__ void field_selftest_kernel(uint64_t seed, uint32_t iterations, int* first_failure) {
    const uint64_t s = seed + 0x1234567ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    const int rc = pw::field_selftest_checks(s, iterations);
    if (rc) atomicCAS(first_failure, 0, rc);
}
}  // namespace

extern "C" int powdr_field_selftest_gpu(uint64_t seed, uint32_t iterations, int* failing_check) {
    int* d = nullptr;
    PW_HIP_TRY(hipMalloc(&d, sizeof(int)));
    PW_HIP_TRY(hipMemsetAsync(d, 0, sizeof(int), pw::stream()));
    hipLaunchKernelGGL(field_selftest_kernel, dim3(64), dim3(256), 0, pw::stream(), seed, iterations, d);
    int rc = 0;
    hipError_t e = hipMemcpyAsync(&rc, d, sizeof(int), hipMemcpyDeviceToHost, pw::stream());
    if (e == hipSuccess) e = hipStreamSynchronize(pw::stream());
    (void)hipFree(d);
    if (e != hipSuccess) return (int)e;
    if (failing_check) *failing_check = rc;
    return (int)hipGetLastError();
}
