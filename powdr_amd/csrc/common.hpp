// Shared host-side plumbing of libpowdr_gpu: the launch stream, optional
// per-kernel HIP-event timing, error helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace pw {

hipStream_t stream();
void set_stream(hipStream_t s);

// Timing: when enabled every PW_LAUNCH is bracketed by two events recorded on
// the launch stream; report() synchronises and aggregates by kernel name.
bool timing_enabled();
void timing_begin(const char* name, hipEvent_t* e0);
void timing_end(const char* name, hipEvent_t e0);

struct ScopedKernelTimer {
    const char* name;
    hipEvent_t e0 = nullptr;
    bool on;
    explicit ScopedKernelTimer(const char* n) : name(n), on(timing_enabled()) {
        if (on) timing_begin(name, &e0);
    }
    ~ScopedKernelTimer() {
        if (on) timing_end(name, e0);
    }
};

inline int hip_status(hipError_t e) { return (int)e; }

#define PW_HIP_TRY(expr)                          \
    do {                                          \
        hipError_t _e = (expr);                   \
        if (_e != hipSuccess) return (int)_e;     \
    } while (0)

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Which code paths the calling thread's library calls took (powdr_gpu_call_stats): parity tests assert that a workload
// really exercised the job forms / kernels it is meant to cover instead of inferring it from sizes.
enum CallStat : int {
    kStatGatherSparseJobs = 0, kStatGatherWholeJobs, kStatGatherChunkJobs, kStatGatherCalls,
    kStatBusFastInteractions, kStatBusInterpretedInteractions, kStatBusBinnedWindows, kStatBusDirectCalls,
    kStatBusXbcCalls, kStatJitKernelLaunches, kStatInterpreterKernelLaunches, kStatCount = 16
};
uint64_t* call_stats();  // this thread's kStatCount counters

}  // namespace pw
