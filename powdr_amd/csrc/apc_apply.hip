// _apc_apply_derived_expr and _apc_apply_bus for gfx950.
//
// Reference semantics:
//   derived columns  /root/reference/openvm/cuda/src/apc_tracegen.cu:72-124
//                    (CPU twin: openvm/src/powdr_extension/trace_generator/cpu/mod.rs:182-202)
//   bus -> histogram /root/reference/openvm/cuda/src/apc_apply_bus.cu:23-169
//                    (CPU twin: .../cpu/periphery.rs:176-237)
//
// One lane per APC row; the bytecode is wave-uniform (see expr_eval.hpp).
// Differences from the reference launch, none of them observable in the result:
//   * interactions whose bus id is none of the three periphery buses are
//     skipped before their multiplicity is evaluated (the reference evaluates
//     it and then ignores it, apc_apply_bus.cu:58-63,111);
//   * a multiplicity m is applied as one atomicAdd(bin, m) instead of m single
//     increments (apc_apply_bus.cu:77,91,106) — same sum mod 2^32;
//   * the interaction list is additionally split over blockIdx.y so that short
//     traces still fill 256 CUs (histogram sums are order independent).
// Defined behaviour where the reference traps or writes out of bounds: a
// histogram index >= the bin count is dropped (index arithmetic is the
// reference's 32-bit wrap-around arithmetic); bitwise operands >= 256 or a
// selector other than 0/1 are ignored (reference: assert(false),
// apc_apply_bus.cu:109).
#include "babybear.hpp"
#include "common.hpp"
#include "expr_eval.hpp"
#include "xbc.hpp"
#include "xbc_compile.hpp"
#include "small_form.hpp"
#include "../../include/powdr_gpu.h"

#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

constexpr int kBlock = 256;

// col_stride = 1: PUSH_APC operands are element offsets col*H (reference encoding); col_stride = H: operands are
// column indices (extension for traces with W*H >= 2^32, which the u32 offsets cannot address).
__global__ __launch_bounds__(kBlock) void apc_apply_derived_expr_kernel(
    uint32_t* d_output, size_t H, int num_calls, const DerivedExprSpec* __restrict__ specs,
    size_t n_cols, const uint32_t* __restrict__ bytecode, size_t col_stride) {
    __shared__ uint32_t stack_lds[pw::kStackCap * kBlock];
    uint32_t* stk = stack_lds + threadIdx.x;
    const size_t total = (size_t)gridDim.x * kBlock;
    for (size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x; r < H; r += total) {
        if (r < (size_t)num_calls) {
            for (size_t i = 0; i < n_cols; ++i) {
                const DerivedExprSpec spec = specs[i];
                // later derived columns may read earlier ones of the same row: plain
                // (non-restrict) accesses by the same thread keep program order.
                uint32_t v = pw::eval_expr<kBlock, true>(bytecode + spec.span.off, spec.span.len,
                                                        d_output, r, stk, col_stride);
                d_output[spec.col_base + r] = v;
            }
        } else {
            for (size_t i = 0; i < n_cols; ++i) d_output[specs[i].col_base + r] = 0u;
        }
    }
}

struct BusParams {
    uint32_t var_bus, tuple_bus, bitwise_bus;
    uint32_t* var_hist;
    uint32_t* tuple_hist;
    uint32_t* bitwise_hist;
    uint32_t var_bins, tuple_sz0, tuple_sz1;
    const uint32_t* slot_range;  // binned mode: per item slot, the expected partitions (in_expected_partitions)
};

// Layout of BitwiseOperationLookup<8>'s count buffer (EXTERNAL to the reference
// checkout, assumption A3 of SURVEY.md): 2^16 range counters followed by 2^16 xor
// counters, both indexed by x * 256 + y.
__host__ __device__ __forceinline__ uint32_t bitwise_index(uint32_t x, uint32_t y, uint32_t selector) {
    return selector * (1u << (2 * POWDR_BITWISE_NUM_BITS)) + (x << POWDR_BITWISE_NUM_BITS) + y;
}

// Binned mode (long traces). The device-wide atomic rate (27 G/s on MI355X, any scope, see
// profiles/r01_microbench_atomics.txt) would bound this stage at ~1 atomic per lookup, and hot bins
// (bytes that are mostly 0, constant range widths) serialise far below that. Instead the
// evaluation kernel writes one packed item (bin | multiplicity << 20) per (periphery interaction,
// row) into a dense buffer — coalesced, no atomics — and a second kernel counts the items of
// one 32 K-bin partition of one table in an LDS histogram (LDS atomics), touching global
// memory with one atomicAdd per non-empty bin per workgroup.
constexpr uint32_t kItemNone = 0xffffffffu;
constexpr uint32_t kItemBinBits = 20;
constexpr uint32_t kItemMaxMult = 1u << (32 - kItemBinBits);
constexpr uint32_t kPartBins = 32768;  // bins per histogram partition: 128 KB of LDS counters

// The partitions of its table a slot's items are EXPECTED in, as (first | last << 16), decided at plan time from the
// constant operands of the interaction (a 12-14-bit range check belongs to partition 0, a bitwise lookup with a
// constant operation to two of four); the histogram pass of a partition skips the slots whose range excludes it. An
// item that falls outside its slot's range (only a dishonest trace produces one) is not packed but counted directly
// with a global atomic, so the result is exact either way.
__device__ __forceinline__ bool in_expected_partitions(uint32_t range, uint32_t bin) {
    const uint32_t part = bin / kPartBins;
    return part >= (range & 0xffffu) && part <= (range >> 16);
}

template <bool BINNED>
__global__ __launch_bounds__(kBlock) void apc_apply_bus_kernel(
    const uint32_t* __restrict__ trace, int num_calls, const uint32_t* __restrict__ bytecode,
    const DevInteraction* __restrict__ interactions, uint32_t n_interactions,
    const ExprSpan* __restrict__ spans, BusParams p, uint32_t per_chunk,
    const int32_t* __restrict__ slot_of, uint32_t* __restrict__ items, size_t item_stride, size_t col_stride,
    size_t row0) {
    __shared__ uint32_t stack_lds[pw::kStackCap * kBlock];
    uint32_t* stk = stack_lds + threadIdx.x;
    const size_t rl = (size_t)blockIdx.x * kBlock + threadIdx.x;  // row within this launch's row window
    const size_t r = row0 + rl;
    const bool live = r < (size_t)num_calls;
    if (!BINNED && !live) return;
    if (BINNED && rl >= item_stride) return;
    const uint32_t i0 = blockIdx.y * per_chunk;
    const uint32_t i1 = min(n_interactions, i0 + per_chunk);

    for (uint32_t i = i0; i < i1; ++i) {
        const DevInteraction intr = interactions[i];
        int kind;
        if (intr.bus_id == p.var_bus) kind = 0;
        else if (intr.bus_id == p.tuple_bus) kind = 1;
        else if (intr.bus_id == p.bitwise_bus) kind = 2;
        else continue;  // execution bridge / memory / pc lookup: no periphery side effect

        const uint32_t range = BINNED ? p.slot_range[slot_of[i]] : 0u;
        uint32_t bin = kItemNone, m = 0u;
        uint32_t* table = kind == 0 ? p.var_hist : kind == 1 ? p.tuple_hist : p.bitwise_hist;
        if (live) {
            const ExprSpan* sp = spans + intr.args_index_off;
            const ExprSpan ms = sp[0];
            m = bb::from_monty(pw::eval_expr<kBlock, true>(bytecode + ms.off, ms.len, trace, r, stk, col_stride));
            if (m != 0u) {
                const ExprSpan s0 = sp[1], s1 = sp[2];
                const uint32_t a0 = bb::from_monty(pw::eval_expr<kBlock, true>(bytecode + s0.off, s0.len, trace, r, stk, col_stride));
                const uint32_t a1 = bb::from_monty(pw::eval_expr<kBlock, true>(bytecode + s1.off, s1.len, trace, r, stk, col_stride));
                if (kind == 0) {
                    // [value, max_bits] -> bin (1 << max_bits) + value - 1   (apc_apply_bus.cu:74)
                    // (shift counts >= 32 give 0, as PTX shl.b32 does for the reference build)
                    const uint32_t idx = (a1 < 32u ? (1u << a1) : 0u) + a0 - 1u;
                    if (idx < p.var_bins) bin = idx;
                } else if (kind == 1) {
                    // [v0, v1] -> bin v0 * sz1 + v1                         (apc_apply_bus.cu:89)
                    const uint32_t idx = a0 * p.tuple_sz1 + a1;
                    if (idx < p.tuple_sz0 * p.tuple_sz1) bin = idx;
                } else {
                    // [x, y, x_xor_y, selector]; arg 2 is never read (apc_apply_bus.cu:94-99)
                    const ExprSpan s3 = sp[4];
                    const uint32_t sel = bb::from_monty(pw::eval_expr<kBlock, true>(bytecode + s3.off, s3.len, trace, r, stk, col_stride));
                    if (sel <= 1u && a0 < 256u && a1 < 256u) bin = bitwise_index(a0, a1, sel);
                }
            }
        }
        if (BINNED) {
            uint32_t item = kItemNone;
            if (bin != kItemNone) {
                if (m < kItemMaxMult - 1u && bin < (1u << kItemBinBits) && in_expected_partitions(range, bin)) item = bin | (m << kItemBinBits);  // m = kItemMaxMult-1 could encode kItemNone: counted directly
                else atomicAdd(table + bin, m);  // does not fit the packed item: rare, direct
            }
            items[(size_t)slot_of[i] * item_stride + rl] = item;
        } else if (bin != kItemNone) {
            atomicAdd(table + bin, m);
        }
    }
}

// The same replay on plan-compiled xbc code (xbc.hpp): only periphery interactions are listed,
// with their table, item slot and the four expression spans resolved on the host.
struct XInteraction {
    uint32_t kind;    // 0 var-range, 1 tuple2, 2 bitwise
    uint32_t slot;    // item slot (binned mode)
    uint32_t off[4];  // instruction offsets of mult, arg0, arg1, selector
    uint32_t len[4];
};

template <bool BINNED>
__global__ __launch_bounds__(kBlock) void apc_apply_bus_xbc_kernel(
    const uint32_t* __restrict__ trace, int num_calls, const uint32_t* __restrict__ code,
    const XInteraction* __restrict__ xint, uint32_t n_xint, BusParams p, uint32_t per_chunk,
    uint32_t* __restrict__ items, size_t item_stride, size_t col_stride, size_t row0) {
    __shared__ uint32_t stack_lds[pw::kStackCap * kBlock];
    uint32_t* stk = stack_lds + threadIdx.x;
    const size_t rl = (size_t)blockIdx.x * kBlock + threadIdx.x;  // row within this launch's row window
    const size_t r = row0 + rl;
    const bool live = r < (size_t)num_calls;
    if (!BINNED && !live) return;
    if (BINNED && rl >= item_stride) return;
    const uint32_t i0 = blockIdx.y * per_chunk;
    const uint32_t i1 = min(n_xint, i0 + per_chunk);
    for (uint32_t i = i0; i < i1; ++i) {
        const XInteraction xi = xint[i];
        const uint32_t range = BINNED ? p.slot_range[xi.slot] : 0u;
        uint32_t bin = kItemNone, m = 0u;
        uint32_t* table = xi.kind == 0 ? p.var_hist : xi.kind == 1 ? p.tuple_hist : p.bitwise_hist;
        if (live) {
            m = bb::from_monty(xbc::eval<kBlock, true>(code + 2 * (size_t)xi.off[0], xi.len[0], trace, r, stk, col_stride));
            if (m != 0u) {
                const uint32_t a0 = bb::from_monty(xbc::eval<kBlock, true>(code + 2 * (size_t)xi.off[1], xi.len[1], trace, r, stk, col_stride));
                const uint32_t a1 = bb::from_monty(xbc::eval<kBlock, true>(code + 2 * (size_t)xi.off[2], xi.len[2], trace, r, stk, col_stride));
                if (xi.kind == 0) {
                    const uint32_t idx = (a1 < 32u ? (1u << a1) : 0u) + a0 - 1u;
                    if (idx < p.var_bins) bin = idx;
                } else if (xi.kind == 1) {
                    const uint32_t idx = a0 * p.tuple_sz1 + a1;
                    if (idx < p.tuple_sz0 * p.tuple_sz1) bin = idx;
                } else {
                    const uint32_t sel = bb::from_monty(xbc::eval<kBlock, true>(code + 2 * (size_t)xi.off[3], xi.len[3], trace, r, stk, col_stride));
                    if (sel <= 1u && a0 < 256u && a1 < 256u) bin = bitwise_index(a0, a1, sel);
                }
            }
        }
        if (BINNED) {
            uint32_t item = kItemNone;
            if (bin != kItemNone) {
                if (m < kItemMaxMult - 1u && bin < (1u << kItemBinBits) && in_expected_partitions(range, bin)) item = bin | (m << kItemBinBits);  // m = kItemMaxMult-1 could encode kItemNone: counted directly
                else atomicAdd(table + bin, m);
            }
            items[(size_t)xi.slot * item_stride + rl] = item;
        } else if (bin != kItemNone) {
            atomicAdd(table + bin, m);
        }
    }
}

// Interactions whose multiplicity and arguments are all small forms (small_form.hpp): no interpreter. All cells an
// interaction reads are requested before the first one is used, so an interaction costs one memory round trip instead
// of one per bytecode instruction.
struct FastInteraction {
    uint32_t kind;  // 0 var-range, 1 tuple2, 2 bitwise
    uint32_t slot;  // item slot (binned mode)
    uint32_t pad[2];
    pw::SmallForm f[4];  // mult, arg0, arg1, selector (bitwise only)
};

template <bool BINNED>
__global__ __launch_bounds__(kBlock) void apc_apply_bus_fast_kernel(
    const uint32_t* __restrict__ trace, int num_calls, const FastInteraction* __restrict__ fint, uint32_t n_fint, BusParams p,
    uint32_t per_chunk, uint32_t* __restrict__ items, size_t item_stride, size_t col_stride, size_t row0) {
    const size_t rl = (size_t)blockIdx.x * kBlock + threadIdx.x;  // row within this launch's row window
    const size_t r = row0 + rl;
    const bool live = r < (size_t)num_calls;
    if (!BINNED && !live) return;
    if (BINNED && rl >= item_stride) return;
    const size_t rr = live ? r : 0;  // dead rows read row 0 and discard it
    const uint32_t i0 = blockIdx.y * per_chunk;
    const uint32_t i1 = min(n_fint, i0 + per_chunk);
    for (uint32_t i = i0; i < i1; ++i) {
        const FastInteraction fi = fint[i];
        const uint32_t range = BINNED ? p.slot_range[fi.slot] : 0u;
        uint32_t ta[4], tb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ta[k] = (fi.f[k].flags & pw::SmallForm::USES_A) ? trace[(size_t)fi.f[k].a * col_stride + rr] : 0u;
            tb[k] = (fi.f[k].flags & pw::SmallForm::USES_B) ? trace[(size_t)fi.f[k].b * col_stride + rr] : 0u;
        }
        uint32_t bin = kItemNone;
        const uint32_t m = live ? bb::from_monty(fi.f[0].eval(ta[0], tb[0])) : 0u;
        uint32_t* table = fi.kind == 0 ? p.var_hist : fi.kind == 1 ? p.tuple_hist : p.bitwise_hist;
        if (m != 0u) {
            const uint32_t a0 = bb::from_monty(fi.f[1].eval(ta[1], tb[1]));
            const uint32_t a1 = bb::from_monty(fi.f[2].eval(ta[2], tb[2]));
            if (fi.kind == 0) {
                const uint32_t idx = (a1 < 32u ? (1u << a1) : 0u) + a0 - 1u;
                if (idx < p.var_bins) bin = idx;
            } else if (fi.kind == 1) {
                const uint32_t idx = a0 * p.tuple_sz1 + a1;
                if (idx < p.tuple_sz0 * p.tuple_sz1) bin = idx;
            } else {
                const uint32_t sel = bb::from_monty(fi.f[3].eval(ta[3], tb[3]));
                if (sel <= 1u && a0 < 256u && a1 < 256u) bin = bitwise_index(a0, a1, sel);
            }
        }
        if (BINNED) {
            uint32_t item = kItemNone;
            if (bin != kItemNone) {
                if (m < kItemMaxMult - 1u && bin < (1u << kItemBinBits) && in_expected_partitions(range, bin)) item = bin | (m << kItemBinBits);  // m = kItemMaxMult-1 could encode kItemNone: counted directly
                else atomicAdd(table + bin, m);
            }
            items[(size_t)fi.slot * item_stride + rl] = item;
        } else if (bin != kItemNone) {
            atomicAdd(table + bin, m);
        }
    }
}

constexpr int kHistBlock = 1024;

__global__ __launch_bounds__(kHistBlock) void bus_histogram_kernel(
    const uint32_t* __restrict__ items, size_t item_stride, const uint32_t* __restrict__ slots, uint32_t n_slots,
    uint32_t* __restrict__ hist, uint32_t bins, uint32_t rows_per_chunk, const uint32_t* __restrict__ slot_range) {
    __shared__ uint32_t lh[kPartBins];
    const uint32_t bin0 = blockIdx.x * kPartBins;
    for (uint32_t b = threadIdx.x; b < kPartBins; b += kHistBlock) lh[b] = 0u;
    __syncthreads();
    const size_t r0 = (size_t)blockIdx.y * rows_per_chunk;
    const size_t r1 = r0 + rows_per_chunk < item_stride ? r0 + rows_per_chunk : item_stride;
    for (uint32_t s = 0; s < n_slots; ++s) {
        const uint32_t range = slot_range[slots[s]];
        if (blockIdx.x < (range & 0xffffu) || blockIdx.x > (range >> 16)) continue;  // no item of this slot lands here
        const uint4* it = reinterpret_cast<const uint4*>(items + (size_t)slots[s] * item_stride);
        for (size_t q = (r0 >> 2) + threadIdx.x; q < (r1 >> 2); q += kHistBlock) {
            const uint4 v = it[q];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            // lanes of this iteration (the last one of a chunk may be partial) and the first of them
            const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
            const bool leader = (unsigned)__lane_id() == (unsigned)__builtin_ctzll(active);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t b = (w[k] & ((1u << kItemBinBits) - 1u)) - bin0;
                const bool hit = w[k] != kItemNone && b < kPartBins;
                // a slot fed from a cell that does not change from row to row sends ONE item 64 times: one LDS atomic for the wave
                // instead of 64 that conflict (same-address LDS atomics of one instruction are serialised)
                if (__builtin_amdgcn_ballot_w64(w[k] != (uint32_t)__builtin_amdgcn_readfirstlane((int)w[k])) == 0ull) {
                    if (hit && leader) atomicAdd(&lh[b], (w[k] >> kItemBinBits) * (uint32_t)__builtin_popcountll(active));
                } else if (hit) {
                    atomicAdd(&lh[b], w[k] >> kItemBinBits);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kPartBins; b += kHistBlock) {
        const uint32_t c = lh[b];
        if (c && bin0 + b < bins) atomicAdd(hist + bin0 + b, c);
    }
}

// host-side classification of the interaction list, cached by content
struct BusPlan {
    int32_t* d_slot_of = nullptr;        // per interaction: item slot or -1
    uint32_t* d_slots[3] = {nullptr, nullptr, nullptr};  // slot ids per table (var, tuple, bitwise)
    uint32_t n_slots[3] = {0, 0, 0};
    uint32_t total_slots = 0;
    // xbc form of the periphery interactions (absent if some expression did not compile)
    bool has_xbc = false;
    XInteraction* d_xint = nullptr;
    uint32_t* d_code = nullptr;
    // the same interactions split into those that are small forms throughout (fast kernel) and the rest (interpreter)
    uint32_t* d_slot_range = nullptr;  // per item slot: expected partitions (first | last << 16)
    FastInteraction* d_fint = nullptr;
    XInteraction* d_xint_slow = nullptr;
    uint32_t n_fast = 0, n_slow = 0;
    // what the plan was compiled from (a cache hit is confirmed by content, not by the 64-bit hash alone)
    std::vector<DevInteraction> key_inter;
    std::vector<ExprSpan> key_spans;
    std::vector<uint32_t> key_bc;
    uint32_t key_ids[3] = {0, 0, 0};
    int device = 0;
    uint64_t last_use = 0;
    BusPlan() = default;
    BusPlan(const BusPlan&) = delete;
    BusPlan& operator=(const BusPlan&) = delete;
    ~BusPlan() {  // hipFree waits for kernels that may still read the tables
        for (void* q : {(void*)d_slot_of, (void*)d_slots[0], (void*)d_slots[1], (void*)d_slots[2], (void*)d_xint, (void*)d_code,
                        (void*)d_slot_range, (void*)d_fint, (void*)d_xint_slow})
            if (q) (void)hipFree(q);
    }
};
constexpr size_t kMaxBusPlans = 64;
std::mutex g_bus_mu;
std::unordered_map<uint64_t, std::shared_ptr<BusPlan>> g_bus_plans;
uint64_t g_bus_clock = 0;
// item buffer of the binned path: one per host thread (= per launch stream)
struct ItemBuffer {
    uint32_t* p = nullptr;
    size_t words = 0;
    int device = -1;  // the device the buffer lives on (a host thread may move to another GPU)
    ~ItemBuffer() { if (p) (void)hipFree(p); }  // a host thread that exits gives its buffer back
};
thread_local ItemBuffer g_item_buf;
#define g_items g_item_buf.p
#define g_items_words g_item_buf.words
#define g_items_device g_item_buf.device

// 8 bytes per step (the bytecode of an un-optimised APC is megabytes)
uint64_t hash_words(const void* p, size_t n, uint64_t h) {
    const uint64_t* w = (const uint64_t*)p;
    size_t k = n / 8;
    for (size_t i = 0; i < k; ++i) { h ^= w[i]; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    const unsigned char* c = (const unsigned char*)p + k * 8;
    for (size_t i = 0; i < n % 8; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}
uint64_t fnv1a64(const void* p, size_t n, uint64_t h) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}

}  // namespace

namespace {
int apply_derived_impl(PowdrFp* d_output, size_t H, int num_apc_calls, const DerivedExprSpec* d_specs, size_t n_cols,
                       const uint32_t* d_bytecode, size_t col_stride) {
    if (n_cols == 0) return 0;  // apc_tracegen.cu:114
    (void)hipGetLastError();    // do not report a stale error of an unrelated earlier call
    if (H == 0) return (int)hipGetLastError();
    unsigned g = pw::div_up(H, kBlock);
    if (g > 65535u * 16u) g = 65535u * 16u;
    pw::ScopedKernelTimer t("apc_apply_derived_expr_kernel");
    hipLaunchKernelGGL(apc_apply_derived_expr_kernel, dim3(g), dim3(kBlock), 0, pw::stream(),
                       d_output, H, num_apc_calls, d_specs, n_cols, d_bytecode, col_stride);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int _apc_apply_derived_expr(PowdrFp* d_output, size_t H, int num_apc_calls,
                                       const DerivedExprSpec* d_specs, size_t n_cols,
                                       const uint32_t* d_bytecode) {
    return apply_derived_impl(d_output, H, num_apc_calls, d_specs, n_cols, d_bytecode, 1);
}

extern "C" int powdr_apc_apply_derived_expr_cols(PowdrFp* d_output, size_t H, int num_apc_calls,
                                                 const DerivedExprSpec* d_specs, size_t n_cols,
                                                 const uint32_t* d_bytecode) {
    return apply_derived_impl(d_output, H, num_apc_calls, d_specs, n_cols, d_bytecode, H);
}

namespace {
int apply_bus_impl(const PowdrFp* d_output, int num_apc_calls,
                   const uint32_t* d_bytecode, size_t bytecode_len,
                   const DevInteraction* d_interactions, size_t n_interactions,
                   const ExprSpan* d_arg_spans, size_t n_arg_spans,
                   uint32_t var_range_bus_id, uint32_t* d_var_hist, size_t var_num_bins,
                   uint32_t tuple2_bus_id, uint32_t* d_tuple2_hist, uint32_t tuple2_sz0,
                   uint32_t tuple2_sz1, uint32_t bitwise_bus_id,
                   uint32_t* d_bitwise_hist, size_t col_stride, const uint32_t* h_bytecode = nullptr,
                   const DevInteraction* h_interactions = nullptr, const ExprSpan* h_arg_spans = nullptr) {
    if (num_apc_calls <= 0) return 0;  // apc_apply_bus.cu:146
    (void)hipGetLastError();
    if (n_interactions == 0) return (int)hipGetLastError();
    const unsigned row_blocks = pw::div_up((size_t)num_apc_calls, kBlock);
    // Enough workgroups to cover 256 CUs x 8 blocks even for short traces.
    const unsigned want_blocks = 256u * 8u;
    BusParams p;
    p.var_bus = var_range_bus_id; p.tuple_bus = tuple2_bus_id; p.bitwise_bus = bitwise_bus_id;
    p.var_hist = d_var_hist; p.tuple_hist = d_tuple2_hist; p.bitwise_hist = d_bitwise_hist;
    p.var_bins = (uint32_t)var_num_bins; p.tuple_sz0 = tuple2_sz0; p.tuple_sz1 = tuple2_sz1;
    p.slot_range = nullptr;
    // ---- plan: host copy of the (small) tables, cached by content --------------------------------
    const char* env = getenv("POWDR_BUS_BINNED");
    const bool want_binned = env ? atoi(env) != 0 : num_apc_calls >= 16384;
    const char* env_x = getenv("POWDR_BUS_XBC");
    const bool want_xbc = env_x ? atoi(env_x) != 0 : true;
    const uint32_t table_bins[3] = {(uint32_t)var_num_bins, tuple2_sz0 * tuple2_sz1, 2u << (2 * POWDR_BITWISE_NUM_BITS)};
    std::vector<DevInteraction> h(n_interactions);
    std::vector<ExprSpan> hs(n_arg_spans);
    std::vector<uint32_t> hb(bytecode_len);
    if (h_bytecode && h_interactions && (h_arg_spans || !n_arg_spans)) {
        // the caller still holds the tables on the host (powdr_apc_apply_bus_host_tables): no copy back, no synchronisation
        if (n_interactions) memcpy(h.data(), h_interactions, n_interactions * sizeof(DevInteraction));
        if (n_arg_spans) memcpy(hs.data(), h_arg_spans, n_arg_spans * sizeof(ExprSpan));
        if (bytecode_len) memcpy(hb.data(), h_bytecode, bytecode_len * 4);
    } else {
        PW_HIP_TRY(hipMemcpyAsync(h.data(), d_interactions, n_interactions * sizeof(DevInteraction), hipMemcpyDeviceToHost, pw::stream()));
        if (n_arg_spans) PW_HIP_TRY(hipMemcpyAsync(hs.data(), d_arg_spans, n_arg_spans * sizeof(ExprSpan), hipMemcpyDeviceToHost, pw::stream()));
        if (bytecode_len) PW_HIP_TRY(hipMemcpyAsync(hb.data(), d_bytecode, bytecode_len * 4, hipMemcpyDeviceToHost, pw::stream()));
        PW_HIP_TRY(hipStreamSynchronize(pw::stream()));
    }
    uint64_t key = fnv1a64(h.data(), h.size() * sizeof(DevInteraction), 1469598103934665603ull);
    key = hash_words(hs.data(), hs.size() * sizeof(ExprSpan), key);
    key = hash_words(hb.data(), hb.size() * 4, key);
    const uint32_t ids[3] = {var_range_bus_id, tuple2_bus_id, bitwise_bus_id};
    key = fnv1a64(ids, sizeof ids, key);
    int device = 0;
    PW_HIP_TRY(hipGetDevice(&device));
    key = fnv1a64(&device, sizeof device, key);
    std::shared_ptr<BusPlan> plan;
    {
        std::lock_guard<std::mutex> lk(g_bus_mu);
        auto it = g_bus_plans.find(key);
        if (it != g_bus_plans.end()) {
            const BusPlan& c = *it->second;
            const bool same = c.device == device && memcmp(c.key_ids, ids, sizeof ids) == 0 && c.key_inter.size() == h.size() &&
                              c.key_spans.size() == hs.size() && c.key_bc == hb &&
                              (h.empty() || memcmp(c.key_inter.data(), h.data(), h.size() * sizeof(DevInteraction)) == 0) &&
                              (hs.empty() || memcmp(c.key_spans.data(), hs.data(), hs.size() * sizeof(ExprSpan)) == 0);
            if (!same) { g_bus_plans.erase(it); it = g_bus_plans.end(); }  // hash collision: the newer tables take the slot
        }
        if (it == g_bus_plans.end()) {
            if (g_bus_plans.size() >= kMaxBusPlans) {
                auto victim = g_bus_plans.begin();
                for (auto j = g_bus_plans.begin(); j != g_bus_plans.end(); ++j) if (j->second->last_use < victim->second->last_use) victim = j;
                g_bus_plans.erase(victim);
            }
            auto bpp = std::make_shared<BusPlan>();
            BusPlan& bp = *bpp;
            std::vector<int32_t> slot_of(n_interactions, -1);
            std::vector<uint32_t> per_table[3];
            std::vector<XInteraction> xints, xints_slow;
            std::vector<uint32_t> slot_range;
            std::vector<FastInteraction> fints;
            std::vector<uint32_t> code;
            xbc::Compiler cc;
            bool ok = true;
            for (size_t i = 0; i < n_interactions; ++i) {
                int kind = h[i].bus_id == ids[0] ? 0 : h[i].bus_id == ids[1] ? 1 : h[i].bus_id == ids[2] ? 2 : -1;
                if (kind < 0) continue;
                slot_of[i] = (int32_t)bp.total_slots;
                XInteraction xi{};
                xi.kind = (uint32_t)kind;
                xi.slot = bp.total_slots;
                per_table[kind].push_back(bp.total_slots++);
                const uint32_t which[4] = {0, 1, 2, 4};  // mult, arg0, arg1, (bitwise) selector = arg 3
                FastInteraction fi{};
                fi.kind = (uint32_t)kind;
                fi.slot = xi.slot;
                fi.f[3].flags = pw::SmallForm::IS_CONST;
                bool fast = true;
                for (int k = 0; k < (kind == 2 ? 4 : 3) && ok; ++k) {
                    const size_t si = (size_t)h[i].args_index_off + which[k];
                    if (si >= hs.size() || (size_t)hs[si].off + hs[si].len > hb.size()) { ok = false; break; }
                    xi.off[k] = (uint32_t)(code.size() / 2);
                    if (!cc.compile(hb.data() + hs[si].off, hs[si].len, code)) { ok = false; break; }
                    xi.len[k] = (uint32_t)(code.size() / 2) - xi.off[k];
                    fast = fast && pw::analyze_small_form(hb.data() + hs[si].off, hs[si].len, fi.f[k]);
                }
                xints.push_back(xi);
                if (fast) fints.push_back(fi);
                else xints_slow.push_back(xi);
                // expected partitions of this slot's items
                uint32_t first = 0, last = (table_bins[kind] - 1) / kPartBins;
                if (ok) {
                    pw::SmallForm c;
                    const auto constant_arg = [&](uint32_t arg, uint32_t& value) {
                        const size_t si = (size_t)h[i].args_index_off + arg;
                        if (!pw::analyze_small_form(hb.data() + hs[si].off, hs[si].len, c) || !(c.flags & pw::SmallForm::IS_CONST)) return false;
                        value = bb::from_monty(c.k0);
                        return true;
                    };
                    uint32_t v;
                    if (kind == 0 && constant_arg(2, v) && v < 31) {  // honest values of a `v`-bit check: [2^v - 1, 2^(v+1) - 2]
                        first = ((1u << v) - 1u) / kPartBins;
                        last = std::min(last, ((2u << v) - 2u) / kPartBins);
                        if (first > last) first = last;
                    } else if (kind == 2 && constant_arg(4, v) && v <= 1) {
                        first = bitwise_index(0, 0, v) / kPartBins;
                        last = bitwise_index(255, 255, v) / kPartBins;
                    }
                }
                slot_range.push_back(first | (last << 16));
            }
            PW_HIP_TRY(hipMalloc(&bp.d_slot_range, (slot_range.size() + 1) * 4));
            if (!slot_range.empty()) PW_HIP_TRY(hipMemcpy(bp.d_slot_range, slot_range.data(), slot_range.size() * 4, hipMemcpyHostToDevice));
            PW_HIP_TRY(hipMalloc(&bp.d_slot_of, (n_interactions + 1) * 4));
            PW_HIP_TRY(hipMemcpy(bp.d_slot_of, slot_of.data(), n_interactions * 4, hipMemcpyHostToDevice));
            for (int t = 0; t < 3; ++t) {
                bp.n_slots[t] = (uint32_t)per_table[t].size();
                PW_HIP_TRY(hipMalloc(&bp.d_slots[t], (per_table[t].size() + 1) * 4));
                if (!per_table[t].empty())
                    PW_HIP_TRY(hipMemcpy(bp.d_slots[t], per_table[t].data(), per_table[t].size() * 4, hipMemcpyHostToDevice));
            }
            if (ok && !xints.empty()) {
                PW_HIP_TRY(hipMalloc(&bp.d_xint, xints.size() * sizeof(XInteraction)));
                PW_HIP_TRY(hipMemcpy(bp.d_xint, xints.data(), xints.size() * sizeof(XInteraction), hipMemcpyHostToDevice));
                PW_HIP_TRY(hipMalloc(&bp.d_code, (code.size() + 2) * 4));
                if (!code.empty()) PW_HIP_TRY(hipMemcpy(bp.d_code, code.data(), code.size() * 4, hipMemcpyHostToDevice));
                bp.has_xbc = true;
                bp.n_fast = (uint32_t)fints.size();
                bp.n_slow = (uint32_t)xints_slow.size();
                PW_HIP_TRY(hipMalloc(&bp.d_fint, (fints.size() + 1) * sizeof(FastInteraction)));
                if (!fints.empty()) PW_HIP_TRY(hipMemcpy(bp.d_fint, fints.data(), fints.size() * sizeof(FastInteraction), hipMemcpyHostToDevice));
                PW_HIP_TRY(hipMalloc(&bp.d_xint_slow, (xints_slow.size() + 1) * sizeof(XInteraction)));
                if (!xints_slow.empty())
                    PW_HIP_TRY(hipMemcpy(bp.d_xint_slow, xints_slow.data(), xints_slow.size() * sizeof(XInteraction), hipMemcpyHostToDevice));
            }
            bp.key_inter = h; bp.key_spans = hs; bp.key_bc = hb;
            memcpy(bp.key_ids, ids, sizeof ids);
            bp.device = device;
            it = g_bus_plans.emplace(key, std::move(bpp)).first;
        }
        plan = it->second;
        plan->last_use = ++g_bus_clock;
    }
    if (plan->total_slots == 0) return (int)hipGetLastError();
    const bool use_xbc = want_xbc && plan->has_xbc;
    const char* env_f = getenv("POWDR_BUS_FAST");
    const bool use_fast = use_xbc && (env_f ? atoi(env_f) != 0 : true);
    // chunking of an interaction list over blockIdx.y so that short traces still fill the chip
    auto chunking = [&](uint32_t n_list, unsigned& chunks, uint32_t& per_chunk) {
        chunks = 1;
        if (n_list == 0) { per_chunk = 1; return; }
        if (row_blocks < want_blocks) {
            chunks = (want_blocks + row_blocks - 1) / row_blocks;
            const unsigned max_chunks = (n_list + 15) / 16;
            if (chunks > max_chunks) chunks = max_chunks;
            if (chunks > 65535u) chunks = 65535u;
            if (chunks == 0) chunks = 1;
        }
        per_chunk = (n_list + chunks - 1) / chunks;
        chunks = (n_list + per_chunk - 1) / per_chunk;
    };
    unsigned xchunks, fchunks = 1;
    uint32_t x_per_chunk, f_per_chunk = 1;
    // the interpreter walks: all periphery interactions (xbc), only the non-small-form ones (fast mode), or the
    // reference's full list (post-fix mode)
    const XInteraction* x_list = use_fast ? plan->d_xint_slow : plan->d_xint;
    const uint32_t n_x = use_fast ? plan->n_slow : plan->total_slots;
    chunking(use_xbc ? n_x : (uint32_t)n_interactions, xchunks, x_per_chunk);
    if (use_fast) chunking(plan->n_fast, fchunks, f_per_chunk);
    uint64_t* stats = pw::call_stats();
    stats[pw::kStatBusFastInteractions] += use_fast ? plan->n_fast : 0;
    stats[pw::kStatBusInterpretedInteractions] += use_xbc ? n_x : plan->total_slots;
    stats[pw::kStatBusXbcCalls] += use_xbc ? 1 : 0;

    // ---- long traces: binned path ---------------------------------------------------------------
    if (want_binned && table_bins[0] <= (1u << kItemBinBits) && table_bins[1] <= (1u << kItemBinBits)) {
        // rows are processed in windows of at most 2^20 so that the item buffer stays bounded
        // (slots x 4 MiB): a 2^22-row trace with 1 600 periphery interactions would otherwise need 27 GB
        const size_t all_rows = ((size_t)num_apc_calls + 3) & ~(size_t)3;
        const size_t window = all_rows < ((size_t)1 << 20) ? all_rows : ((size_t)1 << 20);
        const size_t need = (size_t)plan->total_slots * window;
        bool have = true;
        if (need > g_items_words || g_items_device != device) {
            g_items_device = device;
            if (g_items) (void)hipFree(g_items);
            g_items = nullptr; g_items_words = 0;
            if (hipMalloc(&g_items, need * 4) == hipSuccess) g_items_words = need;
            else { (void)hipGetLastError(); have = false; }
        }
        if (have) {
            uint32_t* tables[3] = {d_var_hist, d_tuple2_hist, d_bitwise_hist};
            p.slot_range = plan->d_slot_range;
            for (size_t row0 = 0; row0 < all_rows; row0 += window) {
                const size_t stride = all_rows - row0 < window ? all_rows - row0 : window;
                stats[pw::kStatBusBinnedWindows] += 1;
                {
                    pw::ScopedKernelTimer t("apc_apply_bus_kernel");
                    if (use_fast && plan->n_fast)
                        hipLaunchKernelGGL(apc_apply_bus_fast_kernel<true>, dim3(pw::div_up(stride, kBlock), fchunks), dim3(kBlock), 0,
                                           pw::stream(), d_output, num_apc_calls, plan->d_fint, plan->n_fast, p, f_per_chunk, g_items,
                                           stride, col_stride, row0);
                    if (use_xbc) {
                        if (n_x)
                            hipLaunchKernelGGL(apc_apply_bus_xbc_kernel<true>, dim3(pw::div_up(stride, kBlock), xchunks), dim3(kBlock), 0,
                                               pw::stream(), d_output, num_apc_calls, plan->d_code, x_list, n_x, p, x_per_chunk, g_items,
                                               stride, col_stride, row0);
                    } else
                        hipLaunchKernelGGL(apc_apply_bus_kernel<true>, dim3(pw::div_up(stride, kBlock), xchunks), dim3(kBlock), 0,
                                           pw::stream(), d_output, num_apc_calls, d_bytecode, d_interactions, (uint32_t)n_interactions,
                                           d_arg_spans, p, x_per_chunk, plan->d_slot_of, g_items, stride, col_stride, row0);
                }
                // one 128-KB-LDS workgroup fits per CU, and the three launches run one after the other, so each
                // launch gets ~2 x 256 workgroups of its own: partitions x row chunks
                for (int t = 0; t < 3; ++t) {
                    if (!plan->n_slots[t]) continue;
                    const unsigned parts = pw::div_up(table_bins[t], kPartBins);
                    unsigned n_chunks = (512 + parts - 1) / parts;
                    const unsigned max_chunks = pw::div_up(stride, 4096);
                    if (n_chunks > max_chunks) n_chunks = max_chunks;
                    if (n_chunks == 0) n_chunks = 1;
                    const uint32_t rows_per_chunk = (uint32_t)(((stride + n_chunks - 1) / n_chunks + 3) & ~(size_t)3);
                    n_chunks = pw::div_up(stride, rows_per_chunk);
                    pw::ScopedKernelTimer tt("bus_histogram_kernel");
                    hipLaunchKernelGGL(bus_histogram_kernel, dim3(parts, n_chunks), dim3(kHistBlock), 0,
                                       pw::stream(), g_items, stride, plan->d_slots[t], plan->n_slots[t], tables[t], table_bins[t], rows_per_chunk,
                                       plan->d_slot_range);
                }
            }
            return (int)hipGetLastError();
        }
    }
    stats[pw::kStatBusDirectCalls] += 1;
    pw::ScopedKernelTimer t("apc_apply_bus_kernel");
    if (use_fast && plan->n_fast)
        hipLaunchKernelGGL(apc_apply_bus_fast_kernel<false>, dim3(row_blocks, fchunks), dim3(kBlock), 0, pw::stream(), d_output,
                           num_apc_calls, plan->d_fint, plan->n_fast, p, f_per_chunk, nullptr, 0, col_stride, 0);
    if (use_xbc) {
        if (n_x)
            hipLaunchKernelGGL(apc_apply_bus_xbc_kernel<false>, dim3(row_blocks, xchunks), dim3(kBlock), 0, pw::stream(),
                               d_output, num_apc_calls, plan->d_code, x_list, n_x, p, x_per_chunk, nullptr, 0, col_stride, 0);
    } else
        hipLaunchKernelGGL(apc_apply_bus_kernel<false>, dim3(row_blocks, xchunks), dim3(kBlock), 0, pw::stream(),
                           d_output, num_apc_calls, d_bytecode, d_interactions,
                           (uint32_t)n_interactions, d_arg_spans, p, x_per_chunk, nullptr, nullptr, 0, col_stride, 0);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int _apc_apply_bus(const PowdrFp* d_output, int num_apc_calls,
                              const uint32_t* d_bytecode, size_t bytecode_len,
                              const DevInteraction* d_interactions, size_t n_interactions,
                              const ExprSpan* d_arg_spans, size_t n_arg_spans,
                              uint32_t var_range_bus_id, uint32_t* d_var_hist, size_t var_num_bins,
                              uint32_t tuple2_bus_id, uint32_t* d_tuple2_hist, uint32_t tuple2_sz0,
                              uint32_t tuple2_sz1, uint32_t bitwise_bus_id,
                              uint32_t* d_bitwise_hist) {
    return apply_bus_impl(d_output, num_apc_calls, d_bytecode, bytecode_len, d_interactions, n_interactions, d_arg_spans,
                          n_arg_spans, var_range_bus_id, d_var_hist, var_num_bins, tuple2_bus_id, d_tuple2_hist, tuple2_sz0,
                          tuple2_sz1, bitwise_bus_id, d_bitwise_hist, 1);
}

extern "C" int powdr_apc_apply_bus_cols(const PowdrFp* d_output, size_t output_height, int num_apc_calls,
                                        const uint32_t* d_bytecode, size_t bytecode_len,
                                        const DevInteraction* d_interactions, size_t n_interactions,
                                        const ExprSpan* d_arg_spans, size_t n_arg_spans,
                                        uint32_t var_range_bus_id, uint32_t* d_var_hist, size_t var_num_bins,
                                        uint32_t tuple2_bus_id, uint32_t* d_tuple2_hist, uint32_t tuple2_sz0,
                                        uint32_t tuple2_sz1, uint32_t bitwise_bus_id,
                                        uint32_t* d_bitwise_hist) {
    return apply_bus_impl(d_output, num_apc_calls, d_bytecode, bytecode_len, d_interactions, n_interactions, d_arg_spans,
                          n_arg_spans, var_range_bus_id, d_var_hist, var_num_bins, tuple2_bus_id, d_tuple2_hist, tuple2_sz0,
                          tuple2_sz1, bitwise_bus_id, d_bitwise_hist, output_height);
}

// Extension: _apc_apply_bus for a caller that still holds the three tables on the host (powdr_apc_generate_witness_gpu
// compiles them itself): h_* are host copies of the device tables. Skips the device-to-host copy of the bytecode
// (megabytes for an un-optimised APC), its hash over the copy and the stream synchronisation of the reference entry.
// output_height = 0: PUSH_APC operands are element offsets (reference encoding); otherwise column indices.
extern "C" int powdr_apc_apply_bus_host_tables(const PowdrFp* d_output, size_t output_height, int num_apc_calls,
                                               const uint32_t* d_bytecode, const uint32_t* h_bytecode, size_t bytecode_len,
                                               const DevInteraction* d_interactions, const DevInteraction* h_interactions,
                                               size_t n_interactions, const ExprSpan* d_arg_spans, const ExprSpan* h_arg_spans,
                                               size_t n_arg_spans, uint32_t var_range_bus_id, uint32_t* d_var_hist,
                                               size_t var_num_bins, uint32_t tuple2_bus_id, uint32_t* d_tuple2_hist,
                                               uint32_t tuple2_sz0, uint32_t tuple2_sz1, uint32_t bitwise_bus_id,
                                               uint32_t* d_bitwise_hist) {
    if (!h_bytecode || !h_interactions || (n_arg_spans && !h_arg_spans)) return (int)hipErrorInvalidValue;
    return apply_bus_impl(d_output, num_apc_calls, d_bytecode, bytecode_len, d_interactions, n_interactions, d_arg_spans,
                          n_arg_spans, var_range_bus_id, d_var_hist, var_num_bins, tuple2_bus_id, d_tuple2_hist, tuple2_sz0,
                          tuple2_sz1, bitwise_bus_id, d_bitwise_hist, output_height ? output_height : 1, h_bytecode,
                          h_interactions, h_arg_spans);
}
