// Per-opcode VALU issue rates on gfx950 (MI355X): the instructions BabyBear arithmetic is made of, each pinned by inline
// assembly, 8 independent dependency chains per lane, 8 waves per SIMD (32 waves per CU), no memory traffic in the loop.
// Reported per opcode: cycles per wave64 instruction per SIMD, from the shader clock (s_memtime) and from wall time at
// the nominal 2.4 GHz. VERDICT r1 item 5: the "integer-VALU issue ceiling" must rest on these numbers.
//
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_opcodes.hip -o tools/microbench_opcodes
// output: profiles/r02_microbench_opcodes.txt (+ one JSON line for bench.py's VALU model)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int CHAINS = 8;
constexpr int REPS = 4;      // instructions per chain per loop iteration
constexpr int ITER = 4096;   // loop iterations
constexpr uint32_t P = 0x78000001u;

enum Op {
    OP_ADD_U32, OP_SUB_U32, OP_ADD3_U32, OP_LSHL_ADD_U32, OP_XOR, OP_AND_OR, OP_LSHRREV, OP_MIN_U32, OP_CNDMASK, OP_MOV,
    OP_ADD_CO, OP_ADDC_CO, OP_BFE, OP_ALIGNBIT, OP_PERM, OP_MUL_LO, OP_MUL_HI, OP_MUL_U24, OP_MAD_U24, OP_MAD_U64_U32,
    OP_LSHL_ADD_U64, OP_FMA_F32, OP_PK_FMA_F32, OP_FMA_F64, OP_DOT4_I32_I8, OP_CNDMASK_SGPR, OP_CMP_CNDMASK, OP_MAX_U32, OP_MONTY5, OP_MODADD3, OP_MAD_I64_I32, OP_MUL_HI_I32, OP_SMONTY3, OP_COUNT
};

struct OpInfo { const char* name; int instrs; };
static const OpInfo kOps[OP_COUNT] = {
    {"v_add_u32", 1}, {"v_sub_u32", 1}, {"v_add3_u32", 1}, {"v_lshl_add_u32", 1}, {"v_xor_b32", 1}, {"v_and_or_b32", 1},
    {"v_lshrrev_b32", 1}, {"v_min_u32", 1}, {"v_cndmask_b32", 1}, {"v_mov_b32", 1}, {"v_add_co_u32", 1}, {"v_addc_co_u32", 1},
    {"v_bfe_u32", 1}, {"v_alignbit_b32", 1}, {"v_perm_b32", 1}, {"v_mul_lo_u32", 1}, {"v_mul_hi_u32", 1}, {"v_mul_u32_u24", 1},
    {"v_mad_u32_u24", 1}, {"v_mad_u64_u32", 1}, {"v_lshl_add_u64", 1}, {"v_fma_f32", 1}, {"v_pk_fma_f32", 1}, {"v_fma_f64", 1},
    {"v_dot4_i32_i8", 1}, {"v_cndmask_b32_e64 (SGPR-pair mask)", 1}, {"v_cmp_lt_u32+v_cndmask_b32 (pair, per instruction)", 2}, {"v_max_u32", 1},
    {"montgomery product (2 mad_u64_u32, mul_lo, sub, min)", 5}, {"modular add (add, sub, min)", 3},
    {"v_mad_i64_i32", 1}, {"v_mul_hi_i32", 1}, {"signed_montgomery product (2 mad_i64_i32, mul_lo)", 3},
};

// One loop iteration = ONE asm block of REPS x CHAINS instructions (operand %c = chain c, %8 = y, %9 = z), so that the
// compiler's hazard recogniser cannot put s_nop between the statements. 64-bit forms take 64-bit chain operands.
#define B8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define BODY(T) B8(T) B8(T) B8(T) B8(T)
#define ASM32(T) asm volatile(BODY(T) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y), "v"(z) : "vcc")
#define ASM64(T) asm volatile(BODY(T) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "v"(y), "v"(z), "v"(yz) : "vcc")
#define T_ADD(c) "v_add_u32 %" #c ", %" #c ", %8\n"
#define T_SUB(c) "v_sub_u32 %" #c ", %" #c ", %8\n"
#define T_ADD3(c) "v_add3_u32 %" #c ", %" #c ", %8, %9\n"
#define T_LSHL_ADD(c) "v_lshl_add_u32 %" #c ", %" #c ", 3, %8\n"
#define T_XOR(c) "v_xor_b32 %" #c ", %" #c ", %8\n"
#define T_AND_OR(c) "v_and_or_b32 %" #c ", %" #c ", %8, %9\n"
#define T_LSHR(c) "v_lshrrev_b32 %" #c ", 1, %" #c "\n"
#define T_MIN(c) "v_min_u32 %" #c ", %" #c ", %8\n"
#define T_CND(c) "v_cndmask_b32 %" #c ", %" #c ", %8, vcc\n"
#define T_MOV(c) "v_mov_b32 %" #c ", %8\n"
#define T_ADDCO(c) "v_add_co_u32 %" #c ", vcc, %" #c ", %8\n"
#define T_ADDC(c) "v_addc_co_u32 %" #c ", vcc, %" #c ", %8, vcc\n"
#define T_BFE(c) "v_bfe_u32 %" #c ", %" #c ", 1, 31\n"
#define T_ALIGN(c) "v_alignbit_b32 %" #c ", %" #c ", %8, 7\n"
#define T_PERM(c) "v_perm_b32 %" #c ", %" #c ", %8, %9\n"
#define T_MULLO(c) "v_mul_lo_u32 %" #c ", %" #c ", %8\n"
#define T_MULHI(c) "v_mul_hi_u32 %" #c ", %" #c ", %8\n"
#define T_MUL24(c) "v_mul_u32_u24 %" #c ", %" #c ", %8\n"
#define T_MAD24(c) "v_mad_u32_u24 %" #c ", %" #c ", %8, %9\n"
#define T_FMA(c) "v_fma_f32 %" #c ", %" #c ", %8, %9\n"
#define T_CNDS(c) "v_cndmask_b32 %" #c ", %" #c ", %8, s[10:11]\n"
#define T_CMPCND(c) "v_cmp_lt_u32 vcc, %" #c ", %8\nv_cndmask_b32 %" #c ", %" #c ", %9, vcc\n"
#define T_MAX(c) "v_max_u32 %" #c ", %" #c ", %8\n"
#define T_DOT4(c) "v_dot4_i32_i8 %" #c ", %8, %9, %" #c "\n"
#define T_MAD64(c) "v_mad_u64_u32 %" #c ", vcc, %8, %9, %" #c "\n"
#define T_MADI64(c) "v_mad_i64_i32 %" #c ", vcc, %8, %9, %" #c "\n"
#define T_MULHII(c) "v_mul_hi_i32 %" #c ", %" #c ", %8\n"
#define T_LSHLADD64(c) "v_lshl_add_u64 %" #c ", %" #c ", 1, %10\n"
#define T_PKFMA(c) "v_pk_fma_f32 %" #c ", %" #c ", %10, %10\n"
#define T_FMA64(c) "v_fma_f64 %" #c ", %" #c ", %10, %10\n"

__device__ __forceinline__ uint32_t monty_mul(uint32_t a, uint32_t b) {  // as bb::mul (babybear.hpp): compiled C++, free scheduling
    const uint64_t t = (uint64_t)a * b;
    const uint32_t m = (uint32_t)t * 0x77ffffffu;
    const uint64_t u = t + (uint64_t)m * P;
    const uint32_t r = (uint32_t)(u >> 32);
    return min(r, r - P);
}

__device__ __forceinline__ int32_t smonty_mul(int32_t a, int32_t b) {  // as bb::smont(bb::smul(a, b)): no conditional subtraction
    const int64_t t = (int64_t)a * b;
    const int32_t m = (int32_t)((uint32_t)t * 0x77ffffffu);
    return (int32_t)((t + (int64_t)m * (int64_t)P) >> 32);
}

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* cycles, uint32_t seed) {
    uint32_t x[CHAINS];
    uint64_t w[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { x[c] = seed + threadIdx.x * 977u + c * 131u + blockIdx.x; w[c] = ((uint64_t)(x[c] ^ 0x5555u) << 32) | x[c]; }
    const uint32_t y = (seed | 1u) + threadIdx.x, z = (seed * 3u) | 1u;
    const uint64_t yz = ((uint64_t)z << 32) | y;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
        if (OP == OP_ADD_U32) ASM32(T_ADD);
        else if (OP == OP_SUB_U32) ASM32(T_SUB);
        else if (OP == OP_ADD3_U32) ASM32(T_ADD3);
        else if (OP == OP_LSHL_ADD_U32) ASM32(T_LSHL_ADD);
        else if (OP == OP_XOR) ASM32(T_XOR);
        else if (OP == OP_AND_OR) ASM32(T_AND_OR);
        else if (OP == OP_LSHRREV) ASM32(T_LSHR);
        else if (OP == OP_MIN_U32) ASM32(T_MIN);
        else if (OP == OP_CNDMASK) ASM32(T_CND);
        else if (OP == OP_MOV) ASM32(T_MOV);
        else if (OP == OP_ADD_CO) ASM32(T_ADDCO);
        else if (OP == OP_ADDC_CO) ASM32(T_ADDC);
        else if (OP == OP_BFE) ASM32(T_BFE);
        else if (OP == OP_ALIGNBIT) ASM32(T_ALIGN);
        else if (OP == OP_PERM) ASM32(T_PERM);
        else if (OP == OP_MUL_LO) ASM32(T_MULLO);
        else if (OP == OP_MUL_HI) ASM32(T_MULHI);
        else if (OP == OP_MUL_U24) ASM32(T_MUL24);
        else if (OP == OP_MAD_U24) ASM32(T_MAD24);
        else if (OP == OP_FMA_F32) ASM32(T_FMA);
        else if (OP == OP_DOT4_I32_I8) ASM32(T_DOT4);
        else if (OP == OP_CNDMASK_SGPR) asm volatile("s_mov_b64 s[10:11], 0x5555\n" BODY(T_CNDS) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y), "v"(z) : "s10", "s11");
        else if (OP == OP_CMP_CNDMASK) ASM32(T_CMPCND);
        else if (OP == OP_MAX_U32) ASM32(T_MAX);
        else if (OP == OP_MAD_U64_U32) ASM64(T_MAD64);
        else if (OP == OP_LSHL_ADD_U64) ASM64(T_LSHLADD64);
        else if (OP == OP_PK_FMA_F32) ASM64(T_PKFMA);
        else if (OP == OP_FMA_F64) ASM64(T_FMA64);
        else if (OP == OP_MAD_I64_I32) ASM64(T_MADI64);
        else if (OP == OP_MUL_HI_I32) ASM32(T_MULHII);
        else if (OP == OP_SMONTY3) {
#pragma unroll
            for (int r = 0; r < REPS; ++r)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) x[c] = (uint32_t)smonty_mul((int32_t)x[c], (int32_t)(y >> 1));
        } else if (OP == OP_MONTY5) {
#pragma unroll
            for (int r = 0; r < REPS; ++r)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) x[c] = monty_mul(x[c], y);
        } else if (OP == OP_MODADD3) {
#pragma unroll
            for (int r = 0; r < REPS; ++r)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) { const uint32_t s_ = x[c] + y; x[c] = min(s_, s_ - P); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc ^= x[c] ^ (uint32_t)w[c] ^ (uint32_t)(w[c] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

struct Result { double cyc_clock, cyc_wall, ms; };

template <int OP>
int run(Result& res, int blocks, uint32_t* out, unsigned long long* d_cyc) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, d_cyc, 12345u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, d_cyc, 12345u + rep);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> cyc((size_t)blocks * 4);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : cyc) mean += (double)c;
    mean /= (double)cyc.size();
    const double instr_per_wave = (double)ITER * REPS * CHAINS * kOps[OP].instrs;
    const double waves_per_simd = (double)blocks * 4 / 1024.0;
    // the shader clock counter ticks at a fixed 100 MHz on some parts; report both views and let the reader pick the consistent one
    res.cyc_clock = mean / instr_per_wave / std::min(waves_per_simd, 8.0);
    res.cyc_wall = best * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);
    res.ms = best;
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs %d, clock %d kHz; %d chains x %d reps x %d iterations per lane, 256-thread blocks\n", prop.gcnArchName,
           prop.multiProcessorCount, prop.clockRate, CHAINS, REPS, ITER);
    for (int waves_per_simd : {8, 4, 1}) {
        const int blocks = 256 * waves_per_simd;  // 4 waves per block, 4 SIMDs per CU
        uint32_t* out; unsigned long long* d_cyc;
        CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
        CHECK(hipMalloc(&d_cyc, (size_t)blocks * 4 * 8));
        printf("\n== %d waves per SIMD (%d blocks) ==\n%-58s %10s %14s %14s\n", waves_per_simd, blocks, "opcode", "ms", "cyc/instr(wall)", "ticks/instr(ctr)");
        Result r[OP_COUNT];
#define RUN(OPX) if (run<OPX>(r[OPX], blocks, out, d_cyc)) return 1; printf("%-58s %10.3f %14.2f %14.2f\n", kOps[OPX].name, r[OPX].ms, r[OPX].cyc_wall, r[OPX].cyc_clock);
        RUN(OP_FMA_F32) RUN(OP_PK_FMA_F32) RUN(OP_FMA_F64) RUN(OP_MOV) RUN(OP_ADD_U32) RUN(OP_SUB_U32) RUN(OP_ADD3_U32) RUN(OP_LSHL_ADD_U32)
        RUN(OP_XOR) RUN(OP_AND_OR) RUN(OP_LSHRREV) RUN(OP_BFE) RUN(OP_ALIGNBIT) RUN(OP_PERM) RUN(OP_MIN_U32) RUN(OP_CNDMASK)
        RUN(OP_ADD_CO) RUN(OP_ADDC_CO) RUN(OP_LSHL_ADD_U64) RUN(OP_MUL_U24) RUN(OP_MAD_U24) RUN(OP_MUL_LO) RUN(OP_MUL_HI) RUN(OP_MAD_U64_U32)
        RUN(OP_DOT4_I32_I8) RUN(OP_CNDMASK_SGPR) RUN(OP_CMP_CNDMASK) RUN(OP_MAX_U32) RUN(OP_MAD_I64_I32) RUN(OP_MUL_HI_I32) RUN(OP_MODADD3) RUN(OP_MONTY5) RUN(OP_SMONTY3)
#undef RUN
        if (waves_per_simd == 8) {
            printf("JSON {\"waves_per_simd\": 8");
            for (int o = 0; o < OP_COUNT; ++o) {
                std::string n = kOps[o].name;
                n = n.substr(0, n.find(' '));
                printf(", \"%s\": %.3f", n.c_str(), r[o].cyc_wall);
            }
            printf("}\n");
        }
        (void)hipFree(out); (void)hipFree(d_cyc);
    }
    return 0;
}
