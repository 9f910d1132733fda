// Source generation for the run-time specialised kernels (jit.hpp): xbc programs (xbc.hpp) -> straight-line HIP.
//
// A program is cut into CHUNKS of bounded code size (a chunk should stay resident in the 64 KB instruction cache a CU pair
// shares); a kernel's grid is (row blocks, chunks): blockIdx.y selects the chunk, so the workgroups resident on a CU at any
// moment run the same few kilobytes of code, and a short trace still fills the chip. Every chunk leaves a partial sum per
// row; small ahead-of-time kernels add the partial sums up (stark_kernels.hip, logup_kernels.hip). Chunks are spread over
// several TRANSLATION UNITS that hiprtc compiles concurrently (instruction selection costs ~0.3 ms per instruction).
#pragma once
#include "prover_internal.hpp"

#include <string>
#include <vector>

namespace pw { namespace jit {

struct XbcView {
    const uint32_t* code;   // xbc instructions, two words each, column-index operands
    const uint32_t* spans;  // {off, len} pairs in instructions
    uint32_t n;             // programs
};
struct LogupView {
    const LogupInteraction* inter;  // {bus (Montgomery), n_args, first span} ; spans laid out [mult, arg0, ...]
    uint32_t n_inter;
    XbcView exprs;
    const uint32_t* gstarts;  // n_groups + 1
    uint32_t n_groups;
};

struct Unit {
    std::string source;
    std::string kernel;     // name of the __global__ function
    uint32_t first_chunk;   // chunks [first_chunk, first_chunk + n_chunks) of the program: gridDim.y = n_chunks
    uint32_t n_chunks;
    uint32_t g0 = 0, g1 = 0;  // the LogUp groups [g0, g1) this unit's chunks cover (g0 == g1: constraint chunks only): the only
                              // permutation columns the unit reads are 4 g0 .. 4 g1 - 1 (the streamed prover extends just those)
};
struct Generated {
    std::vector<Unit> units;
    uint32_t n_chunks = 0;
    size_t est_instructions = 0;
};

// Quotient numerator on the extended domain. Kernel signature (all units):
//   (const uint32_t* lde, const uint32_t* plde, uint64_t N, const Ext* apow, Ext al, const Ext* blpow, uint32_t* part)
// part[(chunk * 4 + k) * N + j] = coordinate k of the chunk's share of
//   sum_c apow[c] C_c(row j) + sum_g apow[nc + g] (q_g prod_i d_i - sum_i m_i prod_{l != i} d_l)      (lg == nullptr: first sum only)
Generated gen_quotient(const XbcView& cons, const LogupView* lg, uint32_t chunk_cost, uint32_t chunks_per_unit);

// LogUp permutation columns on the trace domain. Kernel signature:
//   (const uint32_t* trace, uint64_t H, Ext al, const Ext* blpow, uint32_t* perm, uint32_t* rowsum_part)
// perm[(4 g + k) * H + r] = coordinate k of q_g(r) = sum_{i in g} m_i / d_i;  rowsum_part[(chunk * 4 + k) * H + r] = the chunk's sum_g q_g
Generated gen_logup_perm(const LogupView& lg, uint32_t chunk_cost, uint32_t chunks_per_unit);

}}  // namespace pw::jit
