// Per-kind constants of the thirteen original RV32IM instruction AIRs (POWDR_ORIG_* order, include/powdr_gpu.h), shared by the
// device expanders (original_chips.hip) and the host orchestration (host/apc_host.cpp). Source of every number: the reference's
// snapshot openvm-riscv/tests/openvm_constraints.txt (column counts; one memory access = one timestamp step = one previous
// timestamp in the record) and openvm-riscv/src/isa/opcode.rs (opcode numbers).
#pragma once
#include <cstdint>

#include "../../include/powdr_gpu.h"

namespace orig {

constexpr int kKinds = POWDR_ORIG_KIND_COUNT;
// columns of the AIR, record words per instruction, memory accesses (= timestamp step = trailing previous-timestamp words), first / last opcode
constexpr int kWidths[kKinds] = {36, 53, 41, 26, 18, 37, 32, 28, 36, 59, 39, 31, 20};
constexpr int kRecordWords[kKinds] = {6, 6, 6, 4, 2, 6, 4, 4, 6, 6, 6, 6, 2};
constexpr int kAccesses[kKinds] = {3, 3, 3, 2, 1, 3, 2, 2, 3, 3, 3, 3, 1};
constexpr uint32_t kOpcodeLo[kKinds] = {512, 517, 528, 544, 560, 520, 549, 565, 534, 596, 593, 592, 576};
constexpr uint32_t kOpcodeHi[kKinds] = {516, 519, 533, 545, 561, 521, 552, 565, 535, 599, 595, 592, 576};

inline int kind_of_opcode(uint32_t opcode) {
    for (int k = 0; k < kKinds; ++k)
        if (opcode >= kOpcodeLo[k] && opcode <= kOpcodeHi[k]) return k;
    return -1;
}

}  // namespace orig
