// Packing of an AIR's bus interactions into LogUp groups (one committed extension column per group), shared by the
// prover (pw_prover_create_logup) and the host verifier. Protocol definition: oracle/stark_oracle.cpp `group_starts`
// (the reference's backend calls the groups "chunks"; PowdrAir pushes the interactions one by one,
// /root/reference/openvm/src/powdr_extension/chip.rs:117-129).
#pragma once
#include "../../include/powdr_gpu.h"

#include <stddef.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

namespace pw {

// Degree in the trace columns of a post-fix program; kBadDegree if malformed.
constexpr int kBadDegree = 99;
inline int postfix_degree(const uint32_t* bc, uint32_t len) {
    int st[POWDR_EXPR_STACK_CAPACITY];
    int sp = 0;
    for (uint32_t ip = 0; ip < len;) {
        const uint32_t op = bc[ip++];
        switch (op) {
            case POWDR_OP_PUSH_APC:
            case POWDR_OP_PUSH_CONST:
                if (sp >= POWDR_EXPR_STACK_CAPACITY || ip >= len) return kBadDegree;
                st[sp++] = op == POWDR_OP_PUSH_APC ? 1 : 0;
                ++ip;
                break;
            case POWDR_OP_ADD:
            case POWDR_OP_SUB:
                if (sp < 2) return kBadDegree;
                --sp;
                st[sp - 1] = std::max(st[sp - 1], st[sp]);
                break;
            case POWDR_OP_MUL:
                if (sp < 2) return kBadDegree;
                --sp;
                st[sp - 1] += st[sp];
                break;
            case POWDR_OP_NEG:
                if (sp < 1) return kBadDegree;
                break;
            default:
                return kBadDegree;  // INV_OR_ZERO is not polynomial
        }
    }
    return sp == 1 ? st[0] : kBadDegree;
}

// Every column operand of a (well-formed) post-fix program names a column of the trace. A program that reaches the device without this
// check reads whatever lies behind the matrix: artifacts come from another machine (round 6).
inline bool postfix_columns_below(const uint32_t* bc, uint32_t len, uint32_t width) {
    for (uint32_t ip = 0; ip < len;) {
        const uint32_t op = bc[ip++];
        if (op == POWDR_OP_PUSH_APC || op == POWDR_OP_PUSH_CONST) {
            if (ip >= len) return false;
            if (op == POWDR_OP_PUSH_APC && bc[ip] >= width) return false;
            ++ip;
        }
    }
    return true;
}

// Group boundaries, n_groups + 1 entries ({0} for no interactions). An interaction joins the current group while the
// group's constraint  q * prod d_i - sum_i m_i prod_{j != i} d_j  keeps degree <= 3 (what a blow-up-2 quotient carries):
//   1 + sum deg d_j <= 3  and  deg m_j + sum_{k != j} deg d_k <= 3 for every member.
// `inter` = n x {bus, n_args, first span}; spans {off, len} laid out [mult, arg0, ...]; the caller has bounds-checked them.
inline std::vector<uint32_t> logup_group_starts(const uint32_t* inter, size_t n, const uint32_t* spans, const uint32_t* bc) {
    std::vector<uint32_t> starts{0};
    std::vector<int> deg_m, deg_d;
    int sum_d = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t na = inter[3 * i + 1];
        const uint32_t* sp = spans + 2 * (size_t)inter[3 * i + 2];
        const int m = postfix_degree(bc + sp[0], sp[1]);
        int d = 0;
        for (uint32_t j = 0; j < na; ++j) d = std::max(d, postfix_degree(bc + sp[2 + 2 * j], sp[3 + 2 * j]));
        bool joins = !deg_m.empty();
        if (joins) {
            const int total = sum_d + d;
            joins = 1 + total <= 3 && m + total - d <= 3;
            for (size_t k = 0; joins && k < deg_m.size(); ++k) joins = deg_m[k] + total - deg_d[k] <= 3;
        }
        if (!joins && !deg_m.empty()) {
            starts.push_back((uint32_t)i);
            deg_m.clear();
            deg_d.clear();
            sum_d = 0;
        }
        deg_m.push_back(m);
        deg_d.push_back(d);
        sum_d += d;
    }
    if (n) starts.push_back((uint32_t)n);
    return starts;
}

}  // namespace pw
