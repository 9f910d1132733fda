// "Small forms": expressions that, after simplification, are bilinear in at most two columns,
//     value = k0 + k1 * T[a] + k2 * T[b] + k3 * T[a] * T[b].
// Every multiplicity and argument of the bus interactions of the APCs this library has seen is of this shape (a column,
// a constant, 255 - byte, byte + 256 * byte, is_valid * flag, ...). The interpreter (xbc.hpp) pays a dependent
// global-memory round trip per instruction; a small form is evaluated by fixed code that issues all its loads at once.
// Host side: symbolic execution of the reference post-fix code (cuda/mod.rs:49-81). Device side: SmallForm::eval.
#pragma once
#include "babybear.hpp"
#include "../../include/powdr_gpu.h"

#include <stdint.h>
#include <vector>

namespace pw {

struct SmallForm {
    uint32_t a, b;            // column operands as the bytecode names them (element offset or column index)
    uint32_t k0, k1, k2, k3;  // Montgomery
    uint32_t flags;
    uint32_t pad;
    enum : uint32_t { USES_A = 1, USES_B = 2, HAS_PRODUCT = 4, IS_COLUMN = 8, IS_CONST = 16,
                      NOT_SMALL = 32 };  // set by users that keep a table entry for every expression: evaluate it some other way
#if defined(__HIPCC__)
    // ta / tb: the loaded cells (only read where the flags say so); flags are wave-uniform
    __device__ __forceinline__ uint32_t eval(uint32_t ta, uint32_t tb) const {
        if (flags & IS_CONST) return k0;
        if (flags & IS_COLUMN) return ta;
        uint32_t v = (flags & USES_B) ? bb::mul2(k1, ta, k2, tb) : bb::mul(k1, ta);
        if (flags & HAS_PRODUCT) v = bb::add(v, bb::mul(k3, bb::mul(ta, tb)));
        return bb::add(v, k0);
    }
#endif
};


// Returns false if the expression is not a small form (three columns, a square, an inverse, malformed code).
inline bool analyze_small_form(const uint32_t* bc, uint32_t len, SmallForm& out) {
    // canonical coefficients during the analysis
    struct Poly {
        bool has_a = false, has_b = false;
        uint32_t a = 0, b = 0;
        uint32_t k[4] = {0, 0, 0, 0};  // 1, A, B, AB
    };
    auto fadd = [](uint32_t x, uint32_t y) { return (uint32_t)(((uint64_t)x + y) % bb::P); };
    auto fmul = [](uint32_t x, uint32_t y) { return (uint32_t)((uint64_t)x * y % bb::P); };
    auto fneg = [](uint32_t x) { return x ? bb::P - x : 0u; };
    // bring q's columns into p's naming; false if the union has more than two columns
    auto unify = [](Poly& p, Poly& q) {
        auto slot_in = [](Poly& t, uint32_t col) -> int {  // 0 = A, 1 = B, -1 = no room
            if (t.has_a && t.a == col) return 0;
            if (t.has_b && t.b == col) return 1;
            if (!t.has_a) { t.has_a = true; t.a = col; return 0; }
            if (!t.has_b) { t.has_b = true; t.b = col; return 1; }
            return -1;
        };
        Poly r = p;
        int qa = q.has_a ? slot_in(r, q.a) : 0, qb = q.has_b ? slot_in(r, q.b) : 1;
        if (qa < 0 || qb < 0) return false;
        p.has_a = r.has_a; p.has_b = r.has_b; p.a = r.a; p.b = r.b;
        Poly m = r;
        for (auto& c : m.k) c = 0;
        m.k[0] = q.k[0];
        if (q.has_a) m.k[1 + qa] = q.k[1];
        if (q.has_b) m.k[1 + qb] = q.k[2];
        m.k[3] = q.k[3];  // AB is symmetric
        q = m;
        return true;
    };
    std::vector<Poly> st;
    for (uint32_t ip = 0; ip < len;) {
        const uint32_t op = bc[ip++];
        if (op == POWDR_OP_PUSH_APC || op == POWDR_OP_PUSH_CONST) {
            if (ip >= len || st.size() >= (size_t)POWDR_EXPR_STACK_CAPACITY) return false;
            Poly p;
            if (op == POWDR_OP_PUSH_APC) { p.has_a = true; p.a = bc[ip]; p.k[1] = 1; }
            else p.k[0] = bc[ip] % bb::P;
            ++ip;
            st.push_back(p);
        } else if (op == POWDR_OP_ADD || op == POWDR_OP_SUB || op == POWDR_OP_MUL) {
            if (st.size() < 2) return false;
            Poly y = st.back(); st.pop_back();
            Poly x = st.back(); st.pop_back();
            if (!unify(x, y)) return false;
            Poly r = x;
            if (op == POWDR_OP_MUL) {
                // (x0 + x1 A + x2 B + x3 AB)(y0 + y1 A + y2 B + y3 AB): any A^2 or B^2 term must vanish
                const uint32_t a2 = fmul(x.k[1], y.k[1]), b2 = fmul(x.k[2], y.k[2]);
                const uint32_t a2b = fadd(fmul(x.k[1], y.k[3]), fmul(x.k[3], y.k[1]));
                const uint32_t ab2 = fadd(fmul(x.k[2], y.k[3]), fmul(x.k[3], y.k[2]));
                if (a2 || b2 || a2b || ab2 || fmul(x.k[3], y.k[3])) return false;
                r.k[0] = fmul(x.k[0], y.k[0]);
                r.k[1] = fadd(fmul(x.k[0], y.k[1]), fmul(x.k[1], y.k[0]));
                r.k[2] = fadd(fmul(x.k[0], y.k[2]), fmul(x.k[2], y.k[0]));
                r.k[3] = fadd(fadd(fmul(x.k[0], y.k[3]), fmul(x.k[3], y.k[0])), fadd(fmul(x.k[1], y.k[2]), fmul(x.k[2], y.k[1])));
            } else {
                for (int i = 0; i < 4; ++i) r.k[i] = fadd(x.k[i], op == POWDR_OP_ADD ? y.k[i] : fneg(y.k[i]));
            }
            st.push_back(r);
        } else if (op == POWDR_OP_NEG) {
            if (st.empty()) return false;
            for (auto& c : st.back().k) c = fneg(c);
        } else {
            return false;  // INV_OR_ZERO or unknown
        }
    }
    if (st.size() != 1) return false;
    const Poly& p = st[0];
    out = SmallForm{};
    out.a = p.a;
    out.b = p.b;
    out.k0 = bb::to_monty(p.k[0]); out.k1 = bb::to_monty(p.k[1]); out.k2 = bb::to_monty(p.k[2]); out.k3 = bb::to_monty(p.k[3]);
    const bool ua = p.has_a && (p.k[1] || p.k[3]), ub = p.has_b && (p.k[2] || p.k[3]);
    if (!ua && ub) {  // keep A as the first used column
        out.a = p.b; out.k1 = out.k2; out.k2 = 0;
    }
    const bool a_used = ua || ub, b_used = ua && ub;
    out.flags = (a_used ? SmallForm::USES_A : 0u) | (b_used ? SmallForm::USES_B : 0u) | (p.k[3] ? SmallForm::HAS_PRODUCT : 0u);
    if (!a_used) out.flags |= SmallForm::IS_CONST;
    else if (!b_used && !p.k[3] && p.k[0] == 0 && ((ua ? p.k[1] : p.k[2]) == 1)) out.flags |= SmallForm::IS_COLUMN;
    return true;
}

}  // namespace pw
