// Host-side entry point to the xbc compiler (xbc_compile.hpp) for tests: compile a reference
// post-fix expression and evaluate the resulting xbc program on the host, with exactly the
// semantics of the device evaluator xbc::eval (xbc.hpp).
#include "../xbc_compile.hpp"
#include "../small_form.hpp"
#include "../field_selftest.hpp"
#include "../../../include/powdr_host.h"

#include <vector>

extern "C" int powdr_xbc_eval_host(const uint32_t* postfix, uint32_t len, const uint32_t* trace, size_t r,
                                   uint32_t* result, uint32_t* n_instr) {
    std::vector<uint32_t> code;
    xbc::Compiler cc;
    if (!cc.compile(postfix, len, code)) return -1;
    if (n_instr) *n_instr = (uint32_t)(code.size() / 2);
    uint32_t stack[POWDR_EXPR_STACK_CAPACITY + 1];
    int sp = 0;
    uint32_t top = 0;
    for (size_t ip = 0; ip < code.size(); ip += 2) {
        const uint32_t op = code[ip], a = code[ip + 1];
        switch (op) {
            case xbc::PUSH_COL: stack[sp++] = top; /* fallthrough */
            case xbc::SET_COL: top = trace[(size_t)a + r]; break;
            case xbc::PUSH_CONST: stack[sp++] = top; /* fallthrough */
            case xbc::SET_CONST: top = a; break;
            case xbc::ADD_COL: top = bb::add(top, trace[(size_t)a + r]); break;
            case xbc::SUB_COL: top = bb::sub(top, trace[(size_t)a + r]); break;
            case xbc::RSUB_COL: top = bb::sub(trace[(size_t)a + r], top); break;
            case xbc::MUL_COL: top = bb::mul(top, trace[(size_t)a + r]); break;
            case xbc::ADD_CONST: top = bb::add(top, a); break;
            case xbc::RSUB_CONST: top = bb::sub(a, top); break;
            case xbc::MUL_CONST: top = bb::mul(top, a); break;
            case xbc::ADD: top = bb::add(stack[--sp], top); break;
            case xbc::SUB: top = bb::sub(stack[--sp], top); break;
            case xbc::RSUB: top = bb::sub(top, stack[--sp]); break;
            case xbc::MUL: top = bb::mul(stack[--sp], top); break;
            case xbc::NEG: top = bb::neg(top); break;
            case xbc::INV: top = bb::inv_or_zero(top); break;
            default: return -2;
        }
        if (sp < 0 || sp > POWDR_EXPR_STACK_CAPACITY) return -3;
    }
    *result = top;
    return 0;
}

// Test hook for the small-form analysis (small_form.hpp): 0 and the value if `postfix` is bilinear in at most two
// columns, 1 if it is a well-formed expression of another shape (the kernels then interpret it), -1 never.
extern "C" int powdr_small_form_eval_host(const uint32_t* postfix, uint32_t len, const uint32_t* trace, size_t r, uint32_t* result,
                                          uint32_t* flags) {
    pw::SmallForm f;
    if (!pw::analyze_small_form(postfix, len, f)) return 1;
    if (flags) *flags = f.flags;
    const uint32_t ta = (f.flags & pw::SmallForm::USES_A) ? trace[(size_t)f.a + r] : 0u;
    const uint32_t tb = (f.flags & pw::SmallForm::USES_B) ? trace[(size_t)f.b + r] : 0u;
    // the device evaluator's arithmetic (SmallForm::eval), restated for the host
    uint32_t v;
    if (f.flags & pw::SmallForm::IS_CONST) v = f.k0;
    else if (f.flags & pw::SmallForm::IS_COLUMN) v = ta;
    else {
        v = (f.flags & pw::SmallForm::USES_B) ? bb::mul2(f.k1, ta, f.k2, tb) : bb::mul(f.k1, ta);
        if (f.flags & pw::SmallForm::HAS_PRODUCT) v = bb::add(v, bb::mul(f.k3, bb::mul(ta, tb)));
        v = bb::add(v, f.k0);
    }
    *result = v;
    return 0;
}

// Host entry point of the field self-test (field_selftest.hpp).
extern "C" int powdr_field_selftest(uint64_t seed, uint32_t iterations) { return pw::field_selftest_checks(seed, iterations); }
