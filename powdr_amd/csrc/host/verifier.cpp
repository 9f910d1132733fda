// pw-stark v0 verifier (host C++, Montgomery arithmetic) — the product-side counterpart of the
// reference's CPU verification step (`verify_app_proof::<BabyBearPoseidon2CpuEngine>`,
// /root/reference/openvm-riscv/src/lib.rs:337-341). Protocol: oracle/stark_oracle.cpp header / DESIGN.md §5.
// Independent of the oracle's verifier (different arithmetic representation, own transcript code).
#include "../prover_internal.hpp"
#include "../../../include/powdr_prover.h"
#include "../../../include/powdr_gpu.h"

#include <cstring>
#include <vector>

namespace {

using bb::Ext;
constexpr uint32_t kMagic = 0x31535750u;

struct Transcript {
    uint32_t st[16];
    std::vector<uint32_t> in, out;
    Transcript() { memset(st, 0, sizeof st); }
    void duplex() {
        for (size_t i = 0; i < in.size(); ++i) st[i] = in[i];
        in.clear();
        p2::permute(st, pw::poseidon2_params_host());
        out.assign(st, st + 8);
    }
    void observe(uint32_t m) { out.clear(); in.push_back(m); if (in.size() == 8) duplex(); }
    void observe_n(const uint32_t* w, size_t n) { for (size_t i = 0; i < n; ++i) observe(w[i]); }
    uint32_t sample() { if (!in.empty() || out.empty()) duplex(); uint32_t v = out.back(); out.pop_back(); return v; }
    Ext sample_ext() { Ext e; for (int i = 0; i < 4; ++i) e.c[i] = sample(); return e; }
    uint32_t sample_bits(int b) { return bb::from_monty(sample()) & ((1u << b) - 1u); }
};

struct Digest { uint32_t w[8]; };
bool same(const Digest& a, const Digest& b) { return !memcmp(a.w, b.w, 32); }

Digest hash_row(const uint32_t* row, size_t len) {
    uint32_t st[16] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < k; ++i) st[i] = row[off + i];
        p2::permute(st, pw::poseidon2_params_host());
    }
    Digest d;
    memcpy(d.w, st, 32);
    return d;
}
Digest compress(const Digest& l, const Digest& r) {
    uint32_t st[16];
    memcpy(st, l.w, 32);
    memcpy(st + 8, r.w, 32);
    p2::permute(st, pw::poseidon2_params_host());
    Digest d;
    memcpy(d.w, st, 32);
    return d;
}

// constraint program over opened (extension-field) values; reference post-fix opcodes
bool eval_ext(const uint32_t* bc, uint32_t len, const Ext* vals, uint32_t width, Ext& out) {
    Ext st[POWDR_EXPR_STACK_CAPACITY];
    int sp = 0;
    for (uint32_t ip = 0; ip < len;) {
        uint32_t op = bc[ip++];
        if (op == POWDR_OP_PUSH_APC || op == POWDR_OP_PUSH_CONST) {
            if (ip >= len || sp >= POWDR_EXPR_STACK_CAPACITY) return false;
            uint32_t a = bc[ip++];
            if (op == POWDR_OP_PUSH_APC) { if (a >= width) return false; st[sp++] = vals[a]; }
            else st[sp++] = bb::ext_from_base(bb::to_monty(a % bb::P));
        } else if (op == POWDR_OP_ADD || op == POWDR_OP_SUB || op == POWDR_OP_MUL) {
            if (sp < 2) return false;
            Ext b = st[--sp], a = st[--sp];
            st[sp++] = op == POWDR_OP_ADD ? bb::ext_add(a, b) : op == POWDR_OP_SUB ? bb::ext_sub(a, b) : bb::ext_mul(a, b);
        } else if (op == POWDR_OP_NEG) {
            if (sp < 1) return false;
            st[sp - 1] = bb::ext_neg(st[sp - 1]);
        } else {
            return false;  // INV_OR_ZERO is not a polynomial operation: not allowed in constraints
        }
    }
    if (sp != 1) return false;
    out = st[0];
    return true;
}

}  // namespace

// Returns 0 if the proof is valid, a positive code naming the first failed check otherwise:
// 1 header, 2 constraint identity at zeta, 3 proof of work, 4 query index, 5/6 trace/quotient opening,
// 7 FRI layer opening, 8 final polynomial, 9 trailing words, 10 truncated / malformed.
extern "C" int pw_verify(const PwStarkConfig* cfg, uint32_t width, uint32_t log_h, const uint32_t* bc, size_t bc_len,
                         const uint32_t* spans, size_t n_constraints, const uint32_t* proof, size_t len) {
    if (!cfg || !proof || log_h < 1 || log_h > 26) return 10;
    const size_t H = (size_t)1 << log_h, N = 2 * H;
    const int logN = (int)log_h + 1;
    size_t pos = 0;
    bool short_read = false;
    // proof words are canonical; everything below works on Montgomery words
    auto get = [&]() -> uint32_t { if (pos >= len) { short_read = true; return 0; } return proof[pos++]; };
    auto get_m = [&]() -> uint32_t { uint32_t c = get(); return bb::to_monty(c % bb::P); };
    auto get_digest = [&]() { Digest d; for (auto& w : d.w) w = get_m(); return d; };
    auto get_ext = [&]() { Ext e; for (auto& w : e.c) w = get_m(); return e; };

    const uint32_t hdr[6] = {kMagic, log_h, width, (uint32_t)n_constraints, cfg->num_queries, cfg->pow_bits};
    for (uint32_t h : hdr) if (get() != h) return 1;
    Transcript ch;
    ch.observe(bb::to_monty(kMagic % bb::P));
    for (int i = 1; i < 6; ++i) ch.observe(bb::to_monty(hdr[i] % bb::P));

    const Digest t_root = get_digest();
    ch.observe_n(t_root.w, 8);
    const Ext alpha = ch.sample_ext();
    const Digest q_root = get_digest();
    ch.observe_n(q_root.w, 8);
    const Ext zeta = ch.sample_ext();
    const size_t K = (size_t)width + 8;
    std::vector<Ext> opened(K);
    for (auto& e : opened) { e = get_ext(); ch.observe_n(e.c, 4); }
    if (short_read) return 10;

    // constraint identity: sum_j alpha^(nc-1-j) C_j(opened) == Z_H(zeta) * (Q_lo(zeta) + zeta^H Q_hi(zeta))
    Ext acc = bb::ext_zero();
    for (size_t k = 0; k < n_constraints; ++k) {
        const uint32_t off = spans[2 * k], ln = spans[2 * k + 1];
        Ext v;
        if ((size_t)off + ln > bc_len || !eval_ext(bc + off, ln, opened.data(), width, v)) return 10;
        acc = bb::ext_add(bb::ext_mul(acc, alpha), v);
    }
    const Ext zH = bb::ext_pow(zeta, H);
    const Ext zh = bb::ext_sub(zH, bb::ext_one());
    Ext qlo, qhi;  // sum_k X^k * q_k: the coordinates of the opened chunk columns are the basis coefficients
    {
        auto combine = [&](size_t base) {
            Ext r = bb::ext_zero();
            for (int k = 0; k < 4; ++k) {
                Ext basis = bb::ext_zero();
                basis.c[k] = bb::R_MOD_P;
                r = bb::ext_add(r, bb::ext_mul(basis, opened[base + k]));
            }
            return r;
        };
        qlo = combine(width);
        qhi = combine(width + 4);
    }
    if (!bb::ext_eq(acc, bb::ext_mul(zh, bb::ext_add(qlo, bb::ext_mul(zH, qhi))))) return 2;

    const Ext gamma = ch.sample_ext();
    std::vector<Ext> gpow(K);
    Ext opened_sum = bb::ext_zero();
    {
        Ext g = bb::ext_one();
        for (size_t k = 0; k < K; ++k) { gpow[k] = g; g = bb::ext_mul(g, gamma); }
        for (size_t k = 0; k < K; ++k) opened_sum = bb::ext_add(opened_sum, bb::ext_mul(gpow[k], opened[k]));
    }
    std::vector<Digest> fri_roots(log_h);
    std::vector<Ext> betas(log_h);
    for (uint32_t l = 0; l < log_h; ++l) {
        fri_roots[l] = get_digest();
        ch.observe_n(fri_roots[l].w, 8);
        betas[l] = ch.sample_ext();
    }
    const Ext final_poly = get_ext();
    ch.observe_n(final_poly.c, 4);
    const uint32_t witness = get();
    if (short_read) return 10;
    ch.observe(bb::to_monty(witness % bb::P));
    if (cfg->pow_bits && ch.sample_bits((int)cfg->pow_bits) != 0) return 3;

    auto check_path = [&](Digest leaf, size_t idx, int depth, const Digest& root) {
        for (int l = 0; l < depth; ++l) {
            Digest sib = get_digest();
            leaf = ((idx >> l) & 1) ? compress(sib, leaf) : compress(leaf, sib);
        }
        return same(leaf, root);
    };
    const uint32_t shift0 = bb::to_monty(pw::field::kCosetShift);
    const uint32_t inv2 = bb::inv(bb::to_monty(2));
    std::vector<uint32_t> trow(width), qrow(8);
    for (uint32_t qi = 0; qi < cfg->num_queries; ++qi) {
        const size_t idx = ch.sample_bits(logN);
        if (get() != idx) return short_read ? 10 : 4;
        for (auto& w : trow) w = get_m();
        if (short_read) return 10;
        if (!check_path(hash_row(trow.data(), width), idx, logN, t_root)) return short_read ? 10 : 5;
        for (auto& w : qrow) w = get_m();
        if (!check_path(hash_row(qrow.data(), 8), idx, logN, q_root)) return short_read ? 10 : 6;
        const uint32_t x = bb::mul(shift0, bb::pow_u32(pw::field::root_of_unity(logN), (uint32_t)idx));
        Ext a = bb::ext_zero();
        for (size_t k = 0; k < width; ++k) a = bb::ext_add(a, bb::ext_scale(gpow[k], trow[k]));
        for (size_t k = 0; k < 8; ++k) a = bb::ext_add(a, bb::ext_scale(gpow[width + k], qrow[k]));
        Ext cur = bb::ext_mul(bb::ext_sub(a, opened_sum), bb::ext_inv(bb::ext_sub(bb::ext_from_base(x), zeta)));
        uint32_t shift = shift0;
        for (uint32_t l = 0; l < log_h; ++l) {
            const size_t Nl = N >> l, half = Nl / 2, p = idx & (Nl - 1);
            const Ext sib = get_ext();
            const Ext lo = p < half ? cur : sib, hi = p < half ? sib : cur;
            uint32_t row[8];
            memcpy(row, lo.c, 16);
            memcpy(row + 4, hi.c, 16);
            if (!check_path(hash_row(row, 8), p & (half - 1), logN - 1 - (int)l, fri_roots[l])) return short_read ? 10 : 7;
            const uint32_t xi = bb::mul(shift, bb::pow_u32(pw::field::root_of_unity(logN - (int)l), (uint32_t)(p & (half - 1))));
            const Ext s = bb::ext_scale(bb::ext_add(lo, hi), inv2);
            const Ext d = bb::ext_scale(bb::ext_sub(lo, hi), bb::mul(inv2, bb::inv(xi)));
            cur = bb::ext_add(s, bb::ext_mul(betas[l], d));
            shift = bb::sqr(shift);
        }
        if (short_read) return 10;
        if (!bb::ext_eq(cur, final_poly)) return 8;
    }
    if (short_read) return 10;
    if (pos != len) return 9;
    return 0;
}
