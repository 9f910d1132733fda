// pw-stark v0 verifier (host C++, Montgomery arithmetic) — the product-side counterpart of the
// reference's CPU verification step (`verify_app_proof::<BabyBearPoseidon2CpuEngine>`,
// /root/reference/openvm-riscv/src/lib.rs:337-341). Protocol: oracle/stark_oracle.cpp header / DESIGN.md §5.
// Independent of the oracle's verifier (different arithmetic representation, own transcript code).
#include "../prover_internal.hpp"
#include "../logup_groups.hpp"
#include "../../../include/powdr_prover.h"
#include "../../../include/powdr_gpu.h"

#include <cstring>
#include <vector>

namespace {

using bb::Ext;
constexpr uint32_t kMagic = 0x31535750u;   // "PWS1"
constexpr uint32_t kMagic2 = 0x32535750u;  // "PWS2": with the LogUp extension

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

// the AIR's bus interactions as the verifier is told them (pw_prover_create_logup's arguments)
struct Interactions {
    const uint32_t* inter;   // n x {bus, n_args, first span}
    size_t n;
    const uint32_t* spans;   // {off, len} pairs: [mult, arg0, ...] per interaction
    size_t n_spans;
    const uint32_t* bc;
    size_t bc_len;
};

// Returns 0 if the proof is valid, a positive code naming the first failed check otherwise:
// 1 header, 2 constraint identity at zeta, 3 proof of work, 4 query index, 5/6 trace/quotient opening,
// 7 FRI layer opening, 8 final polynomial, 9 trailing words, 10 truncated / malformed, 11 permutation opening, 12 bus seed.
int verify_impl(const PwStarkConfig* cfg, uint32_t width, uint32_t log_h, const uint32_t* bc, size_t bc_len, const uint32_t* spans,
                size_t n_constraints, const Interactions* lg, const uint32_t* expected_seed, const uint32_t* proof, size_t len,
                uint32_t* sum_out, uint32_t* trace_root_out) {
    if (!cfg || !proof || log_h < 1 || log_h > 26) return 10;
    const uint32_t n_int = lg ? (uint32_t)lg->n : 0;
    uint32_t max_args = 0;
    std::vector<uint32_t> gstarts{0};
    if (lg) {
        for (size_t i = 0; i < lg->n; ++i) {
            const uint32_t na = lg->inter[3 * i + 1], first = lg->inter[3 * i + 2];
            if ((size_t)first + 1 + na > lg->n_spans) return 10;
            for (uint32_t k = 0; k <= na; ++k)
                if ((size_t)lg->spans[2 * (first + k)] + lg->spans[2 * (first + k) + 1] > lg->bc_len) return 10;
            if (na > max_args) max_args = na;
        }
        gstarts = pw::logup_group_starts(lg->inter, lg->n, lg->spans, lg->bc);
    }
    const size_t n_g = gstarts.size() - 1;
    const size_t Wp = lg ? 4 * (n_g + 1) : 0;
    const size_t H = (size_t)1 << log_h, N = 2 * H;
    const int logN = (int)log_h + 1;
    size_t pos = 0;
    bool short_read = false;
    // proof words are canonical; everything below works on Montgomery words
    auto get = [&]() -> uint32_t { if (pos >= len) { short_read = true; return 0; } return proof[pos++]; };
    auto get_m = [&]() -> uint32_t { uint32_t c = get(); return bb::to_monty(c % bb::P); };
    auto get_digest = [&]() { Digest d; for (auto& w : d.w) w = get_m(); return d; };
    auto get_ext = [&]() { Ext e; for (auto& w : e.c) w = get_m(); return e; };

    std::vector<uint32_t> hdr = {lg ? kMagic2 : kMagic, log_h, width, (uint32_t)n_constraints};
    if (lg) hdr.push_back(n_int);
    hdr.push_back(cfg->num_queries);
    hdr.push_back(cfg->pow_bits);
    for (uint32_t h : hdr) if (get() != h) return 1;
    // proof words are canonical field elements (indices, the witness and the header are far below p as well): a word >= p
    // would be a second encoding of the same element, i.e. a malleable proof
    for (size_t i = 0; i < len; ++i) if (proof[i] >= bb::P) return 13;
    Transcript ch;
    for (uint32_t h : hdr) ch.observe(bb::to_monty(h % bb::P));

    const Digest t_root = get_digest();
    ch.observe_n(t_root.w, 8);
    Ext al = bb::ext_zero(), bl = bb::ext_zero(), S = bb::ext_zero();
    Digest p_root{};
    if (lg) {
        // bus challenges: own transcript over the bus seed (shared by the AIRs of a segment; default = own trace root)
        const Digest seed = get_digest();
        if (short_read) return 10;
        Digest want = t_root;
        if (expected_seed) for (int i = 0; i < 8; ++i) want.w[i] = bb::to_monty(expected_seed[i] % bb::P);
        if (!same(seed, want)) return 12;
        ch.observe_n(seed.w, 8);
        Transcript cb;
        cb.observe(bb::to_monty(kMagic2 % bb::P));
        cb.observe_n(seed.w, 8);
        al = cb.sample_ext();
        bl = cb.sample_ext();
        p_root = get_digest();
        ch.observe_n(p_root.w, 8);
        S = get_ext();
        ch.observe_n(S.c, 4);
    }
    const Ext alpha = ch.sample_ext();
    const Digest q_root = get_digest();
    ch.observe_n(q_root.w, 8);
    const Ext zeta = ch.sample_ext();
    const uint32_t g_h = pw::field::root_of_unity((int)log_h), g_inv = bb::inv(g_h);
    const Ext gzeta = bb::ext_scale(zeta, g_h);
    // internal order (= order of the gamma powers): main | perm@zeta | quotient | perm@g*zeta;
    // proof order: main, perm@zeta, perm@g*zeta, quotient
    const size_t K1 = (size_t)width + Wp + 8, K = K1 + Wp;
    std::vector<Ext> opened(K);
    auto read_opened = [&](size_t a, size_t b) { for (size_t k = a; k < b; ++k) { opened[k] = get_ext(); ch.observe_n(opened[k].c, 4); } };
    read_opened(0, width + Wp);
    read_opened(K1, K);
    read_opened(width + Wp, K1);
    if (short_read) return 10;

    // constraint identity: sum_j alpha^(M-1-j) C_j(opened) == Z_H(zeta) * (Q_lo(zeta) + zeta^H Q_hi(zeta))
    Ext acc = bb::ext_zero();
    for (size_t k = 0; k < n_constraints; ++k) {
        const uint32_t off = spans[2 * k], ln = spans[2 * k + 1];
        Ext v;
        if ((size_t)off + ln > bc_len || !eval_ext(bc + off, ln, opened.data(), width, v)) return 10;
        acc = bb::ext_add(bb::ext_mul(acc, alpha), v);
    }
    const Ext zH = bb::ext_pow(zeta, H);
    const Ext zh = bb::ext_sub(zH, bb::ext_one());
    // sum_k X^k * o_k: the coordinates of an opened extension-valued column are the basis coefficients
    auto combine = [&](size_t base) {
        Ext r = bb::ext_zero();
        for (int k = 0; k < 4; ++k) {
            Ext basis = bb::ext_zero();
            basis.c[k] = bb::R_MOD_P;
            r = bb::ext_add(r, bb::ext_mul(basis, opened[base + k]));
        }
        return r;
    };
    if (lg) {
        // LogUp: per group q_g * prod d_i = sum_i m_i prod_{j != i} d_j on every row (i.e. q_g = sum_i m_i / d_i);
        // phi is the running sum of sum_g q_g and ends at S
        std::vector<Ext> blpow(max_args + 2);
        { Ext b = bb::ext_one(); for (auto& x : blpow) { x = b; b = bb::ext_mul(b, bl); } }
        Ext sumq = bb::ext_zero(), sumq_next = bb::ext_zero();
        for (size_t g = 0; g < n_g; ++g) {
            Ext num = bb::ext_zero(), den = bb::ext_one();
            for (size_t i = gstarts[g]; i < gstarts[g + 1]; ++i) {
                const uint32_t bus = lg->inter[3 * i], na = lg->inter[3 * i + 1];
                const uint32_t* sp = lg->spans + 2 * (size_t)lg->inter[3 * i + 2];
                Ext d = bb::ext_add(al, bb::ext_from_base(bb::to_monty(bus % bb::P))), m, a;
                for (uint32_t j = 0; j < na; ++j) {
                    if (!eval_ext(lg->bc + sp[2 + 2 * j], sp[3 + 2 * j], opened.data(), width, a)) return 10;
                    d = bb::ext_add(d, bb::ext_mul(blpow[j + 1], a));
                }
                if (!eval_ext(lg->bc + sp[0], sp[1], opened.data(), width, m)) return 10;
                num = bb::ext_add(bb::ext_mul(num, d), bb::ext_mul(den, m));
                den = bb::ext_mul(den, d);
            }
            const Ext qi = combine(width + 4 * g), qn = combine(K1 + 4 * g);
            sumq = bb::ext_add(sumq, qi);
            sumq_next = bb::ext_add(sumq_next, qn);
            acc = bb::ext_add(bb::ext_mul(acc, alpha), bb::ext_sub(bb::ext_mul(qi, den), num));
        }
        const Ext phi = combine(width + 4 * n_g), phin = combine(K1 + 4 * n_g);
        const Ext is_trans = bb::ext_sub(zeta, bb::ext_from_base(g_inv));
        const Ext is_first = bb::ext_mul(zh, bb::ext_inv(bb::ext_sub(zeta, bb::ext_one())));
        const Ext is_last = bb::ext_mul(zh, bb::ext_inv(is_trans));
        acc = bb::ext_add(bb::ext_mul(acc, alpha), bb::ext_mul(is_first, bb::ext_sub(phi, sumq)));
        acc = bb::ext_add(bb::ext_mul(acc, alpha), bb::ext_mul(is_trans, bb::ext_sub(bb::ext_sub(phin, phi), sumq_next)));
        acc = bb::ext_add(bb::ext_mul(acc, alpha), bb::ext_mul(is_last, bb::ext_sub(phi, S)));
    }
    const Ext qlo = combine(width + Wp), qhi = combine(width + Wp + 4);
    if (!bb::ext_eq(acc, bb::ext_mul(zh, bb::ext_add(qlo, bb::ext_mul(zH, qhi))))) return 2;

    const Ext gamma = ch.sample_ext();
    std::vector<Ext> gpow(K);
    Ext opened_sum = bb::ext_zero(), opened_sum2 = bb::ext_zero();
    {
        Ext g = bb::ext_one();
        for (size_t k = 0; k < K; ++k) { gpow[k] = g; g = bb::ext_mul(g, gamma); }
        for (size_t k = 0; k < K1; ++k) opened_sum = bb::ext_add(opened_sum, bb::ext_mul(gpow[k], opened[k]));
        for (size_t k = K1; k < K; ++k) opened_sum2 = bb::ext_add(opened_sum2, bb::ext_mul(gpow[k], opened[k]));
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
    std::vector<uint32_t> trow(width), prow(Wp), qrow(8);
    for (uint32_t qi = 0; qi < cfg->num_queries; ++qi) {
        const size_t idx = ch.sample_bits(logN);
        if (get() != idx) return short_read ? 10 : 4;
        for (auto& w : trow) w = get_m();
        if (short_read) return 10;
        if (!check_path(hash_row(trow.data(), width), idx, logN, t_root)) return short_read ? 10 : 5;
        if (lg) {
            for (auto& w : prow) w = get_m();
            if (!check_path(hash_row(prow.data(), Wp), idx, logN, p_root)) return short_read ? 10 : 11;
        }
        for (auto& w : qrow) w = get_m();
        if (!check_path(hash_row(qrow.data(), 8), idx, logN, q_root)) return short_read ? 10 : 6;
        const uint32_t x = bb::mul(shift0, bb::pow_u32(pw::field::root_of_unity(logN), (uint32_t)idx));
        Ext a = bb::ext_zero(), a2 = bb::ext_zero();
        for (size_t k = 0; k < width; ++k) a = bb::ext_add(a, bb::ext_scale(gpow[k], trow[k]));
        for (size_t k = 0; k < Wp; ++k) {
            a = bb::ext_add(a, bb::ext_scale(gpow[width + k], prow[k]));
            a2 = bb::ext_add(a2, bb::ext_scale(gpow[K1 + k], prow[k]));
        }
        for (size_t k = 0; k < 8; ++k) a = bb::ext_add(a, bb::ext_scale(gpow[width + Wp + k], qrow[k]));
        Ext cur = bb::ext_mul(bb::ext_sub(a, opened_sum), bb::ext_inv(bb::ext_sub(bb::ext_from_base(x), zeta)));
        if (lg) cur = bb::ext_add(cur, bb::ext_mul(bb::ext_sub(a2, opened_sum2), bb::ext_inv(bb::ext_sub(bb::ext_from_base(x), gzeta))));
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
    if (sum_out) for (int k = 0; k < 4; ++k) sum_out[k] = bb::from_monty(S.c[k]);
    if (trace_root_out) for (int k = 0; k < 8; ++k) trace_root_out[k] = bb::from_monty(t_root.w[k]);
    return 0;
}

}  // namespace

extern "C" int pw_verify(const PwStarkConfig* cfg, uint32_t width, uint32_t log_h, const uint32_t* bc, size_t bc_len,
                         const uint32_t* spans, size_t n_constraints, const uint32_t* proof, size_t len) {
    return verify_impl(cfg, width, log_h, bc, bc_len, spans, n_constraints, nullptr, nullptr, proof, len, nullptr, nullptr);
}

extern "C" int pw_verify_logup(const PwStarkConfig* cfg, uint32_t width, uint32_t log_h, const uint32_t* bc, size_t bc_len,
                               const uint32_t* spans, size_t n_constraints, const uint32_t* interactions, size_t n_interactions,
                               const uint32_t* inter_spans, size_t n_inter_spans, const uint32_t* inter_bytecode,
                               size_t inter_bytecode_len, const uint32_t* expected_bus_seed, const uint32_t* proof, size_t len,
                               uint32_t* cumulative_sum, uint32_t* trace_root) {
    const Interactions lg{interactions, n_interactions, inter_spans, n_inter_spans, inter_bytecode, inter_bytecode_len};
    return verify_impl(cfg, width, log_h, bc, bc_len, spans, n_constraints, &lg, expected_bus_seed, proof, len, cumulative_sum,
                       trace_root);
}

// ---- pw-stark v1: one proof per segment (protocol: oracle/stark_segment.inc; prover: csrc/segment_prover.hip) ----------
namespace {
constexpr uint32_t kMagic3 = 0x33535750u;  // "PWS3"

struct SegShapeV {
    uint32_t W, nc, n_int, log_h;
    size_t H, N, n_g, Wp, K, koff;
    int logN;
    std::vector<uint32_t> gstarts;
    uint32_t max_args;
};
}  // namespace

extern "C" int pw_verify_segment(const PwStarkConfig* cfg, const PwAirDescription* airs, size_t n_airs, int logup_flag,
                                 const uint32_t* proof, size_t len, int check_balance, uint32_t* total_sum4) {
    if (!cfg || !airs || !n_airs || !proof) return 15;
    const bool lg = logup_flag != 0;
    const size_t A = n_airs;
    std::vector<SegShapeV> sh(A);
    size_t K_total = 0;
    int L = 0;
    for (size_t a = 0; a < A; ++a) {
        const PwAirDescription& d = airs[a];
        SegShapeV& s = sh[a];
        if (d.log_height < 1 || d.log_height > 26 || !d.width) return 15;
        s.W = d.width; s.nc = (uint32_t)d.n_constraints; s.log_h = d.log_height; s.n_int = lg ? (uint32_t)d.n_interactions : 0;
        s.H = (size_t)1 << s.log_h; s.N = 2 * s.H; s.logN = (int)s.log_h + 1;
        s.max_args = 0; s.n_g = 0; s.Wp = 0;
        for (size_t k = 0; k < d.n_constraints; ++k)
            if ((size_t)d.cons_spans[2 * k] + d.cons_spans[2 * k + 1] > d.bytecode_len) return 15;
        if (lg) {
            for (size_t i = 0; i < d.n_interactions; ++i) {
                const uint32_t na = d.interactions[3 * i + 1], first = d.interactions[3 * i + 2];
                if ((size_t)first + 1 + na > d.n_inter_spans) return 15;
                for (uint32_t k = 0; k <= na; ++k)
                    if ((size_t)d.inter_spans[2 * (first + k)] + d.inter_spans[2 * (first + k) + 1] > d.inter_bytecode_len) return 15;
                if (na > s.max_args) s.max_args = na;
            }
            s.gstarts = pw::logup_group_starts(d.interactions, d.n_interactions, d.inter_spans, d.inter_bytecode);
            if (s.gstarts.empty()) s.gstarts.push_back(0);
            s.n_g = s.gstarts.size() - 1;
            s.Wp = 4 * (s.n_g + 1);
        }
        s.K = (size_t)s.W + 2 * s.Wp + 8;
        s.koff = K_total;
        K_total += s.K;
        if (s.logN > L) L = s.logN;
    }
    const size_t Nmax = (size_t)1 << L;
    size_t pos = 0;
    bool short_read = false;
    auto get = [&]() -> uint32_t { if (pos >= len) { short_read = true; return 0; } return proof[pos++]; };
    auto get_m = [&]() -> uint32_t { return bb::to_monty(get() % bb::P); };
    auto get_digest = [&]() { Digest d; for (auto& w : d.w) w = get_m(); return d; };
    auto get_ext = [&]() { Ext e; for (auto& w : e.c) w = get_m(); return e; };

    std::vector<uint32_t> hdr = {kMagic3, (uint32_t)A, lg ? 1u : 0u, cfg->num_queries, cfg->pow_bits};
    for (size_t a = 0; a < A; ++a) for (uint32_t x : {sh[a].log_h, sh[a].W, sh[a].nc, sh[a].n_int}) hdr.push_back(x);
    for (uint32_t h : hdr) if (get() != h) return 1;
    for (size_t i = 0; i < len; ++i) if (proof[i] >= bb::P) return 13;
    Transcript ch;
    for (uint32_t h : hdr) ch.observe(bb::to_monty(h % bb::P));

    const Digest t_root = get_digest();
    ch.observe_n(t_root.w, 8);
    Ext al = bb::ext_zero(), bl = bb::ext_zero();
    Digest p_root{};
    std::vector<Ext> S(A, bb::ext_zero());
    if (lg) {
        al = ch.sample_ext();
        bl = ch.sample_ext();
        p_root = get_digest();
        ch.observe_n(p_root.w, 8);
        for (size_t a = 0; a < A; ++a) { S[a] = get_ext(); ch.observe_n(S[a].c, 4); }
    }
    const Ext alpha = ch.sample_ext();
    const Digest q_root = get_digest();
    ch.observe_n(q_root.w, 8);
    const Ext zeta = ch.sample_ext();
    std::vector<Ext> opened(K_total);
    for (auto& e : opened) { e = get_ext(); ch.observe_n(e.c, 4); }
    if (short_read) return 10;

    // constraint identities at zeta, AIR by AIR (per-AIR layout: main | perm at zeta | quotient | perm at g zeta)
    std::vector<Ext> gzeta(A);
    for (size_t a = 0; a < A; ++a) {
        const PwAirDescription& d = airs[a];
        const SegShapeV& s = sh[a];
        const Ext* o = &opened[s.koff];
        const size_t K1 = (size_t)s.W + s.Wp + 8;
        const uint32_t g_h = pw::field::root_of_unity((int)s.log_h), g_inv = bb::inv(g_h);
        gzeta[a] = bb::ext_scale(zeta, g_h);
        auto combine = [&](size_t base) {
            Ext r = bb::ext_zero();
            for (int k = 0; k < 4; ++k) {
                Ext basis = bb::ext_zero();
                basis.c[k] = bb::R_MOD_P;
                r = bb::ext_add(r, bb::ext_mul(basis, o[base + k]));
            }
            return r;
        };
        Ext acc = bb::ext_zero();
        for (size_t k = 0; k < d.n_constraints; ++k) {
            Ext v;
            if (!eval_ext(d.cons_bytecode + d.cons_spans[2 * k], d.cons_spans[2 * k + 1], o, s.W, v)) return 15;
            acc = bb::ext_add(bb::ext_mul(acc, alpha), v);
        }
        const Ext zH = bb::ext_pow(zeta, s.H);
        const Ext zh = bb::ext_sub(zH, bb::ext_one());
        if (lg) {
            std::vector<Ext> blpow(s.max_args + 2);
            { Ext b = bb::ext_one(); for (auto& x : blpow) { x = b; b = bb::ext_mul(b, bl); } }
            Ext sumq = bb::ext_zero(), sumq_next = bb::ext_zero();
            for (size_t g = 0; g < s.n_g; ++g) {
                Ext num = bb::ext_zero(), den = bb::ext_one();
                for (size_t i = s.gstarts[g]; i < s.gstarts[g + 1]; ++i) {
                    const uint32_t bus = d.interactions[3 * i], na = d.interactions[3 * i + 1];
                    const uint32_t* sp = d.inter_spans + 2 * (size_t)d.interactions[3 * i + 2];
                    Ext dd = bb::ext_add(al, bb::ext_from_base(bb::to_monty(bus % bb::P))), m, arg;
                    for (uint32_t j = 0; j < na; ++j) {
                        if (!eval_ext(d.inter_bytecode + sp[2 + 2 * j], sp[3 + 2 * j], o, s.W, arg)) return 15;
                        dd = bb::ext_add(dd, bb::ext_mul(blpow[j + 1], arg));
                    }
                    if (!eval_ext(d.inter_bytecode + sp[0], sp[1], o, s.W, m)) return 15;
                    num = bb::ext_add(bb::ext_mul(num, dd), bb::ext_mul(den, m));
                    den = bb::ext_mul(den, dd);
                }
                const Ext qi = combine(s.W + 4 * g), qn = combine(K1 + 4 * g);
                sumq = bb::ext_add(sumq, qi);
                sumq_next = bb::ext_add(sumq_next, qn);
                acc = bb::ext_add(bb::ext_mul(acc, alpha), bb::ext_sub(bb::ext_mul(qi, den), num));
            }
            const Ext phi = combine(s.W + 4 * s.n_g), phin = combine(K1 + 4 * s.n_g);
            const Ext is_trans = bb::ext_sub(zeta, bb::ext_from_base(g_inv));
            const Ext is_first = bb::ext_mul(zh, bb::ext_inv(bb::ext_sub(zeta, bb::ext_one())));
            const Ext is_last = bb::ext_mul(zh, bb::ext_inv(is_trans));
            acc = bb::ext_add(bb::ext_mul(acc, alpha), bb::ext_mul(is_first, bb::ext_sub(phi, sumq)));
            acc = bb::ext_add(bb::ext_mul(acc, alpha), bb::ext_mul(is_trans, bb::ext_sub(bb::ext_sub(phin, phi), sumq_next)));
            acc = bb::ext_add(bb::ext_mul(acc, alpha), bb::ext_mul(is_last, bb::ext_sub(phi, S[a])));
        }
        const Ext qlo = combine(s.W + s.Wp), qhi = combine(s.W + s.Wp + 4);
        if (!bb::ext_eq(acc, bb::ext_mul(zh, bb::ext_add(qlo, bb::ext_mul(zH, qhi))))) return (int)((a + 1) << 8) | 2;
    }

    const Ext gamma = ch.sample_ext();
    std::vector<Ext> gpow(K_total), sum1(A, bb::ext_zero()), sum2(A, bb::ext_zero());
    { Ext g = bb::ext_one(); for (auto& x : gpow) { x = g; g = bb::ext_mul(g, gamma); } }
    for (size_t a = 0; a < A; ++a) {
        const size_t K1 = (size_t)sh[a].W + sh[a].Wp + 8;
        for (size_t k = 0; k < K1; ++k) sum1[a] = bb::ext_add(sum1[a], bb::ext_mul(gpow[sh[a].koff + k], opened[sh[a].koff + k]));
        for (size_t k = K1; k < sh[a].K; ++k) sum2[a] = bb::ext_add(sum2[a], bb::ext_mul(gpow[sh[a].koff + k], opened[sh[a].koff + k]));
    }
    const int rounds = L - 1;
    std::vector<Digest> fri_roots(rounds);
    std::vector<Ext> betas(rounds);
    for (int l = 0; l < rounds; ++l) {
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

    // one opening of a mixed-height tree: rows[a] = AIR a's row (Montgomery), the siblings follow in the proof
    std::vector<std::vector<uint32_t>> trow(A), prow(A), qrow(A);
    auto check_mixed = [&](const std::vector<std::vector<uint32_t>>& rows, size_t q, const Digest& root) {
        auto level_hash = [&](size_t n, bool& any) {
            std::vector<uint32_t> cat;
            any = false;
            for (size_t a = 0; a < A; ++a) if (sh[a].N == n) { any = true; cat.insert(cat.end(), rows[a].begin(), rows[a].end()); }
            return hash_row(cat.data(), cat.size());
        };
        bool any;
        Digest cur = level_hash(Nmax, any);
        size_t p = q;
        for (size_t n = Nmax / 2; n >= 1; n >>= 1) {
            const Digest sib = get_digest();
            cur = (p & n) ? compress(sib, cur) : compress(cur, sib);
            p &= n - 1;
            const Digest inj = level_hash(n, any);
            if (any) cur = compress(cur, inj);
        }
        return same(cur, root);
    };
    auto check_path = [&](Digest leaf, size_t idx, int depth, const Digest& root) {
        for (int l = 0; l < depth; ++l) {
            const Digest sib = get_digest();
            leaf = ((idx >> l) & 1) ? compress(sib, leaf) : compress(leaf, sib);
        }
        return same(leaf, root);
    };
    const uint32_t shift0 = bb::to_monty(pw::field::kCosetShift);
    const uint32_t inv2 = bb::inv(bb::to_monty(2));
    for (uint32_t qi = 0; qi < cfg->num_queries; ++qi) {
        const size_t q = ch.sample_bits(L);
        if (get() != q) return short_read ? 10 : 4;
        for (size_t a = 0; a < A; ++a) { trow[a].resize(sh[a].W); for (auto& w : trow[a]) w = get_m(); }
        if (short_read) return 10;
        if (!check_mixed(trow, q, t_root)) return short_read ? 10 : 5;
        if (lg) {
            for (size_t a = 0; a < A; ++a) { prow[a].resize(sh[a].Wp); for (auto& w : prow[a]) w = get_m(); }
            if (!check_mixed(prow, q, p_root)) return short_read ? 10 : 11;
        }
        for (size_t a = 0; a < A; ++a) { qrow[a].resize(8); for (auto& w : qrow[a]) w = get_m(); }
        if (!check_mixed(qrow, q, q_root)) return short_read ? 10 : 6;
        // reduced opening of the AIRs of height 2^logn at index q mod 2^logn
        auto ro_at = [&](int logn, bool& any) {
            Ext r = bb::ext_zero();
            any = false;
            for (size_t a = 0; a < A; ++a) {
                if (sh[a].logN != logn) continue;
                any = true;
                const SegShapeV& s = sh[a];
                const size_t K1 = (size_t)s.W + s.Wp + 8;
                const Ext* gp = &gpow[s.koff];
                const uint32_t x = bb::mul(shift0, bb::pow_u32(pw::field::root_of_unity(logn), (uint32_t)(q & (s.N - 1))));
                Ext a1 = bb::ext_zero(), a2 = bb::ext_zero();
                for (size_t k = 0; k < s.W; ++k) a1 = bb::ext_add(a1, bb::ext_scale(gp[k], trow[a][k]));
                for (size_t k = 0; k < s.Wp; ++k) {
                    a1 = bb::ext_add(a1, bb::ext_scale(gp[s.W + k], prow[a][k]));
                    a2 = bb::ext_add(a2, bb::ext_scale(gp[K1 + k], prow[a][k]));
                }
                for (size_t k = 0; k < 8; ++k) a1 = bb::ext_add(a1, bb::ext_scale(gp[s.W + s.Wp + k], qrow[a][k]));
                Ext t = bb::ext_mul(bb::ext_sub(a1, sum1[a]), bb::ext_inv(bb::ext_sub(bb::ext_from_base(x), zeta)));
                if (lg) t = bb::ext_add(t, bb::ext_mul(bb::ext_sub(a2, sum2[a]), bb::ext_inv(bb::ext_sub(bb::ext_from_base(x), gzeta[a]))));
                r = bb::ext_add(r, t);
            }
            return r;
        };
        bool any;
        Ext cur = ro_at(L, any);
        for (int l = 0; l < rounds; ++l) {
            const size_t Nl = Nmax >> l, half = Nl / 2, p = q & (Nl - 1);
            const Ext sib = get_ext();
            const Ext lo = p < half ? cur : sib, hi = p < half ? sib : cur;
            uint32_t row[8];
            memcpy(row, lo.c, 16);
            memcpy(row + 4, hi.c, 16);
            if (!check_path(hash_row(row, 8), p & (half - 1), L - 1 - l, fri_roots[l])) return short_read ? 10 : 7;
            // unshifted subgroup: x_i = w_l^i
            const uint32_t xi = bb::pow_u32(pw::field::root_of_unity(L - l), (uint32_t)(p & (half - 1)));
            const Ext s2 = bb::ext_scale(bb::ext_add(lo, hi), inv2);
            const Ext dd = bb::ext_scale(bb::ext_sub(lo, hi), bb::mul(inv2, bb::inv(xi)));
            cur = bb::ext_add(s2, bb::ext_mul(betas[l], dd));
            const Ext roll = ro_at(L - l - 1, any);
            if (any) cur = bb::ext_add(cur, bb::ext_mul(bb::ext_mul(betas[l], betas[l]), roll));
        }
        if (short_read) return 10;
        if (!bb::ext_eq(cur, final_poly)) return 8;
    }
    if (short_read) return 10;
    if (pos != len) return 9;
    Ext total = bb::ext_zero();
    for (size_t a = 0; a < A; ++a) total = bb::ext_add(total, S[a]);
    if (total_sum4) for (int k = 0; k < 4; ++k) total_sum4[k] = bb::from_monty(total.c[k]);
    if (lg && check_balance && !bb::ext_eq(total, bb::ext_zero())) return 14;
    return 0;
}

// Boundaries of the LogUp groups the prover and the verifier derive from an interaction table (logup_groups.hpp):
// writes up to `cap` entries, returns the number of entries (n_groups + 1), 0 if the table is malformed.
extern "C" size_t pw_logup_group_starts(const uint32_t* interactions, size_t n_interactions, const uint32_t* inter_spans,
                                        size_t n_inter_spans, const uint32_t* inter_bytecode, size_t inter_bytecode_len,
                                        uint32_t* out, size_t cap) {
    for (size_t i = 0; i < n_interactions; ++i) {
        const uint32_t na = interactions[3 * i + 1], first = interactions[3 * i + 2];
        if ((size_t)first + 1 + na > n_inter_spans) return 0;
        for (uint32_t k = 0; k <= na; ++k)
            if ((size_t)inter_spans[2 * (first + k)] + inter_spans[2 * (first + k) + 1] > inter_bytecode_len) return 0;
    }
    const std::vector<uint32_t> g = pw::logup_group_starts(interactions, n_interactions, inter_spans, inter_bytecode);
    for (size_t i = 0; i < g.size() && i < cap; ++i) out[i] = g[i];
    return g.size();
}
