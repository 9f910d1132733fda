/*
 * TEST INFRASTRUCTURE — CPU oracle of the STARK proving path ("pw-stark v0").
 * Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call this file.
 *
 * PARITY STATUS: **parity unpinned** against the reference prover. The prover
 * powdr drives (openvm-stark-backend / openvm-cpu-backend / Plonky3, pinned at
 * /root/reference/Cargo.toml:51-100) is an un-vendored git dependency: no
 * source, no toolchain, no golden proofs here. This file
 * therefore restates the *published* univariate pipeline BASELINE.json's
 * north_star names — radix-2 NTT low-degree extension, constraint/quotient
 * evaluation on the extended domain, Poseidon2 Merkle commitment, FRI — in the
 * shape of Plonky3's uni-stark + two-adic FRI PCS, and is the definition of
 * "pw-stark v0" the HIP prover must match byte for byte. Call sites it stands
 * in for: openvm-riscv/src/lib.rs:327-341 (prove + verify_app_proof),
 * openvm/src/trace_generation.rs:97-139; the AIR it proves is PowdrAir::eval,
 * openvm/src/powdr_extension/chip.rs:94-130 (current-row constraints only, no
 * public values, no preprocessed trace). Conventions marked (†) are recalled
 * Plonky3 conventions that could not be read here; every such constant sits in
 * one table (poseidon2_constants below) so it can be swapped.
 *
 * Arithmetic: canonical representatives, plain `%` (shares no code and no
 * representation with the Montgomery device code).
 *
 * Protocol (all vectors in NATURAL order):
 *   trace T: W columns x H=2^n rows, column-major.
 *   g_n = generator of the order-2^n subgroup (TWO_ADIC_GEN^(2^(27-n))), s = 31 (coset shift)
 *   LDE  L_c[j] = T_c(s * g_{n+1}^j), j < N = 2H            (blow-up 2)
 *   commit(M): Merkle tree, leaf j = Poseidon2 sponge (rate 8, overwrite mode) of row j,
 *              inner node = first 8 words of Poseidon2(left || right)
 *   transcript: duplex sponge challenger over the same permutation (see Challenger)
 *   alpha <- transcript;  Q(x) = sum_j alpha^(nc-1-j) C_j(T(x)) / (x^H - 1) on the coset,
 *   Q = Q_lo + X^H Q_hi (coefficients), the 8 base-field coordinate columns
 *   (lo.c0..c3, hi.c0..c3) are LDE'd and committed like the trace.
 *   zeta <- transcript; open all W + 8 polynomials at zeta; gamma <- transcript;
 *   v_0[j] = sum_k gamma^k (f_k(x_j) - f_k(zeta)) / (x_j - zeta)
 *   FRI: layer l commits pairs (v_l[i], v_l[i + N_l/2]); beta_l <- transcript;
 *        v_{l+1}[i] = (a+b)/2 + beta_l (a-b)/(2 x_i); after n folds v_n has 2 equal entries.
 *   queries: index <- transcript (n+1 bits); open trace row, quotient row, FRI siblings + paths.
 */
#include <cstdint>
#include <cstring>
#include <vector>
#include <array>
#include <cstdio>
#include <stdexcept>

extern "C" {
#include "babybear.h"
}

namespace {

using u32 = uint32_t;
using u64 = uint64_t;
constexpr u32 P = OR_P;
constexpr u32 TWO_ADIC_GEN = 0x1a427a41u; /* 31^15, order 2^27 (checked in tests) (†) */
constexpr u32 COSET_SHIFT = 31u;          /* multiplicative generator (†) */
constexpr u32 EXT_W = 11u;                /* F[X]/(X^4 - 11) (†) */

/* ---------------------------------------------------------------- extension field */
struct Ext { u32 c[4]; };
inline Ext ext_zero() { return {{0, 0, 0, 0}}; }
inline Ext ext_one() { return {{1, 0, 0, 0}}; }
inline Ext ext_from(u32 a) { return {{a, 0, 0, 0}}; }
inline Ext ext_add(const Ext& a, const Ext& b) { Ext r; for (int i = 0; i < 4; ++i) r.c[i] = or_add(a.c[i], b.c[i]); return r; }
inline Ext ext_sub(const Ext& a, const Ext& b) { Ext r; for (int i = 0; i < 4; ++i) r.c[i] = or_sub(a.c[i], b.c[i]); return r; }
inline Ext ext_neg(const Ext& a) { Ext r; for (int i = 0; i < 4; ++i) r.c[i] = or_neg(a.c[i]); return r; }
inline Ext ext_scale(const Ext& a, u32 k) { Ext r; for (int i = 0; i < 4; ++i) r.c[i] = or_mul(a.c[i], k); return r; }
inline Ext ext_mul(const Ext& a, const Ext& b) {
    /* schoolbook product then reduce X^4 = 11 */
    u32 t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t[i + j] = or_add(t[i + j], or_mul(a.c[i], b.c[j]));
    Ext r;
    for (int i = 0; i < 4; ++i) r.c[i] = t[i];
    for (int i = 4; i < 7; ++i) r.c[i - 4] = or_add(r.c[i - 4], or_mul(EXT_W, t[i]));
    return r;
}
inline bool ext_eq(const Ext& a, const Ext& b) { return !memcmp(a.c, b.c, sizeof a.c); }
/* inverse through the tower F -> K = F[Y]/(Y^2-11) -> E = K[X]/(X^2-Y) */
Ext ext_inv(const Ext& a) {
    /* A0 = (a0, a2), A1 = (a1, a3) in K; a = A0 + A1 X */
    auto kmul = [](const u32* x, const u32* y, u32* o) {
        u32 o0 = or_add(or_mul(x[0], y[0]), or_mul(EXT_W, or_mul(x[1], y[1])));
        u32 o1 = or_add(or_mul(x[0], y[1]), or_mul(x[1], y[0]));
        o[0] = o0; o[1] = o1;
    };
    u32 A0[2] = {a.c[0], a.c[2]}, A1[2] = {a.c[1], a.c[3]};
    u32 A0sq[2], A1sq[2];
    kmul(A0, A0, A0sq);
    kmul(A1, A1, A1sq);
    /* Y * A1sq = (11*A1sq[1], A1sq[0]) */
    u32 D[2] = {or_sub(A0sq[0], or_mul(EXT_W, A1sq[1])), or_sub(A0sq[1], A1sq[0])};
    /* D^-1 in K: (u - vY) / (u^2 - 11 v^2) */
    u32 nrm = or_sub(or_mul(D[0], D[0]), or_mul(EXT_W, or_mul(D[1], D[1])));
    u32 ni = or_inv(nrm);
    u32 Di[2] = {or_mul(D[0], ni), or_mul(or_neg(D[1]), ni)};
    u32 R0[2], R1[2], nA1[2] = {or_neg(A1[0]), or_neg(A1[1])};
    kmul(A0, Di, R0);
    kmul(nA1, Di, R1);
    return {{R0[0], R1[0], R0[1], R1[1]}};
}
Ext ext_pow(Ext a, u64 e) { Ext r = ext_one(); while (e) { if (e & 1) r = ext_mul(r, a); a = ext_mul(a, a); e >>= 1; } return r; }

/* ---------------------------------------------------------------- Poseidon2 (width 16) */
/* Shape (†): x^7 S-box, 8 external + 13 internal rounds, external layer circ(2 M4, M4, M4, M4)
 * with M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]], internal layer 1*1^T + diag(mu).
 * Round constants are EXTERNAL to the reference checkout; this table is generated by
 * splitmix64(seed "Poseidon") with rejection sampling and is the single swappable table. */
struct Poseidon2Constants {
    u32 ext_rc[8][16];
    u32 int_rc[13];
    u32 diag[16];
    Poseidon2Constants() {
        u64 s = 0x506F736569646F6Eull;
        auto next = [&]() {
            for (;;) {
                s += 0x9E3779B97F4A7C15ull;
                u64 z = s;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z ^= z >> 31;
                u32 v = (u32)(z & 0x7fffffffu);
                if (v < P) return v;
            }
        };
        for (auto& r : ext_rc) for (auto& c : r) c = next();
        for (auto& c : int_rc) c = next();
        /* (†) [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27] */
        auto inv2k = [](int k) { return or_inv(or_pow(2, (u64)k)); };
        u32 d[16] = {or_neg(2), 1, 2, inv2k(1), 3, 4, or_neg(inv2k(1)), or_neg(3), or_neg(4), inv2k(8),
                     inv2k(2), inv2k(3), inv2k(27), or_neg(inv2k(8)), or_neg(inv2k(4)), or_neg(inv2k(27))};
        memcpy(diag, d, sizeof d);
    }
};
Poseidon2Constants& p2c_mut() { static Poseidon2Constants c; return c; }
const Poseidon2Constants& p2c() { return p2c_mut(); }

inline u32 sbox7(u32 x) { u32 x2 = or_mul(x, x), x3 = or_mul(x2, x), x4 = or_mul(x2, x2); return or_mul(x3, x4); }
void external_layer(u32* s) {
    static const u32 M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    u32 t[16];
    for (int b = 0; b < 4; ++b)
        for (int i = 0; i < 4; ++i) {
            u32 acc = 0;
            for (int j = 0; j < 4; ++j) acc = or_add(acc, or_mul(M4[i][j], s[4 * b + j]));
            t[4 * b + i] = acc;
        }
    for (int i = 0; i < 4; ++i) {
        u32 col = 0;
        for (int b = 0; b < 4; ++b) col = or_add(col, t[4 * b + i]);
        for (int b = 0; b < 4; ++b) s[4 * b + i] = or_add(t[4 * b + i], col);
    }
}
void poseidon2(u32* s) {
    const auto& C = p2c();
    external_layer(s);
    for (int r = 0; r < 4; ++r) {
        for (int i = 0; i < 16; ++i) s[i] = sbox7(or_add(s[i], C.ext_rc[r][i]));
        external_layer(s);
    }
    for (int r = 0; r < 13; ++r) {
        s[0] = sbox7(or_add(s[0], C.int_rc[r]));
        u32 sum = 0;
        for (int i = 0; i < 16; ++i) sum = or_add(sum, s[i]);
        for (int i = 0; i < 16; ++i) s[i] = or_add(sum, or_mul(C.diag[i], s[i]));
    }
    for (int r = 4; r < 8; ++r) {
        for (int i = 0; i < 16; ++i) s[i] = sbox7(or_add(s[i], C.ext_rc[r][i]));
        external_layer(s);
    }
}

using Digest = std::array<u32, 8>;
/* padding-free overwrite-mode sponge, rate 8 (Plonky3 PaddingFreeSponge shape (†)) */
Digest hash_row(const u32* row, size_t len) {
    u32 st[16] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < k; ++i) st[i] = row[off + i];
        poseidon2(st);
    }
    Digest d;
    memcpy(d.data(), st, 32);
    return d;
}
Digest compress(const Digest& l, const Digest& r) {
    u32 st[16];
    memcpy(st, l.data(), 32);
    memcpy(st + 8, r.data(), 32);
    poseidon2(st);
    Digest d;
    memcpy(d.data(), st, 32);
    return d;
}

/* Merkle tree over `n_leaves` (power of two) leaf digests. layers[0] = leaves, back() = root. */
struct Merkle {
    std::vector<std::vector<Digest>> layers;
    void build(std::vector<Digest> leaves) {
        layers.clear();
        layers.push_back(std::move(leaves));
        while (layers.back().size() > 1) {
            const auto& lo = layers.back();
            std::vector<Digest> up(lo.size() / 2);
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)up.size(); ++i) up[i] = compress(lo[2 * i], lo[2 * i + 1]);
            layers.push_back(std::move(up));
        }
    }
    const Digest& root() const { return layers.back()[0]; }
    void path(size_t idx, std::vector<u32>& out) const {
        for (size_t l = 0; l + 1 < layers.size(); ++l) {
            const Digest& sib = layers[l][(idx >> l) ^ 1];
            out.insert(out.end(), sib.begin(), sib.end());
        }
    }
};
/* commit a column-major matrix: leaf j = hash of row j */
void commit_matrix(const u32* m, size_t height, size_t width, Merkle& t) {
    std::vector<Digest> leaves(height);
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)height; ++j) {
        std::vector<u32> row(width);
        for (size_t c = 0; c < width; ++c) row[c] = m[c * height + j];
        leaves[j] = hash_row(row.data(), width);
    }
    t.build(std::move(leaves));
}

/* ---------------------------------------------------------------- challenger */
struct Challenger {
    u32 st[16] = {0};
    std::vector<u32> in, out;
    void duplex() {
        for (size_t i = 0; i < in.size(); ++i) st[i] = in[i];
        in.clear();
        poseidon2(st);
        out.assign(st, st + 8);
    }
    void observe(u32 x) { out.clear(); in.push_back(x); if (in.size() == 8) duplex(); }
    void observe_digest(const Digest& d) { for (u32 x : d) observe(x); }
    void observe_ext(const Ext& e) { for (u32 x : e.c) observe(x); }
    u32 sample() { if (!in.empty() || out.empty()) duplex(); u32 v = out.back(); out.pop_back(); return v; }
    Ext sample_ext() { Ext e; for (int i = 0; i < 4; ++i) e.c[i] = sample(); return e; }
    u32 sample_bits(int b) { return sample() & ((1u << b) - 1u); }
};

/* ---------------------------------------------------------------- NTT (textbook, natural order) */
u32 root_of_unity(int log_n) { return or_pow(TWO_ADIC_GEN, 1ull << (27 - log_n)); }
inline size_t bitrev(size_t x, int bits) { size_t r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r; }
/* in-place DFT: a[j] <- sum_i a[i] w^(ij) */
void dft(u32* a, int log_n, u32 w) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) { size_t j = bitrev(i, log_n); if (i < j) std::swap(a[i], a[j]); }
    for (int s = 1; s <= log_n; ++s) {
        size_t m = (size_t)1 << s, half = m >> 1;
        u32 wm = or_pow(w, n >> s);
        for (size_t k = 0; k < n; k += m) {
            u32 t = 1;
            for (size_t j = 0; j < half; ++j) {
                u32 u = a[k + j], v = or_mul(a[k + j + half], t);
                a[k + j] = or_add(u, v);
                a[k + j + half] = or_sub(u, v);
                t = or_mul(t, wm);
            }
        }
    }
}
void idft(u32* a, int log_n) {
    dft(a, log_n, or_inv(root_of_unity(log_n)));
    u32 ninv = or_inv((u32)(((u64)1 << log_n) % P));
    for (size_t i = 0; i < ((size_t)1 << log_n); ++i) a[i] = or_mul(a[i], ninv);
}
/* evaluations of the polynomial with `n` coefficients on s*<g_{log_n + 1}> (2n points) */
void coset_lde_from_coeffs(const u32* coef, int log_n, u32* out) {
    size_t n = (size_t)1 << log_n;
    u32 sp = 1;
    for (size_t i = 0; i < n; ++i) { out[i] = or_mul(coef[i], sp); sp = or_mul(sp, COSET_SHIFT); }
    for (size_t i = n; i < 2 * n; ++i) out[i] = 0;
    dft(out, log_n + 1, root_of_unity(log_n + 1));
}
void lde_column(const u32* col, int log_n, u32* out) {
    size_t n = (size_t)1 << log_n;
    std::vector<u32> c(col, col + n);
    idft(c.data(), log_n);
    coset_lde_from_coeffs(c.data(), log_n, out);
}

/* ---------------------------------------------------------------- constraints */
/* post-fix programs over the current row; PUSH operand = column index (same opcodes as the
 * trace-generation bytecode, openvm/src/cuda_abi.rs:137-147, minus INV_OR_ZERO) */
enum { OP_PUSH_COL = 0, OP_PUSH_CONST = 1, OP_ADD = 2, OP_SUB = 3, OP_MUL = 4, OP_NEG = 5 };
struct Program { const u32* bc; const u32* spans; size_t n; };

u32 eval_base(const u32* bc, u32 len, const u32* m, size_t stride, size_t row) {
    u32 st[16]; int sp = 0;
    for (u32 ip = 0; ip < len;) {
        u32 op = bc[ip++];
        if (op == OP_PUSH_COL) st[sp++] = m[(size_t)bc[ip++] * stride + row];
        else if (op == OP_PUSH_CONST) st[sp++] = bc[ip++] % P;
        else if (op == OP_NEG) st[sp - 1] = or_neg(st[sp - 1]);
        else { u32 b = st[--sp], a = st[--sp]; st[sp++] = op == OP_ADD ? or_add(a, b) : op == OP_SUB ? or_sub(a, b) : or_mul(a, b); }
    }
    return st[0];
}
Ext eval_ext(const u32* bc, u32 len, const Ext* vals) {
    Ext st[16]; int sp = 0;
    for (u32 ip = 0; ip < len;) {
        u32 op = bc[ip++];
        if (op == OP_PUSH_COL) st[sp++] = vals[bc[ip++]];
        else if (op == OP_PUSH_CONST) st[sp++] = ext_from(bc[ip++] % P);
        else if (op == OP_NEG) st[sp - 1] = ext_neg(st[sp - 1]);
        else { Ext b = st[--sp], a = st[--sp]; st[sp++] = op == OP_ADD ? ext_add(a, b) : op == OP_SUB ? ext_sub(a, b) : ext_mul(a, b); }
    }
    return st[0];
}

/* ---------------------------------------------------------------- proof layout */
struct Config { u32 num_queries; u32 pow_bits; };
constexpr u32 MAGIC = 0x31535750u; /* "PWS1" */

struct Writer {
    std::vector<u32> w;
    void put(u32 x) { w.push_back(x); }
    void put(const Digest& d) { w.insert(w.end(), d.begin(), d.end()); }
    void put(const Ext& e) { w.insert(w.end(), e.c, e.c + 4); }
};

void observe_instance(Challenger& ch, u32 log_h, u32 width, u32 n_constraints, const Config& cfg) {
    ch.observe(MAGIC % P); ch.observe(log_h); ch.observe(width); ch.observe(n_constraints);
    ch.observe(cfg.num_queries); ch.observe(cfg.pow_bits);
}

Ext eval_poly_at(const u32* coef, size_t n, const Ext& z) { /* Horner */
    Ext acc = ext_zero();
    for (size_t i = n; i-- > 0;) acc = ext_add(ext_mul(acc, z), ext_from(coef[i]));
    return acc;
}

std::vector<u32> prove(const Config& cfg, const u32* trace, u32 width, u32 log_h, const Program& prog) {
    const size_t H = (size_t)1 << log_h, N = 2 * H;
    const int logN = (int)log_h + 1;
    Writer pf;
    Challenger ch;
    observe_instance(ch, log_h, width, (u32)prog.n, cfg);
    pf.put(MAGIC); pf.put(log_h); pf.put(width); pf.put((u32)prog.n); pf.put(cfg.num_queries); pf.put(cfg.pow_bits);

    /* 1. coefficients + LDE + commitment of the trace */
    std::vector<u32> coef((size_t)width * H), lde((size_t)width * N);
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)width; ++c) {
        memcpy(&coef[c * H], trace + c * H, H * 4);
        idft(&coef[c * H], (int)log_h);
        coset_lde_from_coeffs(&coef[c * H], (int)log_h, &lde[c * N]);
    }
    Merkle t_tree;
    commit_matrix(lde.data(), N, width, t_tree);
    pf.put(t_tree.root());
    ch.observe_digest(t_tree.root());

    /* 2. quotient */
    Ext alpha = ch.sample_ext();
    std::vector<Ext> apow(prog.n);
    { Ext a = ext_one(); for (size_t j = prog.n; j-- > 0;) { apow[j] = a; a = ext_mul(a, alpha); } }
    u32 sH = or_pow(COSET_SHIFT, H);
    u32 zinv[2] = {or_inv(or_sub(sH, 1)), or_inv(or_sub(or_neg(sH), 1))}; /* x^H = +-s^H */
    std::vector<u32> q(4 * N); /* 4 coordinate columns */
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)N; ++j) {
        Ext acc = ext_zero();
        for (size_t k = 0; k < prog.n; ++k) {
            u32 v = eval_base(prog.bc + prog.spans[2 * k], prog.spans[2 * k + 1], lde.data(), N, (size_t)j);
            acc = ext_add(acc, ext_scale(apow[k], v));
        }
        acc = ext_scale(acc, zinv[j & 1]);
        for (int k = 0; k < 4; ++k) q[k * N + j] = acc.c[k];
    }
    /* coefficients of Q on the coset, split at H, LDE of the 8 chunk columns */
    std::vector<u32> qcoef(8 * H), qlde(8 * N);
    u32 sinv = or_inv(COSET_SHIFT);
    for (int k = 0; k < 4; ++k) {
        idft(&q[k * N], logN);
        u32 sp = 1;
        for (size_t i = 0; i < N; ++i) { q[k * N + i] = or_mul(q[k * N + i], sp); sp = or_mul(sp, sinv); }
        memcpy(&qcoef[(size_t)k * H], &q[k * N], H * 4);           /* lo.c_k */
        memcpy(&qcoef[(size_t)(4 + k) * H], &q[k * N + H], H * 4); /* hi.c_k */
    }
    for (int c = 0; c < 8; ++c) coset_lde_from_coeffs(&qcoef[(size_t)c * H], (int)log_h, &qlde[(size_t)c * N]);
    Merkle q_tree;
    commit_matrix(qlde.data(), N, 8, q_tree);
    pf.put(q_tree.root());
    ch.observe_digest(q_tree.root());

    /* 3. openings at zeta */
    Ext zeta = ch.sample_ext();
    std::vector<Ext> opened(width + 8);
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)width; ++c) opened[c] = eval_poly_at(&coef[c * H], H, zeta);
    for (int c = 0; c < 8; ++c) opened[width + c] = eval_poly_at(&qcoef[(size_t)c * H], H, zeta);
    for (auto& e : opened) { pf.put(e); ch.observe_ext(e); }

    /* 4. reduced opening vector v_0 */
    Ext gamma = ch.sample_ext();
    const size_t K = width + 8;
    std::vector<Ext> gpow(K);
    { Ext g = ext_one(); for (size_t k = 0; k < K; ++k) { gpow[k] = g; g = ext_mul(g, gamma); } }
    Ext opened_sum = ext_zero();
    for (size_t k = 0; k < K; ++k) opened_sum = ext_add(opened_sum, ext_mul(gpow[k], opened[k]));
    u32 wN = root_of_unity(logN);
    std::vector<u32> xs(N);
    { u32 x = COSET_SHIFT; for (size_t j = 0; j < N; ++j) { xs[j] = x; x = or_mul(x, wN); } }
    std::vector<Ext> v(N);
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)N; ++j) {
        Ext acc = ext_zero();
        for (size_t k = 0; k < width; ++k) acc = ext_add(acc, ext_scale(gpow[k], lde[k * N + j]));
        for (size_t k = 0; k < 8; ++k) acc = ext_add(acc, ext_scale(gpow[width + k], qlde[k * N + j]));
        Ext den = ext_sub(ext_from(xs[j]), zeta);
        v[j] = ext_mul(ext_sub(acc, opened_sum), ext_inv(den));
    }

    /* 5. FRI commit phase */
    std::vector<std::vector<Ext>> layers;
    std::vector<Merkle> fri_trees;
    u32 shift = COSET_SHIFT;
    u32 inv2 = or_inv(2);
    for (int l = 0; l < (int)log_h; ++l) {
        size_t Nl = N >> l, half = Nl / 2;
        std::vector<Digest> leaves(half);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)half; ++i) {
            u32 row[8];
            memcpy(row, v[i].c, 16);
            memcpy(row + 4, v[i + half].c, 16);
            leaves[i] = hash_row(row, 8);
        }
        Merkle t;
        t.build(std::move(leaves));
        pf.put(t.root());
        ch.observe_digest(t.root());
        Ext beta = ch.sample_ext();
        u32 wl = root_of_unity(logN - l);
        std::vector<Ext> nv(half);
        std::vector<u32> xinv(half);
        { u32 x = shift; for (size_t i = 0; i < half; ++i) { xinv[i] = x; x = or_mul(x, wl); } }
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)half; ++i) {
            Ext a = v[i], b = v[i + half];
            Ext s = ext_scale(ext_add(a, b), inv2);
            Ext d = ext_scale(ext_sub(a, b), or_mul(inv2, or_inv(xinv[i])));
            nv[i] = ext_add(s, ext_mul(beta, d));
        }
        layers.push_back(std::move(v));
        fri_trees.push_back(std::move(t));
        v = std::move(nv);
        shift = or_mul(shift, shift);
    }
    /* v has N >> log_h = 2 entries, equal for an honest prover */
    pf.put(v[0]);
    ch.observe_ext(v[0]);

    /* 6. proof of work */
    u32 witness = 0;
    if (cfg.pow_bits) {
        for (;; ++witness) {
            Challenger c2 = ch;
            c2.observe(witness);
            if (c2.sample_bits((int)cfg.pow_bits) == 0) break;
        }
    }
    pf.put(witness);
    ch.observe(witness);
    if (cfg.pow_bits) (void)ch.sample_bits((int)cfg.pow_bits);

    /* 7. queries */
    for (u32 qi = 0; qi < cfg.num_queries; ++qi) {
        size_t idx = ch.sample_bits(logN);
        pf.put((u32)idx);
        for (size_t c = 0; c < width; ++c) pf.put(lde[c * N + idx]);
        t_tree.path(idx, pf.w);
        for (size_t c = 0; c < 8; ++c) pf.put(qlde[c * N + idx]);
        q_tree.path(idx, pf.w);
        for (int l = 0; l < (int)log_h; ++l) {
            size_t Nl = N >> l, half = Nl / 2;
            size_t p = idx & (Nl - 1);
            pf.put(layers[l][p ^ half]);
            fri_trees[l].path(p & (half - 1), pf.w);
        }
    }
    return pf.w;
}

/* returns 0 if the proof verifies, otherwise a positive code naming the failed check */
int verify(const Config& cfg, const u32* proof, size_t len, u32 width, u32 log_h, const Program& prog) {
    const size_t H = (size_t)1 << log_h, N = 2 * H;
    const int logN = (int)log_h + 1;
    size_t pos = 0;
    auto need = [&](size_t k) { if (pos + k > len) throw std::runtime_error("short proof"); };
    auto get = [&]() { need(1); return proof[pos++]; };
    auto get_digest = [&]() { need(8); Digest d; memcpy(d.data(), proof + pos, 32); pos += 8; return d; };
    auto get_ext = [&]() { need(4); Ext e; memcpy(e.c, proof + pos, 16); pos += 4; return e; };
    try {
        if (get() != MAGIC || get() != log_h || get() != width || get() != prog.n || get() != cfg.num_queries || get() != cfg.pow_bits) return 1;
        for (size_t i = 0; i < len; ++i) if (proof[i] >= P) return 13; /* non-canonical word: a second encoding of an element */
        Challenger ch;
        observe_instance(ch, log_h, width, (u32)prog.n, cfg);
        Digest t_root = get_digest();
        ch.observe_digest(t_root);
        Ext alpha = ch.sample_ext();
        Digest q_root = get_digest();
        ch.observe_digest(q_root);
        Ext zeta = ch.sample_ext();
        const size_t K = width + 8;
        std::vector<Ext> opened(K);
        for (auto& e : opened) { e = get_ext(); ch.observe_ext(e); }
        /* constraint identity at zeta */
        Ext acc = ext_zero();
        for (size_t k = 0; k < prog.n; ++k)
            acc = ext_add(ext_mul(acc, alpha), eval_ext(prog.bc + prog.spans[2 * k], prog.spans[2 * k + 1], opened.data()));
        Ext zH = ext_pow(zeta, H);
        Ext zh = ext_sub(zH, ext_one());
        Ext basis[4] = {{{1, 0, 0, 0}}, {{0, 1, 0, 0}}, {{0, 0, 1, 0}}, {{0, 0, 0, 1}}};
        Ext qlo = ext_zero(), qhi = ext_zero();
        for (int k = 0; k < 4; ++k) {
            qlo = ext_add(qlo, ext_mul(basis[k], opened[width + k]));
            qhi = ext_add(qhi, ext_mul(basis[k], opened[width + 4 + k]));
        }
        Ext qz = ext_add(qlo, ext_mul(zH, qhi));
        if (!ext_eq(acc, ext_mul(zh, qz))) return 2;

        Ext gamma = ch.sample_ext();
        std::vector<Ext> gpow(K);
        { Ext g = ext_one(); for (size_t k = 0; k < K; ++k) { gpow[k] = g; g = ext_mul(g, gamma); } }
        Ext opened_sum = ext_zero();
        for (size_t k = 0; k < K; ++k) opened_sum = ext_add(opened_sum, ext_mul(gpow[k], opened[k]));
        std::vector<Digest> fri_roots(log_h);
        std::vector<Ext> betas(log_h);
        for (u32 l = 0; l < log_h; ++l) { fri_roots[l] = get_digest(); ch.observe_digest(fri_roots[l]); betas[l] = ch.sample_ext(); }
        Ext final_poly = get_ext();
        ch.observe_ext(final_poly);
        u32 witness = get();
        ch.observe(witness);
        if (cfg.pow_bits && ch.sample_bits((int)cfg.pow_bits) != 0) return 3;

        auto check_path = [&](Digest leaf, size_t idx, int depth, const Digest& root) {
            for (int l = 0; l < depth; ++l) {
                Digest sib = get_digest();
                leaf = ((idx >> l) & 1) ? compress(sib, leaf) : compress(leaf, sib);
            }
            return leaf == root;
        };
        u32 wN = root_of_unity(logN), inv2 = or_inv(2);
        for (u32 qi = 0; qi < cfg.num_queries; ++qi) {
            size_t idx = ch.sample_bits(logN);
            if (get() != idx) return 4;
            need(width);
            const u32* trow = proof + pos; pos += width;
            if (!check_path(hash_row(trow, width), idx, logN, t_root)) return 5;
            need(8);
            const u32* qrow = proof + pos; pos += 8;
            if (!check_path(hash_row(qrow, 8), idx, logN, q_root)) return 6;
            u32 x = or_mul(COSET_SHIFT, or_pow(wN, idx));
            Ext a = ext_zero();
            for (size_t k = 0; k < width; ++k) a = ext_add(a, ext_scale(gpow[k], trow[k]));
            for (size_t k = 0; k < 8; ++k) a = ext_add(a, ext_scale(gpow[width + k], qrow[k]));
            Ext cur = ext_mul(ext_sub(a, opened_sum), ext_inv(ext_sub(ext_from(x), zeta)));
            u32 shift = COSET_SHIFT;
            for (u32 l = 0; l < log_h; ++l) {
                size_t Nl = N >> l, half = Nl / 2, p = idx & (Nl - 1);
                Ext sib = get_ext();
                Ext lo = (p < half) ? cur : sib, hi = (p < half) ? sib : cur;
                u32 row[8];
                memcpy(row, lo.c, 16); memcpy(row + 4, hi.c, 16);
                if (!check_path(hash_row(row, 8), p & (half - 1), logN - 1 - (int)l, fri_roots[l])) return 7;
                u32 xi = or_mul(shift, or_pow(root_of_unity(logN - (int)l), p & (half - 1)));
                Ext s = ext_scale(ext_add(lo, hi), inv2);
                Ext d = ext_scale(ext_sub(lo, hi), or_mul(inv2, or_inv(xi)));
                cur = ext_add(s, ext_mul(betas[l], d));
                shift = or_mul(shift, shift);
            }
            if (!ext_eq(cur, final_poly)) return 8;
        }
        if (pos != len) return 9;
    } catch (const std::exception&) {
        return 10;
    }
    return 0;
}


/* ================================================================== pw-stark v0 + LogUp
 * Optional extension that also proves the AIR's bus interactions (PowdrAir::eval pushes them with
 * `builder.push_interaction(id, args, mult, 1)`, openvm/src/powdr_extension/chip.rs:117-129), in the
 * committed-column LogUp style of the OpenVM-1 generation of the backend (metrics `perm_cols`,
 * `generate_perm_trace_time_ms`, openvm/metrics-viewer/CLAUDE.md:62-73,114). Parity unpinned like the
 * rest of the prover. Differences from v0:
 *   after the trace root: al, bl <- transcript (extension field)
 *   interactions are packed, in order, into groups (the backend's "chunks"): a group is extended while its
 *        constraint below keeps degree <= 3 (group_starts); n_g groups
 *   perm matrix (4 (n_g + 1) base columns): for interaction i, d_i = al + bus_i + sum_j bl^(j+1) a_ij; for group g
 *        q_g = sum_{i in g} m_i / d_i  (one extension column),
 *        phi(row r) = sum_{r' <= r} sum_g q_g(r');  S = phi(last row) goes into the proof
 *   extra (extension-valued) constraints, folded after the base ones:
 *        q_g prod_{i in g} d_i - sum_{i in g} m_i prod_{j in g, j != i} d_j      (every row)
 *        is_first (phi - sum_g q_g)
 *        is_transition (phi' - phi - sum_g q_g')            (' = next row)
 *        is_last (phi - S)
 *     with is_first = Z_H/(x-1), is_last = Z_H/(x-g^-1), is_transition = x - g^-1
 *   openings: main, perm, quotient at zeta; perm also at g*zeta; DEEP with both points.
 */
constexpr u32 MAGIC2 = 0x32535750u; /* "PWS2" */

struct Interactions { const u32* inter; size_t n; const u32* spans; const u32* bc; };  /* inter = n x {bus, n_args, span index} */

struct LogupRow {  /* per-row evaluation shared by prover (base values) and verifier (ext values) */
    static Ext denom_base(const Interactions& I, size_t i, const u32* m, size_t stride, size_t row, const Ext& al, const Ext* blpow) {
        const u32* it = I.inter + 3 * i;
        const u32* sp = I.spans + 2 * (size_t)it[2];
        Ext d = ext_add(al, ext_from(it[0] % P));
        for (u32 j = 0; j < it[1]; ++j) {
            u32 a = eval_base(I.bc + sp[2 + 2 * j], sp[3 + 2 * j], m, stride, row);
            d = ext_add(d, ext_scale(blpow[j + 1], a));
        }
        return d;
    }
    static u32 mult_base(const Interactions& I, size_t i, const u32* m, size_t stride, size_t row) {
        const u32* it = I.inter + 3 * i;
        const u32* sp = I.spans + 2 * (size_t)it[2];
        return eval_base(I.bc + sp[0], sp[1], m, stride, row);
    }
};

size_t max_args(const Interactions& I) { size_t k = 0; for (size_t i = 0; i < I.n; ++i) if (I.inter[3 * i + 1] > k) k = I.inter[3 * i + 1]; return k; }

/* degree in the trace columns of a post-fix program */
int expr_degree(const u32* bc, u32 len) {
    int st[16], sp = 0;
    for (u32 ip = 0; ip < len;) {
        u32 op = bc[ip++];
        if (op == OP_PUSH_COL || op == OP_PUSH_CONST) { if (sp >= 16 || ip >= len) return 99; st[sp++] = op == OP_PUSH_COL ? 1 : 0; ++ip; }
        else if (op == OP_ADD || op == OP_SUB) { if (sp < 2) return 99; --sp; st[sp - 1] = st[sp - 1] > st[sp] ? st[sp - 1] : st[sp]; }
        else if (op == OP_MUL) { if (sp < 2) return 99; --sp; st[sp - 1] += st[sp]; }
        else if (op == OP_NEG) { if (sp < 1) return 99; }
        else return 99;
    }
    return sp == 1 ? st[0] : 99;
}

/* Group boundaries (n_g + 1 entries): interactions are taken in order; interaction i joins the current group while
 *   1 + sum_{j in g} deg d_j <= 3   and   deg m_j + sum_{k in g, k != j} deg d_k <= 3 for every member j,
 * i.e. while the group's constraint stays within the degree the blow-up-2 quotient can carry; otherwise it opens a
 * new group (a single interaction is always accepted: an over-degree one simply cannot be proven). */
std::vector<u32> group_starts(const Interactions& I) {
    std::vector<u32> starts{0};
    std::vector<int> dm, dd;
    int sum_den = 0;
    for (size_t i = 0; i < I.n; ++i) {
        const u32* it = I.inter + 3 * i;
        const u32* sp = I.spans + 2 * (size_t)it[2];
        int m = expr_degree(I.bc + sp[0], sp[1]), d = 0;
        for (u32 j = 0; j < it[1]; ++j) { int a = expr_degree(I.bc + sp[2 + 2 * j], sp[3 + 2 * j]); if (a > d) d = a; }
        bool ok = !dm.empty();
        if (ok) {
            const int ns = sum_den + d;
            ok = 1 + ns <= 3 && m + ns - d <= 3;
            for (size_t k = 0; ok && k < dm.size(); ++k) ok = dm[k] + ns - dd[k] <= 3;
        }
        if (!ok && !dm.empty()) { starts.push_back((u32)i); dm.clear(); dd.clear(); sum_den = 0; }
        dm.push_back(m); dd.push_back(d); sum_den += d;
    }
    if (I.n) starts.push_back((u32)I.n);
    return starts;
}

void observe_instance2(Challenger& ch, u32 log_h, u32 width, u32 nc, u32 n_int, const Config& cfg) {
    ch.observe(MAGIC2 % P); ch.observe(log_h); ch.observe(width); ch.observe(nc); ch.observe(n_int);
    ch.observe(cfg.num_queries); ch.observe(cfg.pow_bits);
}

/* The LogUp challenges (alpha_l, beta_l) come from a transcript of their own that observes only the 8-word
 * `bus seed`, so that every AIR of a segment draws the SAME pair and the per-AIR cumulative sums can be added up.
 * The seed is a digest over the trace commitments of all AIRs on the bus (the segment verifier recomputes it from
 * the trace roots in the proofs); a lone AIR uses its own trace root. */
void bus_challenges(const Digest& seed, Ext& al, Ext& bl) {
    Challenger cb;
    cb.observe(MAGIC2 % P);
    cb.observe_digest(seed);
    al = cb.sample_ext();
    bl = cb.sample_ext();
}

std::vector<u32> prove_logup(const Config& cfg, const u32* trace, u32 width, u32 log_h, const Program& prog, const Interactions& I,
                             const u32* bus_seed) {
    const size_t H = (size_t)1 << log_h, N = 2 * H;
    const int logN = (int)log_h + 1;
    const u32 n_int = (u32)I.n;
    const std::vector<u32> gs = group_starts(I);
    const size_t n_g = gs.size() - 1;
    const size_t Wp = 4 * (n_g + 1);
    /* (num, den) of group g at one row: q_g = num / den with num = sum_i m_i prod_{j != i} d_j, den = prod_i d_i */
    auto group_fraction = [&](size_t g, const u32* m, size_t stride, size_t row, const Ext& al, const Ext* blpow, Ext& num, Ext& den) {
        num = ext_zero(); den = ext_one();
        for (size_t i = gs[g]; i < gs[g + 1]; ++i) {
            Ext d = LogupRow::denom_base(I, i, m, stride, row, al, blpow);
            u32 mu = LogupRow::mult_base(I, i, m, stride, row);
            num = ext_add(ext_mul(num, d), ext_scale(den, mu));
            den = ext_mul(den, d);
        }
    };
    Writer pf;
    Challenger ch;
    observe_instance2(ch, log_h, width, (u32)prog.n, n_int, cfg);
    pf.put(MAGIC2); pf.put(log_h); pf.put(width); pf.put((u32)prog.n); pf.put(n_int); pf.put(cfg.num_queries); pf.put(cfg.pow_bits);

    /* 1. main trace */
    std::vector<u32> coef((size_t)width * H), lde((size_t)width * N);
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)width; ++c) {
        memcpy(&coef[c * H], trace + c * H, H * 4);
        idft(&coef[c * H], (int)log_h);
        coset_lde_from_coeffs(&coef[c * H], (int)log_h, &lde[c * N]);
    }
    Merkle t_tree;
    commit_matrix(lde.data(), N, width, t_tree);
    pf.put(t_tree.root());
    ch.observe_digest(t_tree.root());

    /* 2. permutation (LogUp) trace */
    Digest seed = t_tree.root();
    if (bus_seed) memcpy(seed.data(), bus_seed, 32);
    pf.put(seed);
    ch.observe_digest(seed);
    Ext al, bl;
    bus_challenges(seed, al, bl);
    std::vector<Ext> blpow(max_args(I) + 2);
    { Ext b = ext_one(); for (auto& x : blpow) { x = b; b = ext_mul(b, bl); } }
    std::vector<u32> perm(Wp * H);
    std::vector<Ext> rowsum(H);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)H; ++r) {
        Ext acc = ext_zero();
        for (size_t g = 0; g < n_g; ++g) {
            Ext num, den;
            group_fraction(g, trace, H, (size_t)r, al, blpow.data(), num, den);
            Ext q = ext_mul(num, ext_inv(den));
            for (int k = 0; k < 4; ++k) perm[(4 * g + k) * H + r] = q.c[k];
            acc = ext_add(acc, q);
        }
        rowsum[r] = acc;
    }
    Ext run = ext_zero();
    for (size_t r = 0; r < H; ++r) { run = ext_add(run, rowsum[r]); for (int k = 0; k < 4; ++k) perm[(4 * n_g + k) * H + r] = run.c[k]; }
    const Ext S = run;
    std::vector<u32> pcoef(Wp * H), plde(Wp * N);
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)Wp; ++c) {
        memcpy(&pcoef[c * H], &perm[c * H], H * 4);
        idft(&pcoef[c * H], (int)log_h);
        coset_lde_from_coeffs(&pcoef[c * H], (int)log_h, &plde[c * N]);
    }
    Merkle p_tree;
    commit_matrix(plde.data(), N, Wp, p_tree);
    pf.put(p_tree.root());
    ch.observe_digest(p_tree.root());
    pf.put(S);
    ch.observe_ext(S);

    /* 3. quotient */
    Ext alpha = ch.sample_ext();
    const size_t M = prog.n + n_g + 3;
    std::vector<Ext> apow(M);
    { Ext a = ext_one(); for (size_t j = M; j-- > 0;) { apow[j] = a; a = ext_mul(a, alpha); } }
    u32 sH = or_pow(COSET_SHIFT, H);
    u32 zval[2] = {or_sub(sH, 1), or_sub(or_neg(sH), 1)};
    u32 zinv[2] = {or_inv(zval[0]), or_inv(zval[1])};
    u32 g = root_of_unity((int)log_h), ginv = or_inv(g);
    u32 wN = root_of_unity(logN);
    std::vector<u32> xs(N);
    { u32 x = COSET_SHIFT; for (size_t j = 0; j < N; ++j) { xs[j] = x; x = or_mul(x, wN); } }
    std::vector<u32> q(4 * N);
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)N; ++j) {
        const size_t jn = ((size_t)j + 2) & (N - 1);  /* next row: g * x_j = x_{j+2} */
        Ext acc = ext_zero();
        for (size_t k = 0; k < prog.n; ++k) {
            u32 v = eval_base(prog.bc + prog.spans[2 * k], prog.spans[2 * k + 1], lde.data(), N, (size_t)j);
            acc = ext_add(acc, ext_scale(apow[k], v));
        }
        Ext sumq = ext_zero(), sumq_next = ext_zero();
        for (size_t g = 0; g < n_g; ++g) {
            Ext qi, qn, num, den;
            for (int k = 0; k < 4; ++k) { qi.c[k] = plde[(4 * g + k) * N + j]; qn.c[k] = plde[(4 * g + k) * N + jn]; }
            sumq = ext_add(sumq, qi);
            sumq_next = ext_add(sumq_next, qn);
            group_fraction(g, lde.data(), N, (size_t)j, al, blpow.data(), num, den);
            Ext c = ext_sub(ext_mul(qi, den), num);
            acc = ext_add(acc, ext_mul(apow[prog.n + g], c));
        }
        Ext phi, phin;
        for (int k = 0; k < 4; ++k) { phi.c[k] = plde[(4 * n_g + k) * N + j]; phin.c[k] = plde[(4 * n_g + k) * N + jn]; }
        u32 x = xs[j], Z = zval[j & 1];
        u32 is_first = or_mul(Z, or_inv(or_sub(x, 1)));
        u32 is_last = or_mul(Z, or_inv(or_sub(x, ginv)));
        u32 is_trans = or_sub(x, ginv);
        acc = ext_add(acc, ext_mul(apow[prog.n + n_g], ext_scale(ext_sub(phi, sumq), is_first)));
        acc = ext_add(acc, ext_mul(apow[prog.n + n_g + 1], ext_scale(ext_sub(ext_sub(phin, phi), sumq_next), is_trans)));
        acc = ext_add(acc, ext_mul(apow[prog.n + n_g + 2], ext_scale(ext_sub(phi, S), is_last)));
        acc = ext_scale(acc, zinv[j & 1]);
        for (int k = 0; k < 4; ++k) q[k * N + j] = acc.c[k];
    }
    std::vector<u32> qcoef(8 * H), qlde(8 * N);
    u32 sinv = or_inv(COSET_SHIFT);
    for (int k = 0; k < 4; ++k) {
        idft(&q[k * N], logN);
        u32 sp = 1;
        for (size_t i = 0; i < N; ++i) { q[k * N + i] = or_mul(q[k * N + i], sp); sp = or_mul(sp, sinv); }
        memcpy(&qcoef[(size_t)k * H], &q[k * N], H * 4);
        memcpy(&qcoef[(size_t)(4 + k) * H], &q[k * N + H], H * 4);
    }
    for (int c = 0; c < 8; ++c) coset_lde_from_coeffs(&qcoef[(size_t)c * H], (int)log_h, &qlde[(size_t)c * N]);
    Merkle q_tree;
    commit_matrix(qlde.data(), N, 8, q_tree);
    pf.put(q_tree.root());
    ch.observe_digest(q_tree.root());

    /* 4. openings */
    Ext zeta = ch.sample_ext();
    Ext gzeta = ext_scale(zeta, g);
    const size_t K1 = width + Wp + 8, K = K1 + Wp;
    std::vector<Ext> opened(K);
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)width; ++c) opened[c] = eval_poly_at(&coef[c * H], H, zeta);
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)Wp; ++c) {
        opened[width + c] = eval_poly_at(&pcoef[c * H], H, zeta);
        opened[K1 + c] = eval_poly_at(&pcoef[c * H], H, gzeta);
    }
    for (int c = 0; c < 8; ++c) opened[width + Wp + c] = eval_poly_at(&qcoef[(size_t)c * H], H, zeta);
    /* proof order: main, perm@zeta, perm@g zeta, quotient */
    for (size_t c = 0; c < width + Wp; ++c) { pf.put(opened[c]); ch.observe_ext(opened[c]); }
    for (size_t c = 0; c < Wp; ++c) { pf.put(opened[K1 + c]); ch.observe_ext(opened[K1 + c]); }
    for (int c = 0; c < 8; ++c) { pf.put(opened[width + Wp + c]); ch.observe_ext(opened[width + Wp + c]); }

    /* 5. DEEP vector */
    Ext gamma = ch.sample_ext();
    std::vector<Ext> gpow(K);
    { Ext gg = ext_one(); for (size_t k = 0; k < K; ++k) { gpow[k] = gg; gg = ext_mul(gg, gamma); } }
    Ext sum1 = ext_zero(), sum2 = ext_zero();
    for (size_t k = 0; k < K1; ++k) sum1 = ext_add(sum1, ext_mul(gpow[k], opened[k]));
    for (size_t k = K1; k < K; ++k) sum2 = ext_add(sum2, ext_mul(gpow[k], opened[k]));
    std::vector<Ext> v(N);
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)N; ++j) {
        Ext a1 = ext_zero(), a2 = ext_zero();
        for (size_t k = 0; k < width; ++k) a1 = ext_add(a1, ext_scale(gpow[k], lde[k * N + j]));
        for (size_t k = 0; k < Wp; ++k) {
            u32 pv = plde[k * N + j];
            a1 = ext_add(a1, ext_scale(gpow[width + k], pv));
            a2 = ext_add(a2, ext_scale(gpow[K1 + k], pv));
        }
        for (size_t k = 0; k < 8; ++k) a1 = ext_add(a1, ext_scale(gpow[width + Wp + k], qlde[k * N + j]));
        Ext t1 = ext_mul(ext_sub(a1, sum1), ext_inv(ext_sub(ext_from(xs[j]), zeta)));
        Ext t2 = ext_mul(ext_sub(a2, sum2), ext_inv(ext_sub(ext_from(xs[j]), gzeta)));
        v[j] = ext_add(t1, t2);
    }

    /* 6. FRI, PoW, queries: as v0 plus the perm matrix openings */
    std::vector<std::vector<Ext>> layers;
    std::vector<Merkle> fri_trees;
    u32 shift = COSET_SHIFT, inv2 = or_inv(2);
    for (int l = 0; l < (int)log_h; ++l) {
        size_t Nl = N >> l, half = Nl / 2;
        std::vector<Digest> leaves(half);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)half; ++i) {
            u32 row[8];
            memcpy(row, v[i].c, 16);
            memcpy(row + 4, v[i + half].c, 16);
            leaves[i] = hash_row(row, 8);
        }
        Merkle t;
        t.build(std::move(leaves));
        pf.put(t.root());
        ch.observe_digest(t.root());
        Ext beta = ch.sample_ext();
        u32 wl = root_of_unity(logN - l);
        std::vector<Ext> nv(half);
        std::vector<u32> xl(half);
        { u32 x = shift; for (size_t i = 0; i < half; ++i) { xl[i] = x; x = or_mul(x, wl); } }
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)half; ++i) {
            Ext a = v[i], b = v[i + half];
            Ext s2 = ext_scale(ext_add(a, b), inv2);
            Ext d = ext_scale(ext_sub(a, b), or_mul(inv2, or_inv(xl[i])));
            nv[i] = ext_add(s2, ext_mul(beta, d));
        }
        layers.push_back(std::move(v));
        fri_trees.push_back(std::move(t));
        v = std::move(nv);
        shift = or_mul(shift, shift);
    }
    pf.put(v[0]);
    ch.observe_ext(v[0]);
    u32 witness = 0;
    if (cfg.pow_bits) {
        for (;; ++witness) { Challenger c2 = ch; c2.observe(witness); if (c2.sample_bits((int)cfg.pow_bits) == 0) break; }
    }
    pf.put(witness);
    ch.observe(witness);
    if (cfg.pow_bits) (void)ch.sample_bits((int)cfg.pow_bits);
    for (u32 qi = 0; qi < cfg.num_queries; ++qi) {
        size_t idx = ch.sample_bits(logN);
        pf.put((u32)idx);
        for (size_t c = 0; c < width; ++c) pf.put(lde[c * N + idx]);
        t_tree.path(idx, pf.w);
        for (size_t c = 0; c < Wp; ++c) pf.put(plde[c * N + idx]);
        p_tree.path(idx, pf.w);
        for (size_t c = 0; c < 8; ++c) pf.put(qlde[c * N + idx]);
        q_tree.path(idx, pf.w);
        for (int l = 0; l < (int)log_h; ++l) {
            size_t Nl = N >> l, half = Nl / 2, p = idx & (Nl - 1);
            pf.put(layers[l][p ^ half]);
            fri_trees[l].path(p & (half - 1), pf.w);
        }
    }
    return pf.w;
}

int verify_logup(const Config& cfg, const u32* proof, size_t len, u32 width, u32 log_h, const Program& prog, const Interactions& I,
                 const u32* expected_seed) {
    const size_t H = (size_t)1 << log_h, N = 2 * H;
    const int logN = (int)log_h + 1;
    const u32 n_int = (u32)I.n;
    const std::vector<u32> gs = group_starts(I);
    const size_t n_g = gs.size() - 1;
    const size_t Wp = 4 * (n_g + 1);
    size_t pos = 0;
    auto need = [&](size_t k) { if (pos + k > len) throw std::runtime_error("short proof"); };
    auto get = [&]() { need(1); return proof[pos++]; };
    auto get_digest = [&]() { need(8); Digest d; memcpy(d.data(), proof + pos, 32); pos += 8; return d; };
    auto get_ext = [&]() { need(4); Ext e; memcpy(e.c, proof + pos, 16); pos += 4; return e; };
    try {
        if (get() != MAGIC2 || get() != log_h || get() != width || get() != prog.n || get() != n_int || get() != cfg.num_queries ||
            get() != cfg.pow_bits) return 1;
        for (size_t i = 0; i < len; ++i) if (proof[i] >= P) return 13;
        Challenger ch;
        observe_instance2(ch, log_h, width, (u32)prog.n, n_int, cfg);
        Digest t_root = get_digest();
        ch.observe_digest(t_root);
        Digest seed = get_digest();
        if (expected_seed ? memcmp(seed.data(), expected_seed, 32) != 0 : seed != t_root) return 12;
        ch.observe_digest(seed);
        Ext al, bl;
        bus_challenges(seed, al, bl);
        Digest p_root = get_digest();
        ch.observe_digest(p_root);
        Ext S = get_ext();
        ch.observe_ext(S);
        Ext alpha = ch.sample_ext();
        Digest q_root = get_digest();
        ch.observe_digest(q_root);
        Ext zeta = ch.sample_ext();
        u32 g = root_of_unity((int)log_h), ginv = or_inv(g);
        Ext gzeta = ext_scale(zeta, g);
        const size_t K1 = width + Wp + 8, K = K1 + Wp;
        std::vector<Ext> opened(K);
        for (size_t c = 0; c < width + Wp; ++c) { opened[c] = get_ext(); ch.observe_ext(opened[c]); }
        for (size_t c = 0; c < Wp; ++c) { opened[K1 + c] = get_ext(); ch.observe_ext(opened[K1 + c]); }
        for (int c = 0; c < 8; ++c) { opened[width + Wp + c] = get_ext(); ch.observe_ext(opened[width + Wp + c]); }
        /* constraints at zeta */
        Ext basis[4] = {{{1, 0, 0, 0}}, {{0, 1, 0, 0}}, {{0, 0, 1, 0}}, {{0, 0, 0, 1}}};
        auto ext_col = [&](size_t base) { Ext r = ext_zero(); for (int k = 0; k < 4; ++k) r = ext_add(r, ext_mul(basis[k], opened[base + k])); return r; };
        std::vector<Ext> blpow(max_args(I) + 2);
        { Ext b = ext_one(); for (auto& x : blpow) { x = b; b = ext_mul(b, bl); } }
        Ext acc = ext_zero();
        for (size_t k = 0; k < prog.n; ++k)
            acc = ext_add(ext_mul(acc, alpha), eval_ext(prog.bc + prog.spans[2 * k], prog.spans[2 * k + 1], opened.data()));
        Ext sumq = ext_zero(), sumq_next = ext_zero();
        for (size_t g = 0; g < n_g; ++g) {
            Ext num = ext_zero(), den = ext_one();
            for (size_t i = gs[g]; i < gs[g + 1]; ++i) {
                const u32* it = I.inter + 3 * i;
                const u32* sp = I.spans + 2 * (size_t)it[2];
                Ext d = ext_add(al, ext_from(it[0] % P));
                for (u32 j = 0; j < it[1]; ++j) d = ext_add(d, ext_mul(blpow[j + 1], eval_ext(I.bc + sp[2 + 2 * j], sp[3 + 2 * j], opened.data())));
                Ext m = eval_ext(I.bc + sp[0], sp[1], opened.data());
                num = ext_add(ext_mul(num, d), ext_mul(den, m));
                den = ext_mul(den, d);
            }
            Ext qi = ext_col(width + 4 * g), qn = ext_col(K1 + 4 * g);
            sumq = ext_add(sumq, qi);
            sumq_next = ext_add(sumq_next, qn);
            acc = ext_add(ext_mul(acc, alpha), ext_sub(ext_mul(qi, den), num));
        }
        Ext phi = ext_col(width + 4 * n_g), phin = ext_col(K1 + 4 * n_g);
        Ext zH = ext_pow(zeta, H);
        Ext Z = ext_sub(zH, ext_one());
        Ext is_first = ext_mul(Z, ext_inv(ext_sub(zeta, ext_one())));
        Ext is_last = ext_mul(Z, ext_inv(ext_sub(zeta, ext_from(ginv))));
        Ext is_trans = ext_sub(zeta, ext_from(ginv));
        acc = ext_add(ext_mul(acc, alpha), ext_mul(is_first, ext_sub(phi, sumq)));
        acc = ext_add(ext_mul(acc, alpha), ext_mul(is_trans, ext_sub(ext_sub(phin, phi), sumq_next)));
        acc = ext_add(ext_mul(acc, alpha), ext_mul(is_last, ext_sub(phi, S)));
        Ext qlo = ext_col(width + Wp), qhi = ext_col(width + Wp + 4);
        if (!ext_eq(acc, ext_mul(Z, ext_add(qlo, ext_mul(zH, qhi))))) return 2;

        Ext gamma = ch.sample_ext();
        std::vector<Ext> gpow(K);
        { Ext gg = ext_one(); for (size_t k = 0; k < K; ++k) { gpow[k] = gg; gg = ext_mul(gg, gamma); } }
        Ext sum1 = ext_zero(), sum2 = ext_zero();
        for (size_t k = 0; k < K1; ++k) sum1 = ext_add(sum1, ext_mul(gpow[k], opened[k]));
        for (size_t k = K1; k < K; ++k) sum2 = ext_add(sum2, ext_mul(gpow[k], opened[k]));
        std::vector<Digest> fri_roots(log_h);
        std::vector<Ext> betas(log_h);
        for (u32 l = 0; l < log_h; ++l) { fri_roots[l] = get_digest(); ch.observe_digest(fri_roots[l]); betas[l] = ch.sample_ext(); }
        Ext final_poly = get_ext();
        ch.observe_ext(final_poly);
        u32 witness = get();
        ch.observe(witness);
        if (cfg.pow_bits && ch.sample_bits((int)cfg.pow_bits) != 0) return 3;
        auto check_path = [&](Digest leaf, size_t idx, int depth, const Digest& root) {
            for (int l = 0; l < depth; ++l) { Digest sib = get_digest(); leaf = ((idx >> l) & 1) ? compress(sib, leaf) : compress(leaf, sib); }
            return leaf == root;
        };
        u32 wN = root_of_unity(logN), inv2 = or_inv(2);
        for (u32 qi = 0; qi < cfg.num_queries; ++qi) {
            size_t idx = ch.sample_bits(logN);
            if (get() != idx) return 4;
            need(width);
            const u32* trow = proof + pos; pos += width;
            if (!check_path(hash_row(trow, width), idx, logN, t_root)) return 5;
            need(Wp);
            const u32* prow = proof + pos; pos += Wp;
            if (!check_path(hash_row(prow, Wp), idx, logN, p_root)) return 11;
            need(8);
            const u32* qrow = proof + pos; pos += 8;
            if (!check_path(hash_row(qrow, 8), idx, logN, q_root)) return 6;
            u32 x = or_mul(COSET_SHIFT, or_pow(wN, idx));
            Ext a1 = ext_zero(), a2 = ext_zero();
            for (size_t k = 0; k < width; ++k) a1 = ext_add(a1, ext_scale(gpow[k], trow[k]));
            for (size_t k = 0; k < Wp; ++k) { a1 = ext_add(a1, ext_scale(gpow[width + k], prow[k])); a2 = ext_add(a2, ext_scale(gpow[K1 + k], prow[k])); }
            for (size_t k = 0; k < 8; ++k) a1 = ext_add(a1, ext_scale(gpow[width + Wp + k], qrow[k]));
            Ext cur = ext_add(ext_mul(ext_sub(a1, sum1), ext_inv(ext_sub(ext_from(x), zeta))),
                              ext_mul(ext_sub(a2, sum2), ext_inv(ext_sub(ext_from(x), gzeta))));
            u32 shift = COSET_SHIFT;
            for (u32 l = 0; l < log_h; ++l) {
                size_t Nl = N >> l, half = Nl / 2, p = idx & (Nl - 1);
                Ext sib = get_ext();
                Ext lo = (p < half) ? cur : sib, hi = (p < half) ? sib : cur;
                u32 row[8];
                memcpy(row, lo.c, 16); memcpy(row + 4, hi.c, 16);
                if (!check_path(hash_row(row, 8), p & (half - 1), logN - 1 - (int)l, fri_roots[l])) return 7;
                u32 xi = or_mul(shift, or_pow(root_of_unity(logN - (int)l), p & (half - 1)));
                Ext s2 = ext_scale(ext_add(lo, hi), inv2);
                Ext d = ext_scale(ext_sub(lo, hi), or_mul(inv2, or_inv(xi)));
                cur = ext_add(s2, ext_mul(betas[l], d));
                shift = or_mul(shift, shift);
            }
            if (!ext_eq(cur, final_poly)) return 8;
        }
        if (pos != len) return 9;
    } catch (const std::exception&) {
        return 10;
    }
    return 0;
}

#include "stark_segment.inc"

}  // namespace

/* ---------------------------------------------------------------- C entry points for tests */
extern "C" {

void or_poseidon2_permute(uint32_t* state16) { poseidon2(state16); }
void or_poseidon2_constants(uint32_t* ext_rc /*8*16*/, uint32_t* int_rc /*13*/, uint32_t* diag /*16*/) {
    memcpy(ext_rc, p2c().ext_rc, sizeof p2c().ext_rc);
    memcpy(int_rc, p2c().int_rc, sizeof p2c().int_rc);
    memcpy(diag, p2c().diag, sizeof p2c().diag);
}
/* twin of pw_set_poseidon2_constants (include/powdr_prover.h): canonical words; both NULL = the placeholder stream */
int or_set_poseidon2_constants(const uint32_t* ext_rc /*8*16*/, const uint32_t* int_rc /*13*/) {
    if ((ext_rc == nullptr) != (int_rc == nullptr)) return -1;
    if (!ext_rc) { p2c_mut() = Poseidon2Constants(); return 0; }
    for (int i = 0; i < 128; ++i) if (ext_rc[i] >= P) return -1;
    for (int i = 0; i < 13; ++i) if (int_rc[i] >= P) return -1;
    memcpy(p2c_mut().ext_rc, ext_rc, sizeof p2c().ext_rc);
    memcpy(p2c_mut().int_rc, int_rc, sizeof p2c().int_rc);
    return 0;
}
uint32_t or_root_of_unity(int log_n) { return root_of_unity(log_n); }
void or_ext_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { Ext x, y; memcpy(x.c, a, 16); memcpy(y.c, b, 16); Ext r = ext_mul(x, y); memcpy(o, r.c, 16); }
void or_ext_inv(const uint32_t* a, uint32_t* o) { Ext x; memcpy(x.c, a, 16); Ext r = ext_inv(x); memcpy(o, r.c, 16); }

/* natural-order DFT of one column, in place (w = canonical root of unity for size 2^log_n) */
void or_dft(uint32_t* a, int log_n, int inverse) { if (inverse) idft(a, log_n); else dft(a, log_n, root_of_unity(log_n)); }
/* naive O(n^2) DFT for cross-checking the fast one on small sizes */
void or_dft_naive(const uint32_t* a, uint32_t* out, int log_n) {
    size_t n = (size_t)1 << log_n;
    u32 w = root_of_unity(log_n);
    for (size_t j = 0; j < n; ++j) {
        u32 acc = 0, wj = or_pow(w, j), t = 1;
        for (size_t i = 0; i < n; ++i) { acc = or_add(acc, or_mul(a[i], t)); t = or_mul(t, wj); }
        out[j] = acc;
    }
}
/* column-major LDE of a W x 2^log_h matrix onto the coset (natural order), out is W x 2^(log_h+1) */
void or_lde(const uint32_t* trace, uint32_t width, int log_h, uint32_t* out) {
    size_t H = (size_t)1 << log_h;
#pragma omp parallel for schedule(dynamic)
    for (long c = 0; c < (long)width; ++c) lde_column(trace + c * H, log_h, out + c * 2 * H);
}
/* Merkle commitment of a column-major matrix; digests_out (optional) receives all layers
 * bottom-up, (2*height - 1) * 8 words; root_out 8 words */
void or_merkle_commit(const uint32_t* m, size_t height, size_t width, uint32_t* digests_out, uint32_t* root_out) {
    Merkle t;
    commit_matrix(m, height, width, t);
    memcpy(root_out, t.root().data(), 32);
    if (digests_out) {
        size_t o = 0;
        for (auto& l : t.layers) { memcpy(digests_out + o, l.data(), l.size() * 32); o += l.size() * 8; }
    }
}
/* prove: returns the number of u32 words written (or needed if cap is too small) */
size_t or_prove(uint32_t num_queries, uint32_t pow_bits, const uint32_t* trace, uint32_t width, uint32_t log_h,
                const uint32_t* cons_bc, const uint32_t* cons_spans, size_t n_constraints, uint32_t* proof, size_t cap) {
    Config cfg{num_queries, pow_bits};
    Program pr{cons_bc, cons_spans, n_constraints};
    std::vector<u32> w = prove(cfg, trace, width, log_h, pr);
    if (w.size() <= cap) memcpy(proof, w.data(), w.size() * 4);
    return w.size();
}
int or_verify(uint32_t num_queries, uint32_t pow_bits, const uint32_t* proof, size_t len, uint32_t width, uint32_t log_h,
              const uint32_t* cons_bc, const uint32_t* cons_spans, size_t n_constraints) {
    Config cfg{num_queries, pow_bits};
    Program pr{cons_bc, cons_spans, n_constraints};
    return verify(cfg, proof, len, width, log_h, pr);
}


/* ---- pw-stark v0 + LogUp: interactions = n x {bus id, n_args, first span index}; spans = {off,len} pairs laid out
 * [mult, arg0, arg1, ...] per interaction; bytecode with column-index operands (compile_bus with height 1). */
size_t or_prove_logup(uint32_t num_queries, uint32_t pow_bits, const uint32_t* trace, uint32_t width, uint32_t log_h,
                      const uint32_t* cons_bc, const uint32_t* cons_spans, size_t n_constraints, const uint32_t* inter,
                      size_t n_inter, const uint32_t* ispans, const uint32_t* ibc, const uint32_t* bus_seed /* 8 words or NULL */,
                      uint32_t* proof, size_t cap) {
    Config cfg{num_queries, pow_bits};
    Program pr{cons_bc, cons_spans, n_constraints};
    Interactions I{inter, n_inter, ispans, ibc};
    std::vector<u32> w = prove_logup(cfg, trace, width, log_h, pr, I, bus_seed);
    if (w.size() <= cap) memcpy(proof, w.data(), w.size() * 4);
    return w.size();
}
/* group boundaries of the LogUp packing: writes up to cap entries, returns the number of entries (n_groups + 1) */
size_t or_group_starts(const uint32_t* inter, size_t n_inter, const uint32_t* ispans, const uint32_t* ibc, uint32_t* out, size_t cap) {
    Interactions I{inter, n_inter, ispans, ibc};
    std::vector<u32> g = group_starts(I);
    for (size_t i = 0; i < g.size() && i < cap; ++i) out[i] = g[i];
    return g.size();
}
int or_verify_logup(uint32_t num_queries, uint32_t pow_bits, const uint32_t* proof, size_t len, uint32_t width, uint32_t log_h,
                    const uint32_t* cons_bc, const uint32_t* cons_spans, size_t n_constraints, const uint32_t* inter,
                    size_t n_inter, const uint32_t* ispans, const uint32_t* ibc, const uint32_t* expected_seed /* or NULL */) {
    Config cfg{num_queries, pow_bits};
    Program pr{cons_bc, cons_spans, n_constraints};
    Interactions I{inter, n_inter, ispans, ibc};
    return verify_logup(cfg, proof, len, width, log_h, pr, I, expected_seed);
}


/* ---- pw-stark v1: one proof per segment (stark_segment.inc) ---- */
typedef struct {
    const uint32_t* trace; /* NULL for the verifier */
    uint32_t width, log_h;
    const uint32_t* cons_bc; const uint32_t* cons_spans; size_t n_constraints;
    const uint32_t* inter; size_t n_inter; const uint32_t* ispans; const uint32_t* ibc;
} OrSegAir;

static std::vector<SegAir> seg_airs(const OrSegAir* airs, size_t n) {
    std::vector<SegAir> v(n);
    for (size_t i = 0; i < n; ++i)
        v[i] = SegAir{airs[i].trace, airs[i].width, airs[i].log_h, Program{airs[i].cons_bc, airs[i].cons_spans, airs[i].n_constraints},
                      Interactions{airs[i].inter, airs[i].n_inter, airs[i].ispans, airs[i].ibc}};
    return v;
}
size_t or_prove_segment(uint32_t num_queries, uint32_t pow_bits, int logup, const OrSegAir* airs, size_t n_airs, uint32_t* proof, size_t cap) {
    Config cfg{num_queries, pow_bits};
    std::vector<u32> w = prove_segment(cfg, seg_airs(airs, n_airs), logup != 0);
    if (w.size() <= cap) memcpy(proof, w.data(), w.size() * 4);
    return w.size();
}
int or_verify_segment(uint32_t num_queries, uint32_t pow_bits, int logup, const OrSegAir* airs, size_t n_airs, const uint32_t* proof,
                      size_t len, int check_balance, uint32_t* total_sum4) {
    Config cfg{num_queries, pow_bits};
    return verify_segment(cfg, proof, len, seg_airs(airs, n_airs), logup != 0, check_balance != 0, total_sum4);
}

}  // extern "C"
