// The stages of a STREAMED proof, shared by the one-AIR prover (prover.hip) and the segment prover
// (segment_prover.hip): an AIR whose low-degree extension is not kept in HBM is proven from coefficient arrays — the trace's in
// p->tcoef, the permutation matrix's in place of the matrix (p->perm) — one sub-coset of the extended domain at a time
// (ntt.hip subcoset_lde: rows r + 2^b i). Everything runs on the calling thread's launch stream; p->lde holds the sub-coset being
// processed, p->fscale the sub-coset's twiddle table, p->qpart the partial sums. Not part of the C ABI.
#pragma once
#include "prover_state.hpp"

#include <algorithm>
#include <initializer_list>
#include <vector>

namespace pw { namespace streamed {

#define PW_STRY(x) do { const int _rc = (x); if (_rc) return _rc; } while (0)

struct Ctx {
    PwProver* p;
    uint32_t log_h;
    int b;             // the extended domain is walked as 2^b sub-cosets
    bool perm_panels;  // quotient: the permutation columns come in unit by unit (specialised LogUp kernels)
    size_t H, N, m;    // trace rows, LDE rows, rows of a sub-coset
    uint32_t W, Wp;    // main columns, committed permutation columns
};

struct CoefMatrix { const uint32_t* coef; uint32_t cols; };

// One pass over the sub-cosets (`wanted`: only those): for every r the rows r + 2^b i of the LDE of `mats` (side by side, column
// stride m) are rebuilt in p->lde and handed to body(r).
template <class Body>
inline int for_each_subcoset(const Ctx& c, std::initializer_list<CoefMatrix> mats, const std::vector<char>* wanted, Body&& body) {
    uint32_t* blk = c.p->lde.as<uint32_t>();
    uint32_t* fs = c.p->fscale.as<uint32_t>();
    for (uint32_t r = 0; r < (1u << c.b); ++r) {
        if (wanted && !(*wanted)[r]) continue;
        size_t c0 = 0;
        for (const CoefMatrix& mt : mats) {
            if (mt.cols) PW_STRY(subcoset_lde(mt.coef, blk + c0 * c.m, c.H, c.m, mt.cols, (int)c.log_h, c.b, r, fs));
            c0 += mt.cols;
        }
        PW_STRY(body(r));
    }
    return 0;
}

// Row digests of a matrix given by its coefficient arrays into their slots of `d_leaves` (N x 8 words): every sub-coset's rows are
// hashed as soon as they exist. The inner levels are the caller's (merkle_build_levels / the mixed tree's compress steps).
inline int leaf_hashes(const Ctx& c, const uint32_t* coef, uint32_t cols, uint32_t* d_leaves) {
    return for_each_subcoset(c, {CoefMatrix{coef, cols}}, nullptr, [&](uint32_t r) {
        return merkle_leaf_hash(c.p->lde.as<uint32_t>(), c.m, cols, c.m, d_leaves, (size_t)1 << c.b, r);
    });
}

// The quotient's terms that read the CURRENT row only — sum_k apow[k] C_k + sum_g apow[nc + g] (q_g den_g - num_g) — unscaled, for all N
// rows of the extended domain into d_q (4 x N), sub-coset by sub-coset. The caller adds the boundary terms (quotient_logup_tail with
// part = d_q, one chunk) or divides by Z_H (quotient_combine, in place).
inline int quotient_sums(const Ctx& c, bool jit, bool lg, uint32_t nc, const ConstraintProgram& prog, const LogupProgram& lp, const uint32_t* d_tcoef,
                         const uint32_t* d_pcoef, const bb::Ext* d_apow, bb::Ext al, const bb::Ext* d_blpow, bb::Ext S, int logN, uint32_t* d_q) {
    PwProver* p = c.p;
    uint32_t* d_blk = p->lde.as<uint32_t>();
    uint32_t* d_part = p->qpart.as<uint32_t>();
    const uint32_t one = bb::R_MOD_P;
    if (c.perm_panels) {
        // main columns of the sub-coset once; the permutation columns unit by unit into the panel behind them
        uint32_t* d_panel = d_blk + (size_t)c.W * c.m;
        uint32_t* fs = p->fscale.as<uint32_t>();
        return for_each_subcoset(c, {CoefMatrix{d_tcoef, c.W}}, nullptr, [&](uint32_t r) -> int {
            const uint32_t n_units = quotient_units_jit(p);
            for (uint32_t u = 0; u < n_units; ++u) {
                uint32_t g0 = 0, g1 = 0;
                quotient_unit_groups(p, u, &g0, &g1);
                if (g1 > g0) PW_STRY(subcoset_lde(d_pcoef + (size_t)(4 * g0) * c.H, d_panel, c.H, c.m, 4 * (g1 - g0), (int)c.log_h, c.b, r, fs));
                // Pm = where permutation column 0 would be: the panel holds columns 4 g0 .. 4 g1 - 1
                const uint32_t* Pm = d_panel - (size_t)(4 * g0) * c.m;
                PW_STRY(quotient_unit_jit(p, u, d_blk, Pm, c.m, d_apow, al, d_blpow, d_part));
            }
            return part_scatter(d_part, p->jit.quotient.n_chunks, c.m, c.b, r, c.N, d_q);
        });
    }
    return for_each_subcoset(c, {CoefMatrix{d_tcoef, c.W}, CoefMatrix{d_pcoef, c.Wp}}, nullptr, [&](uint32_t r) -> int {
        const uint32_t* blk_p = d_blk + (size_t)c.W * c.m;
        uint32_t n_parts = 1;
        if (jit && (nc || lg)) {
            PW_STRY(quotient_parts_jit(p, d_blk, lg ? blk_p : nullptr, c.m, d_apow, al, lg ? d_blpow : nullptr, d_part, &n_parts));
        } else if (lg) {
            PW_STRY(quotient_eval_logup(d_blk, blk_p, c.m, logN, prog, lp, d_apow, al, d_blpow, S, one, one, d_part, true));
        } else {
            PW_STRY(quotient_eval(d_blk, c.m, prog, d_apow, one, one, d_part, d_part + 4 * c.m, quotient_chunks(c.m, nc)));
        }
        return part_scatter(d_part, n_parts, c.m, c.b, r, c.N, d_q);
    });
}

// LDE rows idx[q] (q < n) of the matrix given by `coef` into d_rows[q * cols ..] (canonical Montgomery words): one more pass over the
// sub-cosets that hold a queried row, of which only the contiguous stage group runs over every row — the strided stages are finished
// for the queried rows alone (subcoset_rows). d_loc / d_slot: n words of device scratch each (row inside its sub-coset; answer slot).
inline int query_rows(const Ctx& c, const uint32_t* coef, uint32_t cols, const uint32_t* idx, uint32_t n, uint32_t* d_loc, uint32_t* d_slot,
                      uint32_t* d_rows) {
    if (!n || !cols) return 0;
    const uint32_t nb = 1u << c.b;
    std::vector<uint32_t> order(n), loc(n), slot(n), first(nb + 1, 0);
    for (uint32_t q = 0; q < n; ++q) { order[q] = q; first[(idx[q] & (nb - 1)) + 1] += 1; }
    for (uint32_t r = 0; r < nb; ++r) first[r + 1] += first[r];
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) { return (idx[a] & (nb - 1)) < (idx[b2] & (nb - 1)); });
    for (uint32_t k = 0; k < n; ++k) { loc[k] = idx[order[k]] >> c.b; slot[k] = order[k]; }
    hipStream_t st = pw::stream();
    PW_HIP_TRY(hipMemcpyAsync(d_loc, loc.data(), n * 4, hipMemcpyHostToDevice, st));
    PW_HIP_TRY(hipMemcpyAsync(d_slot, slot.data(), n * 4, hipMemcpyHostToDevice, st));
    PW_HIP_TRY(hipStreamSynchronize(st));  // loc / slot are locals
    uint32_t* d_blk = c.p->lde.as<uint32_t>();
    uint32_t* fs = c.p->fscale.as<uint32_t>();
    for (uint32_t r = 0; r < nb; ++r) {
        const uint32_t k0 = first[r], cnt = first[r + 1] - first[r];
        if (!cnt) continue;
        // tall sub-cosets: one pass that keeps the tiles in LDS and sums the queried rows' terms (p->lde is only scratch then)
        const int sel = subcoset_query_rows(coef, c.H, cols, (int)c.log_h, c.b, r, d_loc + k0, cnt, d_slot + k0, d_rows, d_blk, c.p->lde.bytes / 4);
        if (sel == 0) continue;
        if (sel != 1) return sel;
        int done = 0;
        PW_STRY(subcoset_lde_first_group(coef, d_blk, c.H, c.m, cols, (int)c.log_h, c.b, r, fs, &done));
        PW_STRY(subcoset_rows(d_blk, c.m, cols, (int)c.log_h, c.b, r, done, d_loc + k0, cnt, d_slot + k0, d_rows));
    }
    return 0;
}

// v[j] = (sum_k g^k P_k(x_j) - sum1) / (x_j - zeta) [+ (sum_k g^(K1+k) perm_k(x_j) - sum2) / (x_j - g zeta)] from the coefficient arrays:
// the numerator is a polynomial — combined on the arrays, extended as 4 (+ 4) columns through p->gbuf. d_gpow: the AIR's centred gamma
// powers in the order main | perm | quotient (8) | perm at g zeta. sum1 / sum2 are read AFTER host_work() has run: the caller's host
// work (the sums themselves, in the one-AIR prover) that the combination and its extension hide.
template <class HostWork>
inline int deep_from_coefficients(const Ctx& c, bool lg, const uint32_t* d_tcoef, const uint32_t* d_pcoef, const uint32_t* d_qlde, int logN,
                                  const bb::Ext* d_gpow, HostWork&& host_work, const bb::Ext& sum1, const bb::Ext& sum2, bb::Ext zeta, bb::Ext gzeta,
                                  bb::Ext* d_v) {
    uint32_t* d_gcoef = c.p->gbuf.as<uint32_t>();
    uint32_t* d_glde = d_gcoef + 8 * c.H;
    const uint32_t K1 = c.W + c.Wp + 8;
    PW_STRY(ext_lincomb(d_tcoef, c.W, d_pcoef, c.Wp, c.H, d_gpow, lg ? K1 : 0u, d_gcoef));
    PW_STRY(coset_lde_from_coeffs(d_gcoef, d_glde, c.H, c.N, lg ? 8 : 4, (int)c.log_h));
    host_work();
    return deep_from_combo(d_glde, d_qlde, c.N, logN, d_gpow + c.W + c.Wp, sum1, sum2, zeta, gzeta, lg ? 1 : 0, d_v);
}

#undef PW_STRY

}}  // namespace pw::streamed
