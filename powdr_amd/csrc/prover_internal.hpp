// Internal interfaces between the prover's translation units (not part of the C ABI).
#pragma once
#include "babybear.hpp"
#include "ext.hpp"
#include "small_form.hpp"
#include "poseidon2.hpp"
#include "common.hpp"

#include <stddef.h>
#include <stdint.h>

namespace pw {

namespace field {
constexpr uint32_t kTwoAdicGen = 0x1a427a41u;  // canonical; 31^15, order 2^27
constexpr uint32_t kCosetShift = 31u;          // canonical
// Montgomery-form primitive 2^n-th root of unity
inline uint32_t root_of_unity(int n) {
    uint32_t g = bb::to_monty(kTwoAdicGen);
    for (int i = n; i < 27; ++i) g = bb::sqr(g);
    return g;
}
}  // namespace field

// ---- ntt.hip ---------------------------------------------------------------------------
int intt_dif(const uint32_t* in, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n);
int coset_lde_from_coeffs(const uint32_t* coeffs, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n);
// intt_dif + coset_lde_from_coeffs in three (instead of four) passes over HBM: the contiguous stage groups of both transforms
// run in one kernel (ntt.hip lde_fused_kernel). tmp: cols x 2^n words of scratch (unused for n <= 12).
int lde_fused(const uint32_t* in, uint32_t* tmp, uint32_t* out, size_t in_stride, size_t tmp_stride, size_t out_stride, uint32_t cols, int n);
const uint32_t* shift_table(int n);  // s^k / 2^n (Montgomery), k < 2^n, device
// Sub-coset evaluation (the streamed prover): the rows j = r + 2^b i, i < 2^(n+1-b), of the LDE of `cols` polynomials given by their
// coefficient arrays as intt_dif leaves them (2^n words each, bit-reversed, H-scaled): out[c * out_stride + i] = P_c(s g_(n+1)^j).
// d_scratch: 2^13 words (the sub-coset's twiddle table). 1 <= b <= 5.
int subcoset_lde(const uint32_t* coeffs, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n, int b, uint32_t r,
                 uint32_t* d_scratch);
// A FEW rows of a sub-coset (the query phase): only the first stage group of the transform (the contiguous stages; *stages_done of
// log2 m) into `out`, then subcoset_rows finishes the remaining stages for the rows d_local_idx[q] alone — a 2^(log2 m - stages_done)-term
// sum per (row, column) instead of the strided stage groups over every row. rows_out[q * cols + c], canonical Montgomery words.
// the values on <g_n> (natural order, canonical) from coefficient arrays as intt_dif leaves them; d_scratch: >= 2^13 words
int values_from_coefficients(const uint32_t* coeffs, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n, uint32_t* d_scratch);
int subcoset_lde_first_group(const uint32_t* coeffs, uint32_t* out, size_t in_stride, size_t out_stride, uint32_t cols, int n, int b, uint32_t r,
                             uint32_t* d_scratch, int* stages_done);
// The same rows in ONE pass that stores no partial transform (tall transforms; ntt.hip subcoset_query_rows): 0 = done, 1 = the shape
// does not qualify (use the two calls above), else an error. d_work: 8-byte aligned scratch of work_words words.
int subcoset_query_rows(const uint32_t* coeffs, size_t in_stride, uint32_t cols, int n, int b, uint32_t r, const uint32_t* d_local_idx,
                        uint32_t n_idx, const uint32_t* d_slot, uint32_t* rows_out, uint32_t* d_work, size_t work_words);
int subcoset_rows(const uint32_t* part, size_t stride, uint32_t cols, int n, int b, uint32_t r, int stages_done, const uint32_t* d_local_idx,
                  uint32_t n_idx, const uint32_t* d_slot, uint32_t* rows_out);  // row q goes to slot d_slot[q] (null: q)

// ---- merkle.hip ------------------------------------------------------------------------
// Digest tree layout: level 0 = leaves (n_leaves x 8 words), then n_leaves/2, ... , 1;
// all levels concatenated: total (2*n_leaves - 1) * 8 words. root = last 8 words.
const p2::Params& poseidon2_params_host();
// canonical words, 8 x 16 external and 13 internal round constants; both nullptr = back to the placeholder stream
int poseidon2_set_constants(const uint32_t* ext_rc128, const uint32_t* int_rc13);
int poseidon2_upload_params();
int merkle_commit_matrix(const uint32_t* m, size_t height, uint32_t width, size_t col_stride, uint32_t* digests, uint32_t* root_out = nullptr);
// the two halves of merkle_commit_matrix: row digests into the slots j * digest_stride + digest_offset of level 0 (a matrix that holds
// every 2^b-th row of the committed one fills its share of the leaves), then the inner levels over all n_leaves slots
int merkle_leaf_hash(const uint32_t* m, size_t height, uint32_t width, size_t col_stride, uint32_t* digests, size_t digest_stride,
                     size_t digest_offset);
int merkle_build_levels(uint32_t* digests, size_t n_leaves, uint32_t* root_out = nullptr);
// One RUN of a level's row digest (the hashed row = the concatenation of several matrices' rows, absorbed run by run with the sponge
// states parked in `state`, 16 words per row of the level): the columns of a matrix (m, col_stride) or of a column-pointer table
// (d_cols) continue the rate block at position pos0 (= columns absorbed so far mod 8) for the rows j * rstride + roff, j < height;
// `first`: the states start at zero; `last`: a partial block is closed and the digests are written (8 words per row of the level).
int merkle_leaf_absorb(const uint32_t* m, size_t col_stride, const uint32_t* const* d_cols, uint32_t n_cols, uint32_t pos0, size_t height, size_t rstride,
                       size_t roff, uint32_t* state, uint32_t* digests, bool first, bool last);
// the calling thread's host-mapped landing place for a root (`root_out` above is its device address); nullptr: not available
uint32_t* merkle_root_mailbox(uint32_t** device_ptr);
// Mixed-height commitment of a segment (oracle/stark_segment.inc `MixedTree`): by_log[k] = the columns of all matrices of
// height 2^k (device array of device column pointers, AIR order; n_cols = 0 if there is none), k = 0..L, by_log[L] non-empty.
// digests: levels of 2^L, 2^(L-1), ..., 1 nodes concatenated ((2^(L+1) - 1) * 8 words), level of n nodes:
// node j = compress(child j, child j + n) [then compress(node, H(rows j of the height-n matrices))]. d_inject: 2^(L-1) * 8 words
// of scratch. NOTE the same scratch is reused level after level on the launch stream.
struct MixedLevelCols {
    const uint32_t* const* d_cols;
    uint32_t n_cols;
    // non-zero: the level's row digests are ALREADY in place (level L: in `digests`; level k < L: in d_inject + 2^k * 8) — a streamed AIR
    // hashed them sub-coset by sub-coset (prover_stream.hpp leaf_hashes); d_cols / n_cols are ignored
    uint32_t external = 0;
};
// d_inject: 2^L * 8 words (row digests of the smaller heights, the slice of height 2^k at offset 2^k * 8). All heights are hashed
// by ONE launch (widest level first); L <= 27.
int merkle_commit_mixed(const MixedLevelCols* by_log, int L, uint32_t* digests, uint32_t* d_inject);
// leaves = hash of the 8 words (v[i], v[i + half]) of an Ext vector of length 2*half
int merkle_commit_ext_pairs(const bb::Ext* v, size_t half, uint32_t* digests, uint32_t* root_out = nullptr);
inline size_t merkle_words(size_t n_leaves) { return (2 * n_leaves - 1) * 8; }
inline size_t merkle_level_offset(size_t n_leaves, int level) {  // in words
    size_t off = 0, n = n_leaves;
    for (int l = 0; l < level; ++l) { off += n * 8; n >>= 1; }
    return off;
}

// ---- stark_kernels.hip -----------------------------------------------------------------
struct ConstraintProgram {
    const uint32_t* d_bytecode;  // PUSH_COL operands = column index (reference post-fix encoding) or xbc code
    const uint32_t* d_spans;     // {off, len} pairs (u32 words, or xbc instructions when is_xbc)
    uint32_t n_constraints;
    bool is_xbc;
};
// q[k*N + j] (k < 4) = coordinate k of  sum_c alpha_pow[c] * C_c(lde row j) * zinv[j & 1]
// Short traces with many constraints: the constraint list is split over `n_chunks` (quotient_chunks) workgroup rows,
// partial sums go through `part` (4 * n_chunks * N words) and are added up by a second small kernel.
uint32_t quotient_chunks(size_t N, uint32_t n_constraints);
int quotient_eval(const uint32_t* lde, size_t N, const ConstraintProgram& prog, const bb::Ext* d_alpha_pows,
                  uint32_t zinv_even, uint32_t zinv_odd, uint32_t* q, uint32_t* part, uint32_t n_chunks);
// evaluate all constraints on all trace rows; d_first_and_count[0] = min(row * nc + c) over violations
// (caller initialises to ~0), [1] = number of violations
// q = (sum of n_chunks partial sums) * zinv, partial sums as quotient_eval leaves them in `part`
int quotient_combine(const uint32_t* part, uint32_t n_chunks, size_t N, uint32_t zinv_even, uint32_t zinv_odd, uint32_t* q);
int check_constraints(const uint32_t* trace, size_t H, const ConstraintProgram& prog, unsigned long long* d_first_and_count);
// chunk coefficients from the unscaled DIF-iNTT of q over N = 2H points:
// out[(4*ch + k)*H + q'] = cbr[k*N + 2q' + ch] * s^-(bitrev(q') + ch*H) / 2   (H-scaled bit-reversed coefficients)
int quotient_split(const uint32_t* cbr, size_t H, int log_h, uint32_t* out);
// Both weight vectors are stored as CENTRED words (int32 bit patterns, |w| <= p / 2): ext_dot_columns is their only consumer.
// weights[q] = z^(bitrev_n(q)) / 2^n  (Ext), for coefficient vectors as intt_dif leaves them
int zeta_weights(bb::Ext z, int log_h, bb::Ext* weights);
// weights[i] = (zeta^H - 1)/H * g^i / (zeta - g^i): f(zeta) = sum_i f(g^i) weights[i] for natural-order evaluations
int barycentric_weights(bb::Ext zeta, int log_h, bb::Ext* weights);
// out[c] = sum_q cols[c*stride + q] * weights[q]
int ext_dot_columns(const uint32_t* cols, size_t stride, uint32_t n_cols, size_t len, const bb::Ext* weights, bb::Ext* out,
                    bb::Ext* scratch);
// the same for two weight vectors in ONE pass over the columns (out2[c] = sum_q cols[c*stride + q] * weights2[q]); scratch: 2 x
// n_cols x ceil(len / 8192) Ext
int ext_dot_columns2(const uint32_t* cols, size_t stride, uint32_t n_cols, size_t len, const bb::Ext* weights, const bb::Ext* weights2,
                     bb::Ext* out, bb::Ext* out2, bb::Ext* scratch);
// v[j] = (sum_k gpow[k] * M_k[j] - opened_sum) / (x_j - zeta), x_j = s * g^j, over two matrices
int deep_quotient(const uint32_t* lde_a, uint32_t wa, const uint32_t* lde_b, uint32_t wb, size_t N, int logN,
                  const bb::Ext* d_gpow, bb::Ext opened_sum, bb::Ext zeta, bb::Ext* v);
// out[i] = (a+b)/2 + beta (a-b)/(2 x_i), a = v[i], b = v[i+half], x_i = shift * w^i
// d_out[k] = base^k (reversed: base^(n - 1 - k)), k < n <= 2^24, computed on the device; centred: as CENTRED words (the form the DEEP
// kernels and ext_lincomb take). gamma_powers = plain order, centred.
int ext_powers(bb::Ext base, uint32_t n, bool reversed, bool centred, bb::Ext* d_out);
int gamma_powers(bb::Ext gamma, uint32_t K, bb::Ext* d_gpow);
// rows d_indices[idx_off + q] (q < n_idx) of n_jobs column-major matrices into out + out_off (row-major, q-th row at q * width): ONE launch
struct GatherRowsJob { const uint32_t* m; uint64_t height; uint32_t width; uint32_t idx_off; uint64_t out_off; };
int gather_rows_multi(const GatherRowsJob* d_jobs, uint32_t n_jobs, uint32_t max_width, const uint32_t* d_indices, uint32_t n_idx, uint32_t* out);
int fri_fold(const bb::Ext* v, size_t half, int log_size, uint32_t shift, bb::Ext beta, bb::Ext* out);
// y[i] += a * x[i] (a == nullptr: a = 1), Ext vectors of length n
int ext_axpy(bb::Ext* y, const bb::Ext* a_or_null, const bb::Ext* x, size_t n);
// Montgomery -> canonical, in place (proof words leave the device canonical)
int canonicalize_words(uint32_t* d_words, size_t n);
// d_out[i] = *d_ptrs[i]
int gather_words(const uint32_t* const* d_ptrs, uint32_t n, uint32_t* d_out);
// gather row `idx` of a column-major matrix into out[0..width)
int gather_rows(const uint32_t* m, size_t height, uint32_t width, const uint32_t* d_indices, uint32_t n_idx, uint32_t* out);
// ---- logup_kernels.hip (pw-stark v0 + LogUp) ----------------------------------------------------------
struct LogupInteraction {
    uint32_t bus_monty;   // bus id as a field element (Montgomery)
    uint32_t n_args;
    uint32_t first_span;  // index into xspans: [mult, arg0, arg1, ...]
};
struct LogupProgram {
    const LogupInteraction* d_inter;
    uint32_t n;
    const uint32_t* d_xspans;  // {off, len} in xbc instructions
    const uint32_t* d_code;    // xbc code, column-index operands
    const uint32_t* d_gstarts; // n_groups + 1 interaction indices (logup_groups.hpp)
    uint32_t n_groups;
    const SmallForm* d_forms;  // one per span, or nullptr: every multiplicity / argument as a small form (small_form.hpp)
};
// perm (4(n_groups+1) columns x H): q_g coordinates then phi; d_rowsum: H Ext scratch; d_block_totals: H/4096+1 Ext scratch
int logup_perm_trace(const uint32_t* trace, size_t H, const LogupProgram& lp, bb::Ext al, const bb::Ext* d_blpow, uint32_t* perm,
                     bb::Ext* d_rowsum, bb::Ext* d_block_totals);
// the pieces of the two functions above that the run-time specialised path (prover_jit.hip) combines with its own kernels:
// inclusive scan of the row sums into the four phi columns; sum of the chunks' partial row sums -> rowsum (+ as 4 columns);
// partial quotient sums + boundary terms -> q
int logup_scan(const bb::Ext* d_rowsum, size_t H, bb::Ext* d_block_totals, uint32_t* phi_cols);
int logup_rowsum_combine(const uint32_t* part, uint32_t n_chunks, size_t H, bb::Ext* d_rowsum, uint32_t* cols4);
int quotient_logup_tail(const uint32_t* part, uint32_t n_chunks, const uint32_t* plde_phi, const uint32_t* plde_sumq, size_t N, int logN,
                        const bb::Ext* d_apow_tail, bb::Ext S, uint32_t zval_even, uint32_t zval_odd, uint32_t* q);
// main_only: only sum_k apow[k] C_k + sum_g apow[nc + g] (q_g den_g - num_g), unscaled, on `N` rows of (lde | plde) with column stride N —
// the part of the quotient that reads the current row only (the streamed path evaluates it sub-coset by sub-coset and adds the
// boundary terms with quotient_logup_tail)
int quotient_eval_logup(const uint32_t* lde, const uint32_t* plde, size_t N, int logN, const ConstraintProgram& prog,
                        const LogupProgram& lp, const bb::Ext* d_apow, bb::Ext al, const bb::Ext* d_blpow, bb::Ext S,
                        uint32_t zval_even, uint32_t zval_odd, uint32_t* q, bool main_only = false);
int deep_quotient_logup(const uint32_t* lde, uint32_t W, const uint32_t* plde, uint32_t Wp, const uint32_t* qlde, size_t N, int logN,
                        const bb::Ext* d_gpow, bb::Ext sum1, bb::Ext sum2, bb::Ext zeta, bb::Ext gzeta, bb::Ext* v);

// ---- stream_kernels.hip (the streamed proof path) ---------------------------------------------------------------------
// out[k * N + r + (i << b)] = sum_c part[(c * 4 + k) * m + i]: the partial quotient sums of sub-coset r into their rows of the N-row vector
int part_scatter(const uint32_t* part, uint32_t n_chunks, size_t m, int b, uint32_t r, size_t N, uint32_t* out);
// out (4 columns of len, + 4 more when second != 0) = sum_c g[c] ma_c + sum_c g[wa + c] mb_c  |  sum_c g[second + c] mb_c ; g centred
int ext_lincomb(const uint32_t* ma, uint32_t wa, const uint32_t* mb, uint32_t wb, size_t len, const bb::Ext* d_gpow, uint32_t second,
                uint32_t* out);
// v[j] = (G1[j] + sum_k gq[k] qlde_k[j] - sum1) / (x_j - zeta) [+ (G2[j] - sum2) / (x_j - g zeta) when two]; glde = G1 | G2 (8 columns of N)
int deep_from_combo(const uint32_t* glde, const uint32_t* qlde, size_t N, int logN, const bb::Ext* d_gpow_quotient, bb::Ext sum1, bb::Ext sum2,
                    bb::Ext zeta, bb::Ext gzeta, int two, bb::Ext* v);
int ext_to_cols(const bb::Ext* in, size_t len, uint32_t* cols4);

// proof-of-work search: smallest witness w (checked in blocks) such that the transcript state,
// after observing w, samples a value with `bits` low zero bits. state = 16 words sponge state,
// in_len = number of pending absorbed words (they are in pending[]). d_best: one device word of scratch.
int pow_grind(const uint32_t* state16, const uint32_t* pending, uint32_t in_len, uint32_t bits, uint32_t* d_best,
              uint32_t* witness_out);

}  // namespace pw
