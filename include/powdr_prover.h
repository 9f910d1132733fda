/*
 * powdr_prover.h — C ABI of the MI355X STARK prover ("pw-stark v0") in libpowdr_gpu.
 *
 * Boundary B2. The reference reaches its prover through the Rust
 * trait seam `StarkEngine::prove(pk, ProvingContext{per_trace: [(air_id,
 * AirProvingContext{cached_mains, common_main, public_values})]})`:
 *   call sites  /root/reference/openvm/src/trace_generation.rs:97-139 (callback(seg_idx, vm, pk, ctx)),
 *               /root/reference/openvm-riscv/src/lib.rs:327-341 (sdk.app_prover(exe).prove + verify_app_proof)
 *   engines     /root/reference/openvm/src/lib.rs:69-95 (BabyBearPoseidon2CpuEngine / ...GpuEngine)
 *   the AIR     /root/reference/openvm/src/powdr_extension/chip.rs:94-130 (PowdrAir::eval: current-row
 *               constraints `assert_zero(expr)`, no public values, no cached/preprocessed trace)
 * The trait's method list lives in the un-vendored `openvm-stark-backend` crate, so this
 * header is the plain-C surface a third engine `E` (beside the CPU and CUDA engines) would
 * call from its `prove`: one AIR = one prover object built from the AIR's constraint
 * programs (what keygen extracts from `PowdrAir::eval` through the symbolic builder), and
 * `pw_prover_prove` consumes the AirProvingContext's `common_main` as a column-major
 * device matrix — exactly what `PowdrChipGpu::generate_proving_ctx` returns
 * (/root/reference/openvm/src/powdr_extension/trace_generator/cuda/mod.rs:404-421).
 *
 * All `d_` pointers are device pointers owned by the caller. Field words are BabyBear in
 * Montgomery form on the device; proof words are canonical u32 (little endian).
 * Return value 0 = success, otherwise a hipError_t (or -1 for malformed arguments).
 */
#ifndef POWDR_PROVER_H
#define POWDR_PROVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PwProver PwProver;

typedef struct {
    uint32_t num_queries; /* FRI queries (blow-up is fixed at 2: constraint degree <= 3) */
    uint32_t pow_bits;    /* proof-of-work bits before the query phase, 0 = none */
} PwStarkConfig;

/* Constraint programs: post-fix bytecode with the trace-generation opcodes
 * (PUSH_APC=0 with a COLUMN INDEX operand, PUSH_CONST=1, ADD=2, SUB=3, MUL=4, NEG=5),
 * spans = {off, len} pairs in u32 words. Host pointers; copied. Returns NULL for a malformed program — a span past the bytecode, an
 * unknown opcode, an unbalanced or too deep stack, a column index >= width (the same for pw_prover_create_logup's interaction
 * programs) — and when the device tables cannot be allocated. */
PwProver* pw_prover_create(const PwStarkConfig* cfg, uint32_t width, const uint32_t* cons_bytecode,
                           size_t bytecode_len, const uint32_t* cons_spans, size_t n_constraints);
/* The same AIR including its bus interactions (PowdrAir::eval's `push_interaction`, chip.rs:117-129), proven
 * with committed LogUp columns ("pw-stark v0 + LogUp", oracle/stark_oracle.cpp): interactions = n x {bus id,
 * n_args, first span index}, spans = {off,len} pairs laid out [mult, arg0, arg1, ...] per interaction into
 * `inter_bytecode` (post-fix, column-index operands) — i.e. powdr_apc_compile_bus(apc, 1, ...). The proof then
 * also carries the permutation-matrix commitment, the cumulative sum S and the extra openings. */
PwProver* pw_prover_create_logup(const PwStarkConfig* cfg, uint32_t width, const uint32_t* cons_bytecode,
                                 size_t bytecode_len, const uint32_t* cons_spans, size_t n_constraints,
                                 const uint32_t* interactions, size_t n_interactions, const uint32_t* inter_spans,
                                 size_t n_inter_spans, const uint32_t* inter_bytecode, size_t inter_bytecode_len);
/* LogUp across the AIRs of a segment: the challenges of the bus argument are drawn from the 8-word `bus seed`
 * alone, so every AIR proven with the same seed uses the same pair and the per-AIR cumulative sums add up.
 * Flow: pw_prover_trace_root for every AIR -> seed = a digest over all roots (the caller's choice; the segment
 * verifier must recompute it) -> pw_prover_set_bus_seed + pw_prover_prove per AIR. Without a seed (NULL, the
 * default) an AIR uses its own trace root. Canonical words. */
int pw_prover_trace_root(PwProver* p, const uint32_t* d_trace, uint32_t log_height, uint32_t* root8);
int pw_prover_set_bus_seed(PwProver* p, const uint32_t* seed8);
/* How the LogUp kernels evaluate this AIR's multiplicities and arguments: 0 = no LogUp extension, 1 = the bytecode
 * interpreter, 2 = "small forms" (k0 + k1 A + k2 B + k3 A B over at most two columns — fixed code; what the interactions of
 * optimised APCs look like; the few wider expressions of such an AIR still go through the interpreter). Chosen when at least
 * half of the expressions are small forms; POWDR_LOGUP_INTERPRET=1 at creation forces 1 (tests). */
int pw_prover_logup_path(const PwProver* p);
void pw_prover_destroy(PwProver* p);

/* Prove one trace (column-major, width x 2^log_height, Montgomery words, device).
 * *proof_words points at host memory owned by the prover, valid until the next call. */
int pw_prover_prove(PwProver* p, const uint32_t* d_trace, uint32_t log_height, const uint32_t** proof_words,
                    size_t* n_words);

/* The same proof (the same words) of a trace the caller HANDS OVER — what the reference does with `common_main`: the
 * trace generator moves the matrix into the AirProvingContext and the engine owns it from there
 * (/root/reference/openvm/src/powdr_extension/trace_generator/cuda/mod.rs:404-421). The prover may then overwrite
 * d_trace: when the low-degree extension is resident it does not; when the proof is STREAMED (pw_prover_stream_log_blocks)
 * the trace's coefficient arrays replace the trace in place instead of living in a buffer of their own — at BASELINE
 * configs[2] (3 731 x 2^22 + 4 632 permutation columns) that is 62.6 GB, the difference between walking the extended
 * domain as 4 sub-cosets and as 2 (half the coefficient re-reads of every pass). Afterwards d_trace holds, per column,
 * H * (the coefficients) in bit-reversed order; pw_trace_from_coefficients turns that back into the trace (exact).
 * d_trace must be 16-byte aligned (hipErrorInvalidValue otherwise; pw_prover_prove takes any 4-byte aligned trace).
 * pw_prover_stream_log_blocks_consuming: the mode such a proof would run in with the memory free now. */
int pw_prover_prove_consuming(PwProver* p, uint32_t* d_trace, uint32_t log_height, const uint32_t** proof_words,
                              size_t* n_words);
int pw_prover_stream_log_blocks_consuming(const PwProver* p, uint32_t log_height);
/* In place: width columns of 2^log_height H-scaled bit-reversed coefficients (what a streamed pw_prover_prove_consuming
 * leaves) -> the values on the trace domain, natural row order, canonical Montgomery words. d_scratch: 2^13 words. */
int pw_trace_from_coefficients(uint32_t* d_coeffs, uint32_t width, uint32_t log_height, uint32_t* d_scratch); /* d_coeffs 16-byte aligned */

/* "Mock prover": evaluate every constraint on every row of the trace on the device and report violations —
 * the counterpart of the reference's `debug_proving_ctx` used by its `prove_mock` tests
 * (openvm-riscv/src/lib.rs:288-294). *n_violations = number of (row, constraint) pairs that are non-zero;
 * if any, *first_row / *first_constraint name the first one in row-major order. */
int pw_prover_check_constraints(PwProver* p, const uint32_t* d_trace, uint32_t log_height, uint64_t* n_violations,
                                uint64_t* first_row, uint32_t* first_constraint);

/* Verify a proof on the host (no GPU): the counterpart of the reference's CPU verification step
 * `verify_app_proof::<BabyBearPoseidon2CpuEngine>` (openvm-riscv/src/lib.rs:337-341). Constraint
 * programs as for pw_prover_create (post-fix, column-index operands, no INV_OR_ZERO).
 * Returns 0 = valid; 1 header, 2 constraint identity, 3 proof of work, 4 query index, 5 trace opening,
 * 6 quotient opening, 7 FRI layer, 8 final polynomial, 9 trailing words, 10 truncated/malformed,
 * 13 a proof word >= p (non-canonical encodings are rejected: proofs are not malleable). */
int pw_verify(const PwStarkConfig* cfg, uint32_t width, uint32_t log_height, const uint32_t* cons_bytecode,
              size_t bytecode_len, const uint32_t* cons_spans, size_t n_constraints, const uint32_t* proof_words,
              size_t n_words);

/* Verify a "pw-stark v0 + LogUp" proof (pw_prover_create_logup). Interaction arguments as for the prover.
 * Additional failure codes: 11 = permutation-matrix opening, 12 = the proof's bus seed is not
 * `expected_bus_seed` (NULL: not the proof's own trace root). On success `cumulative_sum` (4 canonical words,
 * may be NULL) receives this AIR's bus sum S = sum_rows sum_i m_i / d_i and `trace_root` (8 words, may be NULL)
 * its trace commitment; the caller accepts a segment when every proof verifies against the seed recomputed
 * from all the trace roots and the sums add up to zero (every send matched by a receive). */
int pw_verify_logup(const PwStarkConfig* cfg, uint32_t width, uint32_t log_height, const uint32_t* cons_bytecode,
                    size_t bytecode_len, const uint32_t* cons_spans, size_t n_constraints, const uint32_t* interactions,
                    size_t n_interactions, const uint32_t* inter_spans, size_t n_inter_spans,
                    const uint32_t* inter_bytecode, size_t inter_bytecode_len, const uint32_t* expected_bus_seed,
                    const uint32_t* proof_words, size_t n_words, uint32_t* cumulative_sum, uint32_t* trace_root);

/* How pw_prover_create_logup / pw_verify_logup pack the interactions into committed columns: consecutive
 * interactions share one extension column ("group") while the group's constraint keeps degree <= 3. Writes up to
 * `cap` group boundaries (interaction indices) to `out`, returns their number (n_groups + 1; 0 = malformed table).
 * The permutation matrix has 4 * (n_groups + 1) base columns. */
size_t pw_logup_group_starts(const uint32_t* interactions, size_t n_interactions, const uint32_t* inter_spans,
                             size_t n_inter_spans, const uint32_t* inter_bytecode, size_t inter_bytecode_len,
                             uint32_t* out, size_t cap);

/* ---- one segment = many AIRs -------------------------------------------------------------------------------------
 * The engine call the reference makes once per segment with all chips' traces, `engine.prove(pk, ProvingContext{
 * per_trace})` (openvm/src/trace_generation.rs:97-139, openvm-riscv/src/lib.rs:327-341). */
#define PW_AIR_HAND_OVER 1u /* PwSegmentAir::flags: the trace is the engine's to overwrite (pw_prove_segment_consuming) */
typedef struct PwSegmentAir {
    PwProver* prover;        /* one per AIR (pw_prover_create / _create_logup), reused from segment to segment */
    const uint32_t* d_trace; /* device, column-major width x 2^log_height, Montgomery */
    uint32_t log_height;
    uint32_t flags;          /* read by pw_prove_segment_consuming only (the struct's former padding: size and offsets unchanged) */
} PwSegmentAir;

/* ONE proof for all AIRs of the segment ("pw-stark v1", proof magic PWS3; protocol: oracle/stark_segment.inc) — the
 * shape of the reference's call: all matrices of a phase are committed in one mixed-height Poseidon2 tree, the openings of
 * all AIRs are reduced per height and checked by one FRI over the tallest domain, one query phase answers for everything.
 * logup != 0: every prover must come from pw_prover_create_logup (an AIR without interactions passes n_interactions = 0);
 * the bus challenges are drawn once from the segment transcript and the proof carries every AIR's cumulative sum.
 * Stream contract: the call runs on the calling thread's launch stream (powdr_gpu_set_stream), after everything the
 * caller enqueued there — trace generation included. *proof_words is owned by the library (per host thread) and valid
 * until that thread's next pw_prove_segment. Returns 0 or the first error. */
int pw_prove_segment(const PwSegmentAir* airs, size_t n_airs, int logup, const uint32_t** proof_words, size_t* n_words);

/* The same proof, same words, with the traces HANDED OVER per AIR (flags & PW_AIR_HAND_OVER) — the reference's chips move
 * `common_main` into the engine for every AIR of a segment (openvm/src/powdr_extension/trace_generator/cuda/mod.rs:404-421).
 * An AIR that is proven with its LDE resident leaves its trace alone; an AIR that is STREAMED (its extension does not fit beside
 * the others) keeps the coefficient arrays of a handed-over trace IN the caller's buffer instead of a copy of its own (configs[2]:
 * 62.6 GB, two sub-cosets instead of four). After the call such a buffer holds H-scaled bit-reversed coefficients
 * (pw_trace_from_coefficients restores the trace, exactly); handed-over traces must be 16-byte aligned (hipErrorInvalidValue).
 * pw_segment_last_modes: per AIR of the calling thread's last segment proof, log2(#sub-cosets) (0 = resident) | 0x100 if the
 * trace was overwritten; returns the number of AIRs. */
int pw_prove_segment_consuming(const PwSegmentAir* airs, size_t n_airs, int logup, const uint32_t** proof_words, size_t* n_words);
size_t pw_segment_last_modes(uint32_t* out, size_t cap);
/* The memory plan of the calling thread's last segment proof (bytes; zero when the mode was forced by POWDR_STREAM_LOG_BLOCKS):
 * with every AIR resident | as chosen | what the policy had to work with (free + held, head room, pw_set_device_budget). */
void pw_segment_last_plan(size_t* resident_bytes, size_t* planned_bytes, size_t* available_bytes);

/* Device memory the provers of this process may plan for (bytes; 0 = no limit beyond what the device has free — the default, or
 * POWDR_DEVICE_BUDGET_BYTES read once). A proof whose resident buffers would exceed it runs streamed (one-AIR proofs: pw_prover_prove;
 * segments: the largest AIRs first), exactly as when the device itself is short: an embedder that shares a GPU between engines
 * sets this instead of relying on hipMemGetInfo at the moment of the call. */
void pw_set_device_budget(size_t bytes);
size_t pw_get_device_budget(void);

/* INDEPENDENT proofs, one per AIR (v0 / v0+LogUp), proven concurrently: `n_workers` host threads (0 = 4), each with its
 * own HIP stream on the caller's device, take the AIRs largest first; n_workers = 1 runs inline on the calling thread's
 * stream. This is the flow AIR-level sharding over several GPUs uses (every rank proves its AIRs; powdr_amd/sharding.py).
 * shared_bus_seed != 0: every prover must come from pw_prover_create_logup; phase 1 commits all traces, the bus seed is
 * pw_commitment_digest over the trace roots in AIR order, phase 2 proves every AIR with it (each prover reuses its
 * phase-1 LDE and tree). proofs[i] / n_words[i] are owned by airs[i].prover and valid until its next proof; bus_seed8
 * (may be NULL) receives the seed. Stream contract: the worker streams wait for everything the caller enqueued on its
 * launch stream before the call. Returns 0 or the first error. */
int pw_prove_airs(const PwSegmentAir* airs, size_t n_airs, int shared_bus_seed, unsigned n_workers,
                  const uint32_t** proofs, size_t* n_words, uint32_t* bus_seed8);

typedef struct PwAirDescription {
    uint32_t width, log_height;
    uint32_t logup;                                          /* 0: constraints-only ("PWS1") proof, interaction fields unused */
    const uint32_t* cons_bytecode; size_t bytecode_len;
    const uint32_t* cons_spans; size_t n_constraints;
    const uint32_t* interactions; size_t n_interactions;
    const uint32_t* inter_spans; size_t n_inter_spans;
    const uint32_t* inter_bytecode; size_t inter_bytecode_len;
} PwAirDescription;

/* Host verification (no GPU) of a pw_prove_segment proof: the counterpart of `verify_app_proof` for a whole segment.
 * airs[i].logup is ignored (the segment's `logup` flag applies to all). Returns 0 = valid; ((air index + 1) << 8) | 2 =
 * constraint identity of that AIR; 1 header / shape mismatch, 3 proof of work, 4 query index, 5 main opening,
 * 6 quotient opening, 7 FRI layer, 8 final value, 9 trailing words, 10 truncated, 11 permutation opening, 13 a word >= p,
 * 14 check_balance is set and the AIRs' cumulative bus sums do not add up to zero, 15 malformed description.
 * total_sum4 (may be NULL) receives the sum of the cumulative sums. */
int pw_verify_segment(const PwStarkConfig* cfg, const PwAirDescription* airs, size_t n_airs, int logup,
                      const uint32_t* proof_words, size_t n_words, int check_balance, uint32_t* total_sum4);

/* Host verification of pw_prove_airs' proofs. With shared_bus_seed the seed is recomputed from the trace roots inside the
 * proofs and every proof must have used it. Returns 0; ((air index + 1) << 8) | code of the first failing proof
 * (codes as pw_verify / pw_verify_logup); or 14 when check_balance is set and the cumulative bus sums of all AIRs
 * do not add up to zero. total_sum4 (may be NULL) receives that sum. */
int pw_verify_airs(const PwStarkConfig* cfg, const PwAirDescription* airs, size_t n_airs,
                   const uint32_t* const* proofs, const size_t* n_words, int shared_bus_seed, int check_balance,
                   uint32_t* total_sum4);

/* ---- independent segments over the GPUs of one node ------------------------------------------------------------------
 * The reference proves an execution's segments one after the other on one device (openvm/src/trace_generation.rs:111-141);
 * they are independent once metered execution has fixed their boundaries. pw_prove_segments_multi runs that loop on
 * `n_workers` host threads at once: worker w makes devices[w] current, gets a launch stream of its own and proves the
 * segments placed on it — by `segment_cells` (rows x columns), largest first, each to the least loaded worker
 * (pw_assign_units) — by calling `prove(user, segment, worker, device, commitment8)`: the caller's per-worker replica of the
 * segment pipeline (trace generation into that worker's buffers, pw_prove_segment on that worker's provers; a prover belongs
 * to the device it was created on) which returns the segment's 8-word main commitment (canonical; proof words 5 + 4 n_airs
 * .. + 8 of a pw_prove_segment proof) and keeps or ships the proof itself. No data-path collective. Afterwards the FINAL
 * COMMITMENT MERGE: the commitments are all-gathered over RCCL (one rank per distinct device; workers may share a device) and
 * `commitments` (n_segments x 8 words, host) receives the segment-ordered list every device now holds. RCCL is loaded with
 * dlopen; without it (or POWDR_MULTI_NO_RCCL=1) the merge happens on the host — pw_multi_last_merge(): 1 = RCCL, 2 = host.
 * worker_of_segment (may be NULL) receives who proved each segment in the end. The placement by cells is the PLAN (pw_assign_units):
 * a worker whose queue runs dry steals the smallest unstarted segment of the worker with the most cells still queued — proving time is
 * not proportional to cells (short tails, streamed AIRs) and devices differ by a few percent; POWDR_MULTI_STEAL=0 keeps the plan.
 * Returns 0, the first error of a worker, or a hipError_t. */
typedef int (*PwSegmentProveFn)(void* user, size_t segment, size_t worker, int device, uint32_t* commitment8);
int pw_prove_segments_multi(const int* devices, size_t n_workers, const uint64_t* segment_cells, size_t n_segments,
                            PwSegmentProveFn prove, void* user, uint32_t* commitments, uint32_t* worker_of_segment);
int pw_multi_last_merge(void);
/* Largest-first greedy balance of `n_units` proof units over `n_workers` (ties: lower unit index, lower worker index first);
 * worker_of_unit[u] = the worker unit u is placed on. Returns n_units (0: malformed arguments). */
size_t pw_assign_units(const uint64_t* cells, size_t n_units, size_t n_workers, uint32_t* worker_of_unit);

/* Digest over an ordered list of 8-word commitments (binary Poseidon2 tree; canonical words). */
void pw_commitment_digest(const uint32_t* roots8, size_t n, uint32_t* digest8);

/* Set-up for proofs of 2^log_height-row traces, so that the first pw_prover_prove pays for neither: (1) the run-time specialised
 * kernels are compiled now if the height policy would compile them at the first proof (seconds of host time for a keccak-sized AIR;
 * their partial-sum buffer is part of the reservation), (2) every device buffer such a proof needs is allocated (tens of gigabytes
 * for wide traces) — in the mode the memory free NOW allows (resident, or streamed: see below). Buffers only grow; calling it is
 * optional. Returns hipErrorOutOfMemory when not even the streamed buffers fit. */
int pw_prover_reserve(PwProver* p, uint32_t log_height);

/* STREAMED proofs. A resident proof keeps the low-degree extension of every committed column in HBM (8 bytes per committed cell
 * next to the caller's trace); BASELINE configs[2] — the reference's default segment height 2^22
 * (/root/reference/openvm-riscv/src/lib.rs:366-371) with the bus interactions PowdrAir::eval always pushes
 * (/root/reference/openvm/src/powdr_extension/chip.rs:117-129) — would need 280 GB for 3 731 + 4 632 columns. When that does not
 * fit, pw_prover_prove / pw_prover_trace_root / pw_prover_reserve switch to the streamed mode by themselves: the prover keeps the
 * columns' COEFFICIENT arrays (4 bytes per committed cell) and walks the extended domain as 2^b sub-cosets (rows r + 2^b i),
 * rebuilding the LDE rows of all columns for one sub-coset at a time — for the leaf hashes of the commitments, for the quotient and
 * for the query answers. Same proof words as the resident mode. POWDR_STREAM_LOG_BLOCKS=0 forces resident, =b (>= 1) streamed.
 * pw_prover_stream_log_blocks: the mode a proof of a 2^log_height-row trace would run in with the memory free NOW — 0 resident,
 * b >= 1 streamed over 2^b sub-cosets, -1 not even that fits. */
int pw_prover_stream_log_blocks(const PwProver* p, uint32_t log_height);

/* Highest degree (in the trace columns) among the constraint programs the prover was created with; 99 if one of them
 * is malformed or not polynomial. The blow-up-2 quotient carries degree <= 3 — the reference's bound
 * 2 * DEFAULT_APP_LOG_BLOWUP + 1 (openvm/src/lib.rs:97-101); a prover with a higher value produces proofs that do not
 * verify, so a key generator checks this once. */
int pw_prover_max_constraint_degree(const PwProver* p);

/* Run-time specialised expression kernels. The constraint and interaction programs of a prover are wave-uniform bytecode that
 * the ahead-of-time kernels interpret; for an AIR that is proven segment after segment the library can instead emit them as
 * straight-line HIP, compile that with hiprtc for gfx950 (translation units concurrently on host threads; hiprtc is loaded
 * with dlopen, the library works without it) and run the code objects for the quotient and the LogUp permutation columns.
 * Same proof words either way. By default a prover is specialised at its first proof of a trace of at least 2^18 rows
 * (POWDR_JIT_MIN_LOG_HEIGHT), which costs seconds of host time once; POWDR_JIT=0 never, POWDR_JIT=1 at every first proof.
 * pw_prover_specialise does it now (set-up time, like pw_prover_reserve): 0 = the prover has specialised kernels, 1 = it
 * keeps the interpreter (no hiprtc, POWDR_JIT=0, programs that did not compile to xbc, a compiler error).
 * pw_prover_specialised returns the state (1 specialised, 0 not tried yet, -1 interpreter only) and the number of compiled
 * kernels, their code-object bytes and the number of code chunks (each NULL = skip). */
int pw_prover_specialise(PwProver* p);
/* The same for n provers in ONE concurrent compile batch, whatever their traces' heights (an AIR set that is proven segment after
 * segment is compiled once, at set-up: the reference fixes an APC's AIR at key generation). Returns how many of them run specialised
 * kernels afterwards. Short traces gain most: an interpreted expression kernel walks the AIR's whole program in every lane — ~1.2 ms
 * per launch for a 2 000-column AIR however few rows it has (profiles/r06_tail_segment_c5.txt); measured on the reth-shaped segments
 * (61 AIRs): 5.17 -> 5.41 G cells/s for 145 s of cold compilation. */
size_t pw_provers_specialise(PwProver* const* provers, size_t n);
int pw_prover_specialised(const PwProver* p, size_t* n_kernels, size_t* code_bytes, size_t* n_chunks);
/* Test hook: the HIP source the generator emits for translation unit `unit` of an AIR's specialised kernels (which: 0 = quotient
 * numerator, 1 = LogUp permutation columns; chunk_cost / chunks_per_unit 0 = the defaults), without compiling it: lets a CPU-only
 * test build the generated code for the HOST and execute it. Returns the source length (0: no such unit). */
size_t pw_jit_generated_source(uint32_t width, const uint32_t* cons_bytecode, size_t bytecode_len, const uint32_t* cons_spans, size_t n_constraints,
                               const uint32_t* interactions, size_t n_interactions, const uint32_t* inter_spans, size_t n_inter_spans,
                               const uint32_t* inter_bytecode, size_t inter_bytecode_len, int which, uint32_t chunk_cost, uint32_t chunks_per_unit,
                               size_t unit, char* buf, size_t cap, char* kernel_name, size_t name_cap, uint32_t* first_chunk, uint32_t* n_chunks,
                               uint32_t* total_chunks);
/* Compiled code objects are kept on disk across processes: $POWDR_JIT_CACHE_DIR, else $XDG_CACHE_HOME/powdr_jit, else
 * $HOME/.cache/powdr_jit (POWDR_JIT_CACHE=0: off); an entry is keyed by the unit's source, the embedded headers and the compile options
 * and confirmed by comparing the stored source. Translation units this process compiled / loaded from disk so far: */
void pw_jit_cache_stats(uint64_t* units_compiled, uint64_t* units_from_disk);
/* The same code generation + hiprtc compilation for an AIR given by its tables (as for pw_prover_create / _create_logup;
 * interactions == NULL: constraints only) WITHOUT touching a GPU — hiprtc cross-compiles — so build machines and CPU test
 * suites can check that an AIR's specialised kernels compile. Returns pw_prover_specialise's code (-2: malformed tables);
 * err (may be NULL) receives the compiler's message. */
int pw_jit_compile_check(uint32_t width, const uint32_t* cons_bytecode, size_t bytecode_len, const uint32_t* cons_spans, size_t n_constraints,
                         const uint32_t* interactions, size_t n_interactions, const uint32_t* inter_spans, size_t n_inter_spans,
                         const uint32_t* inter_bytecode, size_t inter_bytecode_len, size_t* n_kernels, size_t* code_bytes, size_t* n_chunks,
                         char* err, size_t err_cap);

/* Number of main-trace columns the prover was created for. */
uint32_t pw_prover_width(const PwProver* p);

/* Bytes of device memory the prover currently holds. */
size_t pw_prover_device_bytes(const PwProver* p);

/* ---- single stages, exposed for parity tests and per-stage measurement ---- */

/* d_coeffs (width x H) receives H-scaled coefficients in bit-reversed order; d_lde (width x 2H)
 * receives natural-order evaluations on the coset 31 * <g_{n+1}>. */
int pw_lde_batch(const uint32_t* d_trace, uint32_t width, uint32_t log_height, uint32_t* d_coeffs, uint32_t* d_lde);

/* The same LDE the way the provers run it: three passes over HBM instead of four — the contiguous stage groups of the
 * inverse and of the forward transform share one kernel, the coefficient array is never written (d_tmp: width x H words of
 * scratch for the strided stages of traces taller than 2^12 rows; its contents are unspecified afterwards). */
int pw_lde_fused(const uint32_t* d_trace, uint32_t width, uint32_t log_height, uint32_t* d_tmp, uint32_t* d_lde);

/* One sub-coset of the LDE from coefficient arrays (the stage the streamed mode is built on): d_coeffs as pw_lde_batch leaves them
 * (width x H, bit-reversed, H-scaled); d_out (width x 2H / 2^log_blocks) receives the rows r + 2^log_blocks * i of the LDE, i.e. the
 * evaluations on (31 g_(n+1)^r) <g_(n+1)^(2^log_blocks)>; d_scale: H words of scratch. 1 <= log_blocks <= min(log_height, 5). */
/* d_coeffs must be 16-byte aligned (hipErrorInvalidValue otherwise). */
int pw_lde_subcoset(const uint32_t* d_coeffs, uint32_t width, uint32_t log_height, uint32_t log_blocks, uint32_t r, uint32_t* d_scale,
                    uint32_t* d_out);

/* Poseidon2 Merkle tree of a column-major matrix; d_digests gets (2*height - 1) * 8 words,
 * leaves first, root last. */
int pw_merkle_commit(const uint32_t* d_matrix, size_t height, uint32_t width, uint32_t* d_digests);

/* Host-side Poseidon2 permutation used by the transcript (canonical words in/out). This is also the known-answer-test
 * hook: after pw_set_poseidon2_constants a maintainer feeds it the vector of the backend's own Poseidon2 test. */
void pw_poseidon2_permute_host(uint32_t* state16);

/* The hash parameters are ONE table installed at run time. The shape is fixed (width 16, x^7, 4 + 13 + 4 rounds, external
 * layer circ(2 M4, M4, M4, M4) with M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]], internal layer 1 1^T + diag(-2, 1, 2,
 * 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27, -2^-8, -1/16, -2^-27) — the shape of p3-baby-bear's width-16 instance);
 * the ROUND CONSTANTS of the reference's permutation live in the un-vendored crate p3-baby-bear 0.5.2
 * (/root/reference/number/Cargo.toml:16-19: BABYBEAR_RC16_EXTERNAL_INITIAL / _FINAL / _INTERNAL), so the library starts with a
 * documented placeholder stream and takes the real table here: ext_rc = 8 x 16 canonical words (the four initial rounds,
 * then the four final ones), int_rc = 13 canonical words. Both NULL = back to the placeholder. Every derived table (folded
 * constants, per-stage scales of the device kernels) is rebuilt; host transcript, host verifiers and the device kernels of
 * every GPU use the new set from the next call on. Call it while no proof is in flight (the device copy is replaced after a
 * device-wide synchronisation); proofs made under different tables do not verify against each other. Returns 0, or -1 for a
 * word >= p / one NULL pointer. */
int pw_set_poseidon2_constants(const uint32_t* ext_rc, const uint32_t* int_rc);
/* The table in use (canonical words): 128 + 13 round constants and the 16 internal-diagonal entries; NULL = skip. */
void pw_get_poseidon2_constants(uint32_t* ext_rc, uint32_t* int_rc, uint32_t* diag);

#ifdef __cplusplus
}
#endif
#endif /* POWDR_PROVER_H */
