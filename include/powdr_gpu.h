/*
 * powdr_gpu.h — C ABI of libpowdr_gpu (MI355X / gfx950 build).
 *
 * Boundary B1: this header declares, byte for byte, the three
 * entry points powdr's Rust host binds in
 *     /root/reference/openvm/src/cuda_abi.rs:8-64      (extern "C" block)
 * and the #[repr(C)] structs of
 *     cuda_abi.rs:66-95 (OriginalAir, Subst, DerivedExprSpec)
 *     cuda_abi.rs:137-169 (OpCode, DevInteraction, ExprSpan)
 * which the reference implements in CUDA at
 *     openvm/cuda/src/apc_tracegen.cu:106-146, apc_apply_bus.cu:119-169.
 * The library must be named `powdr_gpu` (reference openvm/build.rs:16).
 *
 * All pointers are DEVICE pointers owned by the caller; the callee never frees
 * them. Return value: 0 = success, otherwise a hipError_t cast to int (the
 * reference returns `(int)cudaGetLastError()`; Rust maps it through
 * `CudaError::from_result`).
 *
 * Field elements (`Fp` in the reference, `BabyBear` on the Rust side) are
 * 32-bit words in Montgomery form, R = 2^32, p = 0x78000001 (assumption A1 of
 * SURVEY.md). PUSH_CONST operands and histogram indices are canonical.
 */
#ifndef POWDR_GPU_H
#define POWDR_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint32_t PowdrFp; /* BabyBear, Montgomery form */

/* cuda_abi.rs:66-73 / apc_tracegen.cu:10-15 — 24 bytes, pointer at offset 8 */
typedef struct {
    int32_t width;          /* number of columns */
    int32_t height;         /* number of rows (Ha) */
    const PowdrFp* buffer;  /* column-major base: col*height + row (device) */
    int32_t row_block_size; /* stride (in rows) between consecutive APC calls */
} OriginalAir;

/* cuda_abi.rs:75-86 / apc_tracegen.cu:17-22 — 16 bytes */
typedef struct {
    int32_t air_index; /* index into d_original_airs */
    int32_t col;       /* source column within this AIR */
    int32_t row;       /* base row offset within the row-block */
    int32_t apc_col;   /* destination APC column */
} Subst;

/* cuda_abi.rs:162-169 / expr_eval.cuh:91-95 — 8 bytes */
typedef struct {
    uint32_t off; /* offset (u32 words) into the bytecode buffer */
    uint32_t len; /* length (u32 words) of the expression */
} ExprSpan;

/* cuda_abi.rs:88-95 / apc_tracegen.cu:24-29 — 16 bytes */
typedef struct {
    uint64_t col_base; /* apc_col_index * H */
    ExprSpan span;
} DerivedExprSpec;

/* cuda_abi.rs:149-160 / apc_apply_bus.cu:11-17 — 12 bytes */
typedef struct {
    uint32_t bus_id;
    uint32_t num_args;
    uint32_t args_index_off; /* into ExprSpan[]: [mult, arg0, arg1, ...] */
} DevInteraction;

/* cuda_abi.rs:137-147 / expr_eval.cuh:12-20 */
enum PowdrOpCode {
    POWDR_OP_PUSH_APC = 0,   /* followed by element offset col*H */
    POWDR_OP_PUSH_CONST = 1, /* followed by canonical u32 < p */
    POWDR_OP_ADD = 2,
    POWDR_OP_SUB = 3,
    POWDR_OP_MUL = 4,
    POWDR_OP_NEG = 5,
    POWDR_OP_INV_OR_ZERO = 6
};
#define POWDR_EXPR_STACK_CAPACITY 16 /* expr_eval.cuh:22 */
#define POWDR_BITWISE_NUM_BITS 8     /* apc_apply_bus.cu:20 */

/* cuda_abi.rs:12-19 ⇄ apc_tracegen.cu:126-146.
 * out[apc_col*H + r] = r < num_apc_calls
 *        ? air.buffer[col*air.height + row + r*air.row_block_size] : 0
 * for every Subst and every r < output_height. output_height must be a power
 * of two (0 allowed); otherwise hipErrorInvalidValue is returned (the
 * reference aborts through assert). */
int _apc_tracegen(PowdrFp* d_output, size_t output_height,
                  const OriginalAir* d_original_airs, const Subst* d_subs,
                  size_t n_subs, int num_apc_calls);

/* cuda_abi.rs:24-31 ⇄ apc_tracegen.cu:106-124. Sequential over derived
 * columns per row; rows >= num_apc_calls are zero-filled; n_cols == 0 → 0. */
int _apc_apply_derived_expr(PowdrFp* d_output, size_t output_height,
                            int num_apc_calls, const DerivedExprSpec* d_specs,
                            size_t n_cols, const uint32_t* d_bytecode);

/* cuda_abi.rs:36-63 ⇄ apc_apply_bus.cu:119-169. num_apc_calls <= 0 → 0. */
int _apc_apply_bus(const PowdrFp* d_output, int num_apc_calls,
                   const uint32_t* d_bytecode, size_t bytecode_len,
                   const DevInteraction* d_interactions, size_t n_interactions,
                   const ExprSpan* d_arg_spans, size_t n_arg_spans,
                   uint32_t var_range_bus_id, uint32_t* d_var_hist,
                   size_t var_num_bins, uint32_t tuple2_bus_id,
                   uint32_t* d_tuple2_hist, uint32_t tuple2_sz0,
                   uint32_t tuple2_sz1, uint32_t bitwise_bus_id,
                   uint32_t* d_bitwise_hist);

/* ---- extensions (not part of the reference ABI) ------------------------- */

/* The reference encodes PUSH_APC operands as u32 element offsets col*H (cuda/mod.rs:61-63), which cannot
 * address a trace with width*height >= 2^32 (e.g. 3 731 columns x 2^22 rows). These variants take
 * COLUMN INDICES as PUSH_APC operands (cell = d_output[operand * output_height + r]); everything else
 * is as in _apc_apply_derived_expr / _apc_apply_bus. */
int powdr_apc_apply_derived_expr_cols(PowdrFp* d_output, size_t output_height, int num_apc_calls,
                                      const DerivedExprSpec* d_specs, size_t n_cols, const uint32_t* d_bytecode);
int powdr_apc_apply_bus_cols(const PowdrFp* d_output, size_t output_height, int num_apc_calls,
                             const uint32_t* d_bytecode, size_t bytecode_len,
                             const DevInteraction* d_interactions, size_t n_interactions,
                             const ExprSpan* d_arg_spans, size_t n_arg_spans,
                             uint32_t var_range_bus_id, uint32_t* d_var_hist, size_t var_num_bins,
                             uint32_t tuple2_bus_id, uint32_t* d_tuple2_hist, uint32_t tuple2_sz0,
                             uint32_t tuple2_sz1, uint32_t bitwise_bus_id, uint32_t* d_bitwise_hist);

/* _apc_tracegen for a caller that still holds the tables on the host (the reference's host code builds them right
 * before uploading them, cuda/mod.rs:272-332): `h_original_airs` / `h_subs` are host copies of what `d_original_airs`
 * (still needed: the kernels read buffer pointers and heights from it) and the Subst table contain. The reference
 * entry point has to copy both tables back and synchronise the stream twice to find its cached gather plan; this
 * one only enqueues kernels. Same result. With n_airs <= 16 the records travel as kernel arguments and
 * d_original_airs may be NULL (no device table at all); every Subst must name an AIR < n_airs. */
int powdr_apc_tracegen_host_tables(PowdrFp* d_output, size_t output_height, const OriginalAir* d_original_airs,
                                   const OriginalAir* h_original_airs, size_t n_airs, const Subst* h_subs,
                                   size_t n_subs, int num_apc_calls);

/* _apc_apply_bus / powdr_apc_apply_bus_cols for a caller that still holds the three tables on the host: h_* are host
 * copies of the d_* tables. output_height = 0: PUSH_APC operands are element offsets (reference encoding); otherwise
 * they are column indices of a trace of that height. */
int powdr_apc_apply_bus_host_tables(const PowdrFp* d_output, size_t output_height, int num_apc_calls,
                                    const uint32_t* d_bytecode, const uint32_t* h_bytecode, size_t bytecode_len,
                                    const DevInteraction* d_interactions, const DevInteraction* h_interactions,
                                    size_t n_interactions, const ExprSpan* d_arg_spans, const ExprSpan* h_arg_spans,
                                    size_t n_arg_spans, uint32_t var_range_bus_id, uint32_t* d_var_hist,
                                    size_t var_num_bins, uint32_t tuple2_bus_id, uint32_t* d_tuple2_hist,
                                    uint32_t tuple2_sz0, uint32_t tuple2_sz1, uint32_t bitwise_bus_id,
                                    uint32_t* d_bitwise_hist);

/* SURVEY.md §8 row (f)-1, the layout half: the gather for sources that arrive CALL-MAJOR AND COMPACTED — per original AIR
 * a buffer holding, for every APC call r, only the cells the APC uses, contiguous: buffer[r * cells_per_call + slot].
 * out[apc_col * H + r] = r < num_apc_calls ? air[air_index].buffer[r * cells_per_call + slot] : 0; duplicate apc_col:
 * the last SubstCM wins. An original chip would produce this layout by writing, per record, only the cells named by the
 * APC's (row, column) -> slot map instead of its full rows (the reference materialises full column-major traces,
 * cuda/mod.rs:228-253, of which an optimised APC keeps a few percent). Tables are HOST arrays (buffers inside are device
 * pointers); at most 16 AIRs; a slot feeds ONE APC column (two different apc_col for the same (air, slot) are rejected with
 * hipErrorInvalidValue: the producer writes such a cell into two slots). The reference layout stays served by _apc_tracegen. */
typedef struct {
    const PowdrFp* buffer;  /* device: num_apc_calls x cells_per_call, Montgomery */
    int32_t cells_per_call;
    int32_t reserved;
} PowdrCallMajorAir;
typedef struct {
    int32_t air_index, slot, apc_col;
} PowdrSubstCM;
int powdr_apc_tracegen_callmajor(PowdrFp* d_output, size_t output_height, const PowdrCallMajorAir* h_airs, size_t n_airs,
                                 const PowdrSubstCM* h_subs, size_t n_subs, int num_apc_calls);

/* SURVEY.md §8 row (f)-1, the PRODUCER half: the original RV32IM chips an autoprecompile is built from on the device, and the
 * APC gather fused into them. The reference materialises every original AIR's full dummy trace from its record arena
 * (`chip.generate_proving_ctx(record_arena)`, cuda/mod.rs:228-253) and then gathers the cells the APC keeps; a chip computes
 * all cells of a row from one small record, so the expansion can run inside the gather and the dummy traces never exist.
 * All thirteen instruction AIRs of the reference's snapshot openvm-riscv/tests/openvm_constraints.txt (kind: opcodes):
 *   BaseAlu: ADD SUB XOR OR AND 512..516     Shift: SLL SRL SRA 517..519        LessThan: SLT SLTU 520..521
 *   LoadStore: LOADW LOADBU LOADHU STOREW STOREH STOREB 528..533                LoadSignExtend: LOADB LOADH 534..535
 *   BranchEqual: BEQ BNE 544..545            BranchLessThan: BLT BLTU BGE BGEU 549..552
 *   JalLui: JAL LUI 560..561                 Jalr: 565                          Auipc: 576
 *   Multiplication: MUL 592                  MulH: MULH MULHSU MULHU 593..595   DivRem: DIV DIVU REM REMU 596..599
 * Columns and semantics as in that snapshot: every algebraic constraint listed there holds on the rows produced here and every
 * bus interaction of a row is a legal one (range checks in range, PC lookup = the instruction, memory bus = the RV32IM result).
 *
 * Instruction table (host): the block's instructions that keep at least one cell, in program order. Records (device, u32,
 * word-major: d_records[word * num_calls + call]): word 0 = from_state.timestamp of the call's first instruction, then at
 * `rec_off` per instruction
 *   BaseAlu / Shift / LessThan / Multiplication / MulH / DivRem: b, c, rd's previous value, prev_timestamp of rs1, rs2, rd (6 words)
 *   LoadStore / LoadSignExtend: rs1, the aligned word read, the overwritten word, prev_timestamp of rs1, read, write (6)
 *   BranchEqual / BranchLessThan: a, b, prev_timestamp of rs1, rs2 (4)           Jalr: rs1, rd's previous value, prev_timestamp of rs1, rd (4)
 *   JalLui / Auipc: rd's previous value, its prev_timestamp (2)
 * (the layout is this library's: the reference's DenseRecordArena layouts are EXTERNAL). A memory pointer rs1 + imm is taken
 * modulo 2^29 and the access's alignment (rs1 adjusted to match), a jalr target modulo 2^30: records of a real execution satisfy
 * both already. */
enum {
    POWDR_ORIG_BASE_ALU = 0, POWDR_ORIG_SHIFT = 1, POWDR_ORIG_LOAD_STORE = 2, POWDR_ORIG_BRANCH_EQ = 3, POWDR_ORIG_JAL_LUI = 4,
    POWDR_ORIG_LESS_THAN = 5, POWDR_ORIG_BRANCH_LT = 6, POWDR_ORIG_JALR = 7, POWDR_ORIG_LOAD_SIGN_EXTEND = 8, POWDR_ORIG_DIV_REM = 9,
    POWDR_ORIG_MUL_H = 10, POWDR_ORIG_MUL = 11, POWDR_ORIG_AUIPC = 12, POWDR_ORIG_KIND_COUNT = 13
};
typedef struct {
    uint32_t kind;      /* POWDR_ORIG_* */
    uint32_t opcode;    /* global opcode, see above */
    uint32_t pc;        /* from_state.pc of this instruction */
    uint32_t a, b, c;   /* operands of the instruction [opcode, a, b, c, d, e, f, g]; c reduced mod p */
    uint32_t e, f, g;   /* rs2 address space (ALU) / memory address space; needs_write; sign of the immediate */
    uint32_t ts_delta;  /* from_state.timestamp of this instruction minus the call's first timestamp */
    uint32_t air_row;   /* row of this instruction inside its AIR's block of row_block_size rows per call */
    uint32_t rec_off;   /* first record word of this instruction inside a call's record */
} PowdrOrigInstr;
typedef struct {
    int32_t instr;      /* index into the instruction table */
    int32_t col;        /* column of that instruction's AIR */
    int32_t apc_col;
} PowdrRecordSubst;
/* Full column-major dummy traces of the AIRs — what the reference's chips hand to _apc_tracegen: h_airs[kind] = {width,
 * height, device buffer, row_block_size}, POWDR_ORIG_KIND_COUNT entries (an AIR that does not occur: buffer NULL); the row of
 * instruction i of call r is air_row(i) + r * row_block_size; rows beyond the calls are not touched (zero-initialise the buffers). */
int powdr_original_airs_expand(const uint32_t* d_records, size_t num_calls, const PowdrOrigInstr* h_instrs, size_t n_instrs,
                               const OriginalAir* h_airs);
/* The fused form: out[apc_col * H + r] = r < num_apc_calls ? cell `col` of the row instruction `instr` produces in call r : 0,
 * straight from the records (duplicate apc_col: the last PowdrRecordSubst wins). Equal to powdr_original_airs_expand followed
 * by _apc_tracegen on Subst {kind, col, air_row(instr), apc_col}; moves 16 KB of records per call of a keccak block instead of
 * 110 KB of dummy-trace cells written and read back. */
int powdr_apc_tracegen_records(PowdrFp* d_output, size_t output_height, const uint32_t* d_records, size_t num_apc_calls,
                               const PowdrOrigInstr* h_instrs, size_t n_instrs, const PowdrRecordSubst* h_subs, size_t n_subs);

/* Test hook: the row ONE instruction produces from its (up to six) record words, computed on the HOST by the same expander code
 * the kernels run (host-device templates over the cell sink): lets a CPU-only test suite check the product's expanders against the
 * restatement. row_out: the AIR's cells, canonical; returns the AIR's width, or -1 for an instruction no chip accepts. */
int powdr_original_row_expand_host(const PowdrOrigInstr* instr, const uint32_t* record_words6, uint32_t timestamp, uint32_t* row_out);

/* Traces of the shared periphery chips (the RECEIVE side of the three lookup buses) from the histograms
 * _apc_apply_bus filled. The chips are EXTERNAL (openvm-circuit-primitives; instantiated in
 * openvm/src/powdr_extension/trace_generator/cuda/periphery.rs:33-85); in-repo is how a lookup becomes a histogram
 * index (openvm/cuda/src/apc_apply_bus.cu:74,89,104) and these functions invert that map: row i carries the tuple
 * with index i and its count as multiplicity. Column-major Montgomery matrices, layouts ours:
 *   var range : n_bins rows (power of two) x [value, bits, mult],        i = (1 << bits) + value - 1
 *   tuple2    : sz0*sz1 rows (power of two) x [v0, v1, mult],            i = v0 * sz1 + v1
 *   bitwise   : 65 536 rows x [x, y, x ^ y, mult_range, mult_xor],       i = x * 256 + y, hist = [range | xor]
 * All pointers are device pointers; return value as above. */
int powdr_periphery_var_range_trace(const uint32_t* d_var_hist, size_t var_num_bins, PowdrFp* d_out);
int powdr_periphery_tuple2_trace(const uint32_t* d_tuple2_hist, uint32_t tuple2_sz0, uint32_t tuple2_sz1, PowdrFp* d_out);
int powdr_periphery_bitwise_trace(const uint32_t* d_bitwise_hist, PowdrFp* d_out);

/* All launches of this library go to this stream (default: the null stream,
 * like the reference, cuda/mod.rs:374-378). Pass a hipStream_t as void*. */
void powdr_gpu_set_stream(void* hip_stream);
void* powdr_gpu_get_stream(void);

/* Per-kernel timing with HIP events recorded on the launch stream.
 * enable=1 starts collecting (and clears); powdr_gpu_timing_report writes
 * "name count total_ms\n" lines into buf (NUL terminated) and returns the
 * number of bytes needed. It synchronises the stream. */
void powdr_gpu_timing_enable(int enable);
size_t powdr_gpu_timing_report(char* buf, size_t cap);

/* Diagnostics: which code paths the CALLING THREAD's library calls took since the last reset — 16 counters:
 * [0] gather jobs that fetch cell by cell, [1] that stream whole row blocks, [2] that stream row chunks (summed over the
 * _apc_tracegen calls), [3] _apc_tracegen calls; [4] bus interactions replayed by the fixed-shape small-form kernel,
 * [5] by the interpreter, [6] row windows of the binned histogram path, [7] bus calls that used direct atomics, [8] bus
 * calls on plan-compiled (xbc) code; [9] launches of run-time specialised (hiprtc) expression kernels, [10] launches of
 * their interpreter twins; the rest reserved. Parity tests use it to assert that a workload really covered a path. */
void powdr_gpu_call_stats(uint64_t* out16, int reset);

/* Library/ABI self-description for load checks. */
const char* powdr_gpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* POWDR_GPU_H */
