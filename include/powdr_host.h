/*
 * powdr_host.h — host-side mirror (C++ behind a C ABI) of the reference's GPU
 * trace-generation host path, for this repository's MI355X kernels.
 *
 * What it mirrors (paths under /root/reference):
 *   the APC data model and its serde wire format
 *       autoprecompiles/src/lib.rs:177-195 (Apc, Substitution),
 *       autoprecompiles/src/symbolic_machine.rs:115-133,201-209 (SymbolicMachine, main_columns),
 *       expression/src/lib.rs:209-246 + autoprecompiles/src/expression.rs:51-80 (expression / reference serde),
 *       constraint-solver/src/constraint_system.rs:97-137 (DerivedVariable, ComputationMethod)
 *   the bytecode compilers  openvm/src/powdr_extension/trace_generator/cuda/mod.rs:49-177
 *       (emit_expr, compile_derived_to_gpu, compile_bus_to_gpu)
 *   `PowdrTraceGeneratorGpu::try_generate_witness`  cuda/mod.rs:201-401
 *       (OriginalAir/Subst table construction :272-328, the three kernel calls :334-398)
 * The reference's Rust toolchain is absent here, so the host side is C++; a Rust caller
 * would bind these entry points exactly like cuda_abi.rs binds the kernels.
 */
#ifndef POWDR_HOST_H
#define POWDR_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "powdr_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PowdrApc PowdrApc;

/* Parse the JSON document serde_json produces for `Apc` (keys: block, machine{constraints,
 * bus_interactions, derived_columns}, subs, ...). Returns NULL and fills `err` on failure. */
PowdrApc* powdr_apc_from_json(const char* json, size_t len, char* err, size_t err_cap);
void powdr_apc_free(PowdrApc* apc);

/* ---- artifact readers: APCs produced on a machine with the Rust toolchain, replayed here without it ----
 * Every reader accepts any document that CONTAINS `Apc` maps (keys block, machine, subs) and addresses them by their
 * position in document order:
 *   - `apc_candidate_<pcs>_<seq>[_suffix].json`: `ApcWithBusMap{#[serde(flatten)] apc, bus_map}`
 *     (/root/reference/autoprecompiles/src/export.rs:77-93,271-276; the reference's own tests/NAME.json.gz fixtures are such
 *     files): one Apc at the top level, plus the bus map (below);
 *   - the CLI's stage caches `<artifacts-dir>/<stage>/<hash>/artifact.cbor`, written with serde_cbor
 *     (/root/reference/cli-openvm-riscv/src/main.rs:380-407): stage `select` = Vec<ApcWithStats{apc, stats,
 *     evaluation_result}> (autoprecompiles/src/adapter.rs:22-27, main.rs:308-336), stage `setup` = the CompiledProgram
 *     whose PowdrExtension carries the selected APCs (main.rs:340-355). Instructions serialised as structs
 *     ({"opcode": n, "a": ..}) instead of the export's [opcode, a..g] arrays are accepted.
 * index >= count -> NULL with an error message. */
PowdrApc* powdr_apc_from_json_at(const char* json, size_t len, size_t index, char* err, size_t err_cap);
size_t powdr_apc_count_in_json(const char* json, size_t len);
PowdrApc* powdr_apc_from_cbor(const uint8_t* bytes, size_t len, size_t index, char* err, size_t err_cap);
size_t powdr_apc_count_in_cbor(const uint8_t* bytes, size_t len);

/* The bus map of an ApcWithBusMap export (autoprecompiles/src/bus_map.rs:4-16 BusType, OpenVM's custom types
 * openvm-bus-interaction-handler/src/bus_map.rs:17-21). 0 entries when the document had none. */
enum {
    POWDR_BUS_EXECUTION_BRIDGE = 0, POWDR_BUS_MEMORY = 1, POWDR_BUS_PC_LOOKUP = 2, POWDR_BUS_VARIABLE_RANGE_CHECKER = 3,
    POWDR_BUS_BITWISE_LOOKUP = 4, POWDR_BUS_TUPLE_RANGE_CHECKER = 5, POWDR_BUS_OTHER = 6
};
size_t powdr_apc_bus_map_len(const PowdrApc* apc);
/* entry i: bus id, kind (above), tuple sizes (TupleRangeChecker only), variant name. Returns 0, -1 if out of range. */
int powdr_apc_bus_map_entry(const PowdrApc* apc, size_t i, uint64_t* bus_id, uint32_t* kind, uint32_t* sizes2, char* name,
                            size_t name_cap);

/* `apc_candidates.json` of cell PGO (/root/reference/autoprecompiles/src/pgo/cell/mod.rs:34-97, JSON_EXPORT_VERSION 4;
 * versions 0-3 are read as far as their fields go): per candidate the execution frequency, the AIR statistics before
 * and after optimisation (evaluation.rs:12-21,50-59), and the ranking values. The file holds no machines — it tells
 * which candidates exist and how wide / how often; the machines are in the apc_candidate_*.json files. */
typedef struct { uint64_t main_columns, constraints, bus_interactions; } PowdrAirStats;
typedef struct {
    uint64_t execution_frequency;
    uint64_t start_pc;        /* of the first original block */
    uint32_t n_blocks;        /* superblocks: > 1 */
    uint32_t n_instructions;  /* over all original blocks */
    PowdrAirStats before, after;
    uint64_t width_before, value;
    double cost_before, cost_after;
} PowdrApcCandidateInfo;
typedef struct PowdrApcCandidates PowdrApcCandidates;
PowdrApcCandidates* powdr_apc_candidates_from_json(const char* json, size_t len, char* err, size_t err_cap);
void powdr_apc_candidates_free(PowdrApcCandidates* c);
uint64_t powdr_apc_candidates_version(const PowdrApcCandidates* c);
size_t powdr_apc_candidates_count(const PowdrApcCandidates* c);
size_t powdr_apc_candidates_num_labels(const PowdrApcCandidates* c);
int powdr_apc_candidates_get(const PowdrApcCandidates* c, size_t i, PowdrApcCandidateInfo* out);

/* machine.main_columns().count(): unique references of constraints + bus interactions */
uint32_t powdr_apc_width(const PowdrApc* apc);
/* ascending poly ids; column index = position (autoprecompiles/src/powdr.rs:44-57) */
const uint64_t* powdr_apc_poly_ids(const PowdrApc* apc);
uint32_t powdr_apc_num_constraints(const PowdrApc* apc);
uint32_t powdr_apc_num_bus_interactions(const PowdrApc* apc);
uint32_t powdr_apc_num_derived_columns(const PowdrApc* apc);
uint32_t powdr_apc_num_instructions(const PowdrApc* apc);
uint32_t powdr_apc_instruction_opcode(const PowdrApc* apc, uint32_t i);
uint32_t powdr_apc_instruction_num_subs(const PowdrApc* apc, uint32_t i);

/* compile_bus_to_gpu (cuda/mod.rs:143-177) for a trace of `apc_height` rows. Call with NULL
 * outputs to obtain the sizes. Returns the number of bytecode words. */
size_t powdr_apc_compile_bus(const PowdrApc* apc, size_t apc_height, DevInteraction* interactions,
                             ExprSpan* arg_spans, size_t* n_arg_spans, uint32_t* bytecode);
/* compile_derived_to_gpu (cuda/mod.rs:100-141). Returns the number of bytecode words. */
size_t powdr_apc_compile_derived(const PowdrApc* apc, size_t apc_height, DerivedExprSpec* specs, uint32_t* bytecode);
/* Constraint programs for pw_prover_create: PUSH operands are column indices. `spans` gets
 * n_constraints {off,len} pairs. Returns the number of bytecode words. */
size_t powdr_apc_compile_constraints(const PowdrApc* apc, ExprSpan* spans, uint32_t* bytecode);
/* Subst/row_block_size tables (cuda/mod.rs:272-328). instr_air[i] = id of the original AIR of
 * instruction i (what `original_airs.opcode_to_air` yields), any value for instructions
 * without substitutions. AIRs are numbered by first appearance: air_ids_out[k] = caller id of
 * table entry k, row_block_out[k] = instructions per call. Returns the number of Subst
 * records; *n_airs gets the number of table entries. NULL outputs = size query. */
size_t powdr_apc_build_substitutions(const PowdrApc* apc, const int32_t* instr_air, Subst* subs,
                                     int32_t* air_ids_out, int32_t* row_block_out, size_t* n_airs);

/* A column-major device matrix (DeviceMatrix<BabyBear>) */
typedef struct {
    const PowdrFp* buffer;
    int32_t width;
    int32_t height;
} PowdrDeviceMatrix;

/* The shared periphery chips' device histograms + bus ids (cuda/mod.rs:357-372) */
typedef struct {
    uint32_t var_range_bus_id;
    uint32_t* d_var_hist;
    size_t var_num_bins;
    uint32_t tuple2_bus_id;
    uint32_t* d_tuple2_hist;
    uint32_t tuple2_sz0, tuple2_sz1;
    uint32_t bitwise_bus_id;
    uint32_t* d_bitwise_hist;
} PowdrPeriphery;

/* Bus ids (and the tuple checker's sizes) of the three periphery buses from the APC's bus map — what the reference reads
 * from the VM's AIR inventory at run time (openvm/src/lib.rs:337-364). Histogram pointers and var_num_bins are left
 * untouched. Returns how many of the three were found. */
int powdr_apc_periphery_from_bus_map(const PowdrApc* apc, PowdrPeriphery* periphery);

/* try_generate_witness (cuda/mod.rs:201-401): d_output must hold
 * width * next_power_of_two_or_zero(num_apc_calls) words; it is zero-filled here
 * (cuda/mod.rs:266-269), gathered into, derived columns applied, bus interactions replayed
 * into the periphery histograms. dummy_by_air[id] = dummy trace of the AIR with caller id
 * `id` (ids as used in instr_air). Runs on the library stream; tables are cached per height. */
int powdr_apc_generate_witness_gpu(PowdrApc* apc, const int32_t* instr_air, const PowdrDeviceMatrix* dummy_by_air,
                                   size_t n_dummy, size_t num_apc_calls, PowdrFp* d_output,
                                   const PowdrPeriphery* periphery);

/* The same orchestration with the original chips' work folded in: the APC trace comes straight from the
 * call records of the block (include/powdr_gpu.h: powdr_apc_tracegen_records) — the dummy traces of the original AIRs never
 * exist — then derived columns and bus replay as above. The instruction table (which instruction keeps a cell, its pc =
 * block start_pc + 4 j, timestamp offset, row inside its AIR's block, record offset) follows from the APC's own block and
 * substitutions; powdr_apc_instruction_table returns it (out == NULL: only the count; (size_t)-1: an opcode outside the thirteen
 * RV32IM chips) together with the size of one call's record, so that the caller can lay the records out. */
size_t powdr_apc_instruction_table(const PowdrApc* apc, PowdrOrigInstr* out, size_t* words_per_call);
int powdr_apc_generate_witness_from_records(PowdrApc* apc, const uint32_t* d_records, size_t num_apc_calls, PowdrFp* d_output,
                                            const PowdrPeriphery* periphery);

/* Test hook for the plan-time expression compiler (csrc/xbc.hpp): compiles the reference post-fix
 * program `postfix` into the accumulator code the kernels run and evaluates it on the host over a
 * Montgomery-form trace (`trace[operand + r]`). Returns 0, or < 0 if the program is malformed /
 * deeper than the 16-entry stack (then the kernels fall back to the post-fix interpreter). */
int powdr_xbc_eval_host(const uint32_t* postfix, uint32_t len, const uint32_t* trace, size_t r,
                        uint32_t* result, uint32_t* n_instr);

/* Test hook for the small-form analysis (csrc/small_form.hpp) behind the fast bus-replay kernel: returns 0 and the
 * value (Montgomery) when `postfix` simplifies to k0 + k1*T[a] + k2*T[b] + k3*T[a]*T[b], 1 when it does not (the
 * interpreter handles it). `flags`: 1 uses a, 2 uses b, 4 has the product term, 8 is a plain column, 16 is a constant. */
int powdr_small_form_eval_host(const uint32_t* postfix, uint32_t len, const uint32_t* trace, size_t r,
                               uint32_t* result, uint32_t* flags);

/* Self-test of the field arithmetic helpers the kernels are built from (Montgomery products, lazy / loose ranges,
 * 64- and 96-bit accumulators, extension field) against plain modular arithmetic; host code, no GPU.
 * Returns 0 or the number of the first failing check. */
int powdr_field_selftest(uint64_t seed, uint32_t iterations);
/* The same checks on the GPU (16 384 threads with different seeds; exercises the inline multiply-add instructions).
 * Returns a HIP error code; *failing_check receives 0 or the number of the first failing check. */
int powdr_field_selftest_gpu(uint64_t seed, uint32_t iterations, int* failing_check);

#ifdef __cplusplus
}
#endif
#endif /* POWDR_HOST_H */
