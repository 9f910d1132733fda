//! FFI declarations of libpowdr_gpu: one `extern "C"` block per header of `include/`, `#[repr(C)]` mirrors of their
//! structs. The first block IS `openvm/src/cuda_abi.rs:8-64` (same names, same argument order); the reference's
//! `#[repr(C)]` types (`cuda_abi.rs:66-95,149-169`) are re-exported from there when built inside the powdr workspace.
//! Kept in sync with the headers by tests/test_rust_adapter_sync.py (symbol sets and struct field order).
#![allow(non_camel_case_types, clippy::too_many_arguments)]

use core::ffi::{c_char, c_int, c_uint, c_void};

/// BabyBear in Montgomery form, the in-memory representation of `p3_baby_bear::BabyBear` (include/powdr_gpu.h `PowdrFp`).
pub type PowdrFp = u32;

// ---------------------------------------------------------------------------------------------- include/powdr_gpu.h
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct OriginalAir {
    pub width: i32,
    pub height: i32,
    pub buffer: *const PowdrFp,
    pub row_block_size: i32,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct Subst {
    pub air_index: i32,
    pub col: i32,
    pub row: i32,
    pub apc_col: i32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ExprSpan {
    pub off: u32,
    pub len: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct DerivedExprSpec {
    pub col_base: u64,
    pub span: ExprSpan,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct DevInteraction {
    pub bus_id: u32,
    pub num_args: u32,
    pub args_index_off: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct PowdrCallMajorAir {
    pub buffer: *const PowdrFp,
    pub cells_per_call: i32,
    pub reserved: i32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PowdrSubstCM {
    pub air_index: i32,
    pub slot: i32,
    pub apc_col: i32,
}

/// `POWDR_ORIG_*` (include/powdr_gpu.h): the thirteen instruction AIRs of openvm-riscv/tests/openvm_constraints.txt, keyed by the
/// short AIR name `OriginalAirs::opcode_to_air` resolves to
pub const POWDR_ORIG_KINDS: [&str; 13] = ["BaseAlu", "Shift", "LoadStore", "BranchEqual", "JalLui", "LessThan", "BranchLessThan", "Jalr",
                                          "LoadSignExtend", "DivRem", "MulH", "Multiplication", "Auipc"];
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PowdrOrigInstr {
    pub kind: u32,
    pub opcode: u32,
    pub pc: u32,
    pub a: u32,
    pub b: u32,
    pub c: u32,
    pub e: u32,
    pub f: u32,
    pub g: u32,
    pub ts_delta: u32,
    pub air_row: u32,
    pub rec_off: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PowdrRecordSubst {
    pub instr: i32,
    pub col: i32,
    pub apc_col: i32,
}

extern "C" {
    // ---- the reference ABI, openvm/src/cuda_abi.rs:8-64 ----
    pub fn _apc_tracegen(d_output: *mut PowdrFp, output_height: usize, d_original_airs: *const OriginalAir,
                         d_subs: *const Subst, n_subs: usize, num_apc_calls: i32) -> i32;
    pub fn _apc_apply_derived_expr(d_output: *mut PowdrFp, output_height: usize, num_apc_calls: i32,
                                   d_specs: *const DerivedExprSpec, n_cols: usize, d_bytecode: *const u32) -> i32;
    pub fn _apc_apply_bus(d_output: *const PowdrFp, num_apc_calls: i32, d_bytecode: *const u32, bytecode_len: usize,
                          d_interactions: *const DevInteraction, n_interactions: usize, d_arg_spans: *const ExprSpan,
                          n_arg_spans: usize, var_range_bus_id: u32, d_var_hist: *mut u32, var_num_bins: usize,
                          tuple2_bus_id: u32, d_tuple2_hist: *mut u32, tuple2_sz0: u32, tuple2_sz1: u32,
                          bitwise_bus_id: u32, d_bitwise_hist: *mut u32) -> i32;
    // ---- extensions ----
    pub fn powdr_apc_apply_derived_expr_cols(d_output: *mut PowdrFp, output_height: usize, num_apc_calls: i32,
                                             d_specs: *const DerivedExprSpec, n_cols: usize, d_bytecode: *const u32) -> i32;
    pub fn powdr_apc_apply_bus_cols(d_output: *const PowdrFp, output_height: usize, num_apc_calls: i32,
                                    d_bytecode: *const u32, bytecode_len: usize, d_interactions: *const DevInteraction,
                                    n_interactions: usize, d_arg_spans: *const ExprSpan, n_arg_spans: usize,
                                    var_range_bus_id: u32, d_var_hist: *mut u32, var_num_bins: usize, tuple2_bus_id: u32,
                                    d_tuple2_hist: *mut u32, tuple2_sz0: u32, tuple2_sz1: u32, bitwise_bus_id: u32,
                                    d_bitwise_hist: *mut u32) -> i32;
    pub fn powdr_apc_tracegen_host_tables(d_output: *mut PowdrFp, output_height: usize, d_original_airs: *const OriginalAir,
                                          h_original_airs: *const OriginalAir, n_airs: usize, h_subs: *const Subst,
                                          n_subs: usize, num_apc_calls: i32) -> i32;
    pub fn powdr_apc_apply_bus_host_tables(d_output: *const PowdrFp, output_height: usize, num_apc_calls: i32,
                                           d_bytecode: *const u32, h_bytecode: *const u32, bytecode_len: usize,
                                           d_interactions: *const DevInteraction, h_interactions: *const DevInteraction,
                                           n_interactions: usize, d_arg_spans: *const ExprSpan, h_arg_spans: *const ExprSpan,
                                           n_arg_spans: usize, var_range_bus_id: u32, d_var_hist: *mut u32,
                                           var_num_bins: usize, tuple2_bus_id: u32, d_tuple2_hist: *mut u32,
                                           tuple2_sz0: u32, tuple2_sz1: u32, bitwise_bus_id: u32,
                                           d_bitwise_hist: *mut u32) -> i32;
    pub fn powdr_apc_tracegen_callmajor(d_output: *mut PowdrFp, output_height: usize, h_airs: *const PowdrCallMajorAir, n_airs: usize,
                                        h_subs: *const PowdrSubstCM, n_subs: usize, num_apc_calls: i32) -> i32;
    pub fn powdr_original_airs_expand(d_records: *const u32, num_calls: usize, h_instrs: *const PowdrOrigInstr, n_instrs: usize,
                                      h_airs: *const OriginalAir) -> i32;
    pub fn powdr_apc_tracegen_records(d_output: *mut PowdrFp, output_height: usize, d_records: *const u32, num_apc_calls: usize,
                                      h_instrs: *const PowdrOrigInstr, n_instrs: usize, h_subs: *const PowdrRecordSubst, n_subs: usize) -> i32;
    pub fn powdr_original_row_expand_host(instr: *const PowdrOrigInstr, record_words6: *const u32, timestamp: u32, row_out: *mut u32) -> i32;
    pub fn powdr_periphery_var_range_trace(d_var_hist: *const u32, var_num_bins: usize, d_out: *mut PowdrFp) -> i32;
    pub fn powdr_periphery_tuple2_trace(d_tuple2_hist: *const u32, tuple2_sz0: u32, tuple2_sz1: u32, d_out: *mut PowdrFp) -> i32;
    pub fn powdr_periphery_bitwise_trace(d_bitwise_hist: *const u32, d_out: *mut PowdrFp) -> i32;
    pub fn powdr_gpu_set_stream(hip_stream: *mut c_void);
    pub fn powdr_gpu_get_stream() -> *mut c_void;
    pub fn powdr_gpu_timing_enable(enable: c_int);
    pub fn powdr_gpu_timing_report(buf: *mut c_char, cap: usize) -> usize;
    pub fn powdr_gpu_call_stats(out16: *mut u64, reset: c_int);
    pub fn powdr_gpu_version() -> *const c_char;
}

// --------------------------------------------------------------------------------------------- include/powdr_host.h
#[repr(C)]
pub struct PowdrApc {
    _opaque: [u8; 0],
}
#[repr(C)]
pub struct PowdrApcCandidates {
    _opaque: [u8; 0],
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PowdrDeviceMatrix {
    pub buffer: *const PowdrFp,
    pub width: i32,
    pub height: i32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PowdrPeriphery {
    pub var_range_bus_id: u32,
    pub d_var_hist: *mut u32,
    pub var_num_bins: usize,
    pub tuple2_bus_id: u32,
    pub d_tuple2_hist: *mut u32,
    pub tuple2_sz0: u32,
    pub tuple2_sz1: u32,
    pub bitwise_bus_id: u32,
    pub d_bitwise_hist: *mut u32,
}
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct PowdrAirStats {
    pub main_columns: u64,
    pub constraints: u64,
    pub bus_interactions: u64,
}
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct PowdrApcCandidateInfo {
    pub execution_frequency: u64,
    pub start_pc: u64,
    pub n_blocks: u32,
    pub n_instructions: u32,
    pub before: PowdrAirStats,
    pub after: PowdrAirStats,
    pub width_before: u64,
    pub value: u64,
    pub cost_before: f64,
    pub cost_after: f64,
}

extern "C" {
    pub fn powdr_apc_from_json(json: *const c_char, len: usize, err: *mut c_char, err_cap: usize) -> *mut PowdrApc;
    pub fn powdr_apc_from_json_at(json: *const c_char, len: usize, index: usize, err: *mut c_char, err_cap: usize) -> *mut PowdrApc;
    pub fn powdr_apc_count_in_json(json: *const c_char, len: usize) -> usize;
    pub fn powdr_apc_from_cbor(bytes: *const u8, len: usize, index: usize, err: *mut c_char, err_cap: usize) -> *mut PowdrApc;
    pub fn powdr_apc_count_in_cbor(bytes: *const u8, len: usize) -> usize;
    pub fn powdr_apc_free(apc: *mut PowdrApc);
    pub fn powdr_apc_width(apc: *const PowdrApc) -> u32;
    pub fn powdr_apc_poly_ids(apc: *const PowdrApc) -> *const u64;
    pub fn powdr_apc_num_constraints(apc: *const PowdrApc) -> u32;
    pub fn powdr_apc_num_bus_interactions(apc: *const PowdrApc) -> u32;
    pub fn powdr_apc_num_derived_columns(apc: *const PowdrApc) -> u32;
    pub fn powdr_apc_num_instructions(apc: *const PowdrApc) -> u32;
    pub fn powdr_apc_instruction_opcode(apc: *const PowdrApc, i: u32) -> u32;
    pub fn powdr_apc_instruction_num_subs(apc: *const PowdrApc, i: u32) -> u32;
    pub fn powdr_apc_bus_map_len(apc: *const PowdrApc) -> usize;
    pub fn powdr_apc_bus_map_entry(apc: *const PowdrApc, i: usize, bus_id: *mut u64, kind: *mut u32, sizes2: *mut u32,
                                   name: *mut c_char, name_cap: usize) -> c_int;
    pub fn powdr_apc_periphery_from_bus_map(apc: *const PowdrApc, periphery: *mut PowdrPeriphery) -> c_int;
    pub fn powdr_apc_candidates_from_json(json: *const c_char, len: usize, err: *mut c_char, err_cap: usize) -> *mut PowdrApcCandidates;
    pub fn powdr_apc_candidates_free(c: *mut PowdrApcCandidates);
    pub fn powdr_apc_candidates_version(c: *const PowdrApcCandidates) -> u64;
    pub fn powdr_apc_candidates_count(c: *const PowdrApcCandidates) -> usize;
    pub fn powdr_apc_candidates_num_labels(c: *const PowdrApcCandidates) -> usize;
    pub fn powdr_apc_candidates_get(c: *const PowdrApcCandidates, i: usize, out: *mut PowdrApcCandidateInfo) -> c_int;
    pub fn powdr_apc_compile_bus(apc: *const PowdrApc, apc_height: usize, interactions: *mut DevInteraction,
                                 arg_spans: *mut ExprSpan, n_arg_spans: *mut usize, bytecode: *mut u32) -> usize;
    pub fn powdr_apc_compile_derived(apc: *const PowdrApc, apc_height: usize, specs: *mut DerivedExprSpec, bytecode: *mut u32) -> usize;
    pub fn powdr_apc_compile_constraints(apc: *const PowdrApc, spans: *mut ExprSpan, bytecode: *mut u32) -> usize;
    pub fn powdr_apc_build_substitutions(apc: *const PowdrApc, instr_air: *const i32, subs: *mut Subst, air_ids_out: *mut i32,
                                         row_block_out: *mut i32, n_airs: *mut usize) -> usize;
    pub fn powdr_apc_generate_witness_gpu(apc: *mut PowdrApc, instr_air: *const i32, dummy_by_air: *const PowdrDeviceMatrix,
                                          n_dummy: usize, num_apc_calls: usize, d_output: *mut PowdrFp,
                                          periphery: *const PowdrPeriphery) -> c_int;
    pub fn powdr_apc_instruction_table(apc: *const PowdrApc, out: *mut PowdrOrigInstr, words_per_call: *mut usize) -> usize;
    pub fn powdr_apc_generate_witness_from_records(apc: *mut PowdrApc, d_records: *const u32, num_apc_calls: usize, d_output: *mut PowdrFp,
                                                   periphery: *const PowdrPeriphery) -> c_int;
    pub fn powdr_xbc_eval_host(postfix: *const u32, len: u32, trace: *const u32, r: usize, result: *mut u32, n_instr: *mut u32) -> c_int;
    pub fn powdr_small_form_eval_host(postfix: *const u32, len: u32, trace: *const u32, r: usize, result: *mut u32, flags: *mut u32) -> c_int;
    pub fn powdr_field_selftest(seed: u64, iterations: u32) -> c_int;
    pub fn powdr_field_selftest_gpu(seed: u64, iterations: u32, failing_check: *mut c_int) -> c_int;
}

// ------------------------------------------------------------------------------------------- include/powdr_prover.h
/// `PwSegmentProveFn` of include/powdr_prover.h: one worker's replica of the per-segment pipeline.
pub type PwSegmentProveFn = unsafe extern "C" fn(user: *mut c_void, segment: usize, worker: usize, device: c_int, commitment8: *mut u32) -> c_int;

#[repr(C)]
pub struct PwProver {
    _opaque: [u8; 0],
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PwStarkConfig {
    pub num_queries: u32,
    pub pow_bits: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PwSegmentAir {
    pub prover: *mut PwProver,
    pub d_trace: *const u32,
    pub log_height: u32,
    /// read by `pw_prove_segment_consuming` only (`PW_AIR_HAND_OVER`)
    pub flags: u32,
}
/// `PwSegmentAir::flags`: the trace is the engine's to overwrite
pub const PW_AIR_HAND_OVER: u32 = 1;
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PwAirDescription {
    pub width: u32,
    pub log_height: u32,
    pub logup: u32,
    pub cons_bytecode: *const u32,
    pub bytecode_len: usize,
    pub cons_spans: *const u32,
    pub n_constraints: usize,
    pub interactions: *const u32,
    pub n_interactions: usize,
    pub inter_spans: *const u32,
    pub n_inter_spans: usize,
    pub inter_bytecode: *const u32,
    pub inter_bytecode_len: usize,
}

extern "C" {
    pub fn pw_prover_create(cfg: *const PwStarkConfig, width: u32, cons_bytecode: *const u32, bytecode_len: usize,
                            cons_spans: *const u32, n_constraints: usize) -> *mut PwProver;
    pub fn pw_prover_create_logup(cfg: *const PwStarkConfig, width: u32, cons_bytecode: *const u32, bytecode_len: usize,
                                  cons_spans: *const u32, n_constraints: usize, interactions: *const u32,
                                  n_interactions: usize, inter_spans: *const u32, n_inter_spans: usize,
                                  inter_bytecode: *const u32, inter_bytecode_len: usize) -> *mut PwProver;
    pub fn pw_prover_trace_root(p: *mut PwProver, d_trace: *const u32, log_height: u32, root8: *mut u32) -> c_int;
    pub fn pw_prover_set_bus_seed(p: *mut PwProver, seed8: *const u32) -> c_int;
    pub fn pw_prover_logup_path(p: *const PwProver) -> c_int;
    pub fn pw_prover_destroy(p: *mut PwProver);
    pub fn pw_prover_prove(p: *mut PwProver, d_trace: *const u32, log_height: u32, proof_words: *mut *const u32,
                           n_words: *mut usize) -> c_int;
    /// The trace is handed over (the engine owns `common_main`): a streamed proof leaves the coefficient arrays in its place.
    pub fn pw_prover_prove_consuming(p: *mut PwProver, d_trace: *mut u32, log_height: u32, proof_words: *mut *const u32,
                                     n_words: *mut usize) -> c_int;
    pub fn pw_prover_stream_log_blocks_consuming(p: *const PwProver, log_height: u32) -> c_int;
    pub fn pw_trace_from_coefficients(d_coeffs: *mut u32, width: u32, log_height: u32, d_scratch: *mut u32) -> c_int;
    pub fn pw_prover_check_constraints(p: *mut PwProver, d_trace: *const u32, log_height: u32, n_violations: *mut u64,
                                       first_row: *mut u64, first_constraint: *mut u32) -> c_int;
    pub fn pw_verify(cfg: *const PwStarkConfig, width: u32, log_height: u32, cons_bytecode: *const u32, bytecode_len: usize,
                     cons_spans: *const u32, n_constraints: usize, proof_words: *const u32, n_words: usize) -> c_int;
    pub fn pw_verify_logup(cfg: *const PwStarkConfig, width: u32, log_height: u32, cons_bytecode: *const u32,
                           bytecode_len: usize, cons_spans: *const u32, n_constraints: usize, interactions: *const u32,
                           n_interactions: usize, inter_spans: *const u32, n_inter_spans: usize,
                           inter_bytecode: *const u32, inter_bytecode_len: usize, expected_bus_seed: *const u32,
                           proof_words: *const u32, n_words: usize, cumulative_sum: *mut u32, trace_root: *mut u32) -> c_int;
    pub fn pw_logup_group_starts(interactions: *const u32, n_interactions: usize, inter_spans: *const u32,
                                 n_inter_spans: usize, inter_bytecode: *const u32, inter_bytecode_len: usize,
                                 out: *mut u32, cap: usize) -> usize;
    pub fn pw_prove_segment(airs: *const PwSegmentAir, n_airs: usize, logup: c_int, proof_words: *mut *const u32,
                            n_words: *mut usize) -> c_int;
    /// Same proof, same words; AIRs flagged `PW_AIR_HAND_OVER` that are proven STREAMED keep their coefficient arrays in the caller's buffer.
    pub fn pw_prove_segment_consuming(airs: *const PwSegmentAir, n_airs: usize, logup: c_int, proof_words: *mut *const u32,
                                      n_words: *mut usize) -> c_int;
    /// per AIR of the calling thread's last segment proof: log2(#sub-cosets) | 0x100 if the trace was overwritten
    /// the specialised kernels of n provers in one concurrent compile batch, whatever the heights; returns how many are specialised
    pub fn pw_provers_specialise(provers: *const *mut PwProver, n: usize) -> usize;
    pub fn pw_segment_last_modes(out: *mut u32, cap: usize) -> usize;
    /// bytes of the last segment proof's memory plan: all AIRs resident | as chosen | available to the policy
    pub fn pw_segment_last_plan(resident_bytes: *mut usize, planned_bytes: *mut usize, available_bytes: *mut usize);
    /// bytes the provers of this process may plan for on a device (0 = what the device has free)
    pub fn pw_set_device_budget(bytes: usize);
    pub fn pw_get_device_budget() -> usize;
    pub fn pw_verify_segment(cfg: *const PwStarkConfig, airs: *const PwAirDescription, n_airs: usize, logup: c_int,
                             proof_words: *const u32, n_words: usize, check_balance: c_int, total_sum4: *mut u32) -> c_int;
    pub fn pw_prove_airs(airs: *const PwSegmentAir, n_airs: usize, shared_bus_seed: c_int, n_workers: c_uint,
                         proofs: *mut *const u32, n_words: *mut usize, bus_seed8: *mut u32) -> c_int;
    pub fn pw_verify_airs(cfg: *const PwStarkConfig, airs: *const PwAirDescription, n_airs: usize,
                          proofs: *const *const u32, n_words: *const usize, shared_bus_seed: c_int,
                          check_balance: c_int, total_sum4: *mut u32) -> c_int;
    pub fn pw_commitment_digest(roots8: *const u32, n: usize, digest8: *mut u32);
    pub fn pw_prover_reserve(p: *mut PwProver, log_height: u32) -> c_int;
    /// 0 = a proof of this height keeps the LDE resident, b >= 1 = streamed over 2^b sub-cosets, -1 = does not fit.
    pub fn pw_prover_stream_log_blocks(p: *const PwProver, log_height: u32) -> c_int;
    pub fn pw_prover_max_constraint_degree(p: *const PwProver) -> c_int;
    pub fn pw_prover_width(p: *const PwProver) -> u32;
    pub fn pw_prover_device_bytes(p: *const PwProver) -> usize;
    pub fn pw_lde_batch(d_trace: *const u32, width: u32, log_height: u32, d_coeffs: *mut u32, d_lde: *mut u32) -> c_int;
    pub fn pw_lde_fused(d_trace: *const u32, width: u32, log_height: u32, d_tmp: *mut u32, d_lde: *mut u32) -> c_int;
    pub fn pw_lde_subcoset(d_coeffs: *const u32, width: u32, log_height: u32, log_blocks: u32, r: u32, d_scale: *mut u32, d_out: *mut u32) -> c_int;
    pub fn pw_merkle_commit(d_matrix: *const u32, height: usize, width: u32, d_digests: *mut u32) -> c_int;
    pub fn pw_prover_specialise(p: *mut PwProver) -> c_int;
    pub fn pw_prover_specialised(p: *const PwProver, n_kernels: *mut usize, code_bytes: *mut usize, n_chunks: *mut usize) -> c_int;
    pub fn pw_prove_segments_multi(devices: *const c_int, n_workers: usize, segment_cells: *const u64, n_segments: usize,
                                   prove: PwSegmentProveFn, user: *mut c_void, commitments: *mut u32, worker_of_segment: *mut u32) -> c_int;
    pub fn pw_multi_last_merge() -> c_int;
    pub fn pw_assign_units(cells: *const u64, n_units: usize, n_workers: usize, worker_of_unit: *mut u32) -> usize;
    pub fn pw_jit_generated_source(width: u32, cons_bytecode: *const u32, bytecode_len: usize, cons_spans: *const u32, n_constraints: usize,
                                   interactions: *const u32, n_interactions: usize, inter_spans: *const u32, n_inter_spans: usize,
                                   inter_bytecode: *const u32, inter_bytecode_len: usize, which: c_int, chunk_cost: u32, chunks_per_unit: u32,
                                   unit: usize, buf: *mut c_char, cap: usize, kernel_name: *mut c_char, name_cap: usize, first_chunk: *mut u32,
                                   n_chunks: *mut u32, total_chunks: *mut u32) -> usize;
    pub fn pw_jit_cache_stats(units_compiled: *mut u64, units_from_disk: *mut u64);
    pub fn pw_jit_compile_check(width: u32, cons_bytecode: *const u32, bytecode_len: usize, cons_spans: *const u32, n_constraints: usize,
                                interactions: *const u32, n_interactions: usize, inter_spans: *const u32, n_inter_spans: usize,
                                inter_bytecode: *const u32, inter_bytecode_len: usize, n_kernels: *mut usize, code_bytes: *mut usize,
                                n_chunks: *mut usize, err: *mut c_char, err_cap: usize) -> c_int;
    pub fn pw_poseidon2_permute_host(state16: *mut u32);
    pub fn pw_set_poseidon2_constants(ext_rc: *const u32, int_rc: *const u32) -> c_int;
    pub fn pw_get_poseidon2_constants(ext_rc: *mut u32, int_rc: *mut u32, diag: *mut u32);
}

// --------------------------------------------------------------------------------- HIP runtime (libamdhip64), minimal
extern "C" {
    pub fn hipMalloc(ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn hipFree(ptr: *mut c_void) -> c_int;
    pub fn hipMemcpy(dst: *mut c_void, src: *const c_void, bytes: usize, kind: c_int) -> c_int;
    pub fn hipMemset(dst: *mut c_void, value: c_int, bytes: usize) -> c_int;
    pub fn hipDeviceSynchronize() -> c_int;
    pub fn hipSetDevice(device: c_int) -> c_int;
}
pub const HIP_MEMCPY_HOST_TO_DEVICE: c_int = 1;
pub const HIP_MEMCPY_DEVICE_TO_HOST: c_int = 2;
