//! powdr-openvm-hip: the Rust side of this repository's MI355X backend.
//!
//! * [`ffi`]    — `extern "C"` blocks for include/powdr_gpu.h (= openvm/src/cuda_abi.rs:8-64 + extensions),
//!                include/powdr_host.h and include/powdr_prover.h
//! * [`device`] — `DeviceBuffer` / `DeviceMatrix` over hipMalloc (stand-ins for openvm-cuda-common / -backend types)
//! * [`chip`]   — `PowdrChipHip`: `Chip::generate_proving_ctx` -> `powdr_apc_generate_witness_gpu`; from call records without dummy
//!                chips: `PowdrTraceGeneratorHip::try_generate_witness_from_records` -> `powdr_apc_generate_witness_from_records`
//! * [`isa_hip`] — `OpenVmIsaHip`: the builder type and dummy-chip-complex constructor `OpenVmISA` (isa.rs:47-120) has for the CPU and
//!                CUDA engines but not for a third one (an extension trait; the upstream patch is INTEGRATION.md §3c)
//! * [`records_from_arena`] — the per-AIR `DenseRecordArena`s of `PowdrExecutor::execute` -> the word-major call records of
//!                `powdr_apc_generate_witness_from_records` (keccak-block chips)
//! * [`engine`] — `HipEngine` (one `pw_prove_segment` call per segment), `SpecializedConfigHipBuilder`,
//!                `PowdrHipProverExt`
//! * [`multi`]  — the reference's sequential segment loop (trace_generation.rs:111-141) on N GPUs at once:
//!                `pw_prove_segments_multi`, RCCL only for the final commitment merge
//!
//! Wiring inside powdr-openvm (openvm/src/lib.rs:69-95), next to the `cuda` arm of the `cfg_if!`:
//! ```ignore
//! #[cfg(feature = "hip")]
//! pub type PowdrSdkHip<ISA> = GenericSdk<powdr_openvm_hip::HipEngine, powdr_openvm_hip::SpecializedConfigHipBuilder<ISA>>;
//! ```
//! and `powdr_openvm_riscv::prove` (openvm-riscv/src/lib.rs:300-307) picks `PowdrSdkHip` when built with `--features hip`.
pub mod chip;
pub mod device;
pub mod engine;
pub mod ffi;
pub mod isa_hip;
pub mod multi;
pub mod records_from_arena;

pub use chip::{PowdrChipHip, PowdrPeripheryInstancesHip, PowdrTraceGeneratorHip};
pub use isa_hip::{OpenVmIsaHip, OriginalHipChipComplex};
pub use device::{DeviceBuffer, DeviceMatrix, HipError, MemCopyH2D};
pub use engine::{AirProgram, HipAirProver, HipBackend, HipEngine, HipSegmentProof, PowdrHipProverExt, SpecializedConfigHipBuilder};

/// Library / ABI self-description (include/powdr_gpu.h `powdr_gpu_version`)
pub fn backend_version() -> String {
    unsafe { std::ffi::CStr::from_ptr(ffi::powdr_gpu_version()) }.to_string_lossy().into_owned()
}
