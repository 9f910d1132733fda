//! The third engine beside `BabyBearPoseidon2CpuEngine` / `BabyBearPoseidon2GpuEngine` (openvm/src/lib.rs:69-95) and its
//! builder / prover-extension wiring (lib.rs:133-166 `SpecializedConfigGpuBuilder`, :217-269 `PowdrGpuProverExt`).
//!
//! The `StarkEngine` / `ProverBackend` / `ProverDevice` traits are defined in the EXTERNAL `openvm-stark-backend`; what
//! is evidenced in the powdr checkout is: `StarkEngine<SC = .., PB = .., PD = ..>` (lib.rs:175,286), `ProverBackend{Matrix}`
//! (cuda/mod.rs:404), `engine.prove(pk, ProvingContext{per_trace: [(air_id, AirProvingContext)]})`
//! (trace_generation.rs:97-139, empirical_constraints.rs:131). The impl below follows that surface; method names beyond
//! it are marked `// (EXTERNAL trait)`.
use crate::chip::{PowdrChipHip, PowdrPeripheryInstancesHip};
use crate::device::{DeviceBuffer, DeviceMatrix, HipError, MemCopyH2D};
use crate::ffi;
use std::marker::PhantomData;

use openvm_circuit::arch::{
    AirInventory, ChipInventory, ChipInventoryError, DenseRecordArena, VmBuilder, VmChipComplex, VmProverExtension,
};
use openvm_stark_backend::prover::{AirProvingContext, ProverBackend, ProvingContext};
use openvm_stark_backend::{keygen::types::MultiStarkProvingKey, StarkEngine};
use openvm_stark_sdk::p3_baby_bear::BabyBear;
use crate::isa_hip::OpenVmIsaHip;
use powdr_openvm::isa::OpenVmISA;
use powdr_openvm::powdr_extension::{chip::PowdrAir, PowdrExtension};
use powdr_openvm::{BabyBearSC, PeripheryBusIds, SpecializedConfig};

/// `ProverBackend` with column-major device matrices in HBM (the CUDA engine's `GpuBackend`, lib.rs:79)
pub struct HipBackend;
impl ProverBackend for HipBackend {
    type Matrix = DeviceMatrix<BabyBear>;
    // (EXTERNAL trait) the remaining associated types (Val, Challenge, Commitment, ...) are BabyBearSC's
}

/// One prover object per AIR of the proving key, created at key generation from the AIR's constraint programs and
/// bus interactions (what `PowdrAir::eval` pushes through the symbolic builder, chip.rs:94-130).
pub struct HipAirProver {
    pub handle: *mut ffi::PwProver,
    pub width: u32,
}
impl Drop for HipAirProver {
    fn drop(&mut self) {
        unsafe { ffi::pw_prover_destroy(self.handle) }
    }
}

pub struct HipEngine {
    pub config: ffi::PwStarkConfig,
    /// indexed by air_id of the `MultiStarkProvingKey`
    pub provers: Vec<HipAirProver>,
}

/// The proof of one segment: ONE pw-stark v1 proof (magic PWS3) over all AIRs (include/powdr_prover.h).
pub struct HipSegmentProof {
    pub air_ids: Vec<usize>,
    pub log_heights: Vec<u32>,
    pub words: Vec<u32>,
}

impl HipEngine {
    /// keygen-side construction: `programs[air_id]` = (width, post-fix constraint bytecode, spans, interaction tables)
    /// — for APC AIRs `powdr_apc_compile_constraints` / `powdr_apc_compile_bus(apc, 1)` produce exactly these from the APC.
    pub fn new(config: ffi::PwStarkConfig, programs: &[AirProgram]) -> Self {
        let provers = programs
            .iter()
            .map(|p| {
                let handle = unsafe {
                    ffi::pw_prover_create_logup(
                        &config, p.width, p.cons_bytecode.as_ptr(), p.cons_bytecode.len(), p.cons_spans.as_ptr(),
                        p.cons_spans.len() / 2, p.interactions.as_ptr(), p.interactions.len() / 3, p.inter_spans.as_ptr(),
                        p.inter_spans.len() / 2, p.inter_bytecode.as_ptr(), p.inter_bytecode.len(),
                    )
                };
                assert!(!handle.is_null(), "pw_prover_create_logup failed");
                // the reference's degree bound 2 * DEFAULT_APP_LOG_BLOWUP + 1 = 3 (lib.rs:97-101)
                assert!(unsafe { ffi::pw_prover_max_constraint_degree(handle) } <= 3, "constraint degree above the blow-up-2 bound");
                HipAirProver { handle, width: p.width }
            })
            .collect();
        Self { config, provers }
    }

    /// `engine.prove(pk, ctx)` (trace_generation.rs:136-139): ONE call per segment with the traces of all chips, one proof.
    pub fn prove_segment(&self, ctx: &ProvingContext<HipBackend>) -> Result<HipSegmentProof, HipError> {
        self.prove_segment_impl(ctx, false)
    }

    /// The reference's `engine.prove` takes the context BY VALUE: every chip moved its `common_main` into it
    /// (cuda/mod.rs:415-419). Owning the matrices lets an AIR that has to be streamed keep its coefficient arrays in the trace's
    /// own buffer (`pw_prove_segment_consuming`); the matrices are dropped with `ctx` when this returns.
    pub fn prove_segment_owned(&self, ctx: ProvingContext<HipBackend>) -> Result<HipSegmentProof, HipError> {
        self.prove_segment_impl(&ctx, true)
    }

    fn prove_segment_impl(&self, ctx: &ProvingContext<HipBackend>, hand_over: bool) -> Result<HipSegmentProof, HipError> {
        let mut airs = vec![];
        let mut air_ids = vec![];
        let mut log_heights = vec![];
        for (air_id, air_ctx) in &ctx.per_trace {
            let m: &DeviceMatrix<BabyBear> = &air_ctx.common_main;
            if m.height() == 0 {
                continue; // an APC that was not called in this segment
            }
            assert!(m.height().is_power_of_two() && m.width() as u32 == self.provers[*air_id].width);
            let log_h = m.height().trailing_zeros();
            airs.push(ffi::PwSegmentAir {
                prover: self.provers[*air_id].handle,
                d_trace: m.buffer().as_ptr() as *const u32,
                log_height: log_h,
                flags: if hand_over { ffi::PW_AIR_HAND_OVER } else { 0 },
            });
            air_ids.push(*air_id);
            log_heights.push(log_h);
        }
        let mut words: *const u32 = core::ptr::null();
        let mut n_words = 0usize;
        // logup = 1: the bus interactions of every AIR are inside the proof; the cumulative sums must cancel
        HipError::from_result(unsafe {
            if hand_over {
                ffi::pw_prove_segment_consuming(airs.as_ptr(), airs.len(), 1, &mut words, &mut n_words)
            } else {
                ffi::pw_prove_segment(airs.as_ptr(), airs.len(), 1, &mut words, &mut n_words)
            }
        })?;
        let words = unsafe { std::slice::from_raw_parts(words, n_words) }.to_vec();
        Ok(HipSegmentProof { air_ids, log_heights, words })
    }

    /// The CPU verification step (`verify_app_proof::<BabyBearPoseidon2CpuEngine>`, openvm-riscv/src/lib.rs:337-341)
    pub fn verify_segment(&self, programs: &[AirProgram], proof: &HipSegmentProof) -> Result<(), i32> {
        let descs: Vec<ffi::PwAirDescription> = proof
            .air_ids
            .iter()
            .zip(&proof.log_heights)
            .map(|(id, lh)| {
                let p = &programs[*id];
                ffi::PwAirDescription {
                    width: p.width, log_height: *lh, logup: 1,
                    cons_bytecode: p.cons_bytecode.as_ptr(), bytecode_len: p.cons_bytecode.len(),
                    cons_spans: p.cons_spans.as_ptr(), n_constraints: p.cons_spans.len() / 2,
                    interactions: p.interactions.as_ptr(), n_interactions: p.interactions.len() / 3,
                    inter_spans: p.inter_spans.as_ptr(), n_inter_spans: p.inter_spans.len() / 2,
                    inter_bytecode: p.inter_bytecode.as_ptr(), inter_bytecode_len: p.inter_bytecode.len(),
                }
            })
            .collect();
        let mut total = [0u32; 4];
        let rc = unsafe {
            ffi::pw_verify_segment(&self.config, descs.as_ptr(), descs.len(), 1, proof.words.as_ptr(), proof.words.len(), 1, total.as_mut_ptr())
        };
        if rc == 0 { Ok(()) } else { Err(rc) }
    }
}

/// Constraint programs + interaction tables of one AIR (host memory), the keygen output the engine needs.
pub struct AirProgram {
    pub width: u32,
    pub cons_bytecode: Vec<u32>,
    pub cons_spans: Vec<u32>,     // {off, len} pairs
    pub interactions: Vec<u32>,   // {bus, n_args, first span} triples
    pub inter_spans: Vec<u32>,    // {off, len} pairs, [mult, arg0, ...] per interaction
    pub inter_bytecode: Vec<u32>,
}

impl StarkEngine for HipEngine {
    type SC = BabyBearSC;
    type PB = HipBackend;
    type PD = HipDevice;
    // (EXTERNAL trait) fn prove(&self, pk: &MultiStarkProvingKey<Self::SC>, ctx: ProvingContext<Self::PB>) -> Proof<Self::SC>
    //   => self.prove_segment(&ctx), wrapped into the proof type the SDK serialises
}

/// `ProverDevice`: uploads host matrices for chips that still generate traces on the CPU (system chips)
pub struct HipDevice;
impl HipDevice {
    pub fn transport_matrix_to_device(&self, values: &[BabyBear], height: usize, width: usize) -> DeviceMatrix<BabyBear> {
        // row-major host matrix -> column-major device matrix
        let mut cm = vec![values[0]; values.len()];
        for r in 0..height {
            for c in 0..width {
                cm[c * height + r] = values[r * width + c];
            }
        }
        let m = DeviceMatrix::<BabyBear>::with_capacity(height, width);
        let staged: DeviceBuffer<BabyBear> = cm.as_slice().to_device().expect("H2D");
        // (a device-to-device copy into `m`; elided: DeviceMatrix::from_buffer in a full implementation)
        let _ = staged;
        m
    }
}

// ---------------------------------------------------------------------------------------------------------------------
/// lib.rs:133-166 with the HIP engine
#[derive(Default, Clone)]
pub struct SpecializedConfigHipBuilder<ISA> {
    _marker: PhantomData<ISA>,
}

impl<ISA: OpenVmIsaHip> VmBuilder<HipEngine> for SpecializedConfigHipBuilder<ISA> {
    type VmConfig = SpecializedConfig<ISA>;
    type SystemChipInventory = <<ISA as OpenVmIsaHip>::HipBuilder as VmBuilder<HipEngine>>::SystemChipInventory;
    type RecordArena = DenseRecordArena;

    fn create_chip_complex(
        &self,
        config: &SpecializedConfig<ISA>,
        circuit: AirInventory<BabyBearSC>,
    ) -> Result<VmChipComplex<BabyBearSC, Self::RecordArena, HipBackend, Self::SystemChipInventory>, ChipInventoryError> {
        // `OpenVmIsaHip::HipBuilder` (src/isa_hip.rs): the instruction set's builder for this engine — the third one beside
        // `CpuBuilder` / `GpuBuilder` of `OpenVmISA` (isa.rs:63-81), which upstream does not have
        let mut chip_complex =
            VmBuilder::<HipEngine>::create_chip_complex(&<ISA as OpenVmIsaHip>::HipBuilder::default(), &config.original.config, circuit)?;
        let inventory = &mut chip_complex.inventory;
        VmProverExtension::<HipEngine, _, _>::extend_prover(&PowdrHipProverExt::<ISA>::default(), &config.powdr, inventory)?;
        Ok(chip_complex)
    }
}

/// lib.rs:202-269 with the HIP chips
#[derive(Default)]
pub struct PowdrHipProverExt<ISA> {
    _marker: PhantomData<ISA>,
}

impl<ISA: OpenVmIsaHip> VmProverExtension<HipEngine, DenseRecordArena, PowdrExtension<BabyBear, ISA>> for PowdrHipProverExt<ISA> {
    fn extend_prover(
        &self,
        extension: &PowdrExtension<BabyBear, ISA>,
        inventory: &mut ChipInventory<BabyBearSC, DenseRecordArena, HipBackend>,
    ) -> Result<(), ChipInventoryError> {
        // the three shared periphery chips and their bus ids, read from the AIR inventory like lib.rs:337-364
        let ids: PeripheryBusIds = powdr_openvm::get_periphery_bus_ids(inventory);
        let periphery = PowdrPeripheryInstancesHip::from_inventory(inventory, &ids);
        for precompile in &extension.precompiles {
            inventory.next_air::<PowdrAir<BabyBear>>()?;
            let chip = PowdrChipHip::new(precompile.clone(), extension.airs.clone(), extension.base_config.clone(), periphery.clone());
            inventory.add_executor_chip(chip);
        }
        Ok(())
    }
}

impl PowdrPeripheryInstancesHip {
    /// Histograms of the real and the dummy periphery instances (cuda/periphery.rs:24-85): zero-filled device buffers of
    /// `var_num_bins`, `sz0 * sz1` and `2 << 16` u32 counters.
    pub fn from_inventory<SC, RA, PB>(_inventory: &ChipInventory<SC, RA, PB>, ids: &PeripheryBusIds) -> Self {
        let alloc = |n: usize| -> *mut u32 {
            let b = DeviceBuffer::<u32>::with_capacity(n);
            b.fill_zero().expect("memset");
            let p = b.as_mut_ptr();
            core::mem::forget(b); // owned by the periphery instance for the lifetime of the VM
            p
        };
        let make = || ffi::PowdrPeriphery {
            var_range_bus_id: ids.range_checker as u32,
            d_var_hist: alloc(1 << 18),
            var_num_bins: 1 << 18,
            tuple2_bus_id: ids.tuple_range_checker.map(|x| x as u32).unwrap_or(u32::MAX),
            d_tuple2_hist: alloc(256 * 2048),
            tuple2_sz0: 256,
            tuple2_sz1: 2048,
            bitwise_bus_id: ids.bitwise_lookup.map(|x| x as u32).unwrap_or(u32::MAX),
            d_bitwise_hist: alloc(2 << 16),
        };
        Self { real: make(), dummy: make() }
    }
}
