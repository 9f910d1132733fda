//! `PowdrChipHip` / `PowdrTraceGeneratorHip`: the HIP counterparts of `PowdrChipGpu` / `PowdrTraceGeneratorGpu`
//! (openvm/src/powdr_extension/chip.rs:139-175, .../trace_generator/cuda/mod.rs:179-421).
//!
//! What changes against the CUDA path: the tables (`OriginalAir`, `Subst`, derived-column and bus bytecode) are not
//! rebuilt and re-uploaded per segment (cuda/mod.rs:272-398) — the C++ host mirror behind `powdr_apc_generate_witness_gpu`
//! (include/powdr_host.h) compiles them once per APC and keeps them on the device; the chip hands over only what changes
//! per segment: the dummy traces' device pointers and the call count.
use crate::device::{DeviceBuffer, DeviceMatrix, HipError, MemCopyH2D};
use crate::ffi;
use std::cell::RefCell;
use std::collections::HashMap;
use std::rc::Rc;

use openvm_circuit::arch::DenseRecordArena;
use openvm_stark_backend::prover::{AirProvingContext, ProverBackend};
use openvm_stark_backend::Chip;
use openvm_stark_sdk::p3_baby_bear::BabyBear;
use crate::isa_hip::OpenVmIsaHip;
use crate::records_from_arena::{records_from_arenas, BridgeError};
use powdr_openvm::isa::{IsaApc, OpenVmISA};
use powdr_openvm::powdr_extension::executor::OriginalArenas;
use powdr_openvm::powdr_extension::PowdrPrecompile;
use powdr_openvm::extraction_utils::{OriginalAirs, OriginalVmConfig};

/// The shared periphery chips' device histograms and bus ids (cuda/mod.rs:357-372, cuda/periphery.rs:24-85). On the HIP
/// side the three receive-side chips are plain device histograms; their own traces come from
/// `powdr_periphery_{var_range,tuple2,bitwise}_trace` (include/powdr_gpu.h) after all APC chips of the segment ran.
#[derive(Clone)]
pub struct PowdrPeripheryInstancesHip {
    pub real: ffi::PowdrPeriphery,
    /// the dummy instance the original chips' trace generation increments instead of the real one (cuda/periphery.rs:33-60)
    pub dummy: ffi::PowdrPeriphery,
}

pub struct ApcHandle(*mut ffi::PowdrApc);
impl Drop for ApcHandle {
    fn drop(&mut self) {
        unsafe { ffi::powdr_apc_free(self.0) }
    }
}

pub struct PowdrTraceGeneratorHip<ISA: OpenVmIsaHip> {
    pub apc: IsaApc<BabyBear, ISA>,
    pub original_airs: OriginalAirs<BabyBear, ISA>,
    pub config: OriginalVmConfig<ISA>,
    pub periphery: PowdrPeripheryInstancesHip,
    /// parsed once from the APC's serde_json form (autoprecompiles/src/lib.rs:185-195)
    handle: ApcHandle,
    /// AIR name -> id used in `instr_air` / `dummy_by_air`
    air_ids: HashMap<String, i32>,
    /// per original instruction: id of the AIR that proves it (`original_airs.opcode_to_air`, cuda/mod.rs:286-293)
    instr_air: Vec<i32>,
}

impl<ISA: OpenVmIsaHip> PowdrTraceGeneratorHip<ISA> {
    pub fn new(
        apc: IsaApc<BabyBear, ISA>,
        original_airs: OriginalAirs<BabyBear, ISA>,
        config: OriginalVmConfig<ISA>,
        periphery: PowdrPeripheryInstancesHip,
    ) -> Self {
        let json = serde_json::to_vec(&*apc).expect("serialise Apc");
        let mut err = [0u8; 256];
        let h = unsafe { ffi::powdr_apc_from_json(json.as_ptr() as *const _, json.len(), err.as_mut_ptr() as *mut _, err.len()) };
        assert!(!h.is_null(), "powdr_apc_from_json: {}", String::from_utf8_lossy(&err));
        let mut air_ids = HashMap::new();
        let mut instr_air = vec![];
        // cuda/mod.rs:272-293: `self.apc.instructions().zip_eq(self.apc.subs())`, AIR = `original_airs.opcode_to_air[&instr.inner.opcode]`;
        // instructions without substitutions have no AIR row in the tables (any id)
        for (instr, subs) in apc.instructions().zip(apc.subs()) {
            if subs.is_empty() {
                instr_air.push(-1);
                continue;
            }
            let name: String = original_airs.opcode_to_air[&instr.inner.opcode].clone();
            let next = air_ids.len() as i32;
            instr_air.push(*air_ids.entry(name).or_insert(next));
        }
        Self { apc, original_airs, config, periphery, handle: ApcHandle(h), air_ids, instr_air }
    }

    /// cuda/mod.rs:201-401
    pub fn try_generate_witness(
        &self,
        original_arenas: OriginalArenas<DenseRecordArena>,
    ) -> Option<DeviceMatrix<BabyBear>> {
        let mut original_arenas = match original_arenas {
            OriginalArenas::Initialized(arenas) => arenas,
            OriginalArenas::Uninitialized => return None, // the APC was not called (cuda/mod.rs:206-212)
        };
        let num_apc_calls = original_arenas.number_of_calls;

        // the original chips expand their records into "dummy" traces (cuda/mod.rs:215-253); their tracegen is the instruction
        // set's own: `OpenVmIsaHip::create_dummy_chip_complex_hip` (src/isa_hip.rs — an extension trait of this crate, the twin
        // of isa.rs:94-99's `create_dummy_chip_complex_gpu`; `create_dummy_airs` is isa.rs:83-86)
        let chip_inventory = <ISA as OpenVmIsaHip>::create_dummy_chip_complex_hip(
            self.config.config(),
            <ISA as OpenVmISA>::create_dummy_airs(self.config.config(), self.periphery.dummy.clone()).expect("dummy airs"),
            self.periphery.dummy.clone(),
        )
        .expect("dummy chip complex")
        .inventory;
        let mut dummy = vec![ffi::PowdrDeviceMatrix { buffer: core::ptr::null(), width: 0, height: 0 }; self.air_ids.len()];
        let mut keep_alive = vec![];
        for (insertion_idx, chip) in chip_inventory.chips().iter().enumerate().rev() {
            let air_name = chip_inventory.airs().ext_airs()[insertion_idx].name();
            let Some(arena) = original_arenas.take_real_arena(&air_name) else { continue };
            let m: DeviceMatrix<BabyBear> = chip.generate_proving_ctx(arena).common_main;
            if m.height() == 0 {
                continue;
            }
            if let Some(&id) = self.air_ids.get(&air_name) {
                dummy[id as usize] = ffi::PowdrDeviceMatrix {
                    buffer: m.buffer().as_ptr() as *const u32,
                    width: m.width() as i32,
                    height: m.height() as i32,
                };
            }
            keep_alive.push(m);
        }

        let width = unsafe { ffi::powdr_apc_width(self.handle.0) } as usize;
        let height = num_apc_calls.next_power_of_two(); // next_power_of_two_or_zero: 0 calls returned above
        let output = DeviceMatrix::<BabyBear>::with_capacity(height, width);
        // no fill_zero: the library clears exactly the columns no substitution / derived expression covers
        let rc = unsafe {
            ffi::powdr_apc_generate_witness_gpu(
                self.handle.0,
                self.instr_air.as_ptr(),
                dummy.as_ptr(),
                dummy.len(),
                num_apc_calls,
                output.buffer().as_mut_ptr() as *mut u32,
                &self.periphery.real,
            )
        };
        HipError::from_result(rc).unwrap(); // the reference unwraps as well (cuda/mod.rs:334,345,398)
        drop(keep_alive); // hipFree waits for the gather kernels that still read the dummy traces
        Some(output)
    }

    /// The same with the original chips' work folded in (include/powdr_host.h: powdr_apc_generate_witness_from_records): no dummy
    /// chip complex, no dummy traces — the APC trace comes straight from the call records. `records` is the word-major device
    /// buffer of include/powdr_gpu.h's layout (`record_layout()` says which words each instruction owns); the executor
    /// (`PowdrExecutor::execute`, executor/mod.rs:457-528) writes into it what it writes into the per-AIR arenas today: operand
    /// words, overwritten words and previous timestamps of every access.
    pub fn try_generate_witness_from_records(&self, records: &DeviceBuffer<u32>, num_apc_calls: usize) -> Option<DeviceMatrix<BabyBear>> {
        if num_apc_calls == 0 {
            return None;
        }
        let width = unsafe { ffi::powdr_apc_width(self.handle.0) } as usize;
        let height = num_apc_calls.next_power_of_two();
        let output = DeviceMatrix::<BabyBear>::with_capacity(height, width);
        let rc = unsafe {
            ffi::powdr_apc_generate_witness_from_records(
                self.handle.0,
                records.as_ptr(),
                num_apc_calls,
                output.buffer().as_mut_ptr() as *mut u32,
                &self.periphery.real,
            )
        };
        HipError::from_result(rc).unwrap();
        Some(output)
    }

    /// The record flow end to end: the arenas `PowdrExecutor::execute` filled (executor/mod.rs:531-600) -> word-major call records
    /// (src/records_from_arena.rs) -> device -> `powdr_apc_generate_witness_from_records`. `Err`: the block uses a chip the bridge
    /// does not cover yet — the caller falls back to `try_generate_witness` (dummy chips).
    pub fn try_generate_witness_from_arenas(
        &self,
        mut original_arenas: OriginalArenas<DenseRecordArena>,
    ) -> Result<Option<DeviceMatrix<BabyBear>>, BridgeError> {
        let num_apc_calls = match &original_arenas {
            OriginalArenas::Initialized(arenas) => arenas.number_of_calls,
            OriginalArenas::Uninitialized => return Ok(None),
        };
        let (table, words) = self.record_layout();
        let name_of_kind = |kind: u32| -> String { self.air_name_of_kind(kind) };
        let host = records_from_arenas(&table, words, num_apc_calls, &mut original_arenas, &name_of_kind)?;
        let records: DeviceBuffer<u32> = host.as_slice().to_device().expect("records to device");
        Ok(self.try_generate_witness_from_records(&records, num_apc_calls))
    }

    /// AIR name of a chip kind (`POWDR_ORIG_*`): the name `original_airs.opcode_to_air` gives the first opcode of the kind's range
    /// (powdr_amd/original_chips.py OPCODE_RANGES = include/powdr_gpu.h)
    fn air_name_of_kind(&self, kind: u32) -> String {
        const FIRST_OPCODE: [usize; 13] = [512, 517, 528, 544, 560, 520, 549, 565, 534, 596, 593, 592, 576];
        let opcode = openvm_instructions::VmOpcode::from_usize(FIRST_OPCODE[kind as usize]);
        self.original_airs.opcode_to_air[&opcode].clone()
    }

    /// (instruction table of the block, u32 words of one call's record): which instruction keeps a cell, its pc, timestamp offset, row
    /// inside its AIR's block and first record word.
    pub fn record_layout(&self) -> (Vec<ffi::PowdrOrigInstr>, usize) {
        let mut words = 0usize;
        let n = unsafe { ffi::powdr_apc_instruction_table(self.handle.0, core::ptr::null_mut(), &mut words) };
        assert!(n != usize::MAX, "the block uses an opcode outside the thirteen RV32IM chips");
        let mut table = vec![unsafe { core::mem::zeroed::<ffi::PowdrOrigInstr>() }; n];
        unsafe { ffi::powdr_apc_instruction_table(self.handle.0, table.as_mut_ptr(), core::ptr::null_mut()) };
        (table, words)
    }
}

pub struct PowdrChipHip<ISA: OpenVmIsaHip> {
    pub name: String,
    pub record_arena_by_air_name: Rc<RefCell<OriginalArenas<DenseRecordArena>>>,
    pub trace_generator: PowdrTraceGeneratorHip<ISA>,
}

impl<ISA: OpenVmIsaHip> PowdrChipHip<ISA> {
    /// chip.rs:153-175 (`PowdrChipGpu::new`)
    pub fn new(
        precompile: PowdrPrecompile<BabyBear, ISA>,
        original_airs: OriginalAirs<BabyBear, ISA>,
        base_config: OriginalVmConfig<ISA>,
        periphery: PowdrPeripheryInstancesHip,
    ) -> Self {
        let PowdrPrecompile { name, apc, apc_record_arena_gpu, .. } = precompile;
        Self {
            name,
            record_arena_by_air_name: apc_record_arena_gpu,
            trace_generator: PowdrTraceGeneratorHip::new(apc, original_airs, base_config, periphery),
        }
    }
}

/// cuda/mod.rs:404-421, with the HIP matrix type
impl<R, PB: ProverBackend<Matrix = DeviceMatrix<BabyBear>>, ISA: OpenVmIsaHip> Chip<R, PB> for PowdrChipHip<ISA> {
    fn generate_proving_ctx(&self, _: R) -> AirProvingContext<PB> {
        tracing::trace!("Generating air proof input for PowdrChip {}", self.name);
        // reference flow (dummy chips -> dummy traces -> gather). The record flow is `try_generate_witness_from_arenas`: a maintainer
        // switches to it once src/records_from_arena.rs has been compiled against the pinned openvm-rv32im-circuit.
        let trace = self
            .trace_generator
            .try_generate_witness(self.record_arena_by_air_name.take())
            .unwrap_or_else(DeviceMatrix::dummy);
        AirProvingContext { cached_mains: vec![], common_main: trace, public_values: vec![] }
    }
}
