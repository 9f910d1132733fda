//! Segments over the GPUs of one node: the loop of `openvm/src/trace_generation.rs:111-141`
//! (`for (seg_idx, segment) in segments.into_iter().enumerate() { ... vm.generate_proving_ctx(..); callback(seg_idx, vm, pk, ctx) }`)
//! run on N devices at once through `pw_prove_segments_multi` (include/powdr_prover.h): one host thread per worker, segments
//! placed by cells (largest first), no data-path collective, the 8-word main commitments all-gathered over RCCL at the end.
//!
//! Every worker owns a REPLICA of what the reference's single loop owns — a `VirtualMachine` with its chip complex on that
//! worker's device and a `HipEngine` whose provers were created there — because device buffers, prover objects and the
//! periphery histograms belong to one device. What stays shared and read-only: the proving key, the executable, the metered
//! segment boundaries.
use crate::engine::{HipEngine, HipSegmentProof};
use crate::ffi;
use core::ffi::{c_int, c_void};
use std::sync::Mutex;

/// One worker's replica of the per-segment pipeline. `prove_segment` is the body of the reference's loop for segment
/// `seg_idx`: preflight the segment from its saved state, `generate_proving_ctx` (trace generation on this worker's device,
/// through `PowdrChipHip`), then `engine.prove_segment(&ctx)`.
pub trait SegmentWorker: Send {
    fn device(&self) -> i32;
    fn engine(&self) -> &HipEngine;
    fn prove_segment(&mut self, seg_idx: usize) -> Result<HipSegmentProof, crate::device::HipError>;
}

struct Shared<'a> {
    workers: Vec<Mutex<&'a mut dyn SegmentWorker>>,
    proofs: Mutex<Vec<Option<HipSegmentProof>>>,
}

unsafe extern "C" fn prove_cb(user: *mut c_void, segment: usize, worker: usize, device: c_int, commitment8: *mut u32) -> c_int {
    let shared = &*(user as *const Shared);
    let mut w = shared.workers[worker].lock().unwrap();
    debug_assert_eq!(w.device(), device);
    match w.prove_segment(segment) {
        Ok(proof) => {
            // main commitment of a pw-stark v1 proof: after the 5 header words and 4 words per AIR
            let off = 5 + 4 * proof.air_ids.len();
            core::ptr::copy_nonoverlapping(proof.words[off..off + 8].as_ptr(), commitment8, 8);
            shared.proofs.lock().unwrap()[segment] = Some(proof);
            0
        }
        Err(e) => e.code(),
    }
}

/// Proves `segment_cells.len()` segments on the workers; returns the proofs in segment order and the merged commitments
/// (8 words per segment) as every device holds them after the RCCL all-gather.
pub fn prove_segments_multi(workers: &mut [&mut dyn SegmentWorker], segment_cells: &[u64]) -> Result<(Vec<HipSegmentProof>, Vec<[u32; 8]>), i32> {
    let devices: Vec<c_int> = workers.iter().map(|w| w.device()).collect();
    let n = segment_cells.len();
    let shared = Shared {
        workers: workers.iter_mut().map(|w| Mutex::new(&mut **w)).collect(),
        proofs: Mutex::new((0..n).map(|_| None).collect()),
    };
    let mut commitments = vec![[0u32; 8]; n];
    let rc = unsafe {
        ffi::pw_prove_segments_multi(devices.as_ptr(), devices.len(), segment_cells.as_ptr(), n, prove_cb,
                                     &shared as *const Shared as *mut c_void, commitments.as_mut_ptr() as *mut u32, core::ptr::null_mut())
    };
    if rc != 0 {
        return Err(rc);
    }
    let proofs = shared.proofs.into_inner().unwrap().into_iter().map(|p| p.expect("every segment was proven")).collect();
    Ok((proofs, commitments))
}
