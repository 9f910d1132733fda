"""BASELINE configs[2]-scale run on ONE MI355X: a 3 731-column x 2^22-row APC trace (15.6 G cells, 62.6 GB)
generated through the C ABI from dense synthetic sources, then proven (LDE 125 GB resident, coefficients
panel-wise) and verified on the host. Prints per-kernel times and HBM-roofline fractions.
usage: python tools/run_c3_scale.py [log_height] [queries]"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import bench
from powdr_amd import abi, prover


def run(log_h=22, queries=50, pow_bits=16, verbose=True):
    t0 = time.perf_counter()
    wl = bench.build_workload("C3p", log_h, False, seed=0)
    W, H = wl["W"], wl["H"]
    torch.cuda.synchronize()
    if verbose:
        print(f"C3p: W={W} H=2^{log_h} ({W*H/1e9:.2f} G cells), sources {wl['src_bytes']/1e9:.1f} GB, trace {W*H*4/1e9:.1f} GB, "
              f"setup {time.perf_counter()-t0:.1f} s", flush=True)
    abi.lib.powdr_gpu_timing_enable(1)
    t1 = time.perf_counter()
    wl["apc"].generate_witness_gpu(wl["instr_air"], wl["dummy"], wl["calls"], wl["out"].data_ptr(), wl["per"])
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t1
    tg_timing = abi.timing_report()
    # free the sources before the LDE is allocated
    wl["dummy"].clear(); wl["tensors"].clear()
    torch.cuda.empty_cache()
    pr = prover.Prover(W, *wl["cons"], num_queries=queries, pow_bits=pow_bits)
    pr.prove(wl["out"].data_ptr(), log_h, copy=False)  # first call allocates ~128 GB (several seconds of hipMalloc)
    torch.cuda.synchronize()
    abi.lib.powdr_gpu_timing_enable(1)
    t2 = time.perf_counter()
    proof = pr.prove(wl["out"].data_ptr(), log_h)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    timing = abi.timing_report()
    abi.lib.powdr_gpu_timing_enable(0)
    rc = prover.verify(proof, W, log_h, *wl["cons"], num_queries=queries, pow_bits=pow_bits)
    t4 = time.perf_counter()
    cells = W * H
    algo = {"apc_gather_tile_kernel": 8.0, "ntt_group_kernel<dif>": 8.0, "ntt_group_kernel<dit>": 16.0, "leaf_hash_kernel": 8.0,
            "deep_kernel": 8.0, "quotient_kernel": 8.0, "ext_dot_partial_kernel": 4.0}
    hist_mass = [int(t.to(torch.int64).sum()) for t in (wl["per"].var_hist, wl["per"].tuple_hist, wl["per"].bitwise_hist)]
    report = dict(histogram_mass=hist_mass, workload=f"C3p {W} cols x 2^{log_h} rows", cells=cells, trace_gen_ms=t_gen * 1e3, prove_ms=(t3 - t2) * 1e3,
                  verify_ms=(t4 - t3) * 1e3, verify_rc=rc, cells_per_s_prove=cells / (t3 - t2), cells_per_s_total=cells / (t3 - t2 + t_gen),
                  prover_device_bytes=pr.device_bytes(), proof_bytes=int(len(proof) * 4), kernels={})
    timing.update(tg_timing)
    for k, (cnt, ms) in sorted(timing.items(), key=lambda kv: -kv[1][1]):
        e = dict(launches=cnt, ms=ms)
        if k in algo:
            e["algorithmic_GBps"] = algo[k] * cells / (ms * 1e-3) / 1e9
            e["hbm_frac_of_8TBps"] = e["algorithmic_GBps"] / 8000.0
        report["kernels"][k] = e
    if verbose:
        print(json.dumps(report, indent=1))
    pr.close()
    return report


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 22, int(sys.argv[2]) if len(sys.argv) > 2 else 50)
