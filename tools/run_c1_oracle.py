"""BASELINE configs[0] — "guest-sha256 single segment, 2^16 trace rows, reference CPU prover (plumbing, no GPU)": the C1 shape
(1 204 columns, 377 constraints, 954 interactions) at its full 2^16 rows on the CPU ORACLE: the reference's CPU trace
generation convention (row-major dummy traces, sequential row loop, cpu/mod.rs:156-228) + the oracle prover with and without
LogUp, timed on the host cores; then the HIP prover on the same trace (if a GPU is present) and the byte comparison.
usage: python tools/run_c1_oracle.py  -> one JSON line (profiles/r02_c1_oracle.json)"""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np

from oracle import apc_model as om
from oracle import stark_model as sm
from powdr_amd import synth

s = synth.generate("C1", seed=1)
apc = om.load_apc(s.doc)
idx = apc.poly_id_to_index()
calls = (1 << 16) - 3
W, H, log_h = len(idx), 1 << 16, 16
bufs, dims = synth.fill_dummy_traces_numpy(s, calls, 1)
ct = om.build_cpu_tables(apc, idx)
name_to = {n: i for i, (n, _, _, _) in enumerate(dims)}
dummy_rm = [np.ascontiguousarray(bufs[name_to[n]].reshape(dims[name_to[n]][1], dims[name_to[n]][2]).T) for n in ct.air_names]
dummy_w = [dims[name_to[n]][1] for n in ct.air_names]
per = dict(var_bus=3, var_hist=np.zeros(1 << 18, np.uint32), tuple_bus=7, tuple_hist=np.zeros(256 * 2048, np.uint32), sz0=256, sz1=2048,
           bitwise_bus=6, bitwise_hist=np.zeros(2 * 65536, np.uint32))
om.c_oracle()
t0 = time.perf_counter()
vals = om.c_generate_witness(apc, ct, idx, dummy_rm, dummy_w, calls, per)  # row-major [H, W]
t_gen = time.perf_counter() - t0
flat = np.ascontiguousarray(vals.T).reshape(-1)
bc, spans = sm.compile_constraints(apc, idx)
it = sm.compile_interactions(apc, idx)
nq, pb = 100, 16
t0 = time.perf_counter()
p0 = sm.prove(flat, W, log_h, bc, spans, num_queries=nq, pow_bits=pb)
t_v0 = time.perf_counter() - t0
t0 = time.perf_counter()
p1 = sm.prove_logup(flat, W, log_h, bc, spans, *it, num_queries=nq, pow_bits=pb)
t_lg = time.perf_counter() - t0
rec = dict(config="C1 sha256-shaped single segment", cols=W, rows=H, calls=calls, constraints=len(spans), interactions=len(it[0]), host_cores=os.cpu_count(),
           oracle_trace_gen_s=t_gen, oracle_prove_s=t_v0, oracle_prove_logup_s=t_lg, cells=W * H,
           oracle_cells_per_s=W * H / (t_gen + t_v0), oracle_cells_per_s_logup=W * H / (t_gen + t_lg),
           oracle_verify=[int(sm.verify(p0, W, log_h, bc, spans, nq, pb)), int(sm.verify_logup(p1, W, log_h, bc, spans, *it, num_queries=nq, pow_bits=pb))],
           lookups_replayed=int(per["var_hist"].sum() + per["tuple_hist"].sum() + per["bitwise_hist"].sum()))
try:
    import torch

    if torch.cuda.is_available():
        from powdr_amd import prover

        d_t = torch.from_numpy(om.to_monty(flat).view(np.int32)).cuda()
        for key, want, inter in (("hip_v0", p0, None), ("hip_logup", p1, it)):
            pr = prover.Prover(W, bc, spans, num_queries=nq, pow_bits=pb, interactions=inter)
            pr.prove(d_t.data_ptr(), log_h)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = pr.prove(d_t.data_ptr(), log_h)
            rec[key + "_prove_s"] = time.perf_counter() - t0
            rec[key + "_identical_to_oracle"] = bool(len(got) == len(want) and (got == want).all())
            pr.close()
except ImportError:
    pass
print(json.dumps(rec))
