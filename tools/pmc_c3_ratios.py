"""Adds `algorithmic_bytes` and `ratio_to_algorithmic` to the per-kernel HBM traffic of the configs[2] proof
(gpurun_out|profiles/r06_pmc_traffic_c3_logup.json from tools/pmc_c3_logup_r06.sh). The algorithmic figures are those of
the mode the proof ran in — trace handed over, TWO sub-cosets per pass — counted per PROOF for the kernels that move the bytes:
W = 3 731 main, Wp = 4 632 permutation, 8 quotient columns; H = 2^22, N = 2 H, sub-coset m = H.
usage: python tools/pmc_c3_ratios.py <traffic.json>   (rewrites the file in place)"""
import json
import sys

W, Wp, H = 3731, 4632, 1 << 22
N, B = 2 * H, 4
cols = W + Wp
# forward sub-coset transforms: passes over (main + perm) per proof = 2 sub-cosets x {commitment, quotient} with their outputs STORED
# (+ 2 for the query rows: first group only, nothing stored); first group reads the coefficients (fold of 2: every coefficient word once
# per sub-coset), the strided groups (two per transform at 2^22 points) read and write the sub-coset's words once each
# + what the TOOL adds per proof: pw_trace_from_coefficients restores the handed-over trace for the next proof (W columns, one forward
# transform: the same three kernels)
restore = W * H * B
first_reads = 6 * cols * H * B + restore
first_writes = 4 * cols * H * B + restore
strided = 2 * 4 * cols * H * B + 2 * restore  # two strided groups per stored pass, in = out
ALGO = {
    "leaf_hash_kernel": (8 * (cols + 8) * H, 0, "8 B x committed columns x rows (the LDE rows, hashed once)"),
    "ntt_group_kernel<false, 12, 2, true, 256>": (first_reads, first_writes,
                                                  "first stage group of the sub-coset transforms: 6 passes read every coefficient word once, 4 of them store (+ the tool's restoring transform)"),
    "ntt_group_kernel<false, 12, 0, false, 256>": (strided, strided,
                                                   "strided groups: two per stored sub-coset pass, each reads and writes the sub-coset once (+ the tool's restoring transform)"),
    "ntt_group_kernel<true, 12, 0, false, 256>": (2 * cols * H * B, 2 * cols * H * B, "inverse transforms: two strided groups, in = out"),
    "ntt_group_kernel<true, 12, 0, true, 256>": (cols * H * B, cols * H * B, "inverse transforms: the contiguous group"),
    "ext_dot_partial_kernel": ((W + Wp + 8) * H * B, 0, "openings: every column read once (the permutation columns serve both points in one pass)"),
    "ext_lincomb_kernel": (cols * H * B, 0, "DEEP numerator: one pass over the coefficient arrays"),
}
path = sys.argv[1]
t = json.load(open(path))
for k, (rd, wr, why) in ALGO.items():
    e = t["kernels"].get(k)
    if not e:
        continue
    e["algorithmic_bytes"] = dict(fetch=rd, write=wr, what=why)
    e["ratio_to_algorithmic"] = dict(fetch=round(e["fetch_bytes_corrected"] / rd, 3) if rd else None, write=round(e["write_bytes"] / wr, 3) if wr else None)
tot_f = sum(e["fetch_bytes_corrected"] for e in t["kernels"].values())
tot_w = sum(e["write_bytes"] for e in t["kernels"].values())
t["totals"] = dict(fetch_bytes_corrected=tot_f, write_bytes=tot_w)
json.dump(t, open(path, "w"), indent=1)
for k in ALGO:
    e = t["kernels"].get(k)
    if e:
        print(f"{k:48s} fetched {e['fetch_bytes_corrected'] / 1e9:8.1f} GB  written {e['write_bytes'] / 1e9:8.1f} GB  ratio {e['ratio_to_algorithmic']}")
print(f"whole proof: {tot_f / 1e9:.0f} GB fetched + {tot_w / 1e9:.0f} GB written")
