"""VALU instructions per Poseidon2 permutation of the leaf-hash kernels from a rocprofv3 --pmc pass (csv output) of
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d DIR -- python bench.py --log-height 18 --steps 1 --warmup 1 --no-cpu-baseline --no-logup-leg --no-copy-ceiling
plus the measured issue cost of the Montgomery instruction mix (tools/microbench_opcodes.hip, its JSON line).
usage: pmc_valu_json.py PMC_DIR OPCODES_TXT LOG_HEIGHT WIDTH N_PROOFS > profiles/r02_valu_model.json"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

d, opc, log_h, width, n_proofs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
agg, disp = defaultdict(lambda: defaultdict(float)), defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"].replace("pw::(anonymous namespace)::", "")))
        k = re.sub(r"<\d+>$", "", k)  # leaf_hash_kernel<6> (the VGPR-cap variants) -> leaf_hash_kernel
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
lh = agg.get("leaf_hash_kernel") or agg.get("leaf_hash_cols_kernel")
H = 1 << log_h
perms = n_proofs * (2 * H * ((width + 7) // 8) + 2 * H)  # trace LDE rows x ceil(W/8) + quotient LDE rows x 1
instr = lh["SQ_INSTS_VALU"]  # wave instructions summed over waves; x 64 lanes / 64 lanes per wave-permutation
per_perm = instr / (perms / 64)
mix = json.loads([l for l in open(opc) if l.startswith("JSON ")][0][5:])
out = dict(valu_instr_per_perm=per_perm, cycles_per_wave_instr=mix["signed_montgomery"],
           source=f"SQ_INSTS_VALU {instr:.4g} over {len(disp['leaf_hash_kernel'])} leaf-hash dispatches = {perms} permutations "
                  f"(C2 AIR at 2^{log_h} rows, {n_proofs} proofs); issue cost = measured cycles per wave instruction of the signed Montgomery product "
                  f"(2 v_mad_i64_i32 + v_mul_lo_u32, what 97 % of the kernel's instruction stream looks like) at 8 waves/SIMD, nominal 2.4 GHz (profiles/r02_microbench_opcodes.txt)",
           counters={k: v for k, v in lh.items()}, opcode_cycles=mix)
print(json.dumps(out, indent=1))
