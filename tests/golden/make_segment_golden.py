"""Regenerates tests/golden/pw_stark_segment_T.json: SHA-256 of the ORACLE's segment proofs (pw-stark v1, with and without
LogUp) of a fixed synthetic segment. Run from the repository root: python tests/golden/make_segment_golden.py"""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import stark_model as sm  # noqa: E402
from tests.test_segment_proof import synthetic_airs  # noqa: E402

spec = [["T0", 30], ["T1", 200], ["T0", 5], ["T1", 64], ["T0", 30], ["T1", 1500]]
g = dict(spec=spec, seed0=21, num_queries=6, pow_bits=5)
airs = synthetic_airs([tuple(x) for x in spec], seed0=g["seed0"])
for key, logup in (("sha256_v1", False), ("sha256_v1_logup", True)):
    pf = sm.prove_segment(airs, num_queries=g["num_queries"], pow_bits=g["pow_bits"], logup=logup)
    assert sm.verify_segment(pf, airs, g["num_queries"], g["pow_bits"], logup)[0] == 0
    g[key] = hashlib.sha256(pf.astype("<u4").tobytes()).hexdigest()
    g["words" + ("_logup" if logup else "")] = int(len(pf))
(Path(__file__).parent / "pw_stark_segment_T.json").write_text(json.dumps(g, indent=1) + "\n")
print(g)
