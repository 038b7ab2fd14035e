#!/usr/bin/env python3
"""Encrypted inverses with the (l, Bg) = (2, 2^15) bootstrap decomposition (2/3 of the transforms of a CMUX; output noise
2^-16.1 on the 49-bit field, below the l = 3 set's 2^-15.85) - an option beside the north star's l = 3."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
from bmi_amd import tfhe, inverse_bench
out = {}
for qb in (49, 65):
    eng = tfhe.Engine(tfhe.default_params(q_bits=qb, bs_levels=2)); eng.keygen(0x5EED)
    rep = inverse_bench.run(eng, (2, 3, 4) if qb == 49 else (2, 3))
    out[f"q_bits_{qb}_l2"] = {k: {x: v[x] for x in ("evaluate_s", "end_to_end_s", "ms_per_level", "pbs", "depth", "matches_plaintext_circuit", "max_abs_err_vs_numpy")} for k, v in rep.items()}
    eng.close()
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(REPO, "gpurun_out", "inverse_l2.json"), "w"), indent=1)
