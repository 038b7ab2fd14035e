#!/usr/bin/env python3
"""BASELINE config 5 (8 x 8, len 48, ints 16) on the 128-bit-secure torus set on the GPU box: evaluate wall-clock, check against the
plaintext program, the circuit's error budget (profiles/r04_secure_8x8.txt)."""
import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "bounty-matrix-inversion_amd"))
from bmi_amd import tfhe, inverse_bench
eng = tfhe.Engine(tfhe.preset_params("secure128_torus")); eng.keygen(0x5EED)
print("progress: keys ready", flush=True)
r = inverse_bench.run(eng, (8,))
print(json.dumps(r), flush=True)
from bmi_amd.main import EncryptedMatrixInversion
emi = EncryptedMatrixInversion(8, None, 2, 48, 16, False, False, engine=eng)
print(json.dumps({k: (v if not hasattr(v, "tolist") else None) for k, v in emi.error_budget.items() if k in ("p_fail", "lookups", "worst_margin_sigma")} if emi.error_budget else emi.program.failure_probability(eng)["p_fail"]))
