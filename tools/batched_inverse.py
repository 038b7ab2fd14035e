#!/usr/bin/env python3
"""Batched encrypted inverses on the GPU box (EncryptedMatrixInversion.evaluate_many: B matrices in one walk of the levels) on
the default engine, with the unrolled key and on the 128-bit-secure torus set.   usage: batched_inverse.py ["n:batch,..."]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
from bmi_amd import tfhe, inverse_bench

cases = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[1] if len(sys.argv) > 1 else "3:4,3:16,3:32,2:64,4:8").split(",")]
eng = tfhe.Engine(); eng.keygen(0x5EED)
for n, b in cases:
    print("torus64", json.dumps(inverse_bench.run_batched(eng, n, b)), flush=True)
eng.close()
eng = tfhe.Engine(); eng.set_bsk_unroll(2); eng.keygen(0x5EED)
print("torus64 unrolled", json.dumps(inverse_bench.run_batched(eng, 3, 16)), flush=True)
eng.close()
eng = tfhe.Engine(tfhe.preset_params("secure128_torus")); eng.keygen(0x5EED)
print("secure128_torus", json.dumps(inverse_bench.run_batched(eng, 3, 8)), flush=True)
eng.close()
