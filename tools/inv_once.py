#!/usr/bin/env python3
"""One encrypted inverse on a named preset / modulus, for kernel-level profiling of the executor's per-level work:
    rocprofv3 --kernel-trace --stats -d gpurun_out/invprof -- python3 tools/inv_once.py secure128_torus 3 [unroll]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np
from bmi_amd import tfhe
from bmi_amd.main import EncryptedMatrixInversion

spec = sys.argv[1] if len(sys.argv) > 1 else "secure128_torus"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
unroll = len(sys.argv) > 3 and sys.argv[3] == "unroll"
ln, ints = {2: (20, 8), 3: (30, 12), 4: (40, 16)}[n]
eng = tfhe.Engine(tfhe.default_params(q_bits=int(spec)) if spec.isdigit() else tfhe.preset_params(spec))
if unroll:
    eng.set_bsk_unroll(2)
eng.keygen(0x5EED)
emi = EncryptedMatrixInversion(n, None, 2, ln, ints, False, False, engine=eng, unroll=unroll)
np.random.seed(1234 + n)
M = np.random.randn(n, n) * 100
q, s = emi.quantize(M)
enc = emi.encrypt(q, s)
emi._executor()
emi.evaluate(enc)
t0 = time.time()
res = emi.evaluate(enc)
dt = time.time() - t0
ok = np.array_equal(emi.decrypt(res), emi.simulate(q, s))
print(f"{spec} {n}x{n}: evaluate {dt:.3f} s, {emi.program.depth} levels, {dt / emi.program.depth * 1e3:.3f} ms per level, matches the plaintext circuit: {ok}")
eng.close()
