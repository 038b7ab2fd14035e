#!/usr/bin/env python3
"""The README's four precision presets (README.md:107-114) for n = 2, 3, 4 on one MI355X: wall-clock of the encrypted
inverse (compile as found: cached or cold, encrypt + evaluate + decrypt), error against numpy, equality with the plaintext
evaluation of the same program.  The reference ran only "low" at n = 2, 3 in FHE (README.md:129-142)."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np
from bmi_amd import tfhe
from bmi_amd.main import EncryptedMatrixInversion

PRESETS = {"low": (23, 9, False), "medium": (31, 16, False), "medium+": (31, 16, True), "high": (40, 20, True)}
eng = tfhe.Engine(); eng.keygen()
out = {}
for n in (2, 3, 4):
    for name, (ln, ints, td) in PRESETS.items():
        np.random.seed(3234 + n); M = np.random.randn(n, n) * 100
        t0 = time.time(); emi = EncryptedMatrixInversion(n, None, 2, ln, ints, td, False, engine=eng); emi._executor(); tc = time.time() - t0
        q, s = emi.quantize(M)
        if n <= 3: emi.evaluate(emi.encrypt(q, s))
        t0 = time.time(); dec = emi.decrypt(emi.evaluate(emi.encrypt(q, s))); tr = time.time() - t0
        out[f"{n}x{n}_{name}"] = {"len": ln, "ints": ints, "true_division": td, "compile_s": round(tc, 3), "compile_cached": bool(emi.compile_info["cached"]),
                                  "encrypt_run_decrypt_s": round(tr, 3), "pbs": emi.program.n_nodes, "depth": emi.program.depth,
                                  "matches_plaintext_circuit": bool(np.array_equal(dec, emi.simulate(q, s))),
                                  "max_abs_err_vs_numpy": float(np.max(np.abs(emi.dequantize(dec) - np.linalg.inv(M))))}
        print(f"{n}x{n}_{name}", out[f"{n}x{n}_{name}"], flush=True)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "readme_presets.json"), "w"), indent=1)
