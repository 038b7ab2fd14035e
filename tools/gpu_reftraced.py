#!/usr/bin/env python3
"""Times the reference's unmodified qfloat_matrix_inverse (traced by tools/gen_ref_traced.py, stored in
tests/golden/ref_traced_inverse.json.gz) on ciphertexts on the N = 2048 parameter set, next to this repo's restated,
fused circuit of the same function on the same matrix.  Writes gpurun_out/reference_traced_inverse.json."""
import gzip, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
from bmi_amd import tfhe
from bmi_amd.circuit import Circuit
from bmi_amd.executor import Executor
from bmi_amd.main import trace_inverse

data = json.load(gzip.open(os.path.join(REPO, "tests", "golden", "ref_traced_inverse.json.gz"), "rt"))
out = {}
e = tfhe.Engine(tfhe.default_params(q_bits=49, log_N=11))
e.keygen(0x5EED)
e4 = tfhe.Engine(tfhe.default_params(q_bits=49))
e4.keygen(0x5EED)
for case in data["cases"]:
    v = case["vectors"][0]
    rec = {}
    for tag, circ, eng, bits in (("reference_unmodified", Circuit.from_dict(case["circuit"]), e, 5),
                                 ("restated_fused", trace_inverse(case["n"], case["len"], case["ints"], 2, False, False), e4, 4)):
        ex = Executor(circ, eng)
        dl = eng.delta_log(bits)
        ct = eng.encrypt(v["inputs"], dl)
        if case["n"] == 2:
            ex.run(ct)
        t = time.time()
        res = ex.run(ct)
        dt = time.time() - t
        ok = list(eng.decrypt(res, dl)) == v["expected"]
        rec[tag] = {"pbs": len(circ.nodes), "depth": len(circ.levels()), "msg_bits": bits,
                    "N": 1 << eng.P.log_N, "evaluate_s": round(dt, 3), "equals_reference_plaintext": ok}
        print(case["name"], tag, rec[tag], flush=True)
    out[case["name"]] = rec
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "reference_traced_inverse.json"), "w"), indent=1)
