#!/usr/bin/env python3
"""Runs the reference's OWN FHE test file (tests/test_qfloat_fhe.py under /root/reference, unmodified) against the
Concrete-compatible front end bmi_amd/compat (plaintext back end: no GPU in the build container) - `fhe.Compiler(...).compile(inputset)` -> circuit.encrypt / run / decrypt,
the reference's assertions on the decrypted floats included - and keeps what it compiled and ran as data:
tests/golden/ref_own_fhe_tests.json.gz holds, per test function, the compiled circuits (this repo's IR), the inputs the
reference's tests fed them and the outputs they produced when the reference's assertions passed.  The CPU suite
re-simulates them, the GPU suite replays them on ciphertexts.  Build container only (needs /root/reference)."""
import gzip, json, os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
KEEP = {"add_qfloats": 3, "div_qfloats": 2, "neg_qfloats": 2}   # circuits kept per test function (default 1)

with tempfile.TemporaryDirectory() as tmp:
    rec = os.path.join(tmp, "rec.json.gz")
    env = dict(os.environ, ENCSHIM_RECORD=rec, BMI_COMPAT_BACKEND="simulate", PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(REPO, "bounty-matrix-inversion_amd", "bmi_amd", "compat"), os.path.join(REF, "matrix_inversion"),
                                           os.path.join(REPO, "bounty-matrix-inversion_amd")]))
    out = subprocess.run([sys.executable, os.path.join("tests", "test_qfloat_fhe.py")], cwd=REF, env=env,
                         capture_output=True, text=True, timeout=3600)
    tail = out.stderr.strip().splitlines()[-3:]
    print("\n".join(tail))
    assert out.returncode == 0 and tail[-1] == "OK", "the reference's own tests did not pass"
    ran = [l for l in tail if l.startswith("Ran ")][0]
    data = json.load(gzip.open(rec, "rt"))
kept, seen = [], {}
for c in data["cases"]:
    seen[c["function"]] = seen.get(c["function"], 0) + 1
    if seen[c["function"]] <= KEEP.get(c["function"], 1):
        kept.append(c)
dst = os.path.join(REPO, "tests", "golden", "ref_own_fhe_tests.json.gz")
with gzip.GzipFile(dst, "wb", mtime=0) as f:
    f.write(json.dumps({"generator": "tools/gen_ref_fhe_tests.py", "reference_file": "tests/test_qfloat_fhe.py",
                        "unittest_summary": ran + " OK", "circuits_compiled": seen, "cases": kept}).encode())
for c in kept:
    print(c["function"], "msg_bits", c["msg_bits"], "pbs", c["pbs"], "depth", c["depth"], "runs", len(c["runs"]))
print("wrote", dst, os.path.getsize(dst), "bytes")
