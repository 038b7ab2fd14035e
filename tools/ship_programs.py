#!/usr/bin/env python3
"""(Re)writes bounty-matrix-inversion_amd/bmi_amd/programs/: the compiled programs of the BASELINE configurations whose trace takes
minutes (8 x 8 config 5; the 10 x 10 of the reference's own driver), in the compact form of Program.save_compact.  Their file names
carry the tracer fingerprint, so they must be regenerated after ANY change to the tracer's sources - circuit.py, base_p_arrays.py,
qfloat.py, qfloat_matrix_inversion.py, program.py, inverse_circuit.py (NOT the API wrapper main.py) - (tests/test_host_qfloat.py checks
that they are current); stale ones are removed here."""
import glob, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
from bmi_amd import main, program

CONFIGS = [(8, 48, 16), (10, 23, 9)]
os.environ["BMI_CACHE_DIR"] = os.environ.get("BMI_CACHE_DIR") or program.cache_dir()
d = program.shipped_dir()
os.makedirs(d, exist_ok=True)
keep = set()
for n, ln, ints in CONFIGS:
    t = time.time()
    prog, info = main.compile_inverse(n, ln, ints)
    src = info["path"]
    if info.get("shipped"):
        keep.add(src)
        print(n, "current:", os.path.basename(src))
        continue
    dst = os.path.join(d, os.path.basename(src).replace(".npz", ".prog.xz"))
    prog.save_compact(dst)
    back = program.Program.load_compact(dst)
    assert all((getattr(back, k) == getattr(prog, k)).all() for k in program.Program.ARRAYS) and back.meta == prog.meta
    keep.add(dst)
    print(n, ln, ints, f"{time.time() - t:.0f} s", os.path.basename(dst), os.path.getsize(dst) // 1024, "KiB")
for f in glob.glob(os.path.join(d, "*.prog.xz")):
    if f not in keep:
        os.remove(f)
        print("removed stale", os.path.basename(f))
