set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 ${BMI_T:-900} python - "$@" <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/invbench.log
import sys, json, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "bounty-matrix-inversion_amd"))
from bmi_amd import tfhe, inverse_bench
sizes = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,3").split(","))
eng = tfhe.Engine(); eng.keygen(0x5EED)
print(json.dumps(inverse_bench.run(eng, sizes), indent=1))
PY
