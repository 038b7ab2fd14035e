# rocprofv3 kernel stats of one encrypted inverse (default 3x3): where a level's time goes
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/invprof
cat > /tmp/inv_once.py <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "bounty-matrix-inversion_amd"))
from bmi_amd import tfhe, inverse_bench
eng = tfhe.Engine(); eng.keygen(0x5EED)
print(json.dumps(inverse_bench.run(eng, (int(sys.argv[1]),))))
PY
rm -rf gpurun_out/invprof/*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/invprof -- python3 /tmp/inv_once.py ${1:-3} > gpurun_out/invprof/log.txt 2>&1
cat gpurun_out/invprof/*/*_kernel_stats.csv | cut -c1-200
