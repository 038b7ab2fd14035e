# L1 (TCP) hit behaviour of the throughput kernel with and without the workgroup resync (rocprofv3 PMC pass)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/l1
cat > /tmp/one_br.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "bounty-matrix-inversion_amd"))
import numpy as np, torch
from bmi_amd import tfhe
if os.environ.get("BMI_TFHE_LIB"): tfhe.LIB_PATH = os.environ["BMI_TFHE_LIB"]
eng = tfhe.Engine(); eng.keygen(1)
dl = eng.delta_log(); B = 8192
lid = eng.lut_register(np.arange(-8, 8), 4, dl)
ct = eng.encrypt(np.random.default_rng(1).integers(-8, 8, B), dl)
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
d_in = torch.from_numpy(ct.view(np.int64)).to(dev); d_ids = torch.full((B,), lid, dtype=torch.int32, device=dev)
d_out = torch.empty_like(d_in)
for _ in range(2): eng.pbs(d_in, d_ids, B, d_out, s)
torch.cuda.synchronize()
PY
for n in rs0 rs4; do
  rm -rf gpurun_out/l1/$n
  BMI_TFHE_LIB=$GRAFT_REPO_ROOT/bounty-matrix-inversion_amd/lib/ab_$n.so rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d gpurun_out/l1/$n -- python3 /tmp/one_br.py > gpurun_out/l1/$n.log 2>&1 || echo "pmc pass failed for $n"
done
python3 - <<'PY'
import csv, glob
for n in ("rs0", "rs4"):
    for f in glob.glob(f"gpurun_out/l1/{n}/*/*_counter_collection.csv"):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "tpx49" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        print(n, {k: sum(v.values()) / len(v) for k, v in acc.items()})
PY
