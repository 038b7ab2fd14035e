# A/B of kernel-library builds on a wider parameter set: bash tools/gpu_ab_wide.sh "name1 name2" [log_N] [rounds]  (libs: lib/ab_<name>.so)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in $(seq 1 ${3:-2}); do
  for n in $1; do
    echo "== $n"
    BMI_TFHE_LIB=$GRAFT_REPO_ROOT/bounty-matrix-inversion_amd/lib/ab_$n.so timeout -k 10 200 python - ${2:-11} <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "bounty-matrix-inversion_amd"))
import numpy as np, torch
from bmi_amd import tfhe
tfhe.LIB_PATH = os.environ["BMI_TFHE_LIB"]
log_N = int(sys.argv[1])
eng = tfhe.Engine(tfhe.default_params(q_bits=49, log_N=log_N)); eng.keygen(0x5EED)
dl = eng.delta_log(); lid = eng.lut_register(np.arange(-8, 8)[::-1].copy(), 4, dl)
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
for B in (1, 64):
    msgs = np.random.default_rng(B).integers(-8, 8, B)
    d_in = torch.from_numpy(eng.encrypt(msgs, dl).view(np.int64)).to(dev)
    d_ids = torch.full((B,), lid, dtype=torch.int32, device=dev); d_out = torch.empty_like(d_in)
    eng.pbs(d_in, d_ids, B, d_out, s); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): eng.pbs(d_in, d_ids, B, d_out, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    ok = bool(np.array_equal(eng.decrypt(d_out.cpu().numpy().view(np.uint64), dl), -msgs - 1))
    print(json.dumps({"N": 1 << log_N, "B": B, "ms": round(dt * 1e3, 3), "decrypt_ok": ok}), flush=True)
PY
  done
done
