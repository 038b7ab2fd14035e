# timing of the wider parameter sets (argument: log_N = 11 or 12): PBS latency / throughput and encrypted inverses
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python - "$@" <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/wide.log
import sys, os, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "bounty-matrix-inversion_amd"))
import numpy as np, torch
from bmi_amd import tfhe, inverse_bench
log_N = int(sys.argv[1]) if len(sys.argv) > 1 else 11
eng = tfhe.Engine(tfhe.default_params(q_bits=49, log_N=log_N)); eng.keygen(0x5EED)
dl = eng.delta_log(); lid = eng.lut_register(np.arange(-8, 8), 4, dl)
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
for B in (1, 256, 2048):
    msgs = np.random.default_rng(B).integers(-8, 8, B)
    d_in = torch.from_numpy(eng.encrypt(msgs, dl).view(np.int64)).to(dev)
    d_ids = torch.full((B,), lid, dtype=torch.int32, device=dev); d_out = torch.empty_like(d_in)
    eng.pbs(d_in, d_ids, B, d_out, s); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.pbs(d_in, d_ids, B, d_out, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    ok = bool(np.array_equal(eng.decrypt(d_out.cpu().numpy().view(np.uint64), dl), msgs))
    print(json.dumps({"N": 1 << log_N, "B": B, "ms": round(dt * 1e3, 3), "pbs_per_s": round(B / dt, 1), "decrypt_ok": ok}), flush=True)
print(json.dumps(inverse_bench.run(eng, (2, 3) if log_N == 11 else (2,))))
PY
