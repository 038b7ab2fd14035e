#!/usr/bin/env python3
"""Start / end time of every workgroup of ONE launch of k_blind_rotate_lat2u_49 (debug build: make -C csrc prof): where does a
256-workgroup launch lose its 0.4 ms against 256 x the time of a lone workgroup?  usage: wg_times.py [B]"""
import os, sys, json, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np, torch
from bmi_amd import tfhe
tfhe.LIB_PATH = tfhe.LIB_PATH.replace("libbmi_tfhe.so", "libbmi_tfhe_prof.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = tfhe.Engine(tfhe.default_params(q_bits=49)); eng.set_bsk_unroll(2); eng.keygen(0x5EED)
DL = eng.delta_log(); lid = eng.lut_register(np.arange(-8, 8), 4, DL)
ct = eng.encrypt(np.random.default_rng(1).integers(-8, 8, B), DL)
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
d_in = torch.from_numpy(ct.view(np.int64)).to(dev)
d_small = torch.empty((B, eng.P.n + 1), dtype=torch.int64, device=dev)
d_ids = torch.full((B,), lid, dtype=torch.int32, device=dev); d_out = torch.empty((B, 1025), dtype=torch.int64, device=dev)
eng.keyswitch(d_in, B, d_small, s)
for _ in range(3):
    eng.blind_rotate(d_small, d_ids, B, d_out, s)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 2048)()
assert tfhe.load_library().bmi_debug_wg_times_unrolled(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(1024, 2)[:B].astype(np.float64) * 10e-3     # microseconds (100 MHz ticks)
t0 = t[:, 0].min()
start, dur = t[:, 0] - t0, t[:, 1] - t[:, 0]
print(json.dumps({"B": B, "launch_us": round(float(t[:, 1].max() - t0), 1), "start_us": {"max": round(float(start.max()), 1), "p50": round(float(np.median(start)), 1)},
                  "duration_us": {"min": round(float(dur.min()), 1), "p50": round(float(np.median(dur)), 1), "p90": round(float(np.percentile(dur, 90)), 1), "max": round(float(dur.max()), 1)},
                  "duration_by_xcd_us": [round(float(dur[x::8].mean()), 1) for x in range(8)] if B >= 8 else None}))
