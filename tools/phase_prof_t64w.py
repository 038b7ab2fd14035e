#!/usr/bin/env python3
"""Per-phase cycles (s_memtime) of the 16 wavefronts of workgroup 0 of k_blind_rotate_w_t64f (debug build: make -C csrc prof).
The marks wait for the wavefront's outstanding LDS / scalar-memory operations, so a phase's figure includes its own waits."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np, torch
from bmi_amd import tfhe
tfhe.LIB_PATH = tfhe.LIB_PATH.replace("libbmi_tfhe.so", "libbmi_tfhe_prof.so")
NAMES = ["loop head, key requests", "first forward task", "second forward task", "barrier A -> B", "products (2 l rows) + inverse butterfly",
         "barrier, sums to LDS, barrier", "inverse quarter, rounding, atomics", "closing barrier (+ re-centring)"]
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256").split(",")]:
    eng = tfhe.Engine(tfhe.preset_params("secure128_torus")); eng.keygen(0x5EED)
    DL = eng.delta_log(); lid = eng.lut_register(np.arange(-8, 8), 4, DL)
    ct = eng.encrypt(np.random.default_rng(1).integers(-8, 8, B), DL)
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    d_in = torch.from_numpy(ct.view(np.int64)).to(dev)
    d_small = torch.empty((B, eng.P.n + 1), dtype=torch.int64, device=dev)
    d_ids = torch.full((B,), lid, dtype=torch.int32, device=dev)
    d_out = torch.empty((B, eng.P.N + 1), dtype=torch.int64, device=dev)
    eng.keyswitch(d_in, B, d_small, s)
    for _ in range(2): eng.blind_rotate(d_small, d_ids, B, d_out, s)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)(); assert tfhe.load_library().bmi_debug_phase_prof_t64w(buf) == 0
    a = np.array(buf[:128], dtype=np.float64).reshape(16, 8) / eng.P.n
    print(f"B = {B}: s_memtime ticks per CMUX, wavefronts 0-15 of workgroup 0")
    for k, nm in enumerate(NAMES):
        print(f"  {nm:44s} " + " ".join(f"{a[w, k]:6.0f}" for w in range(16)))
    print(f"  {'total':44s} " + " ".join(f"{a[w].sum():6.0f}" for w in range(16)))
    eng.close()
