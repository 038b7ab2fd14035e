#!/usr/bin/env python3
"""Per-phase wall-clock cycles of one workgroup of k_blind_rotate_tp49 (debug build: make -C csrc prof)."""
import os, sys, json, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np, torch
from bmi_amd import tfhe
tfhe.LIB_PATH = tfhe.LIB_PATH.replace("libbmi_tfhe.so", "libbmi_tfhe_prof.so")

NAMES = ["rotate+decompose", "forward NTT x3", "publish+key loads+barrier", "MAC x3", "barrier 2", "inverse+update", "-", "loop head"]
PIPE_NAMES = ["wait accumulator + tiles free", "forward task", "wait all transforms", "multiply", "wait sums of my output", "inverse half", "-", "loop head + key request"]
LAT_NAMES = ["key loads issue + decompose + forward (waves 0-5)", "barrier 1 wait", "MAC", "barrier 2 wait", "inverse + update (waves 0-1)", "barrier 3 wait", "-", "loop head"]

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    log_N = int(sys.argv[3]) if len(sys.argv) > 3 else 10   # 11: k_blind_rotate_wide49 (8 waves; pass variant 0)
    unroll = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # 2: k_blind_rotate_lat2u_49 (pass variant 2 for the phase names); cycles are then per PAIR of coefficients / 2
    q_bits = int(sys.argv[5]) if len(sys.argv) > 5 else 49  # 65 with unroll = 2: k_blind_rotate_lat2u_t64 (the unrolled torus kernel)
    eng = tfhe.Engine(tfhe.default_params(q_bits=q_bits, log_N=log_N)); eng.set_bsk_unroll(unroll); eng.keygen(0x5EED)
    DL = eng.delta_log()
    lid = eng.lut_register(np.arange(-8, 8), 4, DL)
    ct = eng.encrypt(np.random.default_rng(1).integers(-8, 8, B), DL)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d_in = torch.from_numpy(ct.view(np.int64)).to(dev)
    d_small = torch.empty((B, eng.P.n + 1), dtype=torch.int64, device=dev)
    d_ids = torch.full((B,), lid, dtype=torch.int32, device=dev)
    d_out = torch.empty((B, (1 << log_N) + 1), dtype=torch.int64, device=dev)
    variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # 1: pair kernel (4 waves), 2: latency kernel (8 waves)
    eng.set_kernel_variant(variant)
    eng.keyswitch(d_in, B, d_small, s)
    for _ in range(2):
        eng.blind_rotate(d_small, d_ids, B, d_out, s)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    lib = tfhe.load_library()
    rc = (lib.bmi_debug_phase_prof_unrolled_t64 if (unroll == 2 and q_bits == 65) else lib.bmi_debug_phase_prof_unrolled if unroll == 2 else lib.bmi_debug_phase_prof)(buf)
    assert rc == 0, rc
    a = np.array(buf[:], dtype=np.float64).reshape(16, 8)
    WIDE_NAMES = ["decompose + forward tasks (both rounds)", "barrier after tasks", "MAC (both rounds)", "barrier after MAC", "sums + barrier", "inverse + update (waves 0-3)", "barrier after inverse", "loop head"]
    for w in range(8 if log_N > 10 else {1: 4, 2: 16, 3: 4, 4: 8}[variant]):
        tot = a[w].sum()
        print(json.dumps({"wave": w, "total_cycles": tot, "per_cmux": round(tot / eng.P.n, 1),
                          "phases_cycles_per_cmux": {n: round(v / eng.P.n, 0) for n, v in zip(WIDE_NAMES if log_N > 10 else (PIPE_NAMES if (unroll == 2 and q_bits == 49 and os.environ.get('BMI_PIPE_NAMES', '1') == '1') else LAT_NAMES if variant in (2, 4) else NAMES), a[w]) if n != "-"}}))

main()
