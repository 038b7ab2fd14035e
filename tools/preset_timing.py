#!/usr/bin/env python3
"""Blind-rotation / keyswitch timing of a named parameter preset on the GPU box (not the official bench).
usage: preset_timing.py PRESET [batches, comma separated] ;  BMI_TFHE_LIB selects an A/B build of the library"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np, torch
from bmi_amd import tfhe
if os.environ.get('BMI_TFHE_LIB'):
    tfhe.LIB_PATH = os.environ['BMI_TFHE_LIB']


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "secure128_torus"
    batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,256,2048").split(",")]
    over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("BMI_PARAMS", "").split(",") if kv)}   # e.g. BMI_PARAMS=ks_levels=16,ks_base_log=1
    eng = tfhe.Engine(tfhe.preset_params(preset, **over))
    if os.environ.get('BMI_UNROLL') == '2':
        eng.set_bsk_unroll(2)
    eng.keygen(0x5EED)
    DL = eng.delta_log()
    table = np.random.default_rng(9).integers(-8, 8, 16)
    lid = eng.lut_register(table, 4, DL)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    for B in batches:
        msgs = np.random.default_rng(B).integers(-8, 8, B)
        ct = eng.encrypt(msgs, DL)
        d_in = torch.from_numpy(ct.view(np.int64)).to(dev)
        d_ks = torch.empty((B, eng.P.n + 1), dtype=torch.int64, device=dev)
        d_ids = torch.full((B,), lid, dtype=torch.int32, device=dev)
        d_out = torch.empty((B, eng.P.N + 1), dtype=torch.int64, device=dev)
        eng.keyswitch(d_in, B, d_ks, s)
        eng.blind_rotate(d_ks, d_ids, B, d_out, s); torch.cuda.synchronize()
        ok = list(eng.decrypt(d_out.cpu().numpy().view(np.uint64), DL)) == [int(table[m + 8]) for m in msgs]
        reps = 3 if B >= 1024 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): eng.blind_rotate(d_ks, d_ids, B, d_out, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps): eng.keyswitch(d_in, B, d_ks, s)
        e1.record(); torch.cuda.synchronize()
        ks = e0.elapsed_time(e1) / reps
        print(json.dumps({"preset": preset, "B": B, "br_ms": round(ms, 3), "ks_ms": round(ks, 3), "pbs_per_s": round(B / ((ms + ks) * 1e-3), 1),
                          "decrypts_to_lut": bool(ok)}), flush=True)
    eng.close()


main()
