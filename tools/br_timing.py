#!/usr/bin/env python3
"""Quick A/B timing of the blind-rotation kernels on the GPU box (not the official bench)."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np, torch
from bmi_amd import tfhe
from oracle import tfhe_oracle as to
if os.environ.get('BMI_TFHE_LIB'):  # A/B builds of the kernel library (tools only)
    tfhe.LIB_PATH = os.environ['BMI_TFHE_LIB']

def main():
    batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,64,4096").split(",")]
    variants = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1").split(",")]
    qb = int(sys.argv[3]) if len(sys.argv) > 3 else None
    kw = {{"log_n": "log_N"}.get(k[5:].lower(), k[5:].lower()): int(v) for k, v in os.environ.items() if k.startswith("BMIP_")}   # e.g. BMIP_BS_LEVELS=2, BMIP_LOG_N=11
    eng = tfhe.Engine(tfhe.default_params(q_bits=qb, **kw))
    if os.environ.get('BMI_BSK_PRECISION'): eng.set_bsk_precision(int(os.environ['BMI_BSK_PRECISION']))
    unroll = int(os.environ.get('BMI_UNROLL', '1'))   # 2: the unrolled blind rotation (one kernel for every variant / batch)
    eng.set_bsk_unroll(unroll)
    eng.keygen(0x5EED)
    DL = eng.delta_log()
    sk_small, sk_big, bsk, ksk = eng.export_keys()
    octx = to.Ctx(to.default_params(q_bits=eng.q_bits, **kw), bsk, ksk)
    if unroll == 2: octx.set_bsk_unrolled(eng.export_bsk_unrolled())
    lid = eng.lut_register(np.random.default_rng(9).integers(-8, 8, 16), 4, DL)
    tv = eng.lut_get(lid)[None, :]
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    for B in batches:
        msgs = np.random.default_rng(B).integers(-8, 8, B)
        ct = eng.encrypt(msgs, DL)
        small = eng.keyswitch_host(ct)
        d_small = torch.from_numpy(small.view(np.int64)).to(dev)
        d_ids = torch.full((B,), lid, dtype=torch.int32, device=dev)
        d_out = torch.empty((B, eng.P.N + 1), dtype=torch.int64, device=dev)
        d_in = torch.from_numpy(ct.view(np.int64)).to(dev)
        d_ks = torch.empty((B, eng.P.n + 1), dtype=torch.int64, device=dev)
        nchk = min(B, 4)
        want = octx.blind_rotate(small[:nchk], tv, np.zeros(nchk, np.uint32), unrolled=unroll == 2)
        for v in variants:
            eng.set_kernel_variant(v)
            eng.blind_rotate(d_small, d_ids, B, d_out, s); torch.cuda.synchronize()
            ok = np.array_equal(d_out[:nchk].cpu().numpy().view(np.uint64), want)
            reps = 3 if B >= 1024 else 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): eng.blind_rotate(d_small, d_ids, B, d_out, s)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            e0.record()
            for _ in range(reps): eng.keyswitch(d_in, B, d_ks, s)
            e1.record(); torch.cuda.synchronize()
            ks = e0.elapsed_time(e1) / reps
            ks_ok = np.array_equal(d_ks[:nchk].cpu().numpy().view(np.uint64), octx.keyswitch(ct[:nchk]))
            eng.set_keyswitch_variant(1)
            eng.keyswitch(d_in, B, d_ks, s); torch.cuda.synchronize()
            e0.record()
            for _ in range(reps): eng.keyswitch(d_in, B, d_ks, s)
            e1.record(); torch.cuda.synchronize()
            ks_scalar = e0.elapsed_time(e1) / reps
            ks_ok = ks_ok and np.array_equal(d_ks.cpu().numpy().view(np.uint64), small)
            eng.set_keyswitch_variant(0)
            print(json.dumps({"B": B, "variant": v, "br_ms": round(ms, 3), "ks_ms": round(ks, 3), "ks_scalar_ms": round(ks_scalar, 3), "ks_exact": bool(ks_ok), "pbs_per_s": round(B / ((ms + ks) * 1e-3), 1), "bit_exact": bool(ok), "q_bits": eng.q_bits, **kw}), flush=True)

main()
