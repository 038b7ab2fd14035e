#!/usr/bin/env python3
"""Stress run for the synchronisation-heavy kernels (pair counters, resync barrier, split transforms): many batch
sizes around every tiling boundary, every kernel variant, repeated; each output must decrypt to LUT[m], a sample must be
bit-exact against the oracle, and repeated runs of the same input must be identical."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np
from bmi_amd import tfhe
from oracle import tfhe_oracle as to

def main():
    qb = int(sys.argv[1]) if len(sys.argv) > 1 else 49
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    log_N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    kw = {k[5:].lower(): int(v) for k, v in os.environ.items() if k.startswith("BMIP_")}   # e.g. BMIP_BS_LEVELS=2
    unroll = int(os.environ.get("BMI_UNROLL", "1"))     # 2: the unrolled blind rotation (one kernel for every variant / batch size)
    eng = tfhe.Engine(tfhe.default_params(q_bits=qb, log_N=log_N, **kw)); eng.set_bsk_unroll(unroll); eng.keygen(77)
    dl = eng.delta_log()
    _, _, bsk, ksk = eng.export_keys()
    to.set_field(qb)
    octx = to.Ctx(to.default_params(q_bits=qb, log_N=log_N, **kw), bsk, ksk)
    if unroll == 2:
        octx.set_bsk_unrolled(eng.export_bsk_unrolled())
    lb3 = (eng.P.bs_levels, eng.P.bs_base_log) == (3, 15)
    all_variants = (0, 1, 2, 3, 4) if qb != 49 else (0, 2, 3)     # variants 1 / 4 of the 49-bit field are A/B builds (make ab)
    rng = np.random.default_rng(2024)
    table = rng.integers(-8, 8, 16)
    lid = eng.lut_register(table, 4, dl)
    tv = eng.lut_get(lid)[None, :]
    sizes = [1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 515, 1023, 1024, 1025, 2047, 4097, 8191, 8192]
    if log_N != 10:
        sizes = [s_ for s_ in sizes if s_ <= 1025]
    t0 = time.time(); runs = 0; checked = 0
    while time.time() - t0 < budget:
        for B in sizes:
            msgs = rng.integers(-8, 8, B)
            ct = eng.encrypt(msgs, dl)
            ids = np.full(B, lid, np.uint32)
            ref = None
            for variant in ((0,) if log_N != 10 else all_variants if B <= 600 else tuple(v for v in all_variants if v in (0, 1, 3))):
                eng.set_kernel_variant(variant)
                out = eng.pbs_host(ct, ids)
                assert np.array_equal(eng.decrypt(out, dl), table[msgs + 8]), (B, variant)
                if ref is None:
                    ref = out
                else:
                    assert np.array_equal(out, ref), ("variants differ", B, variant)
                runs += 1
            eng.set_kernel_variant(0)
            pick = rng.choice(B, min(B, 3), replace=False)
            assert np.array_equal(ref[pick], octx.pbs(ct[pick], tv, np.zeros(pick.size, np.uint32), unrolled=unroll == 2)), ("oracle", B)
            checked += pick.size
            if time.time() - t0 > budget:
                break
        print(json.dumps({"elapsed_s": round(time.time() - t0, 1), "kernel_runs": runs, "oracle_checked": checked}), flush=True)
    print("stress ok", json.dumps({"q_bits": qb, "log_N": log_N, "unroll": unroll, **kw}))

main()
