#!/usr/bin/env python3
"""Writes tests/golden/pbs_kat.json: known-answer digests of the deterministic PBS path (keys from a seed, fixed
messages and table -> SHA-256 of the keys, the input ciphertexts, the keyswitched and the bootstrapped outputs), computed
with the CPU oracle for every supported (field, N).  The CPU suite checks the oracle against it, the GPU suite the
library: a change of the scheme, of the RNG streams or of any kernel shows up in both."""
import hashlib, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from oracle import tfhe_oracle as to

def h(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()

SEED = 0xC0FFEE
MSGS = [-8, -3, 0, 5, 7]
TABLE = [3, -8, 7, 0, -1, 5, -6, 2, 1, -4, 6, -7, 4, -2, -5, -3]
out = {"seed": SEED, "msgs": MSGS, "table": TABLE, "cases": []}
# (q_bits, log_N, extra parameters): the torus appears with its default set (Bg = 2^10, bootstrap key stored at 48 bits of
# precision - the key the digests cover is the ROUNDED one, ora_round_key - plain and unrolled blind rotation) and with round
# 2's set (Bg = 2^15, exact key)
for q_bits, log_N, kw in ((49, 10, {}), (64, 10, {}), (49, 11, {}), (to.TORUS64, 10, {}), (to.TORUS64, 10, {"bs_base_log": 15})):
    to.set_field(q_bits)
    P = to.default_params(q_bits=q_bits, log_N=log_N, **kw)
    K = to.keygen(P, SEED)
    prec = to.default_bsk_precision(P)
    bsk = to.round_key(K.bsk, prec)
    dl = to.log_q(q_bits) - 1 - 4
    ctx = to.Ctx(P, bsk, K.ksk)
    ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, SEED, 0, to.encode(MSGS, dl))
    tv = to.make_test_vector(log_N, 4, np.array(TABLE), dl)
    small = ctx.keyswitch(ct)
    pbs = ctx.pbs(ct, tv[None, :], np.zeros(len(MSGS), np.uint32))
    assert list(to.decode(to.lwe_phase(K.sk_big, pbs), dl)) == [TABLE[m + 8] for m in MSGS]
    case = {"q_bits": q_bits, "log_N": log_N, "params": kw, "bsk_precision": prec, "sk_big": h(K.sk_big), "bsk": h(bsk),
            "ksk": h(K.ksk), "test_vector": h(tv), "ciphertexts": h(ct), "keyswitched": h(small), "bootstrapped": h(pbs)}
    if (q_bits == 49 and log_N == 10) or (q_bits == to.TORUS64 and prec == 48):     # the unrolled blind rotation
        bsk3 = to.round_key(to.keygen_bsk_unrolled(P, SEED, K.sk_small, K.sk_big), prec)
        ctx.set_bsk_unrolled(bsk3)
        pbs_u = ctx.pbs(ct, tv[None, :], np.zeros(len(MSGS), np.uint32), unrolled=True)
        assert list(to.decode(to.lwe_phase(K.sk_big, pbs_u), dl)) == [TABLE[m + 8] for m in MSGS]
        case["bsk_unrolled"] = h(bsk3)
        case["bootstrapped_unrolled"] = h(pbs_u)
    out["cases"].append(case)
    ctx.close()
json.dump(out, open(os.path.join(REPO, "tests", "golden", "pbs_kat.json"), "w"), indent=1)
print(json.dumps(out)[:300])
