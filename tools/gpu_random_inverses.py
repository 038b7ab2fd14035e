#!/usr/bin/env python3
"""Many random matrices through the encrypted inverse on the GPU, each compared with the plaintext evaluation of the same
compiled program (identical integers expected) and with numpy's inverse: an empirical look at the look-up failure rate
(every 4-bit look-up sits at >= 5.6 sigma).  usage: gpu_random_inverses.py [n] [count] [q_bits] [unroll] [base]
(unroll = 2: the unrolled bootstrap key - at key noise 2^-41 on the 49-bit field; on the torus the 42-bit key through the FFT -, what
EncryptedMatrixInversion(unroll=True) runs; q_bits may also be a preset name, e.g. secure128_torus).  Prints the error budget of the
circuit under the engine's parameters beside the observed mismatches."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np
from bmi_amd import tfhe
from bmi_amd.main import EncryptedMatrixInversion

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    qb = sys.argv[3] if len(sys.argv) > 3 else "65"
    preset = None if qb.isdigit() else qb
    qb = int(qb) if preset is None else None
    ln, ints = {2: (20, 8), 3: (30, 12), 4: (40, 16)}[n]
    unroll = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    base = int(sys.argv[5]) if len(sys.argv) > 5 else 2          # 3: the base-3 configuration of tests/golden/inverse.json (5-bit look-ups)
    if base != 2:
        ln, ints = 14, 6
    eng = tfhe.Engine(tfhe.preset_params(preset) if preset else
                      tfhe.default_params(q_bits=qb, **({"glwe_noise": 2.0 ** -41} if (unroll == 2 and qb == 49) else {})))
    eng.set_bsk_unroll(unroll)
    eng.keygen()          # CSPRNG keys
    emi = EncryptedMatrixInversion(n, None, base, ln, ints, False, False, engine=eng, unroll=(unroll == 2))
    rng = np.random.default_rng(4242 + n)
    wrong = skipped = 0; pbs = emi.program.n_nodes; worst = 0.0; t0 = time.time()
    for i in range(count):
        M = rng.normal(0, 100, (n, n)) if base == 2 else rng.uniform(-100, 100, (n, n))
        q, s = emi.quantize(M)
        if q[:, 0].max() > 2 * base - 1:            # entry beyond the traced leading-digit range (|x| >= 2^(ints+2)): not an input of this circuit
            skipped += 1; continue
        want = emi.simulate(q, s)
        got = emi.decrypt(emi.evaluate(emi.encrypt(q, s)))
        if not np.array_equal(got, want):
            wrong += 1
        if (i + 1) % 5 == 0:     # a silent run is taken to be hung by gpurun
            print(f"# {i + 1} / {count} matrices, {wrong} mismatching, {time.time() - t0:.0f} s", flush=True)
        else:
            err = np.max(np.abs(emi.dequantize(got) - np.linalg.inv(M)))
            if np.isfinite(err): worst = max(worst, float(err))
    done = count - skipped
    budget = emi.error_budget or emi.program.failure_probability(eng)
    print(json.dumps({"n": n, "q_bits": eng.q_bits, "preset": preset, "N": eng.P.N, "lwe_n": eng.P.n, "bsk_precision": eng.bsk_precision,
                      "p_fail_per_inverse_budget": budget["p_fail"], "expected_mismatches": budget["p_fail"] * done, "matrices": done, "mismatching_the_plaintext_circuit": wrong, "lookups_total": done * pbs,
                      "lookups_per_inverse": pbs, "worst_abs_err_vs_numpy_among_matching": worst, "seconds": round(time.time() - t0, 1),
                      "keys": "CSPRNG", "unroll": unroll, "base": base, "len": ln, "ints": ints, "depth": emi.program.depth}))
    eng.close()
    return 1 if wrong else 0

sys.exit(main())
