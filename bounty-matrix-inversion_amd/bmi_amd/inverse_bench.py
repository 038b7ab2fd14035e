"""Wall-clock of the encrypted n x n inverse (BASELINE.json metric, second half), used by `bench.py --inverse`.
Matrices: np.random.seed(1234 + n); M = randn(n, n) * 100.  Reports compile (trace + upload),
encrypt, evaluate (all PBS on the GPU), decrypt, and checks the decrypted result against the circuit's own
plaintext simulation (identical integers) and against numpy's inverse."""
from __future__ import annotations

import time

import numpy as np

from .main import EncryptedMatrixInversion

CONFIGS = {2: (20, 8), 3: (30, 12), 4: (40, 16), 8: (48, 16)}


def run(engine, sizes=(2, 3), shard_threshold=None):
    """With torch.distributed initialised on several ranks every rank must call this (the wide levels are split
    across the ranks' GPUs, executor.py); every rank returns the same report."""
    out = {}
    for n in sizes:
        ln, ints = CONFIGS[n]
        np.random.seed(1234 + n)
        M = np.random.randn(n, n) * 100
        t0 = time.time()
        emi = EncryptedMatrixInversion(n, None, 2, ln, ints, False, False, engine=engine, shard_threshold=shard_threshold)
        t_program = time.time() - t0           # trace + prune + schedule (cold) or a load of the cached program
        t0 = time.time()
        ex = emi._executor()                   # index arrays -> device, LUT uploads, store allocation
        t_exec = time.time() - t0
        t_compile = t_program + t_exec
        q, s = emi.quantize(M)
        t0 = time.time()
        enc = emi.encrypt(q, s)
        t_enc = time.time() - t0
        if n <= 4:
            emi.evaluate(enc)  # warm-up (first launches, LUT uploads); skipped for the long 8x8 run
        t0 = time.time()
        res = emi.evaluate(enc)
        t_eval = time.time() - t0
        t0 = time.time()
        dec = emi.decrypt(res)
        t_dec = time.time() - t0
        sim = emi.simulate(q, s)
        summ = emi.circuit.summary()
        out[f"{n}x{n}"] = {
            "len": ln, "ints": ints, "evaluate_s": round(t_eval, 3), "compile_s": round(t_compile, 3),
            "compile_cached": bool(emi.compile_info["cached"]), "program_s": round(t_program, 3),
            "executor_build_s": round(t_exec, 3),
            "end_to_end_s": round(t_compile + t_enc + t_eval + t_dec, 3),
            "store_gb": round(ex.store_bytes() / 1e9, 3), "store_rows": int(ex.n_rows),
            "traced_pbs": summ.get("traced_pbs"), "pruned_pbs": summ.get("pruned_pbs"),
            "encrypt_s": round(t_enc, 3), "decrypt_s": round(t_dec, 3), "pbs": summ["pbs"], "depth": summ["depth"],
            "ms_per_level": round(t_eval / max(summ["depth"], 1) * 1e3, 3),
            "ranks": emi._executor().world, "sharded_levels": emi._executor().sharded_levels,
            "matches_plaintext_circuit": bool(np.array_equal(dec, sim)),
            "max_abs_err_vs_numpy": float(np.max(np.abs(emi.dequantize(dec) - np.linalg.inv(M)))),
        }
    return out


def run_batched(engine, n=3, batch=16, warm=True):
    """The serving form: `batch` independent encrypted n x n matrices through ONE walk of the circuit's levels
    (EncryptedMatrixInversion.evaluate_many: every level `batch` times wider, look-ups on the throughput kernel).  Reports the
    wall-clock of the batched evaluation, per matrix, and the look-up rate it sustains; every result is checked against the
    plaintext evaluation of the same program."""
    ln, ints = CONFIGS[n]
    rng = np.random.default_rng(4321 + n)
    Ms = [rng.standard_normal((n, n)) * 100 for _ in range(batch)]
    emi = EncryptedMatrixInversion(n, None, 2, ln, ints, False, False, engine=engine)
    qs = [emi.quantize(M) for M in Ms]
    encs = [emi.encrypt(q, s) for q, s in qs]
    t0 = time.time()
    ex = emi._executor(batch)
    t_exec = time.time() - t0
    if warm:
        emi.evaluate_many(encs)
    t0 = time.time()
    res = emi.evaluate_many(encs)
    t_eval = time.time() - t0
    ok = all(np.array_equal(emi.decrypt(r), emi.simulate(q, s)) for r, (q, s) in zip(res, qs))
    summ = emi.circuit.summary()
    return {"n": n, "len": ln, "ints": ints, "batch": batch, "evaluate_s": round(t_eval, 3), "per_matrix_s": round(t_eval / batch, 4),
            "matrices_per_s": round(batch / t_eval, 3), "pbs": summ["pbs"] * batch, "pbs_per_s": round(summ["pbs"] * batch / t_eval, 1),
            "depth": summ["depth"], "widest_level": int(max(w for w, *_ in ex.levels)), "executor_build_s": round(t_exec, 3),
            "store_gb": round(ex.store_bytes() / 1e9, 3), "matches_plaintext_circuit": bool(ok)}


# The reference's own published benchmark (README.md:129-142): "low" precision = array length 23, 9 integer digits, base 2,
# no true division; 2x2 and 3x3, tensorize yes / no; 64-core CPU instance, concrete-python 2.1.0.
README_LOW = {
    (2, True): {"compile_s": 41.0, "run_s": 85.0, "total_s": 126.0},
    (2, False): {"compile_s": 42.0, "run_s": 85.0, "total_s": 127.0},
    (3, True): {"compile_s": 4255.0, "run_s": 1768.0, "total_s": 6024.0},
    (3, False): {"compile_s": 5487.0, "run_s": 1349.0, "total_s": 6837.0},
}


def run_readme_low(engine, cold=True):
    """The reference's README benchmark configurations on this engine: compile (cold = traced from scratch, as the
    reference does on every start; cached = a load of the program file), encrypt, evaluate, decrypt, beside the published
    figures.  Results are checked against the plaintext evaluation of the same program."""
    out = {}
    for (n, tensorize), ref in README_LOW.items():
        ln, ints = 23, 9
        np.random.seed(1234 + n)
        M = np.random.randn(n, n) * 100
        t_cold = None
        if cold:
            t0 = time.time()
            EncryptedMatrixInversion(n, None, 2, ln, ints, False, tensorize, engine=engine, cache=False)
            t_cold = time.time() - t0
        t0 = time.time()
        emi = EncryptedMatrixInversion(n, None, 2, ln, ints, False, tensorize, engine=engine)
        emi._executor()
        t_compile = time.time() - t0
        q, s = emi.quantize(M)
        emi.evaluate(emi.encrypt(q, s))            # warm-up (LUT uploads, first launches)
        t0 = time.time()
        enc = emi.encrypt(q, s)
        res = emi.evaluate(enc)
        dec = emi.decrypt(res)
        t_run = time.time() - t0
        ok = bool(np.array_equal(dec, emi.simulate(q, s)))
        total_cold = (t_cold if t_cold is not None else t_compile) + t_run
        out[f"{n}x{n}_len23_ints9_tensorize_{'yes' if tensorize else 'no'}"] = {
            "compile_cold_s": None if t_cold is None else round(t_cold, 3), "compile_cached_s": round(t_compile, 3),
            "encrypt_run_decrypt_s": round(t_run, 3), "total_cold_s": round(total_cold, 3),
            "total_cached_s": round(t_compile + t_run, 3), "pbs": emi.program.n_nodes, "depth": emi.program.depth,
            "matches_plaintext_circuit": ok,
            "reference_readme": ref, "speedup_run": round(ref["run_s"] / t_run, 1),
            "speedup_total_cold": round(ref["total_s"] / total_cold, 1)}
    return out
