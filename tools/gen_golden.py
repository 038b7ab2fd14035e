#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REFERENCE's plaintext (NumPy-int)
QFloat path in this build container.

Run (build container only; needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

The fixtures are data only: seeded inputs and the integer outputs the reference
produced for them.  No reference source text is stored.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("BMI_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, os.path.join(REF, "matrix_inversion"))

import numpy as np  # noqa: E402
import base_p_arrays as bpa  # noqa: E402  (reference)
import qfloat as rq  # noqa: E402  (reference)
import qfloat_matrix_inversion as rqmi  # noqa: E402  (reference)

QFloat, SignedBinary, Zero = rq.QFloat, rq.SignedBinary, rq.Zero
OUT = os.path.join(REPO, "tests", "golden")


def L(a):
    return [int(x) for x in np.asarray(a).reshape(-1)]


def qf_dump(q):
    return {"array": L(q.array), "ints": int(q.ints), "base": int(q.base), "sign": int(q.sign)}


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(OUT, name)), "bytes")


# --------------------------------------------------------------------------- KATs
def gen_kats():
    k = {}
    q = QFloat.from_float(103.785, 24, 8, 2)
    k["from_float_103.785_24_8_2"] = {"str": str(q), "to_float": q.to_float(), **qf_dump(q)}
    for f, ln, it, b in [(13.75, 10, 5, 2), (-13.75, 10, 5, 2), (0, 10, 5, 2), (300.5, 12, 8, 2),
                         (-0.3, 16, 4, 2), (7.99, 12, 3, 3), (-123.456, 14, 6, 10), (255.999, 20, 8, 2)]:
        q = QFloat.from_float(f, ln, it, b)
        k[f"from_float_{f}_{ln}_{it}_{b}"] = {"str": q.to_str(False) if b <= 10 else None,
                                               "to_float": q.to_float(), **qf_dump(q)}
    q = QFloat(np.array([0, 3, -5, 7, -2, 1]), 3, 2, False)
    k["base_tidy_0_3_-5_7_-2_1"] = qf_dump(q)
    q = QFloat(np.array([5, 0, 0, 0]), 2, 2, False)
    k["base_tidy_5_0_0_0"] = qf_dump(q)
    a = QFloat.from_float(2.5, 10, 5, 2) + QFloat.from_float(-2.5, 10, 5, 2)
    k["add_2.5_-2.5"] = qf_dump(a)
    m = QFloat.from_float(1.75, 8, 4, 2) * QFloat.from_float(0.0625, 8, 4, 2)
    k["mul_1.75_0.0625_8_4"] = {"to_float": m.to_float(), **qf_dump(m)}
    d = QFloat.from_float(5, 10, 5, 2) / QFloat.from_float(3, 10, 5, 2)
    k["div_5_3_10_5"] = {"str": str(d), **qf_dump(d)}
    i = QFloat.from_float(3, 10, 5, 2).invert(1, 10, 0)
    k["invert_3_10_0"] = qf_dump(i)
    z = QFloat.from_float(5, 10, 5, 2) / SignedBinary(0)
    k["div_5_by_sb0"] = qf_dump(z)
    return k


# ------------------------------------------------------------------ base_p_arrays
def gen_bpa(rng):
    cases = []
    for _ in range(60):
        p = int(rng.choice([2, 2, 2, 3, 5]))
        na, nb = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        a = rng.integers(0, p, na)
        b = rng.integers(0, p, nb)
        diff, lt = bpa.base_p_subtraction(a.copy(), b.copy(), p, True)
        diff2 = bpa.base_p_subtraction(a.copy(), b.copy(), p, False)
        c = {"op": "sub", "p": p, "a": L(a), "b": L(b), "diff": L(diff), "lt": int(lt), "diff_noov": L(diff2)}
        cases.append(c)
    for _ in range(40):
        p = int(rng.choice([2, 2, 3]))
        nd, nv = int(rng.integers(2, 14)), int(rng.integers(1, 8))
        dividend = rng.integers(0, p, nd)
        divisor = rng.integers(0, p, nv)
        if rng.random() < 0.15:
            divisor[:] = 0
        quo = bpa.base_p_division(dividend.copy(), divisor.copy(), p)
        cases.append({"op": "div", "p": p, "a": L(dividend), "b": L(divisor), "q": L(quo)})
    for _ in range(40):
        p = 2
        n = int(rng.integers(1, 12))
        a = rng.integers(0, p, n)
        b = a.copy() if rng.random() < 0.3 else rng.integers(0, p, n)
        cases.append({"op": "cmp", "a": L(a), "b": L(b), "ge": int(bpa.is_greater_or_equal(a, b)),
                      "eq": int(bpa.is_equal(a, b))})
    for _ in range(20):
        p = int(rng.choice([2, 3, 10]))
        v = int(rng.integers(-5000, 5000))
        n = int(rng.integers(1, 14))
        arr = bpa.int_to_base_p(v, n, p)
        f = float(rng.random()) * (1 if rng.random() < 0.5 else -1)
        farr = bpa.float_to_base_p(f, n, p)
        cases.append({"op": "codec", "p": p, "v": v, "n": n, "arr": L(arr), "back": int(bpa.base_p_to_int(arr, p)),
                      "f": f, "farr": L(farr), "fback": float(bpa.base_p_to_float(farr, p))})
    return cases


# -------------------------------------------------------------------- QFloat ops
def rand_float(rng, scale):
    return float(rng.integers(-scale * 100, scale * 100)) / 100.0


def gen_qfloat_ops(rng):
    cases = []
    for _ in range(40):
        base = int(rng.choice([2, 2, 2, 3]))
        ln = int(rng.integers(16, 33))
        ints = int(rng.integers(6, min(13, ln - 3)))
        f1, f2 = rand_float(rng, 20), rand_float(rng, 20)
        q1 = QFloat.from_float(f1, ln, ints, base)
        q2 = QFloat.from_float(f2, ln, ints, base)
        c = {"base": base, "len": ln, "ints": ints, "f1": f1, "f2": f2, "q1": qf_dump(q1), "q2": qf_dump(q2)}
        c["add"] = qf_dump(q1 + q2)
        c["sub"] = qf_dump(q1 - q2)
        c["add_int2"] = qf_dump(q1 + 2)
        c["rsub_int2"] = qf_dump(2 - q1)
        c["add_sb1"] = qf_dump(SignedBinary(1) + q1)
        c["rsub_sb1"] = qf_dump(SignedBinary(1) - q1)
        c["mul"] = qf_dump(q1 * q2)
        c["mul_int"] = qf_dump(q1 * int(rng.integers(-2, 3)) if False else q1 * 2)
        c["mul_intm3"] = qf_dump(-3 * q1)
        c["mul_sbm1"] = qf_dump(q1 * SignedBinary(-1))
        c["from_mul"] = qf_dump(QFloat.from_mul(q1, q2))
        nl, ni = int(rng.integers(12, 30)), int(rng.integers(2, 10))
        c["from_mul_fmt"] = {"newlen": nl, "newints": ni, **qf_dump(QFloat.from_mul(q1, q2, nl, ni))}
        c["abs"] = qf_dump(abs(q1))
        c["neg"] = qf_dump(-q1)
        c["gt"] = int(q1 > q2)
        c["ge"] = int(q1 >= q2)
        c["lt"] = int(q1 < q2)
        c["le"] = int(q1 <= q2)
        c["eq"] = int(q1 == q2)
        c["eq_self"] = int(q1 == q1.copy())
        if f2 != 0 and abs(f1) > 0:
            c["div"] = qf_dump(q1 / q2)
            nl2, ni2 = int(rng.integers(14, 34)), int(rng.integers(0, 8))
            c["invert_fmt"] = {"newlen": nl2, "newints": ni2, **qf_dump(q2.invert(1, nl2, ni2))}
            c["invert_m1"] = qf_dump(SignedBinary(-1) / q2)
        z = q1.copy()
        z._sign = 0
        c["zero_sign_add"] = qf_dump(z + q2)
        cases.append(c)
    # tidy on untidy mixed-sign arrays (tests/test_qfloat.py:191-213 pattern)
    tid = []
    for _ in range(40):
        base = int(rng.choice([2, 2, 3, 5]))
        size = int(rng.integers(12, 33))
        ints = int(rng.integers(size // 2 - 2, size // 2 + 2))
        arr = np.zeros(size, dtype=int)
        i1 = size // 4
        i2 = 3 * i1
        arr[i1:i2] = rng.integers(-4 * base, 4 * base, i2 - i1)
        q = QFloat(arr.copy(), ints, base, False)
        bt = qf_dump(q)
        q.tidy()
        tid.append({"base": base, "ints": ints, "in": L(arr), "base_tidy": bt, "tidy": qf_dump(q)})
    # from_mul with the mixed formats of tests/test_qfloat.py:137-143
    fm = []
    for _ in range(10):
        f1 = float(rng.integers(1, 100))
        f2 = float(rng.integers(1, 10000)) / 10000000
        q1 = QFloat.from_float(f1, 18, 18, 2)
        q2 = QFloat.from_float(f2, 25, 0, 2)
        fm.append({"f1": f1, "f2": f2, "q1": qf_dump(q1), "q2": qf_dump(q2),
                   "out": qf_dump(QFloat.from_mul(q1, q2, 18, 1))})
    # QFloat *= integer (qfloat.py:858-865; on a Tracer the integer is encrypted): no further rng draws above this line change
    im = []
    for f, ln, ints in [(13.75, 16, 8), (-7.3125, 18, 9), (0.40625, 14, 6), (101.0, 20, 10), (-0.0, 12, 6)]:
        for kk in (-7, -3, -1, 0, 1, 2, 5, 7):
            q = QFloat.from_float(f, ln, ints, 2)
            q *= kk
            im.append({"f": f, "len": ln, "ints": ints, "k": kk, "q": qf_dump(QFloat.from_float(f, ln, ints, 2)), "out": qf_dump(q)})
    return {"pairs": cases, "tidy": tid, "from_mul_mixed": fm, "imul_int": im}


# ------------------------------------------------------------------------ inverse
def gen_inverse():
    cases = []

    def one(tag, M, ln, ints, base=2, true_division=False, tensorize=False, with_plu=False):
        n = M.shape[0]
        QFloat.reset_stats()
        arrs, signs = rqmi.float_matrix_to_qfloat_arrays(M, ln, ints, base)
        out = rqmi.qfloat_matrix_inverse(arrs.copy(), signs.copy(), n, ln, ints, base, true_division, tensorize)
        c = {"tag": tag, "n": n, "len": ln, "ints": ints, "base": base, "true_division": true_division,
             "tensorize": tensorize, "M": [float(x) for x in M.flatten()],
             "in_arrays": [L(r) for r in arrs], "in_signs": L(signs),
             "out": [L(r) for r in np.asarray(out)],
             "stats": [QFloat.ADDITIONS, QFloat.MULTIPLICATION, QFloat.DIVISION],
             "float": [float(x) for x in rqmi.qfloat_and_signs_arrays_to_float_matrix(np.asarray(out), ints, base).flatten()]}
        if with_plu and n > 2:
            qM = rqmi.qfloat_arrays_to_qfloat_matrix(arrs.copy(), signs.copy(), ints, base)
            P, Lm, U = rqmi.qfloat_lu_decomposition(qM, ln, ints, true_division, tensorize)
            c["P"] = [L(r) for r in np.asarray(rqmi.qfloat_matrix_to_arrays_and_signs(P, ln, ints, base))]
            c["L"] = [L(r) for r in np.asarray(rqmi.qfloat_matrix_to_arrays_and_signs(Lm, ln, ints, base))]
            c["U"] = [L(r) for r in np.asarray(rqmi.qfloat_matrix_to_arrays_and_signs(U, ln, ints, base))]
            qM = rqmi.qfloat_arrays_to_qfloat_matrix(arrs.copy(), signs.copy(), ints, base)
            c["pivot"] = [L(r) for r in np.asarray(rqmi.qfloat_pivot_matrix(qM))]
        cases.append(c)
        print("  inverse", tag, "stats", c["stats"])

    one("survey_2x2", np.array([[10, -3.5], [4.25, 20]]), 20, 8)
    for n, ln, ints in [(2, 20, 8), (2, 23, 9), (3, 30, 12), (3, 23, 9), (4, 40, 16)]:
        np.random.seed(1234 + n)
        M = np.random.randn(n, n) * 100
        one(f"baseline_n{n}_len{ln}_ints{ints}", M, ln, ints, with_plu=True)
    # SURVEY §8 matrix (seed 1237) for config 3
    np.random.seed(1237)
    one("survey_cfg3_seed1237", np.random.randn(3, 3) * 100, 30, 12, with_plu=True)
    # an entry above 2^ints -> non-binary leading digit
    one("overflow_digit_2x2", np.array([[300.5, 12.25], [-7.5, 41.0]]), 20, 8)
    one("overflow_digit_3x3", np.array([[300.5, 12.25, 3.0], [-7.5, 41.0, -60.0], [5.0, -2.0, 77.0]]), 30, 12)
    # small formats, other modes, other base
    np.random.seed(7)
    M2 = np.random.uniform(0, 100, (2, 2))
    one("uniform_2x2_tensorize", M2, 20, 8, tensorize=True)
    one("uniform_2x2_small", M2, 16, 7)
    np.random.seed(8)
    M3 = np.random.uniform(0, 100, (3, 3))
    one("uniform_3x3_small", M3, 18, 8, with_plu=True)
    one("uniform_3x3_small_truediv", M3, 18, 8, true_division=True)
    one("uniform_3x3_small_tensorize", M3, 18, 8, tensorize=True)
    one("uniform_3x3_base3", M3, 14, 6, base=3)
    for s in range(3):
        np.random.seed(100 + s)
        one(f"rand3x3_seed{100 + s}", np.random.randn(3, 3) * 100, 24, 10)
    np.random.seed(1234 + 8)
    if os.environ.get("BMI_GOLDEN_8X8", "1") == "1":
        one("baseline_n8_len48_ints16", np.random.randn(8, 8) * 100, 48, 16)
    # round 2: a 3x3 whose entry really exceeds 2^ints (leading digit 2), and a second matrix per BASELINE config
    one("overflow_digit_3x3_ints8", np.array([[300.5, 12.25, 3.0], [-7.5, 41.0, -60.0], [5.0, -2.0, 77.0]]), 24, 8)
    for n, ln, ints in [(2, 20, 8), (3, 30, 12), (4, 40, 16)]:
        np.random.seed(2234 + n)
        one(f"baseline_b_n{n}_len{ln}_ints{ints}", np.random.randn(n, n) * 100, ln, ints)
    # the README's precision presets beyond "low" (README.md:107-114; the reference could not run them in FHE, :129)
    for name, ln, ints, td in (("medium", 31, 16, False), ("mediumplus", 31, 16, True), ("high", 40, 20, True)):
        for n in (2, 3):
            np.random.seed(3234 + n)
            one(f"readme_{name}_n{n}", np.random.randn(n, n) * 100, ln, ints, true_division=td)
    # the larger sizes of the reference's own driver (main.py:157-201 loops n over 2, 3, 5, 10), low precision
    for n in (5, 10):
        np.random.seed(4234 + n)
        one(f"main_n{n}_len23_ints9", np.random.randn(n, n) * 100, 23, 9)
    return cases


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)
    dump("kats.json", gen_kats())
    dump("base_p_arrays.json", gen_bpa(rng))
    dump("qfloat_ops.json", gen_qfloat_ops(rng))
    dump("inverse.json", gen_inverse())


if __name__ == "__main__":
    main()
