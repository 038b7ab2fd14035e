"""The QFloat oracle (oracle/qfloat_oracle.py) against the golden vectors the
reference itself produced (tools/gen_golden.py).  Integer outputs: bit-exact."""
import json
import os

import numpy as np
import pytest

from oracle import qfloat_oracle as qo

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def same(q, g):
    assert isinstance(q, qo.Q)
    assert [int(x) for x in q.d] == g["array"], (list(q.d), g["array"])
    assert int(q.sign) == g["sign"] and q.ints == g["ints"] and q.base == g["base"]


def mk(g):
    return qo.Q(np.array(g["array"]), g["ints"], g["base"], True, g["sign"])


def test_kats():
    k = load("kats.json")
    q = qo.Q.from_float(103.785, 24, 8, 2)
    assert str(q) == k["from_float_103.785_24_8_2"]["str"] == "01100111.1100100011110101"
    assert q.to_float() == k["from_float_103.785_24_8_2"]["to_float"]
    # reference tests/test_qfloat.py:40-55
    assert str(qo.Q.from_float(13.75, 10, 5, 2)) == "01101.11000"
    assert str(qo.Q.from_float(-13.75, 10, 5, 2)) == "-01101.11000"
    assert str(qo.Q.from_float(0, 10, 5, 2)) == "00000.00000"
    z = qo.Q.from_float(1, 10, 5, 2)
    z.sign = 0
    assert str(z) == "00000.00000"
    for key, g in k.items():
        if key.startswith("from_float_") and key != "from_float_103.785_24_8_2":
            _, _, f, ln, it, b = key.split("_")
            q = qo.Q.from_float(float(f), int(ln), int(it), int(b))
            same(q, g)
            assert q.to_float() == g["to_float"]
            if g["str"] is not None:
                assert q.to_str(False) == g["str"]
    same(qo.Q(np.array([0, 3, -5, 7, -2, 1]), 3, 2, False), k["base_tidy_0_3_-5_7_-2_1"])
    same(qo.Q(np.array([5, 0, 0, 0]), 2, 2, False), k["base_tidy_5_0_0_0"])
    same(qo.Q.from_float(2.5, 10, 5, 2) + qo.Q.from_float(-2.5, 10, 5, 2), k["add_2.5_-2.5"])
    same(qo.Q.from_float(1.75, 8, 4, 2) * qo.Q.from_float(0.0625, 8, 4, 2), k["mul_1.75_0.0625_8_4"])
    d = qo.Q.from_float(5, 10, 5, 2) / qo.Q.from_float(3, 10, 5, 2)
    same(d, k["div_5_3_10_5"])
    assert str(d) == "00001.10101"
    same(qo.Q.from_float(3, 10, 5, 2).invert(1, 10, 0), k["invert_3_10_0"])
    same(qo.Q.from_float(5, 10, 5, 2) / qo.SignedBinary(0), k["div_5_by_sb0"])


def test_base_p_arrays():
    for c in load("base_p_arrays.json"):
        if c["op"] == "sub":
            diff, lt = qo.sub_digits(np.array(c["a"]), np.array(c["b"]), c["p"], True)
            assert list(diff) == c["diff"] and int(lt) == c["lt"]
            assert list(qo.sub_digits(np.array(c["a"]), np.array(c["b"]), c["p"])) == c["diff_noov"]
        elif c["op"] == "div":
            assert list(qo.div_digits(np.array(c["a"]), np.array(c["b"]), c["p"])) == c["q"]
        elif c["op"] == "cmp":
            assert qo.ge_digits(c["a"], c["b"]) == c["ge"] and qo.eq_digits(c["a"], c["b"]) == c["eq"]
        elif c["op"] == "codec":
            arr = qo.int_to_digits(c["v"], c["n"], c["p"])
            assert list(arr) == c["arr"] and qo.digits_to_int(arr, c["p"]) == c["back"]
            farr = qo.frac_to_digits(c["f"], c["n"], c["p"])
            assert [int(x) for x in farr] == c["farr"]
            assert qo.digits_to_frac(farr, c["p"]) == c["fback"]


def test_qfloat_ops():
    g = load("qfloat_ops.json")
    for c in g["pairs"]:
        q1 = qo.Q.from_float(c["f1"], c["len"], c["ints"], c["base"])
        q2 = qo.Q.from_float(c["f2"], c["len"], c["ints"], c["base"])
        same(q1, c["q1"])
        same(q2, c["q2"])
        same(q1 + q2, c["add"])
        same(q1 - q2, c["sub"])
        same(q1 + 2, c["add_int2"])
        same(2 - q1, c["rsub_int2"])
        same(qo.SignedBinary(1) + q1, c["add_sb1"])
        same(qo.SignedBinary(1) - q1, c["rsub_sb1"])
        same(q1 * q2, c["mul"])
        same(q1 * 2, c["mul_int"])
        same(-3 * q1, c["mul_intm3"])
        same(q1 * qo.SignedBinary(-1), c["mul_sbm1"])
        same(qo.Q.from_mul(q1, q2), c["from_mul"])
        f = c["from_mul_fmt"]
        same(qo.Q.from_mul(q1, q2, f["newlen"], f["newints"]), f)
        same(abs(q1), c["abs"])
        same(-q1, c["neg"])
        assert int(q1 > q2) == c["gt"] and int(q1 >= q2) == c["ge"]
        assert int(q1 < q2) == c["lt"] and int(q1 <= q2) == c["le"]
        assert int(q1 == q2) == c["eq"] and int(q1 == q1.copy()) == c["eq_self"]
        if "div" in c:
            same(q1 / q2, c["div"])
            f = c["invert_fmt"]
            same(q2.invert(1, f["newlen"], f["newints"]), f)
            same(qo.SignedBinary(-1) / q2, c["invert_m1"])
        z = q1.copy()
        z.sign = 0
        same(z + q2, c["zero_sign_add"])
    for c in g["tidy"]:
        q = qo.Q(np.array(c["in"]), c["ints"], c["base"], False)
        same(q, c["base_tidy"])
        v = q.to_float()
        q.tidy()
        same(q, c["tidy"])
        assert abs(v - q.to_float()) <= 1e-9  # two-sided (the reference's check is one-sided)
    for c in g["from_mul_mixed"]:
        same(qo.Q.from_mul(mk(c["q1"]), mk(c["q2"]), 18, 1), c["out"])


@pytest.mark.parametrize("case", load("inverse.json"), ids=lambda c: c["tag"])
def test_inverse(case):
    c = case
    M = np.array(c["M"]).reshape(c["n"], c["n"])
    arrs, signs = qo.float_matrix_to_qfloat_arrays(M, c["len"], c["ints"], c["base"])
    assert arrs.tolist() == c["in_arrays"] and signs.tolist() == c["in_signs"]
    qo.Q.reset_stats()
    out = qo.qfloat_matrix_inverse(arrs, signs, c["n"], c["len"], c["ints"], c["base"],
                                   c["true_division"], c["tensorize"])
    assert out.tolist() == c["out"]
    assert qo.Q.stats() == c["stats"]
    assert qo.arrays_to_float_matrix(out, c["ints"], c["base"]).flatten().tolist() == c["float"]
    if "L" in c:
        qM = qo.arrays_to_matrix(arrs, signs, c["ints"], c["base"])
        P, Lm, U = qo.lu_decomposition(qM, c["len"], c["ints"], c["true_division"], c["tensorize"])
        for name, m in (("P", P), ("L", Lm), ("U", U)):
            assert qo.matrix_to_arrays(m, c["len"], c["ints"], c["base"]).tolist() == c[name]
        qM = qo.arrays_to_matrix(arrs, signs, c["ints"], c["base"])
        assert qo.pivot_matrix(qM).tolist() == c["pivot"]


def test_inverse_close_to_numpy():
    # sanity vs floating point: high-ish precision format, well-conditioned matrix
    M = np.array([[40.0, 7.0, -3.0], [2.0, -55.0, 9.0], [6.0, 1.0, 70.0]])
    arrs, signs = qo.float_matrix_to_qfloat_arrays(M, 40, 16, 2)
    out = qo.qfloat_matrix_inverse(arrs, signs, 3, 40, 16, 2, False)
    got = qo.arrays_to_float_matrix(out, 16, 2)
    assert np.max(np.abs(got - np.linalg.inv(M))) < 1e-4
