#!/usr/bin/env python3
"""Traces the reference's UNMODIFIED functions (base_p_arrays.py, qfloat.py under /root/reference) through the
Tracer-compatible shim tools/encshim into circuits of this repo's IR, and writes them - as data: look-up tables, linear
combinations, input ranges, sample inputs and the reference's own plaintext outputs - to tests/golden/ref_traced.json.
The CPU suite simulates the circuits against the recorded outputs; the GPU suite runs them on ciphertexts.
Runs only in the build container (needs /root/reference)."""
import json, os, random, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools", "encshim"))
sys.path.insert(0, "/root/reference/matrix_inversion")
sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np
from concrete import fhe
import base_p_arrays as ref
import qfloat as rq

rng = random.Random(2025)
bits = lambda k: [rng.randint(0, 1) for _ in range(k)]   # noqa: E731
cases = []


def add_case(name, cite, fn, plain, gen, ranges, n_vec=16, n_set=1500):
    inputset = [gen() for _ in range(n_set)]
    circ, _ = fhe.trace(fn, ranges, inputset, msg_bits=4)
    vectors = []
    for _ in range(n_vec):
        args = gen()
        want = plain(*[np.array(a) for a in args])
        want = [int(v) for w in (want if isinstance(want, tuple) else (want,)) for v in np.atleast_1d(w)]
        flat = [v for a in args for v in a]
        assert circ.simulate(flat) == want, name
        vectors.append({"inputs": flat, "expected": want})
    width = max(p for p, _ in circ.luts)
    cases.append({"name": name, "reference": cite, "pbs": len(circ.nodes), "depth": len(circ.levels()),
                  "widest_lookup_bits": width, "circuit": circ.to_dict(), "vectors": vectors})
    print(f"{name}: pbs {len(circ.nodes)} depth {len(circ.levels())} widest look-up {width} bits")


def two(k):
    def gen():
        a, b = bits(k), bits(k)
        r = rng.random()
        if r < 0.15:
            b = list(a)
        elif r < 0.25:
            b = list(a)
            b[rng.randrange(k)] ^= 1
        elif r < 0.30:
            a, b = [0] * k, [1] * k
        elif r < 0.35:
            a, b = [1] * k, [0] * k
        return (a, b)
    return gen
add_case("base_p_addition", "base_p_arrays.py:84-105", lambda a, b: ref.base_p_addition(a, b, 2),
         lambda a, b: ref.base_p_addition(a, b, 2), two(10), [[(0, 1)] * 10] * 2)
add_case("base_p_subtraction_overflow", "base_p_arrays.py:108-139", lambda a, b: ref.base_p_subtraction(a, b, 2, True),
         lambda a, b: ref.base_p_subtraction(a, b, 2, True), two(10), [[(0, 1)] * 10] * 2)
add_case("is_greater_or_equal", "base_p_arrays.py:245-260", ref.is_greater_or_equal, ref.is_greater_or_equal, two(10),
         [[(0, 1)] * 10] * 2)
add_case("is_equal", "base_p_arrays.py:276-280", ref.is_equal, ref.is_equal, two(10), [[(0, 1)] * 10] * 2)
add_case("base_p_division", "base_p_arrays.py:173-203", lambda a, b: ref.base_p_division(a, b, 2),
         lambda a, b: ref.base_p_division(a, b, 2), lambda: (bits(8), bits(4)), [[(0, 1)] * 8, [(0, 1)] * 4])

L, I = 8, 4
QF = rq.QFloat


def qsample():
    a, b = bits(L), bits(L)
    k = rng.random()
    if k < 0.15:
        b = list(a)                                  # equal magnitudes (rare at random, a branch of the comparisons)
    elif k < 0.25:
        b = list(a)
        b[rng.randrange(L)] ^= 1                     # one digit apart
    elif k < 0.30:
        a = [0] * L
    elif k < 0.35:
        b = [0] * L
    elif k < 0.40:
        a, b = [1] * L, [1] * L
    return (a, [rng.choice((-1, 1))], b, [rng.choice((-1, 1))])


def qrun(op):
    def f(a, sa, b, sb):
        r = op(QF(a, I, 2, True, sa[0]), QF(b, I, 2, True, sb[0]))
        return (r._array, r._sign) if isinstance(r, QF) else r
    return f


qr = [[(0, 1)] * L, [(-1, 1)], [(0, 1)] * L, [(-1, 1)]]
for nm, cite, op in (("QFloat.__add__", "qfloat.py:766-850", lambda x, y: x + y),
                     ("QFloat.__sub__", "qfloat.py:938-953", lambda x, y: x - y),
                     ("QFloat.__mul__", "qfloat.py:852-936", lambda x, y: x * y),
                     ("QFloat.__gt__", "qfloat.py:681-764", lambda x, y: x > y)):
    add_case(nm, cite, qrun(op), qrun(op), qsample, qr)

json.dump({"generator": "tools/gen_ref_traced.py (reference functions run unmodified through tools/encshim)",
           "cases": cases}, open(os.path.join(REPO, "tests", "golden", "ref_traced.json"), "w"))
print("wrote tests/golden/ref_traced.json", os.path.getsize(os.path.join(REPO, "tests", "golden", "ref_traced.json")), "bytes")
