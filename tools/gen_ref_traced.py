#!/usr/bin/env python3
"""Traces the reference's UNMODIFIED functions (base_p_arrays.py, qfloat.py under /root/reference) through the
Concrete-compatible front end bmi_amd/compat into circuits of this repo's IR, and writes them - as data: look-up tables, linear
combinations, input ranges, sample inputs and the reference's own plaintext outputs - to tests/golden/ref_traced.json.
The CPU suite simulates the circuits against the recorded outputs; the GPU suite runs them on ciphertexts.
Runs only in the build container (needs /root/reference)."""
import json, os, random, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd", "bmi_amd", "compat"))
sys.path.insert(0, "/root/reference/matrix_inversion")
sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
import numpy as np
from concrete import fhe
import base_p_arrays as ref
import qfloat as rq

rng = random.Random(2025)
bits = lambda k: [rng.randint(0, 1) for _ in range(k)]   # noqa: E731
cases = []


def add_case(name, cite, fn, plain, gen, ranges, n_vec=16, n_set=1500, msg_bits=4, into=None, fuse=False):
    from bmi_amd.circuit import RangeError
    inputset = [gen() for _ in range(n_set)]
    circ, _ = fhe.trace(fn, ranges, inputset, msg_bits=msg_bits, fuse=fuse)
    vectors, outside = [], 0
    while len(vectors) < n_vec:
        args = gen()
        want = plain(*[np.array(a) for a in args])
        want = [int(v) for w in (want if isinstance(want, tuple) else (want,)) for v in np.asarray(w).reshape(-1)]
        flat = [v for a in args for v in a]
        try:
            got = circ.simulate(flat)
        except RangeError:      # an input that leaves the ranges measured on the inputset: undefined, as with Concrete
            outside += 1
            assert outside < 4 * n_vec, name
            continue
        assert got == want, name
        vectors.append({"inputs": flat, "expected": want})
    width = max(p for p, _ in circ.luts)
    (cases if into is None else into).append(
        {"name": name, "reference": cite, "msg_bits": msg_bits, "lazy_lookup_fusion": bool(fuse), "pbs": len(circ.nodes), "depth": len(circ.levels()),
         "widest_lookup_bits": width, "inputset": n_set, "outside_inputset_ranges": outside,
         "circuit": circ.to_dict(), "vectors": vectors})
    print(f"{name}: pbs {len(circ.nodes)} depth {len(circ.levels())} widest look-up {width} bits, "
          f"{outside} fresh inputs outside the measured ranges")


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

# ---- the whole inverse, as the reference composes it (qfloat_matrix_inversion.py:672-720), at BASELINE.json's sizes
import gzip
import qfloat_matrix_inversion as rmi

nrng = np.random.RandomState(99)


def inverse_case(n, ln, ints, n_vec, into, fuse=False):
    def gen():
        M = nrng.randn(n, n) * 100                                   # the sampler of SURVEY.md section 8d
        a, sg = rmi.float_matrix_to_qfloat_arrays(M, ln, ints, 2)
        return ([int(v) for v in np.asarray(a).reshape(-1)], [int(v) for v in np.asarray(sg).reshape(-1)])
    run = lambda a, sg: rmi.qfloat_matrix_inverse(a.reshape(n * n, ln), sg, n, ln, ints, 2, False, False)  # noqa: E731
    add_case(f"qfloat_matrix_inverse_{n}x{n}" + ("_fused" if fuse else ""), "qfloat_matrix_inversion.py:672-720", run, run, gen,
             [[(0, 1)] * (n * n * ln), [(-1, 1)] * (n * n)], n_vec=n_vec, n_set=2000, msg_bits=5, into=into, fuse=fuse)
    into[-1].update(n=n, len=ln, ints=ints)


inverses = []
inverse_case(2, 20, 8, 4, inverses)
inverse_case(3, 30, 12, 2, inverses)
inverse_case(2, 20, 8, 4, inverses, fuse=True)   # the same trace with the shim's lazy look-up fusion switched on
with gzip.GzipFile(os.path.join(REPO, "tests", "golden", "ref_traced_inverse.json.gz"), "wb", mtime=0) as f:
    f.write(json.dumps({"generator": "tools/gen_ref_traced.py", "cases": inverses}).encode())
print("wrote tests/golden/ref_traced_inverse.json.gz",
      os.path.getsize(os.path.join(REPO, "tests", "golden", "ref_traced_inverse.json.gz")), "bytes")

json.dump({"generator": "tools/gen_ref_traced.py (reference functions run unmodified through bmi_amd/compat)",
           "cases": cases}, open(os.path.join(REPO, "tests", "golden", "ref_traced.json"), "w"))
print("wrote tests/golden/ref_traced.json", os.path.getsize(os.path.join(REPO, "tests", "golden", "ref_traced.json")), "bytes")
