"""Minimal stand-in for `concrete.fhe` used ONLY by tools/gen_golden.py, in the
build container, to run the reference's *plaintext NumPy* QFloat path
(SURVEY.md §8c / App. B).  Outside a traced circuit Concrete's `fhe.zeros` /
`fhe.ones` are plain NumPy constructors, which is all this provides.  Never
shipped to, or imported on, the GPU box; not part of the product or oracle."""
import numpy as np
from . import tracing  # noqa: F401


def zeros(shape):
    return np.zeros(shape, dtype=np.int64)


def ones(shape):
    return np.ones(shape, dtype=np.int64)


def univariate(f):
    return f
