"""CPU-side checks of the drop-in boundary: libbmi_tfhe.so loads and exports every symbol that
include/bmi_tfhe.h declares; without a GPU the context constructor fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "bmi_tfhe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bmi_amd import tfhe
    lib = tfhe.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/bmi_tfhe.h but not exported"
    # and the binding wires every one of them
    bound = set(tfhe._SIGS) | {"bmi_ctx_destroy", "bmi_last_error"}
    assert set(syms) <= bound, set(syms) - bound


def test_default_params_are_the_north_star_set():
    from bmi_amd import tfhe
    from oracle import tfhe_oracle as to
    for qb in (64, 49):
        P = tfhe.default_params(q_bits=qb)
        assert (P.n, P.N, P.k, P.bs_levels, P.q_bits) == (630, 1024, 1, 3, qb)
        O = to.default_params(q_bits=qb)  # same numbers as the oracle's default set
        for f, _ in tfhe.Params._fields_:
            assert getattr(P, f) == getattr(O, f), f
    assert tfhe.default_params().q_bits == tfhe.TORUS64          # one default modulus everywhere: q = 2^64, Concrete's own


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bmi_amd import tfhe
    with pytest.raises(tfhe.BmiError) as e:
        tfhe.Engine()
    assert "no HIP device" in str(e.value) or "failed" in str(e.value)


def test_torus64_conversions_are_the_rounded_modulus_switch():
    """2^64-torus interop: bmi_torus64_to_field / bmi_field_to_torus64 against the exact rational
    formulas round(x q / 2^64) mod q and round(x 2^64 / q) mod 2^64 (ties upwards), for both moduli; a round trip
    field -> torus -> field is the identity, torus -> field -> torus moves a word by less than 2^64 / q."""
    import numpy as np
    from bmi_amd import tfhe
    rng = np.random.default_rng(64)
    for qb, q in ((49, (1 << 49) - 720895), (64, (1 << 64) - (1 << 32) + 1)):
        x = np.concatenate([rng.integers(0, 1 << 63, 5000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 5000, dtype=np.uint64),
                            np.array([0, 1, (1 << 64) - 1, 1 << 63, (1 << 63) - 1, (1 << 15), (1 << 64) - (1 << 14)], dtype=np.uint64)])
        got = tfhe.torus64_to_field(x, qb)
        want = [((int(v) * q + (1 << 63)) >> 64) % q for v in x]
        assert [int(v) for v in got] == want
        y = np.concatenate([rng.integers(0, q, 5000, dtype=np.uint64), np.array([0, 1, q - 1, q >> 1, (q >> 1) + 1], dtype=np.uint64)])
        back = tfhe.field_to_torus64(y, qb)
        assert [int(v) for v in back] == [(((int(v) << 64) + (q >> 1)) // q) % (1 << 64) for v in y]
        assert np.array_equal(tfhe.torus64_to_field(back, qb), y)
        step = (1 << 64) // q + 1
        d = (tfhe.field_to_torus64(got, qb).astype(np.uint64) - x).astype(np.int64)
        assert np.abs(d).max() <= step
        with pytest.raises(tfhe.BmiError):
            tfhe.field_to_torus64(np.array([q], dtype=np.uint64), qb)
