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
    assert tfhe.default_params().q_bits in (64, 49)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bmi_amd import tfhe
    with pytest.raises(tfhe.BmiError) as e:
        tfhe.Engine()
    assert "no HIP device" in str(e.value) or "failed" in str(e.value)
