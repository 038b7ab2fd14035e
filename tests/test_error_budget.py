"""Circuit-level error budget (bmi_amd/error_budget.py): what Concrete's `p_error` / `global_p_error` guarantee for the reference
(matrix_inversion/main.py:53-66) - here computed for the library's fixed parameter sets from the compiled program: per look-up the
noise that reaches the blind rotation (the linear combination's amplification included) against half a box, summed over the circuit.
CPU only: the budget is arithmetic on the program's arrays (the GPU tests hold the noise formulas it uses to the measured noise)."""
import math

import numpy as np
import pytest

from bmi_amd import error_budget, tfhe
from bmi_amd.main import EncryptedMatrixInversion, compile_inverse


def test_budget_of_the_baseline_configurations():
    """north-star set (n 630, N 1024, 4-bit look-ups at 6.2 sigma): 2x2 / 3x3 / 4x4 fail by noise with probability ~2e-6 / 3e-5 /
    9e-5 (proportional to the look-up count); N = 2048 and the 128-bit-secure torus set are far below 1e-9"""
    north = tfhe.default_params(q_bits=tfhe.TORUS64)
    last = 0.0
    for n, ln, ints in ((2, 20, 8), (3, 30, 12), (4, 40, 16)):
        prog, _ = compile_inverse(n, ln, ints)
        r = prog.failure_probability(north)
        assert r["lookups"] == prog.n_nodes and 6.0 < r["worst_margin_sigma"] < 6.5
        per = r["p_fail"] / r["lookups"]
        assert 1e-10 < per < 1e-9 and r["p_fail"] > last          # at most the full-width look-up's tail each
        last = r["p_fail"]
        assert r["output_margin_sigma"] > 100                      # decryption never the weak point
        assert prog.failure_probability(tfhe.default_params(q_bits=tfhe.TORUS64, log_N=11))["p_fail"] < 1e-25
        sec = prog.failure_probability(tfhe.preset_params("secure128_torus"))
        assert sec["p_fail"] < 1e-9 and 8.5 < sec["worst_margin_sigma"] < 9.5
    assert 5e-5 < last < 2e-4


def test_the_unrolled_key_and_the_49_bit_field_enter_the_budget():
    prog, _ = compile_inverse(3, 30, 12)
    p49 = tfhe.default_params(q_bits=49)
    plain = prog.failure_probability(p49)
    unrolled = prog.failure_probability(tfhe.default_params(q_bits=49, glwe_noise=2.0 ** -41), unroll=True)
    assert plain["log2_std_pbs_output"] == pytest.approx(-15.85, abs=0.1)
    assert unrolled["log2_std_pbs_output"] == pytest.approx(-16.05, abs=0.15)
    assert 0.3 < unrolled["p_fail"] / plain["p_fail"] < 3.0


def test_parameter_choice_follows_p_error():
    """choose_params: the north-star set where it meets the target, N = 2048 where the look-up count needs the margin; the secure
    set on request; a circuit no set can carry is refused"""
    small, _ = compile_inverse(2, 20, 8)
    big, _ = compile_inverse(4, 40, 16)
    P, rep = error_budget.choose_params(small, 1e-5, q_bits=tfhe.TORUS64)
    assert (P.N, P.n, P.q_bits) == (1024, 630, tfhe.TORUS64) and rep["p_fail"] <= 1e-5 and len(rep["tried"]) == 1
    P, rep = error_budget.choose_params(big, 1e-5, q_bits=tfhe.TORUS64)
    assert (P.N, P.n) == (2048, 630) and rep["p_fail"] <= 1e-5 and len(rep["tried"]) == 2 and rep["tried"][0][1] > 1e-5
    P, rep = error_budget.choose_params(big, 1e-5, q_bits=49)
    assert (P.N, P.q_bits) == (2048, 49)
    P, rep = error_budget.choose_params(big, 1e-5, q_bits=None)          # the library's default modulus
    assert (P.N, P.q_bits) == (2048, tfhe.TORUS64)
    P, rep = error_budget.choose_params(big, 1e-9, q_bits=tfhe.TORUS64, secure=True)
    assert (P.N, P.n) == (2048, 742) and rep["chosen"] == "secure128_torus"
    with pytest.raises(ValueError):
        error_budget.choose_params(big, 1e-40, q_bits=tfhe.TORUS64, secure=True)


def test_wider_look_ups_pick_the_wider_rings_on_the_torus():
    """6-bit look-ups: N = 4096 on the library's default modulus (k_blind_rotate_q_t64f); 5-bit look-ups under security_level 128:
    secure128_torus (N 2048, 4.4 sigma per look-up) while that meets the target, secure128_torus_wide (N 4096, keyswitch 16 x 1 bit:
    5.3 sigma) beyond"""
    from bmi_amd.circuit import Circuit
    from bmi_amd.program import Program

    def chain(bits, n):
        c = Circuit(msg_bits=bits)
        h = 1 << (bits - 1)
        xs = [c.input(-h, h - 1) for _ in range(n)]
        c.set_outputs([c.lut(x, lambda v: -v - 1) for x in xs])
        return Program.from_circuit(c)

    P, rep = error_budget.choose_params(chain(6, 8), 1e-5)
    assert (P.N, P.q_bits, P.n) == (4096, tfhe.TORUS64, 630) and rep["p_fail"] < 1e-7 and 6.0 < rep["worst_margin_sigma"] < 6.5
    P, _ = error_budget.choose_params(chain(6, 8), 1e-5, q_bits=49)
    assert (P.N, P.q_bits) == (4096, 49)
    five = chain(5, 60)
    P, rep = error_budget.choose_params(five, 1e-2, q_bits=tfhe.TORUS64, secure=True)
    assert rep["chosen"] == "secure128_torus" and 4.2 < rep["worst_margin_sigma"] < 4.6
    P, rep = error_budget.choose_params(five, 1e-4, q_bits=tfhe.TORUS64, secure=True)
    assert rep["chosen"] == "secure128_torus_wide" and (P.N, P.n, P.ks_levels, P.ks_base_log) == (4096, 742, 16, 1)
    assert 5.2 < rep["worst_margin_sigma"] < 5.5 and rep["tried"][0][1] > 1e-4
    with pytest.raises(ValueError):
        error_budget.choose_params(chain(6, 8), 1e-5, secure=True)          # no 128-bit-secure set carries 6-bit look-ups


def test_wrapper_reports_and_enforces_its_budget():
    """EncryptedMatrixInversion(p_error=...) picks the set when it creates the engine; with an engine passed in, the budget is
    checked against it (no GPU here: a stand-in with the engine's parameter fields)"""
    class FakeEngine:
        def __init__(self, P):
            self.P, self.q_bits, self.bsk_precision, self.unroll, self.device = P, P.q_bits, 48, 1, 0
    emi = EncryptedMatrixInversion(4, None, 2, 40, 16, engine=FakeEngine(tfhe.default_params(q_bits=tfhe.TORUS64)), p_error=1e-5)
    with pytest.raises(ValueError, match="p_error"):
        emi._engine()
    emi = EncryptedMatrixInversion(4, None, 2, 40, 16, engine=FakeEngine(tfhe.default_params(q_bits=tfhe.TORUS64)))
    emi._engine()
    assert 5e-5 < emi.error_budget["p_fail"] < 2e-4


def test_tail_sum_in_log_space():
    """margins far beyond erfc's underflow still give a finite, monotone log-probability"""
    a, b = error_budget._log_erfc(np.array([5.0, 30.0, 60.0])), None
    assert a[0] == pytest.approx(math.log10(math.erfc(5.0)), rel=1e-9)
    assert a[1] > a[2] and np.isfinite(a).all() and a[2] < -1500
