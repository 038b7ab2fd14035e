"""Circuit-level error budget: the probability that an encrypted evaluation of a compiled program returns a wrong integer
because of NOISE (as opposed to an input outside the traced ranges, which `Program.simulate` catches).

Concrete sizes its parameters for a global `p_error` (the reference relies on it: `fhe.Compiler(...).compile(inputset)` at its
defaults, matrix_inversion/main.py:53-66; `fhe.Configuration(...)` in qfloat_matrix_inversion.py:995-1002).  This module is the
counterpart for the fixed, documented parameter sets of this library: for every look-up of a program it computes the standard
deviation of what reaches the blind rotation and the distance to the nearest decision boundary, and sums the Gaussian tails.

Noise model (variances relative to q^2; the formulas the GPU tests hold the measured noise to within +-15 %,
tests/test_gpu_parity.py, tests/test_gpu_torus_wide.py):

  PBS output      v_pbs = n l (k+1) N (Bg^2 + 2) / 12 * s_eff^2  +  hw(s) (1 + hw(S)) / (12 Bg^(2l)),
                  s_eff^2 = glwe_noise^2 + (1 + hw(S)) 4^(64 - prec) / 12 / 2^128   (torus key stored at prec < 64 bits);
                  unrolled key: 3 x the key term, rounding term 2 x (pairs of key bits not both zero)
  fresh input     v_enc = glwe_noise^2                                                       (bmi_encrypt)
  a look-up's input is a linear combination sum_t c_t leaf_t: v_in = sum_t c_t^2 v(leaf_t)   (independent leaves)
  keyswitch       v_ks = kN l_ks (B^2 + 2) / 12 * lwe_noise^2 + hw(S) / (12 B^(2 l_ks))
  mod-switch      (v_in + v_ks) (2N)^2 + (1 + hw(s)) / 12    positions^2 on the circle of 2N positions

Decision boundary: a table over p bits has boxes of 2N / 2^(p+1) positions (narrower tables than the circuit's message space are
scaled up by the tracer: wider boxes, and their coefficients carry the scale); a value fails when the noise exceeds half a box on
either side: P = erfc(half_box / (sigma sqrt 2)).  Outputs fail at decryption when their noise exceeds Delta / 2.

Key weights are random: hw(s) = n / 2, hw(S) = kN / 2 (their expectations) unless given."""
from __future__ import annotations

import math

import numpy as np

TORUS64 = 65


def default_bsk_precision(P):
    """bits of precision the library stores a bootstrap key at by default (csrc/bmi_host.cpp default_bsk_precision)"""
    if P.q_bits != TORUS64:
        return 64
    if P.log_N == 11:
        return 46
    if P.log_N == 12:
        return 44
    return 48 if P.bs_base_log <= 10 else 64


def pbs_output_variance(P, bsk_precision=None, unroll=False, hw_small=None, hw_big=None):
    N, k, l, n = 1 << P.log_N, P.k, P.bs_levels, P.n
    Bg = 2.0 ** P.bs_base_log
    hs = n / 2.0 if hw_small is None else float(hw_small)
    hb = k * N / 2.0 if hw_big is None else float(hw_big)
    prec = default_bsk_precision(P) if bsk_precision is None else int(bsk_precision)
    s2 = P.glwe_noise ** 2
    if P.q_bits == TORUS64 and prec < 64:
        s2 += (1 + hb) * 4.0 ** (64 - prec) / 12.0 / 2.0 ** 128
    key = n * l * (k + 1) * N * (Bg * Bg + 2) / 12.0 * s2
    rnd = (1 + hb) / (12.0 * Bg ** (2 * l))
    if unroll:
        live_pairs = (n + 1) // 2 * 0.75 if hw_small is None else None
        if live_pairs is None:
            live_pairs = min((n + 1) // 2, hs)          # upper bound from the weight alone
        return 3 * key + 2 * live_pairs * rnd
    return key + hs * rnd


def keyswitch_variance(P, hw_big=None):
    N, k = 1 << P.log_N, P.k
    B = 2.0 ** P.ks_base_log
    hb = k * N / 2.0 if hw_big is None else float(hw_big)
    return k * N * P.ks_levels * (B * B + 2) / 12.0 * P.lwe_noise ** 2 + hb / (12.0 * B ** (2 * P.ks_levels))


def _log_erfc(x):
    """log10 of erfc(x), finite for large x (erfc underflows beyond x ~ 26)"""
    x = np.asarray(x, np.float64)
    out = np.empty_like(x)
    small = x < 25.0
    with np.errstate(divide="ignore"):
        out[small] = np.log10(np.maximum(np.vectorize(math.erfc)(x[small]), 1e-320)) if small.any() else 0
    big = ~small
    # erfc(x) ~ exp(-x^2) / (x sqrt(pi))
    out[big] = (-x[big] ** 2 - np.log(x[big] * math.sqrt(math.pi))) / math.log(10.0)
    return out


def failure_probability(prog, P, bsk_precision=None, unroll=False, hw_small=None, hw_big=None):
    """Error budget of `prog` (a program.Program) under the parameter set `P` (tfhe.Params or anything with its fields).

    Returns a dict: p_fail (probability that at least one look-up or output decodes wrong, union bound), log10_p_fail,
    lookups, worst_margin_sigma (smallest half-box / sigma over the look-ups), p_fail_worst_lookup, widest_amplification
    (largest sum of squared coefficients feeding a look-up), output_margin_sigma, sigma_positions_typical."""
    N = 1 << P.log_N
    n_in, nn = prog.n_inputs, prog.n_nodes
    v_pbs = pbs_output_variance(P, bsk_precision, unroll, hw_small, hw_big)
    v_enc = P.glwe_noise ** 2
    v_ks = keyswitch_variance(P, hw_big)
    hs = P.n / 2.0 if hw_small is None else float(hw_small)
    leaf_var = np.full(n_in + nn, v_pbs)
    leaf_var[:n_in] = v_enc

    def lincomb_var(ptr, leaf, coef):
        ln = np.diff(ptr)
        seg = np.repeat(np.arange(ln.size), ln)
        c2 = np.asarray(coef, np.float64) ** 2
        return (np.bincount(seg, weights=c2 * leaf_var[leaf], minlength=ln.size),
                np.bincount(seg, weights=c2, minlength=ln.size))

    res = {"lookups": int(nn), "params": {"n": int(P.n), "N": int(N), "l": int(P.bs_levels), "log2_Bg": int(P.bs_base_log),
                                          "q_bits": int(P.q_bits), "log2_lwe_noise": round(math.log2(P.lwe_noise), 2)},
           "log2_std_pbs_output": 0.5 * math.log2(v_pbs), "log2_std_keyswitch": 0.5 * math.log2(v_ks)}
    log_terms = []
    if nn:
        v_in, amp = lincomb_var(prog.node_ptr, prog.term_leaf, prog.term_coef)
        sigma_pos = np.sqrt((v_in + v_ks) * (2.0 * N) ** 2 + (1 + hs) / 12.0)
        p_node = prog.lut_p[prog.node_lut].astype(np.int64)
        if (1 << (int(p_node.max()) + 1)) > N:
            raise ValueError(f"a {int(p_node.max())}-bit look-up does not fit N = {N}")
        half_box = N / (2.0 ** (p_node + 1))
        margin = half_box / sigma_pos
        lp = _log_erfc(margin / math.sqrt(2.0))
        log_terms.append(lp)
        res.update(worst_margin_sigma=float(margin.min()), log10_p_fail_worst_lookup=float(lp.max()),
                   widest_amplification=float(amp.max()), sigma_positions_typical=float(np.median(sigma_pos)))
    if prog.n_outputs:
        v_out, _ = lincomb_var(prog.out_ptr, prog.out_leaf, prog.out_coef)
        m_out = 2.0 ** -(prog.msg_bits + 2) / np.sqrt(np.maximum(v_out, 1e-300))     # Delta / 2 over sigma, Delta = q / 2^(msg_bits + 1)
        log_terms.append(_log_erfc(m_out / math.sqrt(2.0)))
        res["output_margin_sigma"] = float(m_out.min())
    if log_terms:
        lt = np.concatenate(log_terms)
        mx = float(lt.max())
        log10_sum = mx + math.log10(float(np.sum(10.0 ** (lt - mx))))       # union bound, summed in log space
        res["log10_p_fail"] = min(log10_sum, 0.0)
        res["p_fail"] = min(1.0, 10.0 ** log10_sum) if log10_sum > -300 else 0.0
    else:
        res.update(log10_p_fail=-math.inf, p_fail=0.0)
    return res


def candidate_sets(msg_bits, q_bits=None, secure=False):
    """Parameter sets to try in order (cheapest first) for a circuit of `msg_bits`-bit look-ups: (label, kwargs of
    tfhe.default_params) or (label, preset name)."""
    from . import tfhe
    torus = q_bits in (None, tfhe.TORUS64)       # None: the library's default modulus
    if secure:
        # the secure LWE noise leaves a 4-bit look-up 9 sigma at N = 2048; a 5-bit one sits at 4.4 sigma there and at 4.9 sigma at
        # N = 4096 (torus only) - whether that meets p_error is the budget's call, circuit by circuit
        if msg_bits > 5 or (msg_bits > 4 and not torus):
            raise ValueError("the 128-bit-secure sets carry 4-bit look-ups (5-bit ones on the 2^64 torus, at 4.4 - 4.9 sigma)")
        return [("secure128_torus", "secure128_torus"), ("secure128_torus_wide", "secure128_torus_wide")] if torus else [("secure128", "secure128")]
    out = []
    for log_n in (10, 11, 12):
        if (1 << (msg_bits + 6)) > (1 << log_n):
            continue                                  # N >= 2^(msg_bits + 6): the box structure of the tracer's message space
        qb = tfhe.TORUS64 if torus else 49
        out.append((f"q_bits={qb}, N={1 << log_n}", dict(q_bits=qb, log_N=log_n) if log_n != 10 else dict(q_bits=qb)))
    if not out:
        raise ValueError(f"no parameter set carries {msg_bits}-bit look-ups on this modulus")
    return out


def choose_params(prog, p_error=1e-5, q_bits=None, secure=False, unroll=False):
    """The first candidate set whose failure probability for `prog` is at most p_error (Concrete's `global_p_error`); the report of
    every set tried.  Raises when none qualifies."""
    from . import tfhe
    tried = []
    for label, spec in candidate_sets(prog.msg_bits, q_bits, secure):
        P = tfhe.preset_params(spec) if isinstance(spec, str) else tfhe.default_params(**spec)
        rep = failure_probability(prog, P, unroll=unroll and P.log_N == 10)
        tried.append((label, rep["p_fail"]))
        if rep["p_fail"] <= p_error:
            return P, dict(rep, chosen=label, tried=tried, p_error=p_error)
    raise ValueError(f"no parameter set reaches p_error = {p_error:g} for this circuit: tried {tried}")
