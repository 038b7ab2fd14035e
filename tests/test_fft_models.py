"""CPU checks of the index algebra of the two floating-point transforms of the 2^64-torus kernels, on their numpy models
(tools/fft_wave_model.py = csrc/fft_wave_f64.hpp: 512 = 8 x 8 x 8 with two LDS exchanges; tools/fft_half_model.py =
csrc/fft_half_f64.hpp: two 256-point halves, 4 x 4 x 4 x 4 with register / lane-bit transposes): forward against the definition
A_k = sum_j (a_j + i a_{j+512}) zeta^j omega^(jk), round trip, and one CMUX-sized sum of products rounding to the exact integers.
The HIP code follows the models step by step; tests/test_gpu_torus_fft.py holds the kernels to the oracle on the GPU."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


def _definition(a):
    j = np.arange(512)
    ang = np.pi * (j.astype(np.longdouble)) / 1024
    u = (a[:512] + 1j * a[512:]) * (np.cos(ang).astype(float) + 1j * np.sin(ang).astype(float))
    return np.array([np.sum(u * np.exp(2j * np.pi * j * k / 512)) for k in range(512)])


def _exact_negacyclic_sum(d, k):
    n = d.shape[1]
    acc = np.zeros(n, dtype=object)
    for p in range(d.shape[0]):
        full = np.convolve(d[p].astype(np.int64).astype(object), k[p].astype(np.int64).astype(object))
        full = np.concatenate([full, np.zeros(2 * n - len(full), dtype=object)])
        acc += full[:n] - full[n:2 * n]
    return acc.astype(float)


def test_wave_transform_model():
    import fft_wave_model as m
    rng = np.random.default_rng(1)
    a = rng.integers(-512, 512, 1024).astype(float)
    V = m.forward(a)
    got = np.zeros(512, complex)
    for c in range(8):
        got[m.freq_of(c, m.LANES)] = V[c]
    assert np.abs(got - _definition(a)).max() < 1e-6
    assert np.abs(m.inverse(V) - a).max() < 1e-9
    d = rng.integers(-512, 512, (6, 1024)).astype(float)
    k = rng.integers(-(1 << 23), 1 << 23, (6, 1024)).astype(float)
    acc = [np.zeros(64, complex) for _ in range(8)]
    for p in range(6):
        D, K = m.forward(d[p]), m.forward(k[p])
        acc = [acc[c] + D[c] * K[c] for c in range(8)]
    r = m.inverse(acc)
    exact = _exact_negacyclic_sum(d, k)
    assert np.all(np.rint(r) == exact) and np.abs(r - exact).max() < 2.0 ** -9


def test_half_transform_model():
    import fft_half_model as m
    rng = np.random.default_rng(2)
    a = rng.integers(-512, 512, 1024).astype(float)
    assert np.abs(m.full_from_halves(a) - _definition(a)).max() < 1e-6
    d = rng.integers(-512, 512, (6, 1024)).astype(float)
    k = rng.integers(-(1 << 23), 1 << 23, (6, 1024)).astype(float)
    Y = sum(m.full_from_halves(d[p]) * m.full_from_halves(k[p]) for p in range(6))
    out = np.zeros(1024)
    S, D = [None] * 4, [None] * 4
    for r in range(4):
        f = m.slot_freq(r, m.LANES)
        S[r] = Y[f] + Y[f + 256]
        D[r] = (Y[f] - Y[f + 256]) * np.conj(m.zeta_pow(4 * f))
    for h, v in ((0, S), (1, D)):
        w = m.inverse_half(v, h)
        for r in range(4):
            idx = 2 * (m.LANES + 64 * r) + h
            out[idx] = w[r].real
            out[idx + 512] = w[r].imag
    exact = _exact_negacyclic_sum(d, k)
    assert np.all(np.rint(out) == exact) and np.abs(out - exact).max() < 2.0 ** -9


def test_quarter_transform_model():
    """N = 2048 (csrc/fft_quarter_f64.hpp): four 256-point quarters + a radix-4 butterfly against the definition
    A_k = sum_j (c_j + i c_{j+1024}) zeta^j omega^(jk), and one CMUX-sized sum of products (digits against 23-bit limbs, the key
    side scaled by 1/2 as the key copy is) rounding to the exact integers"""
    import fft_quarter_model as m
    rng = np.random.default_rng(4)
    c = rng.integers(-512, 512, 2048).astype(float)
    assert np.abs(m.full_from_quarters(c) - m.definition(c)).max() < 1e-5
    d = rng.integers(-512, 512, (6, 2048)).astype(float)
    k = rng.integers(-(1 << 22), 1 << 22, (6, 2048)).astype(float)
    Y = sum(m.full_from_quarters(d[p]) * (0.5 * m.full_from_quarters(k[p])) for p in range(6))
    out = m.inverse_from_products(Y)
    exact = _exact_negacyclic_sum(d, k)
    assert np.all(np.rint(out) == exact) and np.abs(out - exact).max() < 2.0 ** -9


def test_eighth_transform_model():
    """N = 4096 (csrc/fft_eighth_f64.hpp): eight 256-point eighths + a radix-8 butterfly against the definition, the three stored
    tables W_1, W_2, W_4 against W_h, and one CMUX-sized sum of products (digits against 22-bit limbs, the key side scaled by 1/4
    as the key copy is) rounding to the exact integers"""
    import fft_eighth_model as m
    rng = np.random.default_rng(5)
    for h in range(8):
        a, b = m.w_table(h), m.w_from_three(h)
        assert max(np.abs(a[r] - b[r]).max() for r in range(4)) < 1e-14
    c = rng.integers(-512, 512, 4096).astype(float)
    assert np.abs(m.full_from_eighths(c) - m.definition(c)).max() < 1e-4
    d = rng.integers(-512, 512, (6, 4096)).astype(float)
    k = rng.integers(-(1 << 21), 1 << 21, (6, 4096)).astype(float)
    Y = sum(m.full_from_eighths(d[p]) * (0.25 * m.full_from_eighths(k[p])) for p in range(6))
    out = m.inverse_from_products(Y)
    exact = m.exact_negacyclic_sum(d, k)
    assert np.all(np.rint(out) == exact) and np.abs(out - exact).max() < 2.0 ** -8


def test_a_priori_rounding_bounds():
    """tools/fft_bound.py: Percival's bound on a limb sum stays below 1/2 for both torus sets (the twist counted as a stage of its
    own) and fails for 24-bit limbs at N = 2048 - which is why that set stores its key at 46 bits"""
    import fft_bound as b
    n1024, _ = b.limb_sum_bound(10, 3, 10, 24)
    n2048, _ = b.limb_sum_bound(11, 3, 10, 23)
    bad, _ = b.limb_sum_bound(11, 3, 10, 24)
    n4096, _ = b.limb_sum_bound(12, 3, 10, 22)
    bad4, _ = b.limb_sum_bound(12, 3, 10, 23)
    assert 0.3 < n1024 < 0.5 and 0.3 < n2048 < 0.5 and bad > 0.5 and 0.3 < n4096 < 0.5 and bad4 > 0.5
