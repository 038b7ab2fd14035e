"""numpy model of csrc/fft_eighth_f64.hpp: the folded 2,048-point complex transform of a 4,096-coefficient real polynomial
(N = 4096 on the 2^64 torus) split over EIGHT wavefronts by the folded index mod 8.

    u_j = (c_j + i c_{j+2048}) zeta^j,   zeta = exp(i pi / 4096),   A_k = sum_{j < 2048} u_j omega^(jk),   omega = zeta^4

Eighth h transforms the 256 points u_{8m+h}: its twist zeta^(8m) zeta^h is again the twist of the even half of the N = 1024 split
(exp(i pi 2m / 1024)) times zeta^h, so a wavefront runs fft_half_model.forward_half(., 0) unchanged on
re[r] = c[8 (lane + 64 r) + h], im[r] = c[8 (lane + 64 r) + h + 2048] and multiplies slot p (frequency kappa) by
W_h[p] = zeta^(h (4 kappa + 1)) = W_1^(h & 1) W_2^((h >> 1) & 1) W_4^(h >> 2)   (three tables):

    Q'_h[kappa] = W_h[kappa] * half0(c_h)[kappa],      A_{kappa + 256 t} = sum_h e8^(h t) Q'_h[kappa],   e8 = exp(2 pi i / 8)

- a radix-8 butterfly taken where the products are.  The inverse runs backwards:
S_h[kappa] = conj(W_h[kappa]) sum_t e8^(-h t) Y_{kappa + 256 t}, then the inverse even half (1/512; the missing 1/4 of 1/2048 is
folded into the key copy).  Run: python tools/fft_eighth_model.py"""
import numpy as np

import fft_half_model as hm

N = 4096
LANES = hm.LANES
slot_freq = hm.slot_freq


def zeta_pow(e):
    e = np.asarray(e) % (2 * N)
    ang = np.pi * e.astype(np.longdouble) / N
    return np.cos(ang).astype(float) + 1j * np.sin(ang).astype(float)


def w_table(h):
    return [zeta_pow(h * (4 * slot_freq(r, LANES) + 1)) for r in range(4)]


def w_from_three(h):
    """W_h as the product of the stored tables W_1, W_2, W_4 (what the kernel does)"""
    out = [np.ones(64, complex) for _ in range(4)]
    for bit, tab in ((1, w_table(1)), (2, w_table(2)), (4, w_table(4))):
        if h & bit:
            out = [out[r] * tab[r] for r in range(4)]
    return out


def eighth_input(c, h):
    a = np.zeros(1024)
    m = np.arange(256)
    a[2 * m] = c[8 * m + h]
    a[2 * m + 512] = c[8 * m + h + 2048]
    return a


def forward_eighth(c, h):
    v = hm.forward_half(eighth_input(c, h), 0)
    w = w_from_three(h)
    return [v[r] * w[r] for r in range(4)]


E8 = np.exp(2j * np.pi / 8)


def full_from_eighths(c):
    Q = [forward_eighth(c, h) for h in range(8)]
    A = np.zeros(2048, complex)
    for r in range(4):
        kap = slot_freq(r, LANES)
        for t in range(8):
            A[kap + 256 * t] = sum(E8 ** (h * t) * Q[h][r] for h in range(8))
    return A


def inverse_from_products(Y):
    """Y: 2,048 frequency values of a product (already scaled by 1/4) -> 4,096 real coefficients"""
    out = np.zeros(N)
    for h in range(8):
        w = w_from_three(h)
        S = []
        for r in range(4):
            kap = slot_freq(r, LANES)
            S.append(np.conj(w[r]) * sum(E8 ** (-h * t) * Y[kap + 256 * t] for t in range(8)))
        z = hm.inverse_half(S, 0)
        for r in range(4):
            m = LANES + 64 * r
            out[8 * m + h] = z[r].real
            out[8 * m + h + 2048] = z[r].imag
    return out


def definition(c):
    j = np.arange(2048)
    u = (c[:2048] + 1j * c[2048:]) * zeta_pow(j)
    k = np.arange(2048)
    return np.array([np.sum(u * np.exp(2j * np.pi * j * kk / 2048)) for kk in k])


def exact_negacyclic_sum(d, k):
    n = d.shape[1]
    acc = np.zeros(n, dtype=object)
    for p in range(d.shape[0]):
        full = np.convolve(d[p].astype(np.int64).astype(object), k[p].astype(np.int64).astype(object))
        full = np.concatenate([full, np.zeros(2 * n - len(full), dtype=object)])
        acc += full[:n] - full[n:2 * n]
    return acc.astype(float)


def main():
    rng = np.random.default_rng(6)
    c = rng.integers(-512, 512, N).astype(float)
    for h in range(8):
        a, b = w_table(h), w_from_three(h)
        assert max(np.abs(a[r] - b[r]).max() for r in range(4)) < 1e-14
    err = np.abs(full_from_eighths(c) - definition(c)).max()
    print("forward (eight eighths) max err", err)
    assert err < 1e-4
    d = rng.integers(-512, 512, (6, N)).astype(float)
    kk = rng.integers(-(1 << 21), 1 << 21, (6, N)).astype(float)     # 22-bit balanced limbs
    Y = sum(full_from_eighths(d[p]) * (0.25 * full_from_eighths(kk[p])) for p in range(6))
    out = inverse_from_products(Y)
    exact = exact_negacyclic_sum(d, kk)
    assert np.all(np.rint(out) == exact)
    print("largest distance of a limb sum from its integer: 2^%.1f" % np.log2(np.abs(out - exact).max()))


if __name__ == "__main__":
    main()
