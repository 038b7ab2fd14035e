"""numpy model of csrc/fft_quarter_f64.hpp: the folded 1,024-point complex transform of a 2,048-coefficient real polynomial
(N = 2048 on the 2^64 torus: the secure128_torus set) split over FOUR wavefronts by the folded index mod 4.

    u_j = (c_j + i c_{j+1024}) zeta^j,   zeta = exp(i pi / 2048),   A_k = sum_{j < 1024} u_j omega^(jk),   omega = zeta^4
    (A_k = c(zeta^(4k+1)): the values of c at 1,024 roots of X^2048 + 1, the other 1,024 are the conjugates)

Quarter h transforms the 256 points u_{4m+h}: its twist zeta^(4m) zeta^h is the twist of the EVEN half of the N = 1024 split
(fft_half_model.forward_half(., 0): exp(i pi 2m / 1024)) times the constant zeta^h, so wavefront h runs that code unchanged on
re[r] = c[4 (lane + 64 r) + h], im[r] = c[4 (lane + 64 r) + h + 1024] and multiplies slot p (frequency kappa) by
W_h[p] = zeta^(h (4 kappa + 1)):

    Q'_h[kappa] = W_h[kappa] * half0(c_h)[kappa],      A_{kappa + 256 t} = sum_h i^(h t) Q'_h[kappa]          (omega^256 = i)

The inverse runs backwards: S_h[kappa] = conj(W_h[kappa]) sum_t i^(-h t) Y_{kappa + 256 t}, then the inverse even half
(which carries 1/512; the missing factor 1/2 of 1/1024 is folded into the key copy).
Index algebra only, checked against the definition and through a CMUX-sized sum of negacyclic products.
Run: python tools/fft_quarter_model.py"""
import numpy as np

import fft_half_model as hm

N = 2048
LANES = hm.LANES
slot_freq = hm.slot_freq


def zeta_pow(e):
    e = np.asarray(e) % (2 * N)
    ang = np.pi * e.astype(np.longdouble) / N
    return np.cos(ang).astype(float) + 1j * np.sin(ang).astype(float)


def w_table(h):
    """W_h[reg][lane] = zeta^(h (4 kappa + 1)) at slot (reg, lane)"""
    return [zeta_pow(h * (4 * slot_freq(r, LANES) + 1)) for r in range(4)]


def quarter_input(c, h):
    """the 1,024 reals the even-half code of the N = 1024 split reads for quarter h: position 2 m is c[4 m + h], 2 m + 512 is c[4 m + h + 1024]"""
    a = np.zeros(1024)
    m = np.arange(256)
    a[2 * m] = c[4 * m + h]
    a[2 * m + 512] = c[4 * m + h + 1024]
    return a


def forward_quarter(c, h):
    """[reg][lane] -> Q'_h at slot order"""
    v = hm.forward_half(quarter_input(c, h), 0)
    if h:
        w = w_table(h)
        v = [v[r] * w[r] for r in range(4)]
    return v


def full_from_quarters(c):
    Q = [forward_quarter(c, h) for h in range(4)]
    A = np.zeros(1024, complex)
    for r in range(4):
        kap = slot_freq(r, LANES)
        for t in range(4):
            A[kap + 256 * t] = sum((1j) ** (h * t) * Q[h][r] for h in range(4))
    return A


def inverse_from_products(Y):
    """Y: 1,024 frequency values of a product (already scaled by 1/2) -> 2,048 real coefficients"""
    out = np.zeros(N)
    for h in range(4):
        w = w_table(h)
        S = []
        for r in range(4):
            kap = slot_freq(r, LANES)
            S.append(np.conj(w[r]) * sum((-1j) ** (h * t) * Y[kap + 256 * t] for t in range(4)))
        z = hm.inverse_half(S, 0)
        for r in range(4):
            m = LANES + 64 * r
            out[4 * m + h] = z[r].real
            out[4 * m + h + 1024] = z[r].imag
    return out


def definition(c):
    j = np.arange(1024)
    u = (c[:1024] + 1j * c[1024:]) * zeta_pow(j)
    return np.array([np.sum(u * np.exp(2j * np.pi * j * k / 1024)) for k in range(1024)])


def exact_negacyclic_sum(d, k):
    n = d.shape[1]
    acc = np.zeros(n, dtype=object)
    for p in range(d.shape[0]):
        full = np.convolve(d[p].astype(np.int64).astype(object), k[p].astype(np.int64).astype(object))
        full = np.concatenate([full, np.zeros(2 * n - len(full), dtype=object)])
        acc += full[:n] - full[n:2 * n]
    return acc.astype(float)


def main():
    rng = np.random.default_rng(5)
    c = rng.integers(-512, 512, N).astype(float)
    err = np.abs(full_from_quarters(c) - definition(c)).max()
    print("forward (four quarters) max err", err)
    assert err < 1e-5
    worst = 0.0
    for _ in range(4):
        d = rng.integers(-512, 512, (6, N)).astype(float)
        kk = rng.integers(-(1 << 22), 1 << 22, (6, N)).astype(float)     # 23-bit balanced limbs
        Y = sum(full_from_quarters(d[p]) * (0.5 * full_from_quarters(kk[p])) for p in range(6))
        out = inverse_from_products(Y)
        exact = exact_negacyclic_sum(d, kk)
        assert np.all(np.rint(out) == exact)
        worst = max(worst, np.abs(out - exact).max())
    print("largest distance of a limb sum from its integer: 2^%.1f" % np.log2(worst))


if __name__ == "__main__":
    main()
