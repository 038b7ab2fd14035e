"""numpy model of csrc/fft_wave_f64.hpp: the folded 512-point complex FFT one wavefront runs for a 1024-coefficient negacyclic
product on the 2^64 torus (64 lanes x 8 complex points, three register DFT8 passes, two LDS exchanges).  Index algebra only:
the model follows the kernel's registers / lanes / scratch addresses step by step and is checked against the definition

    A_k = sum_j (a_j + i a_{j+512}) zeta^j omega^{jk},   zeta = exp(i pi / 1024), omega = exp(2 pi i / 512)

(the values of a real polynomial at the roots zeta^(4k+1) of X^1024 + 1), and through a negacyclic product against the schoolbook
one.  Also prints the constants the header embeds.  Run: python tools/fft_wave_model.py"""
import numpy as np

N, H, ROWC = 1024, 512, 72
LANES = np.arange(64)
ZETA = np.exp(1j * np.pi / N)


def zeta_pow(e):
    """zeta^e from the reduced angle (what the host tables hold: long double cos / sin rounded to doubles)"""
    e = np.asarray(e) % (2 * N)
    ang = np.pi * e.astype(np.longdouble) / N
    return np.cos(ang).astype(float) + 1j * np.sin(ang).astype(float)

W8 = np.exp(2j * np.pi / 8)
HSQ = np.sqrt(0.5)


def br3(r):
    return ((r & 1) << 2) | (r & 2) | ((r & 4) >> 2)


def dft8(x, inv):
    """x: list of 8 complex lane-vectors; radix-2 DIF exactly as the header writes it; returns natural-order frequencies"""
    s = -1.0 if inv else 1.0
    muli = lambda v: s * 1j * v
    a0, d0 = x[0] + x[4], x[0] - x[4]
    a1, t1 = x[1] + x[5], x[1] - x[5]
    a2, t2 = x[2] + x[6], x[2] - x[6]
    a3, t3 = x[3] + x[7], x[3] - x[7]
    d1 = (t1.real - s * t1.imag) + 1j * (t1.imag + s * t1.real)       # t1 (1 + s i), the factor 1/sqrt 2 deferred
    d2 = muli(t2)
    d3 = (-t3.real - s * t3.imag) + 1j * (-t3.imag + s * t3.real)     # t3 (-1 + s i), deferred
    b0, b2 = a0 + a2, a0 - a2
    b1, b3 = a1 + a3, muli(a1 - a3)
    e0, e2 = d0 + d2, d0 - d2
    e1, e3 = d1 + d3, muli(d1 - d3)
    y = [None] * 8
    y[0], y[4] = b0 + b1, b0 - b1
    y[2], y[6] = b2 + b3, b2 - b3
    y[1], y[5] = e0 + HSQ * e1, e0 - HSQ * e1
    y[3], y[7] = e2 + HSQ * e3, e2 - HSQ * e3
    return y


def tables():
    t1 = np.array([zeta_pow(LANES * (4 * k2 + 1)) for k2 in range(8)])      # [k2][lane]
    t2 = np.array([zeta_pow(32 * np.arange(8) * d) for d in range(8)])     # [d][a] = omega_64^(a d)
    return t1, t2


def forward(a):
    """a: 1024 reals -> (V, where V[c][lane] is the value at frequency k = 64 c + 8 (lane & 7) + (lane >> 3))"""
    t1, t2 = tables()
    v = [(a[LANES + 64 * r] + 1j * a[LANES + 64 * r + 512]) * zeta_pow(64 * r) for r in range(8)]
    v = dft8(v, False)
    v = [v[k2] * t1[k2] for k2 in range(8)]
    scratch = np.zeros(8 * ROWC, complex)
    for k2 in range(8):
        scratch[k2 * ROWC + LANES] = v[k2]
    v = [scratch[(LANES >> 3) * ROWC + (LANES & 7) + 8 * b] for b in range(8)]
    v = dft8(v, False)
    v = [v[d] * t2[d][LANES & 7] for d in range(8)]
    for d in range(8):
        scratch[(LANES >> 3) * ROWC + d * 8 + (((LANES & 7) + d) & 7)] = v[d]
    v = [scratch[(LANES >> 3) * ROWC + (LANES & 7) * 8 + ((a_ + (LANES & 7)) & 7)] for a_ in range(8)]
    return dft8(v, False)


def inverse(V):
    t1, t2 = tables()
    v = dft8(list(V), True)                               # over c -> a
    scratch = np.zeros(8 * ROWC, complex)
    for a_ in range(8):
        scratch[(LANES >> 3) * ROWC + (LANES & 7) * 8 + ((a_ + (LANES & 7)) & 7)] = v[a_]
    v = [scratch[(LANES >> 3) * ROWC + d * 8 + (((LANES & 7) + d) & 7)] for d in range(8)]
    v = [v[d] * np.conj(t2[d][LANES & 7]) for d in range(8)]
    v = dft8(v, True)                                     # over d -> b
    for b in range(8):
        scratch[(LANES >> 3) * ROWC + (LANES & 7) + 8 * b] = v[b]
    v = [scratch[k2 * ROWC + LANES] for k2 in range(8)]
    v = [v[k2] * np.conj(t1[k2]) for k2 in range(8)]
    v = dft8(v, True)                                     # over k2 -> r
    out = np.zeros(N)
    for r in range(8):
        w = v[r] * np.conj(zeta_pow(64 * r)) / H
        out[LANES + 64 * r] = w.real
        out[LANES + 64 * r + 512] = w.imag
    return out


def freq_of(c, lane):
    return 64 * c + 8 * (lane & 7) + (lane >> 3)


def main():
    rng = np.random.default_rng(1)
    a = rng.integers(-512, 512, N).astype(float)
    V = forward(a)
    j = np.arange(H)
    u = (a[:H] + 1j * a[H:]) * ZETA ** j
    want = np.array([np.sum(u * np.exp(2j * np.pi * j * k / H)) for k in range(H)])
    got = np.zeros(H, complex)
    for c in range(8):
        got[freq_of(c, LANES)] = V[c]
    print("forward max err", np.abs(got - want).max())
    assert np.abs(got - want).max() < 1e-6
    back = inverse(V)
    print("round trip max err", np.abs(back - a).max())
    assert np.abs(back - a).max() < 1e-9
    # negacyclic product of digits with 24-bit limbs, exactness margin
    worst = 0.0
    for _ in range(20):
        d = rng.integers(-512, 512, (6, N)).astype(float)
        k = rng.integers(-(1 << 23), 1 << 23, (6, N)).astype(float)
        acc = [np.zeros(64, complex) for _ in range(8)]
        for p in range(6):
            D, K = forward(d[p]), forward(k[p])
            acc = [acc[c] + D[c] * K[c] for c in range(8)]
        r = inverse(acc)
        exact = np.zeros(N, dtype=object)
        for p in range(6):
            full = np.convolve(d[p].astype(np.int64).astype(object), k[p].astype(np.int64).astype(object))
            full = np.concatenate([full, np.zeros(2 * N - len(full), dtype=object)])
            exact += full[:N] - full[N:2 * N]
        err = np.abs(r - exact.astype(float)).max()
        assert np.all(np.rint(r) == exact.astype(float))
        worst = max(worst, err)
    print("largest distance of a limb sum from its integer over 20 CMUX-sized sums: 2^%.1f" % np.log2(worst))
    print("twist constants zeta^(64 r): cos / sin of pi r / 16")
    for r in range(8):
        print("    %.17g, %.17g," % (np.cos(np.pi * r / 16), np.sin(np.pi * r / 16)))


if __name__ == "__main__":
    main()
