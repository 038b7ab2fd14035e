"""numpy model of csrc/fft_half_f64.hpp: the folded 512-point complex transform of a 1,024-coefficient real polynomial split over
TWO wavefronts by the parity of the folded index (latency kernel of the 2^64 torus): wavefront h transforms the 256 points
u_{2m+h} (4 complex points per lane, 256 = 4 x 4 x 4 x 4: register DFT4s with the lane bits moved into the registers by
2 x 2 transposes - v_permlane32_swap / v_permlane16_swap for lane bits 5 and 4, DPP moves for bits 3..0 - no LDS), the odd
half multiplies its output by omega_512^k, and the pair (E + O', E - O') is formed where the products are taken.
Index algebra only, checked against the definition and through a negacyclic product.  Run: python tools/fft_half_model.py"""
import numpy as np

N = 1024
LANES = np.arange(64)


def zeta_pow(e):
    e = np.asarray(e) % (2 * N)
    ang = np.pi * e.astype(np.longdouble) / N
    return np.cos(ang).astype(float) + 1j * np.sin(ang).astype(float)


def dft4(x, inv):
    s = -1j if inv else 1j
    a, b = x[0] + x[2], x[0] - x[2]
    c, d = x[1] + x[3], s * (x[1] - x[3])
    return [a + c, b + d, a - c, b - d]


def transpose(v, rb, lb):
    """v[reg][lane] -> the 2 x 2 transposes between register bit rb and lane bit lb"""
    out = [np.empty(64, complex) for _ in range(4)]
    for R in range(4):
        for L in range(64):
            rbit, lbit = (R >> rb) & 1, (L >> lb) & 1
            Rs = (R & ~(1 << rb)) | (lbit << rb)
            Ls = (L & ~(1 << lb)) | (rbit << lb)
            out[R][L] = v[Rs][Ls]
    return out


def slot_freq(reg, lane):
    """frequency k (of the 256-point half transform) held by register `reg` of lane `lane` after the forward half"""
    return 64 * reg + 16 * (lane & 3) + 4 * ((lane >> 2) & 3) + (lane >> 4)


def tables(h):
    t1 = np.array([zeta_pow(h + LANES * (2 + 8 * k2)) for k2 in range(4)])     # [k2][lane]: twist of the lane + omega_256^(lane k2)
    t2 = np.array([zeta_pow(32 * (np.arange(16)) * k0) for k0 in range(4)])     # [kappa0][n0]: omega_64^(n0 kappa0)
    t3 = np.array([zeta_pow(128 * np.arange(4) * l0) for l0 in range(4)])       # [lambda0][l0]: omega_16^(l0 lambda0)
    return t1, t2, t3


def forward_half(a, h):
    """a: 1024 reals; returns v[reg][lane] = E_k (h = 0) or omega_512^k O_k (h = 1), k = slot_freq(reg, lane)"""
    t1, t2, t3 = tables(h)
    v = [(a[2 * (LANES + 64 * r) + h] + 1j * a[2 * (LANES + 64 * r) + h + 512]) * zeta_pow(128 * r) for r in range(4)]
    v = dft4(v, False)
    v = [v[k2] * t1[k2] for k2 in range(4)]
    v = transpose(transpose(v, 1, 5), 0, 4)
    v = dft4(v, False)
    v = [v[k0] * t2[k0][LANES & 15] for k0 in range(4)]
    v = transpose(transpose(v, 1, 3), 0, 2)
    v = dft4(v, False)
    v = [v[l0] * t3[l0][LANES & 3] for l0 in range(4)]
    v = transpose(transpose(v, 1, 1), 0, 0)
    v = dft4(v, False)
    if h:
        v = [v[r] * zeta_pow(4 * slot_freq(r, LANES)) for r in range(4)]
    return v


def inverse_half(v, h):
    """v[reg][lane]: S (h = 0) or D (h = 1) in slot order -> (coefficients 2m+h, 2m+h+512 for m = lane + 64 r) as [r][lane] complex"""
    t1, t2, t3 = tables(h)
    v = dft4(list(v), True)
    v = transpose(transpose(v, 0, 0), 1, 1)
    v = [v[l0] * np.conj(t3[l0][LANES & 3]) for l0 in range(4)]
    v = dft4(v, True)
    v = transpose(transpose(v, 0, 2), 1, 3)
    v = [v[k0] * np.conj(t2[k0][LANES & 15]) for k0 in range(4)]
    v = dft4(v, True)
    v = transpose(transpose(v, 0, 4), 1, 5)
    v = [v[k2] * np.conj(t1[k2]) for k2 in range(4)]
    v = dft4(v, True)
    return [v[r] * np.conj(zeta_pow(128 * r)) / 512 for r in range(4)]


def full_from_halves(a):
    E, O = forward_half(a, 0), forward_half(a, 1)
    F = np.zeros(512, complex)
    for r in range(4):
        k = slot_freq(r, LANES)
        F[k] = E[r] + O[r]
        F[k + 256] = E[r] - O[r]
    return F


def main():
    rng = np.random.default_rng(3)
    a = rng.integers(-512, 512, N).astype(float)
    j = np.arange(512)
    u = (a[:512] + 1j * a[512:]) * zeta_pow(j)
    want = np.array([np.sum(u * np.exp(2j * np.pi * j * k / 512)) for k in range(512)])
    got = full_from_halves(a)
    print("forward (two halves) max err", np.abs(got - want).max())
    assert np.abs(got - want).max() < 1e-6
    # product: Y = F(d) F(k) pointwise, then S / D and the two inverse halves
    worst = 0.0
    for _ in range(10):
        d = rng.integers(-512, 512, (6, N)).astype(float)
        kk = rng.integers(-(1 << 23), 1 << 23, (6, N)).astype(float)
        Y = sum(full_from_halves(d[p]) * full_from_halves(kk[p]) for p in range(6))
        S = [None] * 4
        D = [None] * 4
        for r in range(4):
            k = slot_freq(r, LANES)
            S[r] = Y[k] + Y[k + 256]
            D[r] = (Y[k] - Y[k + 256]) * np.conj(zeta_pow(4 * k))
        out = np.zeros(N)
        for h, v in ((0, S), (1, D)):
            w = inverse_half(v, h)
            for r in range(4):
                m = LANES + 64 * r
                out[2 * m + h] = w[r].real
                out[2 * m + h + 512] = w[r].imag
        exact = np.zeros(N, dtype=object)
        for p in range(6):
            full = np.convolve(d[p].astype(np.int64).astype(object), kk[p].astype(np.int64).astype(object))
            full = np.concatenate([full, np.zeros(2 * N - len(full), dtype=object)])
            exact += full[:N] - full[N:2 * N]
        assert np.all(np.rint(out) == exact.astype(float))
        worst = max(worst, np.abs(out - exact.astype(float)).max())
    print("largest distance of a limb sum from its integer: 2^%.1f" % np.log2(worst))


if __name__ == "__main__":
    main()
