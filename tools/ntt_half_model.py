#!/usr/bin/env python3
"""Index/twiddle model of the TWO-WAVE 1024-point negacyclic NTT of the latency kernel (csrc/ntt_half_f64.hpp), in
exact arithmetic over q = 2^49 - 720895.

A polynomial a is split by parity, a_e[m] = a[2m], a_o[m] = a[2m+1]; one wavefront transforms each half with a
512-point negacyclic NTT (root phi = psi^2), 8 registers x 64 lanes, three register DFT8 passes with two LDS
transposes between them:
    lane l, reg j           : b[l + 64 j]
    P1  twist zeta^j, DFT8 over j (zeta = phi^64)      -> reg k1
    W1  * phi^((2 k1 + 1) l)
    T1  lane (k1, l0) = 8 k1 + l0, reg l1              <- value of lane l0 + 8 l1, reg k1
    P2  DFT8 over l1                                   -> reg k2a
    W2  * w64^(l0 k2a)      (w64 = phi^16)
    T2  lane (k1, k2a), reg l0                         <- value of lane (k1, l0), reg k2a
    P3  DFT8 over l0                                   -> reg k2b
    slot p = 64 reg + lane holds B[kk],  kk = k1 + 8 k2a + 64 k2b.
Combination (done where the values are consumed, not by an extra pass):
    A[kk] = E[kk] + T[kk] O[kk],   A[kk + 512] = E[kk] - T[kk] O[kk],   T[kk] = psi^(2 kk + 1);
the odd-half wavefront stores O' = T * O.  The inverse runs the same steps backwards from S = A_lo + A_hi (even
half) and (A_lo - A_hi) / T (odd half) with 1/1024 folded into the W1 table."""
import random

Q = 562949952700417
N = 1024
GEN = 5
PSI = pow(GEN, (Q - 1) // (2 * N), Q)
assert pow(PSI, N, Q) == Q - 1
PHI = PSI * PSI % Q
ZETA = pow(PHI, 64, Q)
W8 = pow(PHI, 128, Q)
W64 = pow(PHI, 16, Q)


def inv(x):
    return pow(x, Q - 2, Q)


def dft(xs, root):
    n = len(xs)
    return [sum(xs[j] * pow(root, j * k, Q) for j in range(n)) % Q for k in range(n)]


def kk_of(lane, reg):
    k1, k2a = lane >> 3, lane & 7
    return k1 + 8 * k2a + 64 * reg


def half_forward(b):
    """b: 512 coefficients -> out[lane][reg] = B[kk_of(lane, reg)]"""
    C = [[0] * 8 for _ in range(64)]
    for l in range(64):
        x = [b[l + 64 * j] * pow(ZETA, j, Q) % Q for j in range(8)]
        X = dft(x, W8)
        C[l] = [X[k1] * pow(PHI, (2 * k1 + 1) * l, Q) % Q for k1 in range(8)]
    D = [[0] * 8 for _ in range(64)]
    for k1 in range(8):
        for l0 in range(8):
            x = [C[l0 + 8 * l1][k1] for l1 in range(8)]            # T1
            X = dft(x, W8)
            D[8 * k1 + l0] = [X[k2a] * pow(W64, l0 * k2a, Q) % Q for k2a in range(8)]
    out = [[0] * 8 for _ in range(64)]
    for k1 in range(8):
        for k2a in range(8):
            x = [D[8 * k1 + l0][k2a] for l0 in range(8)]           # T2
            out[8 * k1 + k2a] = dft(x, W8)
    return out


def half_inverse(V):
    """V[lane][reg] in the forward's output layout -> 512 coefficients, scaled by 1/512"""
    D = [[0] * 8 for _ in range(64)]
    for k1 in range(8):
        for k2a in range(8):
            X = dft(V[8 * k1 + k2a], inv(W8))                      # over k2b -> l0
            for l0 in range(8):
                D[8 * k1 + l0][k2a] = X[l0] * pow(inv(W64), l0 * k2a, Q) % Q
    C = [[0] * 8 for _ in range(64)]
    for k1 in range(8):
        for l0 in range(8):
            X = dft(D[8 * k1 + l0], inv(W8))                       # over k2a -> l1
            for l1 in range(8):
                l = l0 + 8 * l1
                C[l][k1] = X[l1] * pow(inv(PHI), (2 * k1 + 1) * l, Q) % Q * inv(512) % Q
    b = [0] * 512
    for l in range(64):
        X = dft(C[l], inv(W8))                                     # over k1 -> j
        for j in range(8):
            b[l + 64 * j] = X[j] * pow(inv(ZETA), j, Q) % Q
    return b


def main():
    rng = random.Random(7)
    a = [rng.randrange(Q) for _ in range(N)]
    want = [sum(a[m] * pow(PSI, (2 * k + 1) * m, Q) for m in range(N)) % Q for k in range(0, N, 37)]
    E = half_forward(a[0::2])
    O = half_forward(a[1::2])
    # definition of the half transform
    b = a[0::2]
    for lane, reg in ((0, 0), (5, 3), (63, 7), (17, 2)):
        kk = kk_of(lane, reg)
        assert E[lane][reg] == sum(b[m] * pow(PHI, (2 * kk + 1) * m, Q) for m in range(512)) % Q
    # combination
    A = {}
    for lane in range(64):
        for reg in range(8):
            kk = kk_of(lane, reg)
            t = pow(PSI, 2 * kk + 1, Q) * O[lane][reg] % Q
            A[kk] = (E[lane][reg] + t) % Q
            A[kk + 512] = (E[lane][reg] - t) % Q
    assert [A[k] for k in range(0, N, 37)] == want
    # inverse: S = A_lo + A_hi -> even half * 2 ; (A_lo - A_hi) / T -> odd half * 2
    S = [[(A[kk_of(l, r)] + A[kk_of(l, r) + 512]) % Q for r in range(8)] for l in range(64)]
    Dm = [[(A[kk_of(l, r)] - A[kk_of(l, r) + 512]) * inv(pow(PSI, 2 * kk_of(l, r) + 1, Q)) % Q for r in range(8)] for l in range(64)]
    half = inv(2)
    assert [v * half % Q for v in half_inverse(S)] == a[0::2]
    assert [v * half % Q for v in half_inverse(Dm)] == a[1::2]
    print("half-transform model OK: psi =", hex(PSI))


if __name__ == "__main__":
    main()
