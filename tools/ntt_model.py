#!/usr/bin/env python3
"""Index/twiddle model of the wave-level 1024-point negacyclic NTT used by the HIP kernels
(csrc/ntt_wave.hpp): 16 registers x 64 lanes, passes 16 (regs) -> 16 (regs, after an LDS
transpose) -> 4 (regs, after a second transpose inside each group of 4 lanes).  Pure-Python exact arithmetic; validates the decomposition
against the definition A[k] = sum_n a[n] psi^(n(2k+1)) and prints the root to hard-code."""
import random

Q = 0xFFFFFFFF00000001
N = 1024


def find_psi():
    psi0 = pow(7, (Q - 1) // (2 * N), Q)
    # want psi^32 == 8 (order-64 root as a shift) => psi^64 == 64
    w = pow(psi0, 32, Q)
    for t in range(1, 2 * N, 2):
        if pow(w, t, Q) == 8:
            return pow(psi0, t, Q)
    raise SystemExit("no root")


PSI = find_psi()
assert pow(PSI, 32, Q) == 8 and pow(PSI, 64, Q) == 64 and pow(PSI, N, Q) == Q - 1


def dft(xs, root):
    n = len(xs)
    return [sum(xs[j] * pow(root, j * k, Q) for j in range(n)) % Q for k in range(n)]


def fwd(a):
    """returns out[thread][reg] with the frequency index each slot holds"""
    # P1: lane l, reg j holds a[l + 64 j]; twist by rho^j = 2^(6j); 16-pt DFT root 2^12 -> reg k1
    Y = [[0] * 16 for _ in range(64)]
    for l in range(64):
        x = [a[l + 64 * j] * pow(2, 6 * j, Q) % Q for j in range(16)]
        Y[l] = dft(x, pow(2, 12, Q))
    # general twiddle W1[l][k1] = psi^(l (2 k1 + 1))
    Z = [[Y[l][k1] * pow(PSI, l * (2 * k1 + 1), Q) % Q for k1 in range(16)] for l in range(64)]
    # transpose: thread (k1, t) = 4*k1 + t holds reg u = Z[t + 4u][k1]
    # P2: 16-pt DFT over u, root 2^12 -> reg v ; twiddle 8^(t v)
    V = [[0] * 16 for _ in range(64)]
    for k1 in range(16):
        for t in range(4):
            x = [Z[t + 4 * u][k1] for u in range(16)]
            X = dft(x, pow(2, 12, Q))
            V[4 * k1 + t] = [X[v] * pow(8, t * v, Q) % Q for v in range(16)]
    # T2 + P3: inside each group of 4 lanes (same k1) transpose so that lane (k1, g) holds register 4*vl + tt =
    # V[k1][tt][v = 4*vl + g]; then four 4-point DFTs over tt in registers (root 2^48): register 4*vl + s
    out = [[0] * 16 for _ in range(64)]
    freq = [[0] * 16 for _ in range(64)]
    for k1 in range(16):
        for g in range(4):
            for vl in range(4):
                v = 4 * vl + g
                x = [V[4 * k1 + tt][v] for tt in range(4)]
                X = dft(x, pow(2, 48, Q))
                for s in range(4):
                    out[4 * k1 + g][4 * vl + s] = X[s]
                    freq[4 * k1 + g][4 * vl + s] = k1 + 16 * v + 256 * s
    return out, freq


def main():
    random.seed(1)
    a = [random.randrange(Q) for _ in range(N)]
    out, freq = fwd(a)
    for th in (0, 1, 5, 63):
        for v in (0, 3, 15):
            k = freq[th][v]
            want = sum(a[n] * pow(PSI, n * (2 * k + 1), Q) for n in range(N)) % Q
            assert want == out[th][v], (th, v)
    assert sorted(f for r in freq for f in r) == list(range(N))
    print("PSI =", hex(PSI), " PSI^-1 =", hex(pow(PSI, Q - 2, Q)), " N^-1 =", hex(pow(N, Q - 2, Q)))
    print("model OK")


if __name__ == "__main__":
    main()
