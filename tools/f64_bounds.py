#!/usr/bin/env python3
"""Magnitude bounds (in units of p) through the lazy f64 NTT pipeline of ntt_wave_f64.hpp, to place the few
explicit reductions.  Every value must stay an exactly representable integer: |v| < 2^53 = 16 p (p ~ 2^49).
mul(a, b) with |b| <= p/2 returns |r| <= (0.5 + a/8) p  (quotient estimate off by <= 3 a 2^-5, low product word
<= a 2^-5);  red(a) returns <= 0.5 p for a < 4, else 0.5 + a 2^-4... (quotient error a * 2^-52 * 2^49 / p)."""
import itertools, sys

LIM = 15.9

def M(a):
    assert a < LIM, a
    return 0.5 + a / 8.0

def R(a):
    assert a < LIM, a
    return 0.5 + a / 16.0 if a >= 4 else 0.5

def br4(r):
    return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3)

def dft16(x, reds):
    """x: 16 bounds; reds: set of (stage, index) positions reduced right after that stage (stage 1..4).
    Mirrors dft16<>: stage s has half = 8 >> (s-1); the difference outputs with i != 0 are multiplied."""
    x = list(x)
    half = 8
    for stage in (1, 2, 3, 4):
        for b in range(0, 16, 2 * half):
            for i in range(half):
                u = x[b + i] + x[b + i + half]
                assert u < LIM, (stage, b, i, u)
                x[b + i] = u
                x[b + i + half] = u if (i == 0 or stage == 4) else M(u)
        for idx in range(16):
            if (stage, idx) in reds:
                x[idx] = R(x[idx])
        half //= 2
    return [x[br4(r)] for r in range(16)]

def forward(reds1, reds2, red_x0=True, verbose=False):
    x = [0.001] + [0.5] * 15                       # digits (tiny) after the psi^(64 j) twist
    x = dft16(x, reds1)
    a1 = max(x)
    x = [M(v) for v in x]                          # W1
    x = dft16(x, reds2)                            # (the LDS transpose permutes lanes, not bounds)
    a2 = max(x)
    x = [R(x[0]) if red_x0 else x[0]] + [M(v) for v in x[1:]]   # W2 (v = 0 is not multiplied)
    # T2 then four DFT4: each output is a sum of 4 values (one of them multiplied by psi^512)
    m = max(x)
    o1 = M(2 * m)
    d4 = max(2 * m + 2 * m, 2 * m + o1)
    assert d4 < LIM, d4
    if verbose:
        print(f"  forward: dft16#1 out {a1:.2f}  dft16#2 out {a2:.2f}  after W2 {m:.2f}  eval {d4:.2f}")
    return d4

def mac_and_inverse(ev, reds3, reds4, verbose=False):
    prod = M(ev)
    s = 3 * prod                                   # three levels, lazily summed
    both = 2 * s                                   # + the partner's partial
    assert both < LIM, both
    x = [R(both)] * 16
    m = max(x)
    o1 = M(2 * m)
    d4 = max(4 * m, 2 * m + o1)
    x = [R(d4)] + [M(d4)] * 15                     # W2I
    x = dft16(x, reds3)
    a3 = max(x)
    x = [M(v) for v in x]                          # W1I
    x = dft16(x, reds4)
    a4 = max(x)
    out = max(x[0], max(M(v) for v in x[1:]))      # psi^(-64 j) twist, j = 0 untouched
    tot = out + 0.5                                # + accumulator (reduced)
    assert tot < LIM, tot
    if verbose:
        print(f"  mac: product {prod:.2f} sum {both:.2f};  inverse: dft16#3 out {a3:.2f} dft16#4 out {a4:.2f} final {out:.2f}")
    return tot

CUR = {(2, i) for i in (0, 1, 2, 3, 4, 8, 9, 10, 11, 12)}   # what dft16<> does today: sums + the untwiddled difference

def search(pipeline_fn, name):
    """greedy: smallest reduction sets (by count) for two consecutive dft16 that keep everything < LIM"""
    cands = [(s, i) for s in (1, 2, 3) for i in range(16)]
    best = None
    for n1 in range(0, 4):
        for n2 in range(0, 4):
            if best and n1 + n2 >= best[0]:
                continue
            for r1 in itertools.combinations(cands, n1):
                ok = False
                for r2 in itertools.combinations(cands, n2):
                    try:
                        v = pipeline_fn(set(r1), set(r2))
                    except AssertionError:
                        continue
                    best = (n1 + n2, r1, r2, v)
                    ok = True
                    break
                if ok:
                    break
    print(name, best)
    return best

if __name__ == "__main__":
    print("current placement:")
    ev = forward(CUR, CUR, verbose=True)
    mac_and_inverse(ev, CUR, CUR, verbose=True)
    bf = search(lambda a, b: forward(a, b), "forward")
    ev = forward(set(bf[1]), set(bf[2]), verbose=True)
    bi = search(lambda a, b: mac_and_inverse(ev, a, b), "inverse")
    mac_and_inverse(ev, set(bi[1]), set(bi[2]), verbose=True)
