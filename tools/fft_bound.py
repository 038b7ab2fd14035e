#!/usr/bin/env python3
"""A-priori bound on the rounding error of the limb sums the 2^64-torus kernels compute through a floating-point transform
(csrc/fft_wave_f64.hpp / fft_half_f64.hpp at N = 1024, csrc/fft_quarter_f64.hpp at N = 2048), to be compared with 1/2: below
it, rounding the inverse transform to the nearest integer returns the exact integer sum whatever the order of the
floating-point operations.

Percival (2003), error of a weighted (right-angle) FFT product z = x * y of length 2^n computed in floating point:

    |z' - z|_inf  <  ||x|| ||y|| ((1 + e)^(3n) (1 + e sqrt 5)^(3n + 1) (1 + b)^(3n) - 1)

e = 2^-53 (f64 unit round-off), b = largest error of a table entry (tables are rounded from long double: b <= e).  n is the
number of butterfly stages of the COMPLEX transform (the real polynomial of N coefficients is folded into N / 2 complex
points: n = log2 N - 1); `extra` counts additional full multiplication stages a particular split adds (the separate twist
of the folded form; the W_h twiddle of the quarter split is the twiddle of its radix-4 stage and is NOT extra).

A limb sum is 2 l products of a digit polynomial (|d| <= Bg / 2) with a balanced limb polynomial (|k| <= 2^(bits - 1)):
||d|| <= sqrt(N) Bg / 2, ||k|| <= sqrt(N) 2^(bits - 1); the errors of the 2 l products add."""
import math
import sys

E = 2.0 ** -53


def percival_factor(n, beta=E):
    # (1 + 2^-53 is not representable: evaluate through log1p / expm1)
    return math.expm1(3 * n * math.log1p(E) + (3 * n + 1) * math.log1p(E * math.sqrt(5)) + 3 * n * math.log1p(beta))


def limb_sum_bound(log_n_poly, levels, base_log, limb_bits, extra=1):
    """bound on |computed - exact| of one limb sum; extra = multiplication stages counted on top of the log2(N) - 1 butterflies"""
    n_poly = 1 << log_n_poly
    norm = n_poly * 2.0 ** (base_log - 1) * 2.0 ** (limb_bits - 1)      # ||d|| ||k||, both at their largest in every coefficient
    per_product = norm * percival_factor(log_n_poly - 1 + extra)
    return 2 * levels * per_product, per_product


SETS = {
    "north_star_torus64 (N 1024, l 3, Bg 2^10, 24-bit limbs)": (10, 3, 10, 24),
    "secure128_torus   (N 2048, l 3, Bg 2^10, 23-bit limbs)": (11, 3, 10, 23),
    "N 2048 with 24-bit limbs (NOT used: bound fails)": (11, 3, 10, 24),
    "N 4096 torus sets (l 3, Bg 2^10, 22-bit limbs)": (12, 3, 10, 22),
    "N 4096 with 23-bit limbs (NOT used: bound fails)": (12, 3, 10, 23),
}

if __name__ == "__main__":
    for name, (lg, l, bg, lb) in SETS.items():
        total, per = limb_sum_bound(lg, l, bg, lb)
        total0, _ = limb_sum_bound(lg, l, bg, lb, extra=0)
        print(f"{name}: per product {per:.4f}, limb sum {total:.3f} (twist counted as a stage; {total0:.3f} without)  "
              f"{'< 1/2 ok' if total < 0.5 else '>= 1/2 NOT certified'}")
    sys.exit(0)
