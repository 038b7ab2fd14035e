// N = 2048 on the 2^64 torus: the folded 1,024-point complex transform of a REAL polynomial of 2,048 coefficients, split over
// FOUR wavefronts by the folded index mod 4 (bmi_kernels_t64w.hip; model of the index algebra: tools/fft_quarter_model.py).
//
//   u_j = (c_j + i c_{j+1024}) zeta^j,   zeta = exp(i pi / 2048),   A_k = sum_{j < 1024} u_j omega^(jk),   omega = zeta^4
//
// (A_k = c(zeta^(4k+1)): the values of c at 1,024 roots of X^2048 + 1; the other 1,024 are their conjugates, c is real.)
// Quarter h owns the 256 points u_{4m+h}.  Its twist zeta^(4m) zeta^h is the twist exp(i pi 2m / 1024) of the EVEN half of the
// N = 1024 split (fft_half_f64.hpp, H = 0) times the constant zeta^h, so a wavefront runs ffth::forward_half<0> unchanged on
// re[r] = c[4 (lane + 64 r) + h], im[r] = c[4 (lane + 64 r) + h + 1024] and multiplies slot p (frequency kappa = slot_freq) by
// W_h[p] = zeta^(h (4 kappa + 1)):
//
//   Q'_h[kappa] = W_h[kappa] half0(c_h)[kappa],       A_{kappa + 256 t} = sum_h i^(h t) Q'_h[kappa]            (omega^256 = i)
//
// - a radix-4 butterfly over the four quarters, taken where the products are (phase B of the kernel).  The inverse runs
// backwards: S_h[kappa] = conj(W_h[kappa]) sum_t i^(-h t) Y_{kappa + 256 t}, then ffth::inverse_half<0>, which carries 1/512;
// the missing factor 1/2 of 1/1024 is folded into the key copy (exact: a power of two).
//
// Why rounding the inverse to the nearest integer is exact here (the a-priori bound, re-derived for this length - the measured
// distance is ~2^-12, bmi_fft_margin_host): a limb sum is 2 l = 6 products of a digit polynomial (|d| <= 2^9, 2,048
// coefficients: ||d|| <= 2^14.5) with a 23-bit balanced key limb polynomial (|k| <= 2^22: ||k|| <= 2^27.5).  Percival's bound
// for a weighted floating-point FFT product of length 2^n: ||d|| ||k|| ((1 + e)^(3n) (1 + e sqrt 5)^(3n + 1) (1 + b)^(3n) - 1)
// with e = 2^-53, b = e (tables rounded from long double).  The quarter split is a 10-stage transform (8 stages of the
// 256-point halves + the radix-4 butterfly, W_h being the butterfly's twiddle; the twist rides in the first table); taking
// n = 11 to count the separate twist multiplication of the folded form as a stage of its own: 2^42 x 1.57e-14 = 0.069 per
// product, 0.41 < 1/2 for the six of a limb sum (tools/fft_bound.py prints both sets).  24-bit limbs would give 0.83: hence 46
// bits of key precision at this length, not 48.
#pragma once
#include "fft_half_f64.hpp"

namespace fftq {

using ffth::C;
using ffth::cmul;
using ffth::slot_freq;
using ffth::static_for;

constexpr int N = 2048;
constexpr int QUARTER = 256;   // complex points (= slots) per quarter
// Tables, as doubles (complex = (re, im) pairs).  ffth::forward_half<0> / inverse_half<0> read the H = 0 block of HT_T1 (256
// complex words at 0), HT_T2 (at 1024) and HT_T3 (at 1152); the H = 1 block of HT_T1 ([512, 1024)) and the HT_W block
// ([1184, 1696)) are not read by the H = 0 code: W_1 and W_2 live there, W_3 follows.
constexpr int QT_W1 = 512;
constexpr int QT_W2 = ffth::HT_W;
constexpr int QT_W3 = ffth::HT_WORDS;
constexpr int QT_WORDS = ffth::HT_WORDS + 2 * QUARTER;
__host__ __device__ constexpr int w_offset(int h) { return h == 1 ? QT_W1 : (h == 2 ? QT_W2 : QT_W3); }

// the QT_WORDS doubles of tables (host)
inline void build_tables(double *t) {
    ffth::build_tables(t);   // HT_T1 (both halves), HT_T2, HT_T3, HT_W; the H = 1 half of HT_T1 and HT_W are overwritten below
    auto zeta_pow = [](unsigned e, double *dst) {
        const long double ang = 3.14159265358979323846264338327950288L * (long double)(e % 4096) / 2048.0L;
        dst[0] = (double)cosl(ang);
        dst[1] = (double)sinl(ang);
    };
    for (unsigned h = 1; h < 4; h++)
        for (unsigned reg = 0; reg < 4; reg++)
            for (unsigned lane = 0; lane < 64; lane++)
                zeta_pow(h * (4 * (unsigned)slot_freq((int)reg, (int)lane) + 1), t + w_offset((int)h) + (reg * 64 + lane) * 2);
}

// Forward quarter H: re[r] = c[4 (lane + 64 r) + H], im[r] = c[4 (lane + 64 r) + H + 1024]; v = Q'_H in slot order (slot p = 64 r + lane)
template <int H>
__device__ __forceinline__ void forward_quarter(const double (&re)[4], const double (&im)[4], C (&v)[4], int lane, const double *tw) {
    ffth::forward_half<0>(re, im, v, lane, tw);
    if constexpr (H != 0) {
        const double2 *w = reinterpret_cast<const double2 *>(tw + w_offset(H));
        static_for<0, 4>([&](auto R) {
            const double2 t = w[R * 64 + lane];
            v[R] = cmul<false>(v[R], t.x, t.y);
        });
    }
}
__device__ __forceinline__ void forward_quarter(int h, const double (&re)[4], const double (&im)[4], C (&v)[4], int lane, const double *tw) {
    ffth::forward_half<0>(re, im, v, lane, tw);
    if (h != 0) {   // (uniform over the wavefront)
        const double2 *w = reinterpret_cast<const double2 *>(tw + w_offset(h));
        static_for<0, 4>([&](auto R) {
            const double2 t = w[R * 64 + lane];
            v[R] = cmul<false>(v[R], t.x, t.y);
        });
    }
}

// Inverse quarter: v = S_h in slot order (already multiplied by conj W_h); re[r] / im[r] = coefficients 4 (lane + 64 r) + h and + 1024
__device__ __forceinline__ void inverse_quarter(C (&v)[4], double (&re)[4], double (&im)[4], int lane, const double *tw) {
    ffth::inverse_half<0>(v, re, im, lane, tw);
}

}  // namespace fftq
