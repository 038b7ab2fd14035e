// N = 4096 on the 2^64 torus: the folded 2,048-point complex transform of a REAL polynomial of 4,096 coefficients, split over
// EIGHT wavefronts by the folded index mod 8 (bmi_kernels_t64q.hip; model of the index algebra: tools/fft_eighth_model.py).
//
//   u_j = (c_j + i c_{j+2048}) zeta^j,   zeta = exp(i pi / 4096),   A_k = sum_{j < 2048} u_j omega^(jk),   omega = zeta^4
//
// Eighth h owns the 256 points u_{8m+h}; its twist zeta^(8m) zeta^h is again the twist exp(i pi 2m / 1024) of the EVEN half of the
// N = 1024 split (fft_half_f64.hpp, H = 0) times the constant zeta^h, so a wavefront runs ffth::forward_half<0> unchanged on
// re[r] = c[8 (lane + 64 r) + h], im[r] = c[8 (lane + 64 r) + h + 2048] and multiplies slot p (frequency kappa = slot_freq) by
// W_h[p] = zeta^(h (4 kappa + 1)) = W_1^(h & 1) W_2^(h >> 1 & 1) W_4^(h >> 2)   (three tables in LDS):
//
//   Q'_h[kappa] = W_h[kappa] half0(c_h)[kappa],       A_{kappa + 256 t} = sum_h e8^(h t) Q'_h[kappa],   e8 = exp(2 pi i / 8)
//
// - a radix-8 butterfly (fftw::dft8) taken where the products are.  The inverse runs backwards: S_h[kappa] = conj(W_h[kappa])
// sum_t e8^(-h t) Y_{kappa + 256 t}, then ffth::inverse_half<0> (1/512); the missing factor 1/4 of 1/2048 is folded into the key copy.
//
// Exactness of the rounded limb sums at this length (tools/fft_bound.py): 2 l = 6 products of a digit polynomial (|d| <= 2^9, 4,096
// coefficients) with a 22-bit balanced key limb polynomial: ||d|| ||k|| <= 2^42; Percival's factor at 11 stages + the twist = 1.72e-14:
// 0.076 per product, 0.45 < 1/2 for a limb sum (23-bit limbs: 0.91) - hence 44 bits of key precision at N = 4096.
#pragma once
#include "fft_half_f64.hpp"
#include "fft_wave_f64.hpp"

namespace ffte {

using ffth::C;
using ffth::cmul;
using ffth::slot_freq;
using ffth::static_for;

constexpr int N = 4096;
constexpr int EIGHTH = 256;   // complex points (= slots) per eighth
// Tables (doubles): the H = 0 code of fft_half_f64.hpp reads HT_T1's first block ([0, 512)), HT_T2 (at 1024) and HT_T3 (at 1152);
// W_1 sits in the unused H = 1 block of HT_T1, W_2 in the HT_W block, W_4 follows.
constexpr int ET_W1 = 512;
constexpr int ET_W2 = ffth::HT_W;
constexpr int ET_W4 = ffth::HT_WORDS;
constexpr int ET_WORDS = ffth::HT_WORDS + 2 * EIGHTH;

inline void build_tables(double *t) {
    ffth::build_tables(t);
    auto zeta_pow = [](unsigned e, double *dst) {
        const long double ang = 3.14159265358979323846264338327950288L * (long double)(e % 8192) / 4096.0L;
        dst[0] = (double)cosl(ang);
        dst[1] = (double)sinl(ang);
    };
    const int off[3] = {ET_W1, ET_W2, ET_W4};
    for (unsigned b = 0; b < 3; b++)
        for (unsigned reg = 0; reg < 4; reg++)
            for (unsigned lane = 0; lane < 64; lane++)
                zeta_pow((1u << b) * (4 * (unsigned)slot_freq((int)reg, (int)lane) + 1), t + off[b] + (reg * 64 + lane) * 2);
}

// v[r] (slot 64 r + lane) times W_h (INV: conj W_h), h uniform over the wavefront
template <bool INV>
__device__ __forceinline__ void times_w(C (&v)[4], int h, int lane, const double *tw) {
    if (h & 1) {
        const double2 *w = reinterpret_cast<const double2 *>(tw + ET_W1);
        static_for<0, 4>([&](auto R) { const double2 t = w[R * 64 + lane]; v[R] = cmul<INV>(v[R], t.x, t.y); });
    }
    if (h & 2) {
        const double2 *w = reinterpret_cast<const double2 *>(tw + ET_W2);
        static_for<0, 4>([&](auto R) { const double2 t = w[R * 64 + lane]; v[R] = cmul<INV>(v[R], t.x, t.y); });
    }
    if (h & 4) {
        const double2 *w = reinterpret_cast<const double2 *>(tw + ET_W4);
        static_for<0, 4>([&](auto R) { const double2 t = w[R * 64 + lane]; v[R] = cmul<INV>(v[R], t.x, t.y); });
    }
}

// Forward eighth h: re[r] = c[8 (lane + 64 r) + h], im[r] = c[8 (lane + 64 r) + h + 2048]; v = Q'_h in slot order
__device__ __forceinline__ void forward_eighth(int h, const double (&re)[4], const double (&im)[4], C (&v)[4], int lane, const double *tw) {
    ffth::forward_half<0>(re, im, v, lane, tw);
    times_w<false>(v, h, lane, tw);
}

// Inverse eighth h: v = sum_t e8^(-h t) Y_t in slot order (NOT yet multiplied by conj W_h); re[r] / im[r] = coefficients
// 8 (lane + 64 r) + h and + 2048
__device__ __forceinline__ void inverse_eighth(int h, C (&v)[4], double (&re)[4], double (&im)[4], int lane, const double *tw) {
    times_w<true>(v, h, lane, tw);
    ffth::inverse_half<0>(v, re, im, lane, tw);
}

}  // namespace ffte
