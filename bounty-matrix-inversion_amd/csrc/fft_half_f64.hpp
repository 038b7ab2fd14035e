// The folded 512-point complex transform of fft_wave_f64.hpp split over TWO wavefronts by the parity of the folded index, for
// the latency kernel of the 2^64 torus (bmi_kernels_t64f.hip): wavefront h transforms the 256 points u_{2m+h},
//
//   u_j = (a_j + i a_{j+512}) zeta^j,   E_k = sum_m u_{2m} w^(mk),   O_k = sum_m u_{2m+1} w^(mk),   w = exp(2 pi i / 256),
//   F_k = E_k + omega_512^k O_k,   F_{k+256} = E_k - omega_512^k O_k                      (k < 256),
//
// the odd half multiplies its output by omega_512^k, and the pair (E + O', E - O') is formed where the products are taken
// (phase B of the kernel).  The inverse splits by decimation in frequency: S_k = Y_k + Y_{k+256} gives the even points,
// D_k = (Y_k - Y_{k+256}) omega_512^-k the odd ones.
//
// One wavefront, 4 complex points per lane, 256 = 4 x 4 x 4 x 4: four register DFT4s; between them the next two lane bits are
// moved into the register index by 2 x 2 transposes that never touch LDS - v_permlane32_swap / v_permlane16_swap for lane
// bits 5 and 4 (one instruction per 32-bit register pair), DPP moves for bits 3..0 - so a half transform is a straight run of
// vector instructions (~120 f64 + ~130 32-bit) with no LDS round trip inside.  Model of the index algebra: tools/fft_half_model.py.
//
// Layouts.  Coefficient side: register r of lane l holds point m = l + 64 r, i.e. coefficients 2m + h (real part) and
// 2m + h + 512 (imaginary part).  Evaluation side ("slot order"): register rho of lane l holds frequency
// k = 64 rho + 16 (l & 3) + 4 ((l >> 2) & 3) + (l >> 4); slot p = 64 rho + l.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "fft_wave_f64.hpp"
#include "lane_transpose.hpp"

namespace ffth {

using fftw::C;
using fftw::cmul;
using fftw::static_for;

constexpr int N = 1024;
// tables (complex = (re, im) pairs), built on the host from long double cos / sin, staged into LDS by the workgroup
constexpr int HT_T1 = 0;          // [h][k2][lane]   zeta^(h + lane (2 + 8 k2))      512 complex
constexpr int HT_T2 = 1024;       // [kappa0][n0]    omega_64^(n0 kappa0) = zeta^(32 n0 kappa0), n0 = lane & 15    64 complex
constexpr int HT_T3 = 1024 + 128; // [lambda0][l0]   omega_16^(l0 lambda0) = zeta^(128 l0 lambda0), l0 = lane & 3   16 complex
constexpr int HT_W = 1024 + 128 + 32;   // [slot p]  omega_512^k(p) = zeta^(4 k(p))                                256 complex
constexpr int HT_WORDS = HT_W + 512;

__host__ __device__ __forceinline__ int slot_freq(int reg, int lane) { return 64 * reg + 16 * (lane & 3) + 4 * ((lane >> 2) & 3) + (lane >> 4); }

// cos / sin of pi r / 8: the constant part zeta^(128 r) of the twist
__device__ constexpr double TW_C[4] = {1.0, 0.92387953251128674, 0.70710678118654757, 0.38268343236508984};
__device__ constexpr double TW_S[4] = {0.0, 0.38268343236508978, 0.70710678118654746, 0.92387953251128674};

template <bool INV>
__device__ __forceinline__ void dft4(C (&x)[4]) {
    const C a = x[0] + x[2], b = x[0] - x[2];
    const C c = x[1] + x[3], d = fftw::mul_i<INV>(x[1] - x[3]);
    x[0] = a + c;
    x[1] = b + d;
    x[2] = a - c;
    x[3] = b - d;
}

using lanetr::tr_double;
// register bit 1 <-> lane bit HI, register bit 0 <-> lane bit HI - 1
template <int HI>
__device__ __forceinline__ void transpose(C (&v)[4], int lane) {
    tr_double<HI>(v[0].r, v[2].r, lane);
    tr_double<HI>(v[0].i, v[2].i, lane);
    tr_double<HI>(v[1].r, v[3].r, lane);
    tr_double<HI>(v[1].i, v[3].i, lane);
    tr_double<HI - 1>(v[0].r, v[1].r, lane);
    tr_double<HI - 1>(v[0].i, v[1].i, lane);
    tr_double<HI - 1>(v[2].r, v[3].r, lane);
    tr_double<HI - 1>(v[2].i, v[3].i, lane);
}

// Forward half.  re[r] = a[2 (lane + 64 r) + H], im[r] = a[2 (lane + 64 r) + H + 512]; v = slot order out (H = 1: times omega_512^k).
// (each table read is issued one stage ahead of its use and no earlier: the kernel runs 16 wavefronts per CU at 128 registers)
template <int H>
__device__ __forceinline__ void forward_half(const double (&re)[4], const double (&im)[4], C (&v)[4], int lane, const double *tw) {
    const double2 *tw2 = reinterpret_cast<const double2 *>(tw);
    double2 wa[4], wb[4];
    static_for<0, 4>([&](auto K) { wa[K] = tw2[HT_T1 / 2 + (H * 4 + K) * 64 + lane]; });
    v[0] = C{re[0], im[0]};
    static_for<1, 4>([&](auto R) { v[R] = cmul<false>(C{re[R], im[R]}, TW_C[R], TW_S[R]); });
    dft4<false>(v);
    static_for<1, 4>([&](auto K) { wb[K] = tw2[HT_T2 / 2 + K * 16 + (lane & 15)]; });
    static_for<0, 4>([&](auto K) { v[K] = cmul<false>(v[K], wa[K].x, wa[K].y); });
    transpose<5>(v, lane);
    dft4<false>(v);
    static_for<1, 4>([&](auto K) { wa[K] = tw2[HT_T3 / 2 + K * 4 + (lane & 3)]; });
    static_for<1, 4>([&](auto K) { v[K] = cmul<false>(v[K], wb[K].x, wb[K].y); });
    transpose<3>(v, lane);
    dft4<false>(v);
    if constexpr (H == 1) static_for<0, 4>([&](auto R) { wb[R] = tw2[HT_W / 2 + R * 64 + lane]; });
    static_for<1, 4>([&](auto K) { v[K] = cmul<false>(v[K], wa[K].x, wa[K].y); });
    transpose<1>(v, lane);
    dft4<false>(v);
    if constexpr (H == 1) static_for<0, 4>([&](auto R) { v[R] = cmul<false>(v[R], wb[R].x, wb[R].y); });
}

// Inverse half (includes 1/512).  v = S (H = 0) or D (H = 1) in slot order; re[r] / im[r] = coefficients 2 (lane + 64 r) + H and + 512.
template <int H>
__device__ __forceinline__ void inverse_half(C (&v)[4], double (&re)[4], double (&im)[4], int lane, const double *tw) {
    const double2 *tw2 = reinterpret_cast<const double2 *>(tw);
    double2 wa[4], wb[4];
    static_for<1, 4>([&](auto K) { wa[K] = tw2[HT_T3 / 2 + K * 4 + (lane & 3)]; });
    dft4<true>(v);
    transpose<1>(v, lane);
    static_for<1, 4>([&](auto K) { wb[K] = tw2[HT_T2 / 2 + K * 16 + (lane & 15)]; });
    static_for<1, 4>([&](auto K) { v[K] = cmul<true>(v[K], wa[K].x, wa[K].y); });
    dft4<true>(v);
    transpose<3>(v, lane);
    static_for<0, 4>([&](auto K) { wa[K] = tw2[HT_T1 / 2 + (H * 4 + K) * 64 + lane]; });
    static_for<1, 4>([&](auto K) { v[K] = cmul<true>(v[K], wb[K].x, wb[K].y); });
    dft4<true>(v);
    transpose<5>(v, lane);
    static_for<0, 4>([&](auto K) { v[K] = cmul<true>(v[K], wa[K].x, wa[K].y); });
    dft4<true>(v);
    re[0] = v[0].r * (1.0 / 512);
    im[0] = v[0].i * (1.0 / 512);
    static_for<1, 4>([&](auto R) {
        const C t = cmul<true>(v[R], TW_C[R] * (1.0 / 512), TW_S[R] * (1.0 / 512));
        re[R] = t.r;
        im[R] = t.i;
    });
}

// the HT_WORDS doubles of tables (host)
inline void build_tables(double *t) {
    auto zeta_pow = [](unsigned e, double *dst) {
        const long double ang = 3.14159265358979323846264338327950288L * (long double)(e % 2048) / 1024.0L;
        dst[0] = (double)cosl(ang);
        dst[1] = (double)sinl(ang);
    };
    for (unsigned h = 0; h < 2; h++)
        for (unsigned k2 = 0; k2 < 4; k2++)
            for (unsigned lane = 0; lane < 64; lane++) zeta_pow(h + lane * (2 + 8 * k2), t + HT_T1 + ((h * 4 + k2) * 64 + lane) * 2);
    for (unsigned k0 = 0; k0 < 4; k0++)
        for (unsigned n0 = 0; n0 < 16; n0++) zeta_pow(32 * n0 * k0, t + HT_T2 + (k0 * 16 + n0) * 2);
    for (unsigned l0 = 0; l0 < 4; l0++)
        for (unsigned x = 0; x < 4; x++) zeta_pow(128 * x * l0, t + HT_T3 + (l0 * 4 + x) * 2);
    for (unsigned reg = 0; reg < 4; reg++)
        for (unsigned lane = 0; lane < 64; lane++) zeta_pow(4 * (unsigned)slot_freq((int)reg, (int)lane), t + HT_W + (reg * 64 + lane) * 2);
}

}  // namespace ffth
