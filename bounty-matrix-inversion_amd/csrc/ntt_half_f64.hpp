// Two-wave 1024-point negacyclic NTT over q = 2^49 - 720895 for the LATENCY kernel: a polynomial is split by
// parity and each half (512 coefficients) is transformed by its own wavefront, 8 coefficients per lane, so the
// instruction stream on the critical path of one CMUX is about 0.4x that of the one-wave transform of
// ntt_wave_f64.hpp (a single wavefront runs at ~1/3 of the VALU rate: its f64 dependency chains and LDS round trips
// are not covered by anything else).  tools/ntt_half_model.py is the exact-arithmetic index/twiddle model.
//
//   lane l, reg j           : b[l + 64 j]                       (b = a[2m] or a[2m + 1])
//   P1  twist zeta^j, DFT8 over j   (zeta = psi^128)            -> reg k1
//   W1  * psi^(2 (2 k1 + 1) l)                                  (LDS table)
//   T1  lane (k1, l0) = 8 k1 + l0, reg l1   <- lane l0 + 8 l1, reg k1      (LDS tile, rows padded 64 -> 72)
//   P2  DFT8 over l1                                            -> reg k2a
//   W2  * psi^(32 l0 k2a)                                       (LDS table)
//   T2  lane (k1, k2a), reg l0              <- lane (k1, l0), reg k2a      (inside groups of 8 lanes, swizzled)
//   P3  DFT8 over l0                                            -> reg k2b
//   slot p = 64 reg + lane holds B[kk], kk = k1 + 8 k2a + 64 k2b; the odd half is then multiplied by psi^(2 kk + 1).
// The two halves are never combined by a pass of their own: the consumer forms  A[kk] = E + O',  A[kk + 512] = E - O'
// when it reads them, and produces  S = A_lo + A_hi  (even half) and  A_lo - A_hi  (odd half) for the inverse, which
// runs the same steps backwards with 1/1024 folded into its W1 table.  Both LDS transposes are conflict-free.
// Magnitudes (tools/ntt_half_model.py / f64_bounds.py conventions): no explicit reduction is needed inside the
// DFT8s; the largest value anywhere is < 10.3 q against the exact-integer limit 16 q.
#pragma once
#include "ntt_wave_f64.hpp"

namespace ntth {

using nttf::psi_pow;
using nttf::sched_fence;
using nttf::static_for;
using nttf::wave_sync;

constexpr int HALF = 512;
constexpr int HROW = 72;                 // padded tile row (words): 8 rows x 72 -> both transposes conflict-free
constexpr int HSCRATCH = 8 * HROW;       // 576 words >= 512: the tile also takes the finished half transform
// twiddle tables (doubles, centred), built on the host (bmi_host.cpp: build_twiddles_half)
constexpr int HT_W1 = 0;                 // [k1][l]      psi^(2 (2 k1 + 1) l)
constexpr int HT_W2 = 512;               // [k2a][l0]    psi^(32 l0 k2a)
constexpr int HT_T = 576;                // [reg][lane]  psi^(2 kk + 1)
constexpr int HT_W1I = 1088;             // [k1][l]      psi^-(2 (2 k1 + 1) l) / 1024
constexpr int HT_W2I = 1600;             // [k2a][l0]    psi^-(32 l0 k2a)
constexpr int HT_TI = 1664;              // [reg][lane]  psi^-(2 kk + 1)
constexpr int HT_WORDS = 2176;

__host__ __device__ __forceinline__ int kk_of(int lane, int reg) { return (lane >> 3) + 8 * (lane & 7) + 64 * reg; }

constexpr int br3(int r) { return ((r & 1) << 2) | (r & 2) | ((r & 4) >> 2); }

// 8-point DFT over the register array, root psi^256 (INV: its inverse), natural order in and out, lazy
template <bool INV>
__device__ __forceinline__ void dft8(double (&x)[8]) {
    static_for<0, 4>([&](auto I) {
        constexpr int i = I;
        const double u = x[i] + x[i + 4], d = x[i] - x[i + 4];
        x[i] = u;
        if constexpr (i == 0) x[i + 4] = d;
        else x[i + 4] = f49::mul(d, psi_pow<INV, 256 * i>());
    });
    static_for<0, 2>([&](auto B) {
        static_for<0, 2>([&](auto I) {
            constexpr int b = B * 4, i = I;
            const double u = x[b + i] + x[b + i + 2], d = x[b + i] - x[b + i + 2];
            x[b + i] = u;
            if constexpr (i == 0) x[b + i + 2] = d;
            else x[b + i + 2] = f49::mul(d, psi_pow<INV, 512>());
        });
    });
    static_for<0, 4>([&](auto B) {
        constexpr int b = B * 2;
        const double u = x[b] + x[b + 1], d = x[b] - x[b + 1];
        x[b] = u;
        x[b + 1] = d;
    });
    double y[8];
    static_for<0, 8>([&](auto R) { y[br3(R)] = x[R]; });
    static_for<0, 8>([&](auto R) { x[R] = y[R]; });
}

// Forward half transform.  x[j] = b[lane + 64 j] (|.| <= 0.8 q) in; slot layout out (|.| <= 8.1 q; ODD: <= 1.6 q).
template <bool ODD>
__device__ __forceinline__ void forward_half(double (&x)[8], int lane, const double *tw, double *scratch) {
    double wa[8], wb[8];
#if BMI_LAT2_PRIO == 2
    __builtin_amdgcn_s_setprio(2);
#endif
    static_for<0, 8>([&](auto K) { wa[K] = tw[HT_W1 + K * 64 + lane]; });
    sched_fence();
    static_for<1, 8>([&](auto J) { x[J] = f49::mul(x[J], psi_pow<false, 128 * J>()); });
    dft8<false>(x);
    static_for<0, 8>([&](auto K) { x[K] = f49::mul(x[K], wa[K]); });
    wave_sync();
    static_for<0, 8>([&](auto K) { scratch[K * HROW + lane] = x[K]; });
    wave_sync();
    const int k1 = lane >> 3, l0 = lane & 7;
    double *row = scratch + k1 * HROW;
#if BMI_LAT2_PRIO
    __builtin_amdgcn_s_setprio(BMI_LAT2_PRIO == 2 ? 1 : 2);   // a wave that is ahead yields issue slots to the ones sharing its SIMD
#endif
    static_for<0, 8>([&](auto L1) { x[L1] = row[l0 + 8 * L1]; });
    static_for<1, 8>([&](auto K) { wb[K] = tw[HT_W2 + K * 8 + l0]; });
    if constexpr (ODD) static_for<0, 8>([&](auto R) { wa[R] = tw[HT_T + R * 64 + lane]; });
    sched_fence();
    dft8<false>(x);
    x[0] = f49::red(x[0]);
    static_for<1, 8>([&](auto K) { x[K] = f49::mul(x[K], wb[K]); });
    wave_sync();
    static_for<0, 8>([&](auto K) { row[8 * K + ((l0 + K) & 7)] = x[K]; });   // lane (k1, l0), reg k2a = K
    wave_sync();
    static_for<0, 8>([&](auto L0) { x[L0] = row[8 * l0 + ((L0 + l0) & 7)]; });  // this lane is (k1, k2a = lane & 7)
#if BMI_LAT2_PRIO
    __builtin_amdgcn_s_setprio(BMI_LAT2_PRIO == 2 ? 0 : 1);
#endif
    dft8<false>(x);
    if constexpr (ODD) static_for<0, 8>([&](auto R) { x[R] = f49::mul(x[R], wa[R]); });
}

// Inverse half transform: slot layout in (|.| <= 1.1 q; for ODD the raw difference, divided by psi^(2 kk + 1) here),
// x[j] = b[lane + 64 j] out (|.| <= 8.9 q), scaled by 1/1024 (1/512 of the transform and the 1/2 of the split).
template <bool ODD>
__device__ __forceinline__ void inverse_half(double (&x)[8], int lane, const double *tw, double *scratch) {
    const int k1 = lane >> 3, l0 = lane & 7;
    double *row = scratch + k1 * HROW;
    double wa[8], wb[8];
    if constexpr (ODD) {
        static_for<0, 8>([&](auto R) { wa[R] = tw[HT_TI + R * 64 + lane]; });
        static_for<0, 8>([&](auto R) { x[R] = f49::mul(x[R], wa[R]); });
    }
    static_for<1, 8>([&](auto K) { wb[K] = tw[HT_W2I + K * 8 + l0]; });
    sched_fence();
    dft8<true>(x);                                                             // k2b -> l0
    wave_sync();
    static_for<0, 8>([&](auto L0) { row[8 * l0 + ((L0 + l0) & 7)] = x[L0]; });  // this lane is (k1, k2a = lane & 7)
    wave_sync();
    static_for<0, 8>([&](auto K) { x[K] = row[8 * K + ((l0 + K) & 7)]; });     // now lane (k1, l0), reg k2a = K
    static_for<0, 8>([&](auto K) { wa[K] = tw[HT_W1I + K * 64 + lane]; });
    sched_fence();
    x[0] = f49::red(x[0]);
    static_for<1, 8>([&](auto K) { x[K] = f49::mul(x[K], wb[K]); });
    dft8<true>(x);                                                             // k2a -> l1
    wave_sync();
    static_for<0, 8>([&](auto L1) { row[l0 + 8 * L1] = x[L1]; });
    wave_sync();
    static_for<0, 8>([&](auto K) { x[K] = f49::mul(scratch[K * HROW + lane], wa[K]); });
    dft8<true>(x);                                                             // k1 -> j
    static_for<1, 8>([&](auto J) { x[J] = f49::mul(x[J], psi_pow<true, 128 * J>()); });
}

}  // namespace ntth
