// Wave-level negacyclic transform of a REAL polynomial of 1,024 coefficients through a folded 512-point complex FFT in f64,
// for the EXACT integer products of the 2^64-torus kernels (bmi_kernels_t64f.hip).
//
//   A_k = sum_{j < 512} (a_j + i a_{j+512}) zeta^j omega^(jk),   zeta = exp(i pi / 1024), omega = zeta^4
//
// is the value of a(X) at the root zeta^(4k+1) of X^1024 + 1 (the other 512 roots are the conjugates: a is real), so a
// pointwise product of two such vectors followed by the inverse is the negacyclic product.  Why a floating-point transform may
// carry exact integer arithmetic here: the products summed per limb (6 polynomials of digits |d| <= 2^9 against balanced 24-bit
// key limbs) are integers below 2^45, and the rounding error of this transform on them stays below 2^-11 (measured:
// bmi_fft_margin_host; model: tools/fft_wave_model.py).  A priori (tools/fft_bound.py): the error analysis of floating-point FFT
// products (Percival 2003) bounds one product's error by ||d|| ||k|| ((1 + eps)^(3n) (1 + eps sqrt 5)^(3n + 1) (1 + beta)^(3n) - 1),
// n = the number of butterfly stages.  Both forms used here have n = 9 stages - this 8 x 8 x 8 transform, and the parity split of
// fft_half_f64.hpp (eight stages of the 256-point halves + the E +- O' butterfly, whose twiddle omega_512^k is that stage's own) -
// plus the twist multiplication of the folded form, counted as a tenth stage: ||d|| ||k|| <= 2^14 * 2^28 (digits and limbs at their
// largest magnitudes in every coefficient) times 1.43e-14 = 0.063 per product, 0.38 < 1/2 for the six of a limb sum (0.34 without
// the extra stage), so
// rounding the inverse to the nearest integer returns the exact sum - the result does not depend on the order of the
// floating-point operations, and the oracle's integer arithmetic is the specification.
//
// One wavefront per polynomial, 8 complex points per lane, 512 = 8 x 8 x 8:
//   registers r = j2, lane = j1 = a + 8 b (j = j1 + 64 j2):  constant twist zeta^(64 r), DFT8 over r -> k2,
//   table twiddle zeta^(lane (4 k2 + 1)) (carries the rest of the twist), LDS exchange to lane = a + 8 k2, registers b,
//   DFT8 over b -> d, table twiddle omega_64^(a d), LDS exchange inside groups of eight lanes to lane = d + 8 k2, registers a,
//   DFT8 over a -> c.   Evaluation layout: register c of lane d + 8 k2 holds frequency k = 64 c + 8 d + k2.
// The inverse runs the same passes backwards with conjugated twiddles and the factor 1/512 in its last constants.
// Register convention of a transform's 16 doubles: x[r] = real part, x[r + 8] = imaginary part of complex point r; in
// coefficient form that is x[J] = a[lane + 64 J], J = 0..15 - the layout of the accumulator tiles of the other kernels.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>

#include "lane_transpose.hpp"

namespace fftw {

constexpr int LOG_N = 10;
constexpr int N = 1 << LOG_N;
constexpr int ROWC = 72;                       // complex words per scratch row (64 + 8: rows 8 lanes apart in the banks)
constexpr int SCRATCH_WORDS = 8 * ROWC * 2;    // doubles of LDS per wavefront
// twiddle tables (built on the host from long double cos / sin, staged into LDS by every workgroup), complex = (re, im) pairs
constexpr int TW_T1 = 0;                       // [k2][lane]  zeta^(lane (4 k2 + 1))          512 complex
constexpr int TW_T2 = 1024;                    // [d][a]      omega_64^(a d) = zeta^(32 a d)   64 complex
constexpr int TW_WORDS = 1024 + 128;

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

struct C {
    double r, i;
};
__device__ __forceinline__ C operator+(C a, C b) { return C{a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ C operator-(C a, C b) { return C{a.r - b.r, a.i - b.i}; }
// a * (S i)
template <bool INV>
__device__ __forceinline__ C mul_i(C a) {
    if constexpr (INV) return C{a.i, -a.r};
    else return C{-a.i, a.r};
}
// a * w (INV: a * conj w)
template <bool INV>
__device__ __forceinline__ C cmul(C a, double wr, double wi) {
    if constexpr (INV) return C{__builtin_fma(a.r, wr, a.i * wi), __builtin_fma(a.i, wr, -(a.r * wi))};
    else return C{__builtin_fma(a.r, wr, -(a.i * wi)), __builtin_fma(a.r, wi, a.i * wr)};
}
// a + s h b with h = sqrt(1/2)
__device__ __forceinline__ C fma_h(C a, C b, double sh) { return C{__builtin_fma(sh, b.r, a.r), __builtin_fma(sh, b.i, a.i)}; }

// cos / sin of pi r / 16: the constant part zeta^(64 r) of the twist
__device__ constexpr double TWIST_C[8] = {1.0, 0.98078528040323043, 0.92387953251128674, 0.83146961230254524,
                                          0.70710678118654757, 0.55557023301960229, 0.38268343236508984, 0.19509032201612833};
__device__ constexpr double TWIST_S[8] = {0.0, 0.19509032201612825, 0.38268343236508978, 0.55557023301960218,
                                          0.70710678118654746, 0.83146961230254524, 0.92387953251128674, 0.98078528040323043};

// 8-point DFT over the register array, root exp(+-2 pi i / 8) (INV: minus), natural order in and out; radix-2 decimation in
// frequency, the two factors sqrt(1/2) folded into the last stage's multiply-adds.  52 f64 instructions.
template <bool INV>
__device__ __forceinline__ void dft8(C (&x)[8]) {
    constexpr double s = INV ? -1.0 : 1.0;
    constexpr double h = 0.70710678118654752440;
    const C a0 = x[0] + x[4], d0 = x[0] - x[4];
    const C a1 = x[1] + x[5], t1 = x[1] - x[5];
    const C a2 = x[2] + x[6], t2 = x[2] - x[6];
    const C a3 = x[3] + x[7], t3 = x[3] - x[7];
    const C d1 = C{t1.r - s * t1.i, t1.i + s * t1.r};        // t1 (1 + s i), times h later
    const C d2 = mul_i<INV>(t2);
    const C d3 = C{-t3.r - s * t3.i, s * t3.r - t3.i};       // t3 (-1 + s i), times h later
    const C b0 = a0 + a2, b2 = a0 - a2;
    const C b1 = a1 + a3, b3 = mul_i<INV>(a1 - a3);
    const C e0 = d0 + d2, e2 = d0 - d2;
    const C e1 = d1 + d3, e3 = mul_i<INV>(d1 - d3);
    x[0] = b0 + b1;
    x[4] = b0 - b1;
    x[2] = b2 + b3;
    x[6] = b2 - b3;
    x[1] = fma_h(e0, e1, h);
    x[5] = fma_h(e0, e1, -h);
    x[3] = fma_h(e2, e3, h);
    x[7] = fma_h(e2, e3, -h);
}

#ifndef BMI_FFT_EX1
#define BMI_FFT_EX1 0   // exchange 1 (lane bits 5..3 <-> register index): 0 = through LDS, 1 = v_permlane32/16_swap + DPP
#endif
#ifndef BMI_FFT_EX2
#define BMI_FFT_EX2 0   // exchange 2 (lane bits 2..0 <-> register index): 0 = through LDS, 1 = DPP moves
#endif
// register bit t <-> lane bit LO + t (t = 0, 1, 2) of the 8 complex registers
template <int LO>
__device__ __forceinline__ void transpose8(C (&v)[8], int lane) {
    static_for<0, 4>([&](auto R) {
        lanetr::tr_double<LO + 2>(v[R].r, v[R + 4].r, lane);
        lanetr::tr_double<LO + 2>(v[R].i, v[R + 4].i, lane);
    });
    static_for<0, 4>([&](auto Q) {
        constexpr int r = (Q & 1) + 4 * (Q >> 1);
        lanetr::tr_double<LO + 1>(v[r].r, v[r + 2].r, lane);
        lanetr::tr_double<LO + 1>(v[r].i, v[r + 2].i, lane);
    });
    static_for<0, 4>([&](auto Q) {
        lanetr::tr_double<LO>(v[2 * Q].r, v[2 * Q + 1].r, lane);
        lanetr::tr_double<LO>(v[2 * Q].i, v[2 * Q + 1].i, lane);
    });
}
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// scratch addresses (complex words) of the two exchanges
__device__ __forceinline__ int ex1_row(int lane) { return (lane >> 3) * ROWC + (lane & 7); }          // + 8 b
__device__ __forceinline__ int ex2_base(int lane) { return (lane >> 3) * ROWC; }

// Forward transform.  x[J] = a[lane + 64 J] on entry, evaluation layout on exit (x[c] real, x[c + 8] imaginary part).
// mid() runs before the second DFT8: callers issue global loads there whose latency the rest of the transform hides.
template <class Mid = NoHook>
__device__ __forceinline__ void forward(double (&x)[16], int lane, const double *tw, double *scratch, Mid mid = Mid()) {
    const double2 *tw2 = reinterpret_cast<const double2 *>(tw);
    double2 *sc = reinterpret_cast<double2 *>(scratch);
    double2 w[8];
    static_for<0, 8>([&](auto K) { w[K] = tw2[TW_T1 / 2 + K * 64 + lane]; });
    sched_fence();
    C v[8];
    v[0] = C{x[0], x[8]};
    static_for<1, 8>([&](auto R) { v[R] = cmul<false>(C{x[R], x[R + 8]}, TWIST_C[R], TWIST_S[R]); });
    dft8<false>(v);
    static_for<0, 8>([&](auto K) { v[K] = cmul<false>(v[K], w[K].x, w[K].y); });
    const int r1 = ex1_row(lane), a = lane & 7, base = ex2_base(lane);
    if constexpr (BMI_FFT_EX1) {
        transpose8<3>(v, lane);
    } else {
        wave_sync();
        static_for<0, 8>([&](auto K) { sc[K * ROWC + lane] = double2{v[K].r, v[K].i}; });
        wave_sync();
        static_for<0, 8>([&](auto B) {
            const double2 t = sc[r1 + 8 * B];
            v[B] = C{t.x, t.y};
        });
    }
    static_for<1, 8>([&](auto D) { w[D] = tw2[TW_T2 / 2 + D * 8 + a]; });
    mid();
    sched_fence();
    dft8<false>(v);
    static_for<1, 8>([&](auto D) { v[D] = cmul<false>(v[D], w[D].x, w[D].y); });
    if constexpr (BMI_FFT_EX2) {
        transpose8<0>(v, lane);
    } else {
        wave_sync();
        static_for<0, 8>([&](auto D) { sc[base + D * 8 + ((a + D) & 7)] = double2{v[D].r, v[D].i}; });
        wave_sync();
        static_for<0, 8>([&](auto A) {
            const double2 t = sc[base + a * 8 + ((A + a) & 7)];
            v[A] = C{t.x, t.y};
        });
    }
    dft8<false>(v);
    static_for<0, 8>([&](auto Cc) {
        x[Cc] = v[Cc].r;
        x[Cc + 8] = v[Cc].i;
    });
}

// Inverse transform (includes 1/512): evaluation layout in, x[J] = a[lane + 64 J] out (not yet rounded to integers).
__device__ __forceinline__ void inverse(double (&x)[16], int lane, const double *tw, double *scratch) {
    const double2 *tw2 = reinterpret_cast<const double2 *>(tw);
    double2 *sc = reinterpret_cast<double2 *>(scratch);
    const int r1 = ex1_row(lane), a = lane & 7, base = ex2_base(lane);
    double2 w[8];
    static_for<1, 8>([&](auto D) { w[D] = tw2[TW_T2 / 2 + D * 8 + a]; });
    sched_fence();
    C v[8];
    static_for<0, 8>([&](auto Cc) { v[Cc] = C{x[Cc], x[Cc + 8]}; });
    dft8<true>(v);   // over c -> a
    if constexpr (BMI_FFT_EX2) {
        transpose8<0>(v, lane);
    } else {
        wave_sync();
        static_for<0, 8>([&](auto A) { sc[base + a * 8 + ((A + a) & 7)] = double2{v[A].r, v[A].i}; });
        wave_sync();
        static_for<0, 8>([&](auto D) {
            const double2 t = sc[base + D * 8 + ((a + D) & 7)];
            v[D] = C{t.x, t.y};
        });
    }
    static_for<1, 8>([&](auto D) { v[D] = cmul<true>(v[D], w[D].x, w[D].y); });
    static_for<0, 8>([&](auto K) { w[K] = tw2[TW_T1 / 2 + K * 64 + lane]; });
    sched_fence();
    dft8<true>(v);   // over d -> b
    if constexpr (BMI_FFT_EX1) {
        transpose8<3>(v, lane);
    } else {
        wave_sync();
        static_for<0, 8>([&](auto B) { sc[r1 + 8 * B] = double2{v[B].r, v[B].i}; });
        wave_sync();
        static_for<0, 8>([&](auto K) {
            const double2 t = sc[K * ROWC + lane];
            v[K] = C{t.x, t.y};
        });
    }
    static_for<0, 8>([&](auto K) { v[K] = cmul<true>(v[K], w[K].x, w[K].y); });
    dft8<true>(v);   // over k2 -> r
    x[0] = v[0].r * (1.0 / 512);
    x[8] = v[0].i * (1.0 / 512);
    static_for<1, 8>([&](auto R) {
        const C t = cmul<true>(v[R], TWIST_C[R] * (1.0 / 512), TWIST_S[R] * (1.0 / 512));
        x[R] = t.r;
        x[R + 8] = t.i;
    });
}

// word offset (in doubles) of evaluation value (lane, register c) inside a transform-domain key polynomial: complex words in
// [c][lane] order, so that a wavefront requests one 16-byte word per lane and register
__host__ __device__ __forceinline__ int eval_offset(int lane, int c) { return (c * 64 + lane) * 2; }

}  // namespace fftw
