// Shared by the 2^64-torus kernels (bmi_kernels_t64.hip, bmi_kernels_t64u.hip) and the host (bmi_host.cpp): the limb
// schemes of the bootstrap key and the word <-> exact-double conversions.
//
// No transform exists mod 2^64, so a torus external product is computed EXACTLY over the integers: the digit polynomials
// (|d| <= 2^(Bg-1)) are transformed mod p = 2^49 - 720895 and multiplied with LIMBS balanced limb polynomials of every
// key word (read as a signed integer),  k = 2^PRE sum_j k_j 2^(BITS j);  per limb the sum over the 2 l N digit x limb terms
// is an integer below p / 2, so its centred residue mod p IS that integer; the limb results are recombined with shifts
// mod 2^64.  The scheme follows from the PRECISION the key is stored at (bmi_set_bsk_precision):
//
//   64 bits  3 limbs of 22 bits            the exact key; digits up to 2^14 (Bg <= 2^15): 2 l N 2^14 2^21 = 2^47.6 < p/2
//   48 bits  2 limbs of 24 bits, PRE = 16  key words rounded (half up, as signed integers) to multiples of 2^16; digits up
//                                          to 2^9 (Bg <= 2^10) leave room for the factor 6 of the UNROLLED step (three keys,
//                                          each product scaled by X^c - 1): 6 * 2 l N 2^9 2^23 = 2^47.2 < p/2.  The default
//                                          of the torus set (Bg = 2^10): 2/3 of the work of the exact key, and the rounding
//                                          error (2^15.x per word, summed over the GLWE key's set bits: 2^18.7 per row) stays
//                                          under the key noise 2^20 - output noise 2^-22.6 against 2^-19.85 of (Bg 2^15, exact)
//   42 bits  2 limbs of 21 bits, PRE = 22  round 2's throughput option at Bg = 2^15 (effective key noise 2^-39.3)
//   44 bits  2 limbs of 22 bits, PRE = 20  N = 4096 (Bg = 2^10, l = 3) through the floating-point transform only (bmi_kernels_t64q.hip,
//                                          fft_eighth_f64.hpp): the a-priori error bound of the 2,048-point transform is 0.45 at this width
//   46 bits  2 limbs of 23 bits, PRE = 18  N = 2048 (the secure128_torus set, Bg = 2^10, l = 3) through the floating-point
//                                          transform only (bmi_kernels_t64w.hip): a limb sum 2 l N 2^9 2^22 = 2^44.6 keeps the
//                                          a-priori error bound of the 1,024-point transform below 1/2 (fft_quarter_f64.hpp)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define T64_HD __host__ __device__ __forceinline__
#else
#define T64_HD inline
#endif

namespace t64 {

typedef uint64_t u64;
typedef int64_t i64;

template <int PREC> struct Scheme;
template <> struct Scheme<64> { static constexpr int LIMBS = 3, BITS = 22, PRE = 0; };
template <> struct Scheme<48> { static constexpr int LIMBS = 2, BITS = 24, PRE = 16; };
template <> struct Scheme<42> { static constexpr int LIMBS = 2, BITS = 21, PRE = 22; };
template <> struct Scheme<46> { static constexpr int LIMBS = 2, BITS = 23, PRE = 18; };
template <> struct Scheme<44> { static constexpr int LIMBS = 2, BITS = 22, PRE = 20; };

T64_HD bool precision_ok(int prec) { return prec == 64 || prec == 48 || prec == 46 || prec == 44 || prec == 42; }
T64_HD int limbs_of(int prec) { return prec == 64 ? 3 : 2; }
T64_HD int limb_bits(int prec) { return prec == 64 ? 22 : prec / 2; }
T64_HD int limb_pre(int prec) { return 64 - prec; }
// largest bootstrap base log a precision admits: 2 l N 2^(b-1) 2^(BITS-1) < p/2 at l = 3, N = 1024 means b + BITS <= 37;
// the unrolled step needs b + BITS <= 34
T64_HD int max_base_log(int prec, bool unrolled) { return (unrolled ? 34 : 37) - limb_bits(prec); }

// The key word stored at `prec` bits of precision: rounded half up (as a signed integer) to a multiple of 2^(64 - prec);
// unsigned arithmetic, the wrap at the top of the range is the torus's own.
T64_HD u64 round_key_word(u64 w, int prec) {
    const int drop = 64 - prec;
    if (drop == 0) return w;
    return ((w + ((u64)1 << (drop - 1))) >> drop) << drop;
}

// balanced limb j of a signed 64-bit word at precision `prec`: limb_j in [-2^(BITS-1), 2^(BITS-1)), the last one takes the rest
T64_HD i64 limb_of(i64 k, int j, int prec) {
    const int bits = limb_bits(prec), limbs = limbs_of(prec);
    const i64 B = (i64)1 << bits, H = B >> 1;
    k >>= limb_pre(prec);
    for (int t = 0; t < j; t++) {
        const i64 d = ((k + H) & (B - 1)) - H;
        k = (k - d) >> bits;
    }
    if (j == limbs - 1) return k;
    return ((k + H) & (B - 1)) - H;
}

#if defined(__HIPCC__)
// round(a * 2N / 2^64) mod 2N, ties up
template <int LOG_2N>
__device__ __forceinline__ uint32_t modswitch(u64 a) {
    return (uint32_t)(((a >> (63 - LOG_2N)) + 1) >> 1) & ((1u << LOG_2N) - 1);
}
// exact integer |v| < 2^52 held in a double -> two's complement 64-bit word
__device__ __forceinline__ u64 f64_to_word(double v) {
    const double hi = __builtin_floor(v * 0x1p-32);
    const double lo = __builtin_fma(-0x1p32, hi, v);          // in [0, 2^32)
    return ((u64)(uint32_t)(int32_t)hi << 32) | (u64)(uint32_t)lo;
}
// signed integer |t| < 2^52 held in an int64 -> double (exact)
__device__ __forceinline__ double word_to_f64(i64 t) {
    return __builtin_fma((double)(int32_t)(t >> 32), 0x1p32, (double)(uint32_t)t);
}
// The oracle's decomposition rule on a torus word (oracle/tfhe_oracle.c ora_decompose): the word as a signed integer, rounded
// half up to its top L * BG bits -> that rounded value as an exact double (the balanced digits are peeled off it in f64).
template <int L, int BG>
__device__ __forceinline__ double rounded_top(u64 v) {
    const i64 t = (i64)v >> (64 - L * BG - 1);                                   // L BG + 1 signed bits
    if constexpr (L * BG + 1 <= 32) return __builtin_floor(__builtin_fma((double)(int32_t)t, 0.5, 0.5));
    else return __builtin_floor(__builtin_fma(word_to_f64(t), 0.5, 0.5));
}
#endif

}  // namespace t64
