// Signed gadget decomposition of the 49-bit field in f64 (shared by bmi_kernels_f64.hip and bmi_kernels_f64u.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dec49 {

// Decomposition rule of every modulus of this library (and of the oracle): the centred coefficient is rounded HALF UP to its
// top l * Bg bits, r = floor(v / 2^shift + 1/2), and r is split into balanced signed digits d in [-Bg/2, Bg/2) from the
// least significant one up, r <- floor(r / Bg + 1/2), the top digit taking the last carry - the closest-representative
// signed decomposition of the TFHE literature.  In f64 every step is exact: one v_fma_f64 and one v_floor_f64.
__device__ __forceinline__ double round_half_up(double x, double scale) { return __builtin_floor(__builtin_fma(x, scale, 0.5)); }

// Rounded value r of a centred coefficient v (l = 3, base 2^15: 45 of the 49 bits are kept, r = round_half_up(v, 2^-4)); the
// three signed digits recovered from r.
__device__ __forceinline__ double digit_of(double r, int lev) {
    const double r1 = round_half_up(r, 0x1p-15);
    if (lev == 2) return __builtin_fma(-32768.0, r1, r);
    const double r2 = round_half_up(r1, 0x1p-15);
    if (lev == 1) return __builtin_fma(-32768.0, r2, r1);
    return r2;
}

// The same for any (levels L, base 2^BG) with L * BG <= 48: SC rounds a centred coefficient to its top L * BG bits.
template <int L, int BG>
struct Dec {
    static constexpr double SC = 1.0 / (double)(1ull << (49 - L * BG));
    static constexpr double B = (double)(1ull << BG);
    static constexpr double BINV = 1.0 / (double)(1ull << BG);
    static_assert(L >= 1 && L <= 3 && L * BG <= 48, "decomposition must fit the 49-bit field");
    // digit `lev` (0 = most significant) of the rounded value r
    static __device__ __forceinline__ double digit(double r, int lev) {
#pragma unroll
        for (int t = L - 1; t > 0; t--) {
            const double rn = round_half_up(r, BINV);
            if (t == lev) return __builtin_fma(-B, rn, r);
            r = rn;
        }
        return r;
    }
    // peels the least significant remaining digit off r
    static __device__ __forceinline__ double peel(double &r) {
        const double rn = round_half_up(r, BINV);
        const double d = __builtin_fma(-B, rn, r);
        r = rn;
        return d;
    }
};

}  // namespace dec49
