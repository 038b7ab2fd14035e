// Arithmetic in Z_q, q = 2^64 - 2^32 + 1 (the "Goldilocks" prime), for host and gfx950 device code.
// All values are canonical (in [0, q)) on entry and exit unless a function says otherwise.
// 2^64 = 2^32 - 1 (=: EPS), 2^96 = -1, so 2 has order 192 and the roots of unity of order
// 4/8/16/32/64 are 2^48, 2^24, 2^12, 2^6, 2^3: multiplications by them are shifts (mul_pow2).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

namespace gl {

typedef uint64_t u64;
typedef int64_t i64;

constexpr u64 P = 0xFFFFFFFF00000001ULL;
constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod P

GL_HD u64 add(u64 a, u64 b) {
    u64 s = a + b;
    if (s < a || s >= P) s += EPS;  // s - P (mod 2^64)
    return s;
}
GL_HD u64 sub(u64 a, u64 b) {
    u64 d = a - b;
    if (a < b) d -= EPS;  // d + P (mod 2^64)
    return d;
}
GL_HD u64 neg(u64 a) { return a ? P - a : 0; }

GL_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// x * EPS for a 32-bit x, without a multiplier: (x << 32) - x
// x * EPS for a 32-bit x.  Measured on gfx950: the 64x32 multiply-add (v_mad_u64_u32) is cheaper here than the
// shift/subtract form with its extra carry-dependent instructions.
GL_HD u64 times_eps32(u64 x32) { return x32 * EPS; }

// (hi * 2^64 + lo) mod P, any hi/lo
GL_HD u64 reduce128(u64 hi, u64 lo) {
    u64 hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;  // 2^96 = -1: subtract hh, fix the borrow with +P
    u64 t1 = times_eps32(hl);  // 2^64 = EPS
    u64 r = t0 + t1;
    if (r < t1) r += EPS;
    if (r >= P) r -= P;
    return r;
}
// (hi32 * 2^64 + lo) mod P for hi32 < 2^32 (one fold, no borrow step)
GL_HD u64 reduce96(u64 hi32, u64 lo) {
    u64 t1 = times_eps32(hi32);
    u64 r = lo + t1;
    if (r < t1) r += EPS;
    if (r >= P) r -= P;
    return r;
}
GL_HD u64 mul(u64 a, u64 b) { return reduce128(mulhi64(a, b), a * b); }

// x * 2^S mod P for a compile-time S in [0, 192)
template <int S>
GL_HD u64 mul_pow2(u64 x) {
    static_assert(S >= 0 && S < 192, "shift out of range");
    if constexpr (S == 0) {
        return x;
    } else if constexpr (S <= 32) {
        return reduce96(x >> (64 - S), x << S);
    } else if constexpr (S < 64) {
        return reduce128(x >> (64 - S), x << S);
    } else if constexpr (S == 64) {
        return reduce128(x, 0);
    } else if constexpr (S < 96) {
        // x*2^(S-64) = h*2^64 + l with h < 2^32 ; times 2^64: l*2^64 + h*2^128, 2^128 = -2^32
        constexpr int T = S - 64;
        u64 h = x >> (64 - T), l = x << T;
        return sub(reduce128(l, 0), h << 32);
    } else {
        return neg(mul_pow2<S - 96>(x));
    }
}

// signed small integer -> Z_q
GL_HD u64 from_i64(i64 v) { return v >= 0 ? (u64)v : (u64)v - EPS; }  // |v| < 2^63: P + v = 2^64 + v - EPS

// centred lift to a signed 64-bit integer in (-q/2, q/2]
GL_HD i64 centered(u64 a) { return (i64)(a > (P >> 1) ? a + EPS : a); }  // a - P = a + EPS (mod 2^64)

// round(a * 2^log2N / q) mod 2^log2N, exact (a canonical, log2N <= 16)
GL_HD uint32_t modswitch(u64 a, uint32_t log2N) {
    // t = a*2^log2N + floor(q/2) = hi*2^64 + lo ; floor(t/q) = hi + [hi*EPS + lo >= q]
    u64 hi = a >> (64 - log2N), lo = a << log2N;
    u64 lo2 = lo + (P >> 1);
    hi += (lo2 < lo);
    u64 s = lo2 + hi * EPS;
    uint32_t qt = (uint32_t)hi + ((s < lo2 || s >= P) ? 1u : 0u);
    return qt & ((1u << log2N) - 1u);
}

}  // namespace gl
