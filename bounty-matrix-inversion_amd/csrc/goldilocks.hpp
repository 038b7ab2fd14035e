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

// ---- 32-bit limb helpers.  Measured on gfx950 (tools/microbench/valu_cost.hip, wall clock, chip-wide):
// plain 32-bit ALU ops issue at ~1 per 2 cycles per SIMD, but EVERYTHING this arithmetic is made of is half
// rate (~1 per 4 cycles): 64-bit forms (v_lshl_add_u64, v_lshlrev_b64, v_cmp_*_u64), v_mad_u64_u32 / v_mul_*,
// carry-producing or -consuming adds (v_add_co_u32, v_addc_co_u32, v_subb_co_u32), v_cndmask and DPP moves;
// two waves per SIMD saturate the pipe.  A modular add therefore costs about the same written with 64-bit
// compares or with hardware carries (both were built and timed); the carry form below needs no compares and
// composes better with the reductions, so it is the one kept.
GL_HD uint32_t lo32(u64 x) { return (uint32_t)x; }
GL_HD uint32_t hi32(u64 x) { return (uint32_t)(x >> 32); }
GL_HD u64 mk64(uint32_t lo, uint32_t hi) { return ((u64)hi << 32) | (u64)lo; }

#if defined(__clang__)
GL_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t cin, uint32_t &cout) { return __builtin_addc(a, b, cin, &cout); }
GL_HD uint32_t subc32(uint32_t a, uint32_t b, uint32_t bin, uint32_t &bout) { return __builtin_subc(a, b, bin, &bout); }
#else
GL_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t cin, uint32_t &cout) {
    u64 t = (u64)a + b + cin;
    cout = (uint32_t)(t >> 32);
    return (uint32_t)t;
}
GL_HD uint32_t subc32(uint32_t a, uint32_t b, uint32_t bin, uint32_t &bout) {
    u64 t = (u64)a - b - bin;
    bout = (uint32_t)(t >> 63);
    return (uint32_t)t;
}
#endif

// x + (flag ? EPS : 0) on halves, returning the carry out
GL_HD u64 add_eps_if(u64 x, uint32_t flag, uint32_t &cout) {
    uint32_t c;
    const uint32_t m = flag ? 0xFFFFFFFFu : 0u;
    const uint32_t l = addc32(lo32(x), m, 0u, c);
    const uint32_t h = addc32(hi32(x), 0u, c, cout);
    return mk64(l, h);
}
// x - (flag ? EPS : 0) on halves
GL_HD u64 sub_eps_if(u64 x, uint32_t flag) {
    uint32_t b, b2;
    const uint32_t m = flag ? 0xFFFFFFFFu : 0u;
    const uint32_t l = subc32(lo32(x), m, 0u, b);
    const uint32_t h = subc32(hi32(x), 0u, b, b2);
    return mk64(l, h);
}
// canonical representative of a value in [0, 2^64): x >= P  <=>  x + EPS carries
GL_HD u64 canon(u64 x) {
    uint32_t c1, c2;
    const uint32_t l = addc32(lo32(x), 0xFFFFFFFFu, 0u, c1);
    const uint32_t h = addc32(hi32(x), 0u, c1, c2);
    return c2 ? mk64(l, h) : x;
}

GL_HD u64 add(u64 a, u64 b) {
    uint32_t c1, c2, c3, c4;
    const uint32_t s0 = addc32(lo32(a), lo32(b), 0u, c1);
    const uint32_t s1 = addc32(hi32(a), hi32(b), c1, c2);
    const uint32_t t0 = addc32(s0, 0xFFFFFFFFu, 0u, c3);  // s + EPS = s - P (mod 2^64); carries iff s >= P
    const uint32_t t1 = addc32(s1, 0u, c3, c4);
    const bool sel = (c2 | c4) != 0;
    return mk64(sel ? t0 : s0, sel ? t1 : s1);
}
GL_HD u64 sub(u64 a, u64 b) {
    uint32_t b1, b2;
    const uint32_t d0 = subc32(lo32(a), lo32(b), 0u, b1);
    const uint32_t d1 = subc32(hi32(a), hi32(b), b1, b2);
    return sub_eps_if(mk64(d0, d1), b2);  // borrow: + P = - EPS (mod 2^64)
}
GL_HD u64 neg(u64 a) { return sub(0, a); }

GL_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// x * EPS for a 32-bit x (the 64x32 multiply-add is as cheap as the shift/subtract form on gfx950)
GL_HD u64 times_eps32(u64 x32) { return x32 * EPS; }

// (hi * 2^64 + lo) mod P, any hi/lo:  2^64 = EPS, 2^96 = -1
GL_HD u64 reduce128(u64 hi, u64 lo) {
    uint32_t b1, b2, c;
    // t0 = lo - hi_hi  (+P on borrow)
    const uint32_t l0 = subc32(lo32(lo), hi32(hi), 0u, b1);
    const uint32_t l1 = subc32(hi32(lo), 0u, b1, b2);
    const u64 t0 = sub_eps_if(mk64(l0, l1), b2);
    // + hi_lo * EPS  (+EPS on carry)
    const u64 t1 = times_eps32(lo32(hi));
    uint32_t c1, c2;
    const uint32_t r0 = addc32(lo32(t0), lo32(t1), 0u, c1);
    const uint32_t r1 = addc32(hi32(t0), hi32(t1), c1, c2);
    const u64 r = add_eps_if(mk64(r0, r1), c2, c);
    return canon(r);
}
// (hi32 * 2^64 + lo) mod P for hi32 < 2^32 (one fold, no borrow step)
GL_HD u64 reduce96(u64 hi32v, u64 lo) {
    const u64 t1 = times_eps32(hi32v);
    uint32_t c1, c2, c;
    const uint32_t r0 = addc32(lo32(lo), lo32(t1), 0u, c1);
    const uint32_t r1 = addc32(hi32(lo), hi32(t1), c1, c2);
    const u64 r = add_eps_if(mk64(r0, r1), c2, c);
    return canon(r);
}
// 64 x 64 -> 128 schoolbook on 32-bit halves: four 32x32+64 multiply-adds (v_mad_u64_u32), then the fold
GL_HD u64 mul(u64 a, u64 b) {
    const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    const u64 p00 = (u64)a0 * b0;
    const u64 m1 = (u64)a0 * b1 + hi32(p00);   // < 2^64: (2^32-1)^2 + 2^32 - 1
    const u64 m2 = (u64)a1 * b0 + lo32(m1);
    const u64 hi = (u64)a1 * b1 + hi32(m1) + hi32(m2);
    return reduce128(hi, mk64(lo32(p00), lo32(m2)));
}

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
