// The second ciphertext field: q = 2^49 - 720895 (prime, q = 1 mod 2^16).  Exact integer arithmetic carried in
// IEEE doubles on the GPU: every element is an integer of magnitude < 2^53, kept as a CENTRED representative, and
// every reduction is  x - q * rint(x / q)  (v_mul_f64, v_rndne_f64, v_fma_f64 - no compares, no carries).
// Measured on gfx950 (tools/microbench/butterfly_cost.hip): one radix-2 butterfly costs 56.9 cycles per wave in
// this form against 148 for the Goldilocks field (whose 64-bit integer ops are all half rate), and the 4 spare
// mantissa bits allow sums of up to 16 reduced values before any reduction (lazy butterflies, lazy MAC).
//
// Bounds (p = q): red() returns |r| <= p/2 exactly (the centred residue) for |x| < 4p and |r| < 0.51p up to 2^53;
// mul(a, b) is exact for |a| < 2^53, |b| <= p/2 and returns |r| <= (0.5 + |a| / 8p) p: the quotient estimate is off by
// at most 3 |a|/32p and the low product word adds |a|/32p (tools/f64_bounds.py tracks these through the transforms).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define F49_HD __host__ __device__ __forceinline__
#else
#define F49_HD inline
#endif

namespace f49 {

typedef uint64_t u64;
typedef int64_t i64;

constexpr u64 Q = 562949952700417ULL;
constexpr int QBITS = 49;
constexpr double P = 562949952700417.0;
constexpr double PINV = 1.0 / 562949952700417.0;
constexpr u64 GEN = 5;  // generator of Z_q^*

// ---- compile-time integer helpers (twiddle constants)
constexpr u64 mulmod_c(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % Q); }
constexpr u64 powmod_c(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = mulmod_c(r, b);
        b = mulmod_c(b, b);
        e >>= 1;
    }
    return r;
}
constexpr double centred_c(u64 v) { return v > (Q >> 1) ? -(double)(Q - v) : (double)v; }

// ---- integer arithmetic mod q (host code and the integer kernels: keyswitch, lincomb)
F49_HD u64 addq(u64 a, u64 b) { u64 s = a + b; return s >= Q ? s - Q : s; }
F49_HD u64 subq(u64 a, u64 b) { return a >= b ? a - b : a + Q - b; }
F49_HD u64 negq(u64 a) { return a ? Q - a : 0; }
F49_HD u64 mulq(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % Q); }
F49_HD u64 from_i64(i64 v) { return v >= 0 ? (u64)v % Q : Q - ((u64)(-v) % Q); }
F49_HD i64 centered(u64 a) { return a > (Q >> 1) ? (i64)a - (i64)Q : (i64)a; }
F49_HD uint32_t modswitch(u64 a, uint32_t log2N) {
    return (uint32_t)(((a << log2N) + (Q >> 1)) / Q) & ((1u << log2N) - 1u);  // a < 2^49, log2N <= 14: no overflow
}

// ---- f64 arithmetic (device hot path)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
F49_HD double red(double x) { return __builtin_fma(-__builtin_rint(x * PINV), P, x); }
F49_HD double mul(double a, double b) {
    const double h = a * b;
    const double l = __builtin_fma(a, b, -h);
    return __builtin_fma(-__builtin_rint(h * PINV), P, h) + l;
}
F49_HD double to_f(u64 v) { return v > (Q >> 1) ? -(double)(Q - v) : (double)v; }  // canonical -> centred
F49_HD u64 to_u(double x) {                                                        // any |x| < 2^53 -> canonical
    double r = red(x);
    if (r < 0) r += P;
    return (u64)r;
}

}  // namespace f49
