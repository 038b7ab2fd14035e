// Integer kernels shared by both ciphertext fields (policy F): LWE keyswitch and leveled linear combinations.
// F provides the arithmetic mod q on canonical 64-bit words:
//   static void digits(u64 a, uint32_t levels, uint32_t base_log, unsigned char *d)  // d[lev] = digit + B/2
//   static u64 add(u64, u64), sub(u64, u64), neg(u64), mul_small(i64 coef, u64 v)
//   static u64 reduce96(uint32_t hi, u64 lo), reduce128(u64 hi, u64 lo)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ksl {

typedef uint64_t u64;
typedef int64_t i64;

// Keyswitch: out = (0,...,0,b) - sum_j sum_lev dec_lev(a_j) * KSK[j][lev].  One workgroup handles KS_TILE
// ciphertexts; thread t owns output columns t, t+256, t+512.  The signed digits d in [-B/2, B/2] are staged in
// LDS as d + B/2 (unsigned), so the inner multiply-accumulate is unsigned: acc(96 bit) += d' * K is two
// v_mad_u64_u32 and one add; the bias sum_rows (B/2) * K is a per-key constant (ks_bias, computed at keygen)
// and is added back at the end:  -sum d K = bias - sum d' K.
constexpr int KS_TILE = 8;
constexpr int KS_THREADS = 256;
constexpr int KS_COLS = 3;  // ceil(631 / 256)

struct acc96 {
    u64 lo;       // bits 0..63
    uint32_t hi;  // bits 64..95
};

__device__ __forceinline__ void mac96(acc96 &a, uint32_t d, u64 k) {
    // a += d * k, d < 2^8: (t1:t0) = d*k0 + a0 ; (u1:u0) = d*k1 + (a2:a1) + t1
    const u64 t = (u64)d * (uint32_t)k + (uint32_t)a.lo;
    const u64 u = (u64)d * (uint32_t)(k >> 32) + (((u64)a.hi << 32) | (a.lo >> 32)) + (t >> 32);
    a.lo = (u << 32) | (uint32_t)t;
    a.hi = (uint32_t)(u >> 32);
}

// SPLIT = false: one workgroup walks all rows and writes the finished small ciphertexts (throughput form).
// SPLIT = true : blockIdx.y selects a slice of the coefficients; the workgroup writes its 96-bit partial sums
//                and k_keyswitch_reduce finishes (latency form for small batches: the row walk is the latency).
template <bool SPLIT, class F>
__global__ void __launch_bounds__(KS_THREADS)
    k_keyswitch(const u64 *__restrict__ in, const u64 *__restrict__ ksk, const u64 *__restrict__ ks_bias,
                u64 *__restrict__ out, unsigned __int128 *__restrict__ partial, uint32_t count, uint32_t n,
                uint32_t big_n, uint32_t levels, uint32_t base_log, uint32_t ks_stride, uint32_t coefs_per_slice) {
    extern __shared__ unsigned char digits[];  // [KS_TILE][slice coefficients * levels], value d + B/2
    const uint32_t first = blockIdx.x * KS_TILE;
    const uint32_t tile = min((uint32_t)KS_TILE, count - first);
    const uint32_t j0 = SPLIT ? blockIdx.y * coefs_per_slice : 0;
    const uint32_t nj = SPLIT ? min(coefs_per_slice, big_n - j0) : big_n;
    const uint32_t rows = nj * levels;
    const i64 half = (i64)1 << (base_log - 1);
    for (uint32_t idx = threadIdx.x; idx < KS_TILE * nj; idx += KS_THREADS) {
        const uint32_t b = idx / nj, j = idx % nj;
        unsigned char *d = digits + (size_t)b * rows + (size_t)j * levels;
        if (b >= tile) {  // unused slots of a ragged last tile: digit value 0 -> stored bias only (result discarded)
            for (uint32_t lev = 0; lev < levels; lev++) d[lev] = (unsigned char)half;
            continue;
        }
        F::digits(in[(size_t)(first + b) * (big_n + 1) + j0 + j], levels, base_log, d);
    }
    __syncthreads();

    acc96 acc[KS_TILE][KS_COLS];
#pragma unroll
    for (int b = 0; b < KS_TILE; b++)
#pragma unroll
        for (int cc = 0; cc < KS_COLS; cc++) acc[b][cc] = acc96{0, 0};
    bool col_ok[KS_COLS];
#pragma unroll
    for (int cc = 0; cc < KS_COLS; cc++) col_ok[cc] = threadIdx.x + cc * KS_THREADS <= n;

    const u64 *kbase = ksk + (size_t)j0 * levels * ks_stride;
#pragma unroll 2
    for (uint32_t r = 0; r < rows; r++) {
        const u64 *krow = kbase + (size_t)r * ks_stride;
        u64 kv[KS_COLS];
#pragma unroll
        for (int cc = 0; cc < KS_COLS; cc++) kv[cc] = col_ok[cc] ? krow[threadIdx.x + cc * KS_THREADS] : 0;
#pragma unroll
        for (int b = 0; b < KS_TILE; b++) {
            const uint32_t d = digits[(size_t)b * rows + r];
#pragma unroll
            for (int cc = 0; cc < KS_COLS; cc++) mac96(acc[b][cc], d, kv[cc]);
        }
    }
#pragma unroll
    for (int b = 0; b < KS_TILE; b++) {
        if (b >= (int)tile) break;
#pragma unroll
        for (int cc = 0; cc < KS_COLS; cc++) {
            const uint32_t col = threadIdx.x + cc * KS_THREADS;
            if (col > n) continue;
            if constexpr (SPLIT) {
                partial[((size_t)blockIdx.y * count + first + b) * ks_stride + col] =
                    ((unsigned __int128)acc[b][cc].hi << 64) | acc[b][cc].lo;
            } else {
                u64 v = F::sub(ks_bias[col], F::reduce96(acc[b][cc].hi, acc[b][cc].lo));
                if (col == n) v = F::add(v, in[(size_t)(first + b) * (big_n + 1) + big_n]);
                out[(size_t)(first + b) * (n + 1) + col] = v;
            }
        }
    }
}

template <class F>
__global__ void __launch_bounds__(256)
    k_keyswitch_reduce(const u64 *__restrict__ in, const unsigned __int128 *__restrict__ partial,
                       const u64 *__restrict__ ks_bias, u64 *__restrict__ out, uint32_t count, uint32_t n,
                       uint32_t big_n, uint32_t ks_stride, uint32_t slices) {
    const uint32_t ct = blockIdx.x;
    for (uint32_t col = threadIdx.x; col <= n; col += blockDim.x) {
        unsigned __int128 a = 0;  // <= 64 slices of < 2^81
        for (uint32_t s = 0; s < slices; s++) a += partial[((size_t)s * count + ct) * ks_stride + col];
        u64 v = F::sub(ks_bias[col], F::reduce128((u64)(a >> 64), (u64)a));
        if (col == n) v = F::add(v, in[(size_t)ct * (big_n + 1) + big_n]);
        out[(size_t)ct * (n + 1) + col] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// One workgroup per output row; a thread owns the columns tid, tid + 256, ... and walks the row's terms once, so the
// loads of one term (LC_COLS per thread) are independent and the term loop is the only serial chain (the first
// version looped columns outermost: one load in flight per thread, 135 us per level of the encrypted inverse).
constexpr int LC_COLS = 5;  // 5 x 256 >= k N + 1 = 1025

template <class F>
__global__ void __launch_bounds__(256)
    k_lincomb(const u64 *__restrict__ store, const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ idx,
              const i64 *__restrict__ coef, const u64 *__restrict__ const_body, u64 *__restrict__ out, uint32_t width) {
    const uint32_t row = blockIdx.x;
    const uint32_t e0 = row_ptr[row], e1 = row_ptr[row + 1];
    for (uint32_t x0 = 0; x0 < width; x0 += LC_COLS * 256) {
        // Coefficients are small integers (weights of packed look-up inputs), so the terms are summed exactly in
        // signed 128-bit and reduced once per output word; F::mul_small (a 128-bit remainder for the 49-bit field)
        // is only used for a coefficient too large for that (|cf| >= 2^31: at most 2^32 such terms fit).
        __int128 acc[LC_COLS];
        u64 big[LC_COLS];
#pragma unroll
        for (int c = 0; c < LC_COLS; c++) {
            acc[c] = 0;
            big[c] = 0;
        }
#pragma unroll 2
        for (uint32_t e = e0; e < e1; e++) {
            const i64 cf = coef[e];
            const u64 *src = store + (size_t)idx[e] * width + x0;
            u64 v[LC_COLS];
#pragma unroll
            for (int c = 0; c < LC_COLS; c++) {
                const uint32_t x = threadIdx.x + 256 * c;
                v[c] = x0 + x < width ? src[x] : 0;
            }
            if (cf > -((i64)1 << 31) && cf < ((i64)1 << 31)) {
#pragma unroll
                for (int c = 0; c < LC_COLS; c++) acc[c] += (__int128)cf * (__int128)v[c];
            } else {
#pragma unroll
                for (int c = 0; c < LC_COLS; c++) big[c] = F::add(big[c], F::mul_small(cf, v[c]));
            }
        }
#pragma unroll
        for (int c = 0; c < LC_COLS; c++) {
            const uint32_t x = x0 + threadIdx.x + 256 * c;
            if (x >= width) continue;
            const bool neg = acc[c] < 0;
            const unsigned __int128 m = neg ? (unsigned __int128)(-acc[c]) : (unsigned __int128)acc[c];
            u64 a = F::reduce128((u64)(m >> 64), (u64)m);
            if (neg) a = F::neg(a);
            a = F::add(a, big[c]);
            if (x == width - 1) a = F::add(a, const_body[row]);
            out[(size_t)row * width + x] = a;
        }
    }
}


#define KSL_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

template <class F>
int launch_keyswitch(const u64 *in, const u64 *ksk, const u64 *ks_bias, u64 *out, void *partial, uint32_t slices,
                     uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels, uint32_t base_log, uint32_t ks_stride,
                     hipStream_t s) {
    if (count == 0) return 0;
    const dim3 tiles((count + KS_TILE - 1) / KS_TILE);
    if (slices <= 1 || partial == nullptr) {
        const size_t lds = (size_t)KS_TILE * big_n * levels;
        hipLaunchKernelGGL((k_keyswitch<false, F>), tiles, dim3(KS_THREADS), lds, s, in, ksk, ks_bias, out,
                           (unsigned __int128 *)nullptr, count, n, big_n, levels, base_log, ks_stride, big_n);
        KSL_LAUNCH_CHECK();
        return 0;
    }
    const uint32_t per = (big_n + slices - 1) / slices;
    const size_t lds = (size_t)KS_TILE * per * levels;
    hipLaunchKernelGGL((k_keyswitch<true, F>), dim3(tiles.x, slices), dim3(KS_THREADS), lds, s, in, ksk, ks_bias, out,
                       (unsigned __int128 *)partial, count, n, big_n, levels, base_log, ks_stride, per);
    KSL_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_keyswitch_reduce<F>), dim3(count), dim3(256), 0, s, in, (const unsigned __int128 *)partial,
                       ks_bias, out, count, n, big_n, ks_stride, slices);
    KSL_LAUNCH_CHECK();
    return 0;
}

template <class F>
int launch_lincomb(const u64 *store, const uint32_t *row_ptr, const uint32_t *idx, const i64 *coef,
                   const u64 *const_body, u64 *out, uint32_t count, uint32_t width, hipStream_t s) {
    if (count == 0) return 0;
    hipLaunchKernelGGL((k_lincomb<F>), dim3(count), dim3(256), 0, s, store, row_ptr, idx, coef, const_body, out, width);
    KSL_LAUNCH_CHECK();
    return 0;
}

}  // namespace ksl
