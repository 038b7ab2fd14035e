// LWE keyswitch as an int8 matrix product on the matrix cores (gfx950 v_mfma_i32_32x32x32_i8).
//
// The keyswitch  out = (0,...,0,b) - sum_j sum_lev dec_lev(a_j) * KSK[j][lev]  IS a matrix product:
//     S[ct][col] = sum_r D[ct][r] * K[r][col],      r = j * levels + lev  (8192 rows at the north-star set)
// with D the signed gadget digits (|d| <= B/2 = 8: int8) and K the keyswitch key.  K's words are wider than any
// MFMA operand, so each centred word is split once, at keygen, into L balanced base-256 limbs k_l in [-128, 127]
// (L = 7 for the 49-bit field, 9 for the 64-bit one):  K = sum_l k_l 2^(8 l).  The product then runs per limb in
// exact int32 (|sum| <= 8192 * 8 * 128 < 2^24) and the limbs are recombined mod q afterwards:
//     k_ks_digits      decomposes the big ciphertexts' mask coefficients into the int8 matrix D   (per call)
//     k_ksk_to_limbs   lays the key out limb-wise in MFMA operand order                            (per keygen)
//     k_ks_mfma        one wavefront = 32 ciphertexts x 32 output columns x all L limbs; operands straight from
//                      global memory (coalesced 1 KiB fragments, a ring of KS_PF k-steps in flight), optional split over K
//     k_ks_combine     sums the K-slices, recombines the limbs in 128-bit, reduces mod q, subtracts from (0, b)
// Against the scalar kernel of ks_lincomb.hpp (8 ciphertexts x 631 columns per workgroup, 96-bit v_mad chains)
// this moves the 5.2 M multiply-accumulates per ciphertext from half-rate 64-bit VALU work to the matrix cores.
//
// Fragment order: lane l = (h = l >> 5, r = l & 31) holds 16 consecutive k (bytes) of row/column r for k-half h.
// D and K use the SAME (h, byte) -> k map, and k is summed over, so the product does not depend on how the
// hardware numbers k inside a step; the C/D map (column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h) does
// matter and is the documented dtype-independent one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ksm {

typedef uint64_t u64;
typedef int64_t i64;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int TILE = 32;     // ciphertexts per wavefront tile, and output columns per column block
constexpr int KSTEP = 32;    // k consumed per MFMA
constexpr int WAVES = 4;     // wavefronts (ciphertext tiles) per workgroup
#ifndef BMI_KS_PF_WIDE
#define BMI_KS_PF_WIDE 3   // ring depth for the 9-limb moduli (their operand registers leave room for three k-steps in flight)
#endif
#ifndef BMI_KS_PF
#define BMI_KS_PF 4   // measured at 8,192 ciphertexts, 7 limbs (keyswitch total, three kernels): 1.13 ms with the round-2 double buffer, 0.72 / 0.70 / 0.65 ms at depth 1 / 2 / 4 of this ring
#endif

// ---- digits: one thread per mask coefficient; requires levels <= 16 (8-byte stores when levels is 8 or 16)
template <class F>
__global__ void __launch_bounds__(256)
    k_ks_digits(const u64 *__restrict__ in, signed char *__restrict__ D, uint32_t count, uint32_t big_n, uint32_t levels,
                uint32_t base_log) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)count * big_n) return;
    const uint32_t ct = (uint32_t)(idx / big_n), j = (uint32_t)(idx % big_n);
    unsigned char d[16];
    F::digits(in[(size_t)ct * (big_n + 1) + j], levels, base_log, d);
    const int half = 1 << (base_log - 1);
    signed char *o = D + ((size_t)ct * big_n + j) * levels;
    if ((levels & 7) == 0) {
        for (uint32_t g = 0; g < levels; g += 8) {
            u64 w = 0;
#pragma unroll
            for (int lev = 0; lev < 8; lev++) w |= (u64)(unsigned char)(signed char)((int)d[g + lev] - half) << (8 * lev);
            *reinterpret_cast<u64 *>(o + g) = w;
        }
    } else {
        for (uint32_t lev = 0; lev < levels; lev++) o[lev] = (signed char)((int)d[lev] - half);
    }
}

// ---- key layout: Bm[cb][ks][limb][h][r][16 bytes], element (k = 32 ks + 16 h + byte, col = 32 cb + r)
__host__ __device__ __forceinline__ size_t limb_offset(uint32_t cb, uint32_t ks, uint32_t limb, uint32_t h, uint32_t r,
                                                       uint32_t ksteps, uint32_t L) {
    return (((((size_t)cb * ksteps + ks) * L + limb) * 2 + h) * 32 + r) * 16;
}

template <class F>
__global__ void __launch_bounds__(256)
    k_ksk_to_limbs(const u64 *__restrict__ ksk, signed char *__restrict__ Bm, uint32_t rows, uint32_t n, uint32_t ks_stride,
                   uint32_t L) {
    const uint32_t cbs = (n + 1 + TILE - 1) / TILE, ksteps = rows / KSTEP;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * cbs * TILE) return;
    const uint32_t col = (uint32_t)(idx % (cbs * TILE)), k = (uint32_t)(idx / (cbs * TILE));
    __int128 c = col <= n ? (__int128)F::centered(ksk[(size_t)k * ks_stride + col]) : 0;
    const uint32_t cb = col / TILE, r = col % TILE, ks = k / KSTEP, h = (k % KSTEP) / 16, byte = k % 16;
    for (uint32_t l = 0; l < L; l++) {
        const int dl = (int)((c + 128) & 255) - 128;  // balanced digit in [-128, 127]
        c = (c - dl) >> 8;
        Bm[limb_offset(cb, ks, l, h, r, ksteps, L) + byte] = (signed char)dl;
    }
}

// ---- the product.  grid = (ceil(tiles / WAVES), column blocks, slices); S[slice][ct][limb][cbs * 32] (int32)
// k-steps of operands in flight per wavefront (7 limbs: the 49-bit field; 9 limbs: the 64-bit moduli, whose 144 accumulator
// registers leave room for fewer stages at two wavefronts per SIMD)
template <int L>
constexpr int ks_pf() { return L <= 7 ? BMI_KS_PF : BMI_KS_PF_WIDE; }

template <int L>
__global__ void __launch_bounds__(64 * WAVES)
    k_ks_mfma(const signed char *__restrict__ D, const signed char *__restrict__ Bm, int *__restrict__ S, uint32_t count,
              uint32_t K, uint32_t cbs, uint32_t ksteps_per_slice) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * WAVES + wave;
    if (tile * TILE >= count) return;  // whole wavefront: no barrier in this kernel
    const uint32_t cb = blockIdx.y, slice = blockIdx.z;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t ksteps = K / KSTEP;
    const uint32_t ks0 = slice * ksteps_per_slice;
    const uint32_t ks1 = min(ks0 + ksteps_per_slice, ksteps);
    const uint32_t ct_row = min(tile * TILE + r, count - 1);  // ragged last tile: clamped rows are computed and dropped
    const v4i *ap = reinterpret_cast<const v4i *>(D + (size_t)ct_row * K + 16 * h) + (size_t)ks0 * 2;
    const v4i *bp = reinterpret_cast<const v4i *>(Bm + limb_offset(cb, ks0, 0, h, r, ksteps, L));
    constexpr int BSTEP = L * 64;  // 16-byte words per k-step of the key

    v16i acc[L];
#pragma unroll
    for (int l = 0; l < L; l++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[l][i] = 0;

    // Operands come straight from global memory (L2-resident key, streamed digits): a ring of KS_PF k-steps is kept in flight
    // per wavefront.  With one step ahead (round 2) the kernel ran at the latency of an L2 access per k-step: 0.96 ms for 8,192
    // ciphertexts against 0.12 ms of matrix-core time.
    constexpr int KS_PF = ks_pf<L>();
    v4i a_buf[KS_PF], b_buf[KS_PF][L];
#pragma unroll
    for (int s = 0; s < KS_PF; s++) {
        a_buf[s] = v4i{0, 0, 0, 0};
#pragma unroll
        for (int l = 0; l < L; l++) b_buf[s][l] = v4i{0, 0, 0, 0};
        if (ks0 + s < ks1) {  // an empty K-slice (slices not dividing the k-steps) touches no memory and writes zeros
            a_buf[s] = ap[2 * s];
#pragma unroll
            for (int l = 0; l < L; l++) b_buf[s][l] = bp[s * BSTEP + l * 64];
        }
    }
    for (uint32_t ks = ks0; ks < ks1; ks += KS_PF) {
#pragma unroll
        for (int s = 0; s < KS_PF; s++) {
            if (ks + s < ks1) {   // uniform over the wavefront
#pragma unroll
                for (int l = 0; l < L; l++) acc[l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_buf[s], b_buf[s][l], acc[l], 0, 0, 0);
                if (ks + s + KS_PF < ks1) {   // refill this stage with the step KS_PF ahead
                    a_buf[s] = ap[2 * (s + KS_PF)];
#pragma unroll
                    for (int l = 0; l < L; l++) b_buf[s][l] = bp[(s + KS_PF) * BSTEP + l * 64];
                }
            }
        }
        ap += 2 * KS_PF;
        bp += (size_t)BSTEP * KS_PF;
    }

    const uint32_t cw = cbs * TILE;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const uint32_t ct = tile * TILE + row;
        if (ct >= count) continue;
        int *o = S + (((size_t)slice * count + ct) * L) * cw + cb * TILE + r;
#pragma unroll
        for (int l = 0; l < L; l++) o[(size_t)l * cw] = acc[l][i];
    }
}

// ---- recombination: one thread per (ciphertext, column)
template <class F, int L>
__global__ void __launch_bounds__(256)
    k_ks_combine(const u64 *__restrict__ in, const int *__restrict__ S, u64 *__restrict__ out, uint32_t count, uint32_t n,
                 uint32_t big_n, uint32_t cbs, uint32_t slices) {
    const uint32_t ct = blockIdx.x;
    const uint32_t cw = cbs * TILE;
    for (uint32_t col = threadIdx.x; col <= n; col += blockDim.x) {
        i64 sum[L];
#pragma unroll
        for (int l = 0; l < L; l++) sum[l] = 0;
#pragma unroll 4
        for (uint32_t sl = 0; sl < slices; sl++) {  // L independent loads per slice, four slices in flight
            const int *base = S + (((size_t)sl * count + ct) * L) * cw + col;
#pragma unroll
            for (int l = 0; l < L; l++) sum[l] += base[(size_t)l * cw];
        }
        __int128 v = 0;
#pragma unroll
        for (int l = L - 1; l >= 0; l--) v = v * 256 + sum[l];
        const bool neg = v < 0;
        const unsigned __int128 m = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
        u64 red = F::reduce128((u64)(m >> 64), (u64)m);
        if (neg) red = F::neg(red);
        const u64 base = col == n ? in[(size_t)ct * (big_n + 1) + big_n] : 0;
        out[(size_t)ct * (n + 1) + col] = F::sub(base, red);
    }
}

inline size_t limb_bytes(uint32_t rows, uint32_t n, uint32_t L) {
    const uint32_t cbs = (n + 1 + TILE - 1) / TILE;
    return (size_t)cbs * (rows / KSTEP) * L * 2 * 32 * 16;
}
inline size_t s_bytes(uint32_t count, uint32_t n, uint32_t L, uint32_t slices) {
    const uint32_t cbs = (n + 1 + TILE - 1) / TILE;
    return (size_t)slices * count * L * cbs * TILE * sizeof(int);
}

#define KSM_CHECK()                              \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

template <class F>
int launch_ksk_to_limbs(const u64 *ksk, signed char *Bm, uint32_t rows, uint32_t n, uint32_t ks_stride, uint32_t L,
                        hipStream_t s) {
    const uint32_t cbs = (n + 1 + TILE - 1) / TILE;
    const size_t total = (size_t)rows * cbs * TILE;
    hipLaunchKernelGGL((k_ksk_to_limbs<F>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ksk, Bm, rows, n,
                       ks_stride, L);
    KSM_CHECK();
    return 0;
}

// D: count * big_n * levels bytes;  S: s_bytes(count, n, L, slices)
template <class F, int L>
int launch_keyswitch(const u64 *in, const signed char *Bm, signed char *D, int *S, u64 *out, uint32_t slices,
                     uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels, uint32_t base_log, hipStream_t s) {
    if (count == 0) return 0;
    const uint32_t K = big_n * levels, cbs = (n + 1 + TILE - 1) / TILE, ksteps = K / KSTEP;
    const size_t coefs = (size_t)count * big_n;
    hipLaunchKernelGGL((k_ks_digits<F>), dim3((unsigned)((coefs + 255) / 256)), dim3(256), 0, s, in, D, count, big_n, levels,
                       base_log);
    KSM_CHECK();
    const uint32_t tiles = (count + TILE - 1) / TILE;
    const uint32_t per_slice = (ksteps + slices - 1) / slices;
    hipLaunchKernelGGL((k_ks_mfma<L>), dim3((tiles + WAVES - 1) / WAVES, cbs, slices), dim3(64 * WAVES), 0, s, D, Bm, S, count,
                       K, cbs, per_slice);
    KSM_CHECK();
    hipLaunchKernelGGL((k_ks_combine<F, L>), dim3(count), dim3(256), 0, s, in, S, out, count, n, big_n, cbs, slices);
    KSM_CHECK();
    return 0;
}

}  // namespace ksm
