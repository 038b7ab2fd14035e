// HIP kernels (gfx950 / CDNA4) of the TFHE programmable-bootstrap engine.
//
//   k_bsk_to_ntt        standard-domain GGSW rows -> NTT (evaluation) domain, lane-layout, once per keygen
//   k_blind_rotate_tp   THROUGHPUT variant: one pair of wavefronts = one ciphertext; mod-switch, n CMUXes
//                       (decompose -> 6 forward NTTs -> 12 pointwise MACs -> 2 inverse NTTs), sample extraction
//   k_blind_rotate_lat  LATENCY variant: one workgroup of 8 wavefronts = one ciphertext
//   k_keyswitch         big-key LWE -> small-key LWE (signed base-2^4 decomposition, 128-bit accumulators)
//   k_lincomb           leveled linear combinations of ciphertexts (CSR)
//   k_negacyclic_mul    test hook: c = a * b mod (X^N + 1, q) through the wave NTT
//
// Data layout in HBM
//   bootstrap key : [n][(k+1)l = 6][k+1 = 2][N] 64-bit words, NTT domain, each polynomial stored in the
//                   wave's register layout (nttw::eval_offset) so one lane reads 16 B and one wave 1 KiB
//                   per load instruction (fully coalesced); 61,931,520 B for the default set.
//   keyswitch key : [k*N][ks_levels][n+1 padded to KS_STRIDE] words, row-major: consecutive threads read
//                   consecutive columns of one row.
//   ciphertexts   : big  [count][k*N+1], small [count][n+1] words.
#include <hip/hip_runtime.h>

#include "bmi_internal.hpp"
#include "ks_lincomb.hpp"
#include "ks_mfma.hpp"
#include "ntt_wave.hpp"

using gl::u64;
using gl::i64;
using namespace nttw;

namespace {

__device__ __forceinline__ void stage_twiddles(u64 *lds_tw, const u64 *__restrict__ g_tw) {
    for (int i = threadIdx.x; i < TW_WORDS; i += blockDim.x) lds_tw[i] = g_tw[i];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bsk_to_ntt(const u64 *__restrict__ std_polys, u64 *__restrict__ ntt_polys,
                                                    const u64 *__restrict__ g_tw, uint32_t n_polys) {
    __shared__ u64 lds[TW_WORDS + 4 * SCRATCH_WORDS];
    stage_twiddles(lds, g_tw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t poly = blockIdx.x * 4 + wave;
    if (poly >= n_polys) return;
    u64 *scratch = lds + TW_WORDS + wave * SCRATCH_WORDS;
    u64 x[16];
    static_for<0, 16>([&](auto J) { x[J] = std_polys[(size_t)poly * N + lane + 64 * J]; });
    forward(x, lane, lds, scratch);
    static_for<0, 16>([&](auto V) { ntt_polys[(size_t)poly * N + eval_offset(lane, V)] = x[V]; });
}

// c = a * b  (negacyclic), one wave per product
__global__ void __launch_bounds__(256) k_negacyclic_mul(const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                        u64 *__restrict__ c, const u64 *__restrict__ g_tw,
                                                        uint32_t count) {
    __shared__ u64 lds[TW_WORDS + 4 * SCRATCH_WORDS];
    stage_twiddles(lds, g_tw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t p = blockIdx.x * 4 + wave;
    if (p >= count) return;
    u64 *scratch = lds + TW_WORDS + wave * SCRATCH_WORDS;
    u64 x[16], y[16];
    static_for<0, 16>([&](auto J) {
        x[J] = a[(size_t)p * N + lane + 64 * J];
        y[J] = b[(size_t)p * N + lane + 64 * J];
    });
    forward(x, lane, lds, scratch);
    forward(y, lane, lds, scratch);
    static_for<0, 16>([&](auto V) { x[V] = gl::mul(x[V], y[V]); });
    inverse(x, lane, lds, scratch);
    static_for<0, 16>([&](auto J) { c[(size_t)p * N + lane + 64 * J] = x[J]; });
}

// ------------------------------------------------------------------------------------------------
// Signed decomposition of one coefficient for the bootstrap gadget (l = 3, base 2^15): digits of the
// centred lift rounded to its top 45 bits; d[0] is the most significant and absorbs the final carry.
__device__ __forceinline__ void decompose3x15(u64 a, int (&d)[3]) {
    i64 c = gl::centered(a);
    i64 r = (c >> 19) + ((c >> 18) & 1);
    int d2 = (int)(r & 0x7FFF);
    r >>= 15;
    if (d2 >= 0x4000) { d2 -= 0x8000; r += 1; }
    int d1 = (int)(r & 0x7FFF);
    r >>= 15;
    if (d1 >= 0x4000) { d1 -= 0x8000; r += 1; }
    d[0] = (int)r;
    d[1] = d1;
    d[2] = d2;
}

// THROUGHPUT blind rotation: one PAIR of wavefronts per ciphertext, CTS ciphertexts per workgroup.
// Wave c of the pair owns GLWE component c: its accumulator polynomial ACC_c (registers), the three digit
// polynomials decomposed from it, and output polynomial c of the external product.  Per CMUX and level the
// wave transforms its digit polynomial, publishes the result in its LDS tile, and both waves multiply-
// accumulate their own and the partner's transform with the bootstrap-key rows of their output polynomial;
// then each wave inverse-transforms its own sum.  ~128 live registers per lane (no spills at 2 waves/SIMD,
// which saturates the integer VALU on gfx950); two workgroup barriers per level order the tile hand-off.
// (A first version kept a whole ciphertext - both accumulators, both sums - in one wave: 251 registers, and
// it spilled once the arithmetic moved to 32-bit limbs.)
template <int CTS>
__global__ void __launch_bounds__(128 * CTS, 2)
    k_blind_rotate_tp(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                      const u64 *__restrict__ luts, const u64 *__restrict__ bsk, const u64 *__restrict__ g_tw,
                      u64 *__restrict__ out, uint32_t count, uint32_t n) {
    constexpr int AT_WORDS = BMI_AT_WORDS;
    __shared__ u64 lds[TW_WORDS + 2 * CTS * SCRATCH_WORDS + CTS * AT_WORDS];
    stage_twiddles(lds, g_tw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ctl = wave >> 1, c = wave & 1;
    // the last workgroup may hold a ciphertext slot beyond the batch: it computes on slot count-1 again
    // (all waves must reach every barrier) and simply does not write a result
    const uint32_t ct_raw = blockIdx.x * CTS + ctl;
    const bool live = ct_raw < count;
    const uint32_t ct = live ? ct_raw : count - 1;
    u64 *tile = lds + TW_WORDS + wave * SCRATCH_WORDS;
    const u64 *ptile = lds + TW_WORDS + (wave ^ 1) * SCRATCH_WORDS;
    uint16_t *at = reinterpret_cast<uint16_t *>(lds + TW_WORDS + 2 * CTS * SCRATCH_WORDS + ctl * AT_WORDS);
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);

    // modulus switch of the small ciphertext (mask + body) to Z_{2N}; both waves of the pair share the array
    for (uint32_t i = lane + 64 * c; i <= n; i += 128) at[i] = (uint16_t)gl::modswitch(lwe[i], LOG_N + 1);
    __syncthreads();

    // ACC = X^(-b~) * (0, tv): component 0 starts at zero, component 1 at the rotated test polynomial
    u64 acc[16];
    {
        const u64 *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at[n];
        static_for<0, 16>([&](auto J) {
            const uint32_t e = (lane + 64 * J + bt) & (2 * N - 1);
            const u64 v = tv[e & (N - 1)];
            acc[J] = c ? ((e & N) ? gl::neg(v) : v) : 0;
        });
    }

    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];  // a_t == 0 adds exactly zero; not skipped so that barriers stay uniform
        const u64 *bsk_i = bsk + (size_t)i * 12 * N;
        // diff = X^(a~) * ACC_c - ACC_c through the wave's own tile (natural order), then its three digits,
        // packed: dlo = d2 | d1 << 16 (16-bit signed fields), dhi = d0
        wave_sync();
        static_for<0, 16>([&](auto J) { tile[lane + 64 * J] = acc[J]; });
        wave_sync();
        int dlo[16], dhi[16];
        static_for<0, 16>([&](auto J) {
            const uint32_t e = (lane + 64 * J + 2 * N - a_t) & (2 * N - 1);
            u64 v = tile[e & (N - 1)];
            v = (e & N) ? gl::neg(v) : v;
            int d[3];
            decompose3x15(gl::sub(v, acc[J]), d);
            dlo[J] = (d[2] & 0xFFFF) | (d[1] << 16);
            dhi[J] = d[0];
        });
        u64 accn[16];
        static_for<0, 3>([&](auto LEV) {
            constexpr int lev = LEV;
            u64 x[16];
            static_for<0, 16>([&](auto J) {
                const int dg = lev == 0 ? dhi[J] : (lev == 1 ? (dlo[J] >> 16) : (int)(short)dlo[J]);
                x[J] = gl::from_i64((i64)dg);
            });
            forward(x, lane, lds, tile);
            wave_sync();
            static_for<0, 16>([&](auto V) { tile[eval_offset(lane, V)] = x[V]; });
            __syncthreads();  // both transforms of this level are published
            // rows (component, level) x output polynomial c: own row from registers, partner's from its tile
            const u64 *row_own = bsk_i + ((size_t)(c * 3 + lev) * 2 + c) * N;
            const u64 *row_par = bsk_i + ((size_t)((c ^ 1) * 3 + lev) * 2 + c) * N;
            static_for<0, 8>([&](auto VP) {
                const ulonglong2 bo = reinterpret_cast<const ulonglong2 *>(row_own)[VP * 64 + lane];
                const ulonglong2 bp = reinterpret_cast<const ulonglong2 *>(row_par)[VP * 64 + lane];
                const ulonglong2 xp = reinterpret_cast<const ulonglong2 *>(ptile)[VP * 64 + lane];
                u64 s0 = gl::add(gl::mul(x[2 * VP], bo.x), gl::mul(xp.x, bp.x));
                u64 s1 = gl::add(gl::mul(x[2 * VP + 1], bo.y), gl::mul(xp.y, bp.y));
                if constexpr (lev == 0) {
                    accn[2 * VP] = s0;
                    accn[2 * VP + 1] = s1;
                } else {
                    accn[2 * VP] = gl::add(accn[2 * VP], s0);
                    accn[2 * VP + 1] = gl::add(accn[2 * VP + 1], s1);
                }
            });
            __syncthreads();  // the partner has read this tile: it may be overwritten
        });
        inverse(accn, lane, lds, tile);
        static_for<0, 16>([&](auto J) { acc[J] = gl::add(acc[J], accn[J]); });
    }

    // sample extraction of coefficient 0: a_out[0] = A[0], a_out[t] = -A[N - t] (wave 0), b_out = B[0] (wave 1)
    if (!live) return;
    u64 *o = out + (size_t)ct * (N + 1);
    if (c == 0) {
        static_for<0, 16>([&](auto J) {
            const uint32_t m = lane + 64 * J;
            if (m == 0) o[0] = acc[J];
            else o[N - m] = gl::neg(acc[J]);
        });
    } else if (lane == 0) {
        o[N] = acc[0];
    }
}

// LATENCY blind rotation: one workgroup of 8 wavefronts per ciphertext.  Per CMUX:
//   phase A  waves 0..5: wave r = (component c, level lev) builds its digit polynomial from the LDS accumulator
//            (rotation + decomposition) and forward-transforms it; the result stays in the wave's LDS tile.
//            Every thread meanwhile prefetches its 24 bootstrap-key words for phase B.
//   phase B  all 512 threads: 2 x 1024 output slots, 4 per thread, 6 multiply-accumulates each -> LDS.
//   phase C  waves 0,1: inverse transform of one output polynomial each, accumulated into the LDS accumulator.
// LDS: twiddles | acc[2][1024] | 6 tiles | Y[2][1024] | mod-switched mask.  ~101 KB -> one workgroup per CU.
constexpr int LAT_THREADS = 512;
constexpr int LAT_LDS_WORDS = TW_WORDS + 2 * N + 6 * SCRATCH_WORDS + 2 * N + BMI_AT_WORDS;
static_assert(LAT_LDS_WORDS <= BMI_LDS_WORDS_MAX, "LAT_LDS_WORDS exceeds the 160 KB of LDS");

__global__ void __launch_bounds__(LAT_THREADS)
    k_blind_rotate_lat(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                       const u64 *__restrict__ luts, const u64 *__restrict__ bsk, const u64 *__restrict__ g_tw,
                       u64 *__restrict__ out, uint32_t count, uint32_t n) {
    extern __shared__ u64 lds[];
    u64 *acc = lds + TW_WORDS;                  // [2][N], natural coefficient order
    u64 *tiles = acc + 2 * N;                   // [6][SCRATCH_WORDS]
    u64 *Y = tiles + 6 * SCRATCH_WORDS;         // [2][N], evaluation layout
    uint16_t *at = reinterpret_cast<uint16_t *>(Y + 2 * N);
    stage_twiddles(lds, g_tw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += LAT_THREADS) at[i] = (uint16_t)gl::modswitch(lwe[i], LOG_N + 1);
    __syncthreads();
    {
        const u64 *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at[n];
        for (int m = tid; m < N; m += LAT_THREADS) {
            const uint32_t e = (m + bt) & (2 * N - 1);
            const u64 v = tv[e & (N - 1)];
            acc[m] = 0;
            acc[N + m] = (e & N) ? gl::neg(v) : v;
        }
    }
    __syncthreads();

    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];
        if (a_t == 0) continue;  // uniform over the workgroup
        const u64 *bsk_i = bsk + (size_t)i * 12 * N;
        // prefetch for phase B: slot = tid + 512 m -> (oc, idx); rows r = 0..5
        u64 b[4][6];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int slot = tid + LAT_THREADS * m, oc = slot >> LOG_N, idx = slot & (N - 1);
#pragma unroll
            for (int r = 0; r < 6; r++) b[m][r] = bsk_i[(size_t)(r * 2 + oc) * N + idx];
        }
        if (wave < 6) {
            const int c = wave / 3, lev = wave - 3 * c;
            const u64 *a = acc + c * N;
            u64 x[16];
            static_for<0, 16>([&](auto J) {
                const uint32_t mm = lane + 64 * J;
                const uint32_t e = (mm + 2 * N - a_t) & (2 * N - 1);
                u64 v = a[e & (N - 1)];
                v = (e & N) ? gl::neg(v) : v;
                int d[3];
                decompose3x15(gl::sub(v, a[mm]), d);
                x[J] = gl::from_i64((i64)(lev == 0 ? d[0] : (lev == 1 ? d[1] : d[2])));
            });
            u64 *tile = tiles + wave * SCRATCH_WORDS;
            forward(x, lane, lds, tile);
            wave_sync();
            static_for<0, 16>([&](auto V) { tile[eval_offset(lane, V)] = x[V]; });
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int slot = tid + LAT_THREADS * m, idx = slot & (N - 1);
            u64 y = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) y = gl::add(y, gl::mul(tiles[r * SCRATCH_WORDS + idx], b[m][r]));
            Y[slot] = y;
        }
        __syncthreads();
        if (wave < 2) {
            u64 x[16];
            static_for<0, 16>([&](auto V) { x[V] = Y[wave * N + eval_offset(lane, V)]; });
            u64 *tile = tiles + wave * SCRATCH_WORDS;
            inverse(x, lane, lds, tile);
            u64 *a = acc + wave * N;
            static_for<0, 16>([&](auto J) { a[lane + 64 * J] = gl::add(a[lane + 64 * J], x[J]); });
        }
        __syncthreads();
    }
    u64 *o = out + (size_t)ct * (N + 1);
    for (int m = tid; m < N; m += LAT_THREADS) {
        if (m == 0) {
            o[0] = acc[0];
            o[N] = acc[N];
        } else {
            o[N - m] = gl::neg(acc[m]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Keyswitch and lincomb live in ks_lincomb.hpp (shared with the 49-bit field); the Goldilocks policy:
struct FieldG {
    static __device__ __forceinline__ void digits(u64 a, uint32_t levels, uint32_t base_log, unsigned char *d) {
        // centred lift, round half up to the top levels*base_log bits, signed digits in [-B/2, B/2), top absorbs the carry
        const uint32_t shift = 64 - levels * base_log;
        const i64 B = (i64)1 << base_log, half = B >> 1;
        const i64 c = gl::centered(a);
        i64 r = (c >> shift) + ((c >> (shift - 1)) & 1);
        for (int lev = (int)levels - 1; lev >= 1; lev--) {
            i64 v = r & (B - 1);
            r >>= base_log;
            if (v >= half) { v -= B; r += 1; }
            d[lev] = (unsigned char)(v + half);
        }
        d[0] = (unsigned char)(r + half);
    }
    static __device__ __forceinline__ i64 centered(u64 a) { return gl::centered(a); }
    static __device__ __forceinline__ u64 add(u64 a, u64 b) { return gl::add(a, b); }
    static __device__ __forceinline__ u64 sub(u64 a, u64 b) { return gl::sub(a, b); }
    static __device__ __forceinline__ u64 neg(u64 a) { return gl::neg(a); }
    static __device__ __forceinline__ u64 mul_small(i64 cf, u64 v) { return gl::mul(gl::from_i64(cf), v); }
    static __device__ __forceinline__ u64 reduce96(uint32_t hi, u64 lo) { return gl::reduce96(hi, lo); }
    static __device__ __forceinline__ u64 reduce128(u64 hi, u64 lo) { return gl::reduce128(hi, lo); }
};

}  // namespace

// ------------------------------------------------------------------------------------- launchers
namespace bmi {

#define BMI_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int launch_bsk_to_ntt(const u64 *std_polys, u64 *ntt_polys, const u64 *g_tw, uint32_t n_polys, hipStream_t s) {
    hipLaunchKernelGGL(k_bsk_to_ntt, dim3((n_polys + 3) / 4), dim3(256), 0, s, std_polys, ntt_polys, g_tw, n_polys);
    BMI_LAUNCH_CHECK();
    return 0;
}

int launch_negacyclic_mul(const u64 *a, const u64 *b, u64 *c, const u64 *g_tw, uint32_t count, hipStream_t s) {
    hipLaunchKernelGGL(k_negacyclic_mul, dim3((count + 3) / 4), dim3(256), 0, s, a, b, c, g_tw, count);
    BMI_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_tp(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const u64 *bsk,
                           const u64 *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    if (count == 0) return 0;
    constexpr int CTS = BMI_TP_CTS;
    hipLaunchKernelGGL((k_blind_rotate_tp<CTS>), dim3((count + CTS - 1) / CTS), dim3(128 * CTS), 0, s, small_cts,
                       lut_ids, luts, bsk, g_tw, out, count, n);
    BMI_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_lat(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const u64 *bsk,
                            const u64 *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    if (count == 0) return 0;
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)LAT_LDS_WORDS * sizeof(u64);
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(k_blind_rotate_lat), lds, configured)) return rc;
    hipLaunchKernelGGL(k_blind_rotate_lat, dim3(count), dim3(LAT_THREADS), lds, s, small_cts, lut_ids, luts, bsk, g_tw, out,
                       count, n);
    BMI_LAUNCH_CHECK();
    return 0;
}

int launch_keyswitch(const u64 *in, const u64 *ksk, const u64 *ks_bias, u64 *out, void *partial, uint32_t slices,
                     uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels, uint32_t base_log, uint32_t ks_stride,
                     hipStream_t s) {
    return ksl::launch_keyswitch<FieldG>(in, ksk, ks_bias, out, partial, slices, count, n, big_n, levels, base_log,
                                         ks_stride, s);
}

int launch_ksk_to_limbs(const u64 *ksk, signed char *limbs, uint32_t rows, uint32_t n, uint32_t ks_stride, hipStream_t s) {
    return ksm::launch_ksk_to_limbs<FieldG>(ksk, limbs, rows, n, ks_stride, KS_LIMBS, s);
}

int launch_keyswitch_mfma(const u64 *in, const signed char *limbs, signed char *digits, int *sums, u64 *out,
                          uint32_t slices, uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels,
                          uint32_t base_log, hipStream_t s) {
    return ksm::launch_keyswitch<FieldG, KS_LIMBS>(in, limbs, digits, sums, out, slices, count, n, big_n, levels, base_log, s);
}

int launch_lincomb(const u64 *store, const uint32_t *row_ptr, const uint32_t *idx, const i64 *coef,
                   const u64 *const_body, u64 *out, uint32_t count, uint32_t width, hipStream_t s) {
    return ksl::launch_lincomb<FieldG>(store, row_ptr, idx, coef, const_body, out, count, width, s);
}

// store[rows[i]] = src[i] for i < count (rows of `width` words): the executor's level buffer -> recycled store rows
__global__ void __launch_bounds__(256) k_scatter_rows(const u64 *__restrict__ src, u64 *__restrict__ store,
                                                      const uint32_t *__restrict__ rows, uint32_t width) {
    const u64 *s = src + (size_t)blockIdx.x * width;
    u64 *d = store + (size_t)rows[blockIdx.x] * width;
    for (uint32_t x = threadIdx.x; x < width; x += 256) d[x] = s[x];
}

int launch_scatter_rows(const u64 *src, u64 *store, const uint32_t *rows, uint32_t count, uint32_t width, hipStream_t s) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_scatter_rows, dim3(count), dim3(256), 0, s, src, store, rows, width);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace bmi
