// HIP kernels (gfx950) for the 49-bit ciphertext field (q = 2^49 - 720895): the same pipeline as bmi_kernels.hip
// with exact integer arithmetic carried in f64 (field49.hpp, ntt_wave_f64.hpp).  Ciphertexts, keyswitch key and
// linear combinations stay canonical 64-bit integers in HBM; the bootstrap key (NTT domain), the twiddles and
// the test polynomials are stored as centred doubles so that the hot loop does no conversions.
//
//   k_bsk_to_ntt49        standard-domain GGSW rows (u64) -> NTT domain (f64), lane layout
//   k_blind_rotate_tp49   THROUGHPUT: one pair of wavefronts per ciphertext (see bmi_kernels.hip for the scheme)
//   k_blind_rotate_lat49  LATENCY: one workgroup of 8 wavefronts per ciphertext
//   keyswitch / lincomb   ks_lincomb.hpp with the Field49 policy
#include <hip/hip_runtime.h>

#include "bmi_internal.hpp"
#include "ks_lincomb.hpp"
#include "ntt_wave_f64.hpp"

using f49::i64;
using f49::u64;
using namespace nttf;

namespace {

__device__ __forceinline__ void stage_twiddles(double *lds_tw, const double *__restrict__ g_tw) {
    for (int i = threadIdx.x; i < TW_WORDS; i += blockDim.x) lds_tw[i] = g_tw[i];
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_bsk_to_ntt49(const u64 *__restrict__ std_polys, double *__restrict__ ntt_polys,
                                                      const double *__restrict__ g_tw, uint32_t n_polys) {
    __shared__ double lds[TW_WORDS + 4 * SCRATCH_WORDS];
    stage_twiddles(lds, g_tw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t poly = blockIdx.x * 4 + wave;
    if (poly >= n_polys) return;
    double *scratch = lds + TW_WORDS + wave * SCRATCH_WORDS;
    double x[16];
    static_for<0, 16>([&](auto J) { x[J] = f49::to_f(std_polys[(size_t)poly * N + lane + 64 * J]); });
    forward(x, lane, lds, scratch);
    static_for<0, 16>([&](auto V) { ntt_polys[(size_t)poly * N + eval_offset(lane, V)] = f49::red(x[V]); });
}

__global__ void __launch_bounds__(256) k_negacyclic_mul49(const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                          u64 *__restrict__ c, const double *__restrict__ g_tw,
                                                          uint32_t count) {
    __shared__ double lds[TW_WORDS + 4 * SCRATCH_WORDS];
    stage_twiddles(lds, g_tw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t p = blockIdx.x * 4 + wave;
    if (p >= count) return;
    double *scratch = lds + TW_WORDS + wave * SCRATCH_WORDS;
    double x[16], y[16];
    static_for<0, 16>([&](auto J) {
        x[J] = f49::to_f(a[(size_t)p * N + lane + 64 * J]);
        y[J] = f49::to_f(b[(size_t)p * N + lane + 64 * J]);
    });
    forward(x, lane, lds, scratch);
    forward(y, lane, lds, scratch);
    static_for<0, 16>([&](auto V) { x[V] = f49::mul(x[V], f49::red(y[V])); });
    inverse(x, lane, lds, scratch);
    static_for<0, 16>([&](auto J) { c[(size_t)p * N + lane + 64 * J] = f49::to_u(x[J]); });
}

// Rounded value r = rint(v / 2^4) of a centred coefficient v (l = 3, base 2^15: 45 of the 49 bits are kept); the
// three signed digits are recovered from r with round-half-even steps (digit_lev in [-2^14, 2^14]).
__device__ __forceinline__ double digit_of(double r, int lev) {
    const double r1 = __builtin_rint(r * 0x1p-15);
    if (lev == 2) return __builtin_fma(-32768.0, r1, r);
    const double r2 = __builtin_rint(r1 * 0x1p-15);
    if (lev == 1) return __builtin_fma(-32768.0, r2, r1);
    return r2;
}

template <int CTS>
__global__ void __launch_bounds__(128 * CTS, BMI_TP49_WAVES_PER_SIMD)
    k_blind_rotate_tp49(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                        const double *__restrict__ luts, const double *__restrict__ bsk, const double *__restrict__ g_tw,
                        u64 *__restrict__ out, uint32_t count, uint32_t n) {
    constexpr int AT_WORDS = 160;
    __shared__ double lds[TW_WORDS + 2 * CTS * SCRATCH_WORDS + CTS * AT_WORDS];
    stage_twiddles(lds, g_tw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ctl = wave >> 1, c = wave & 1;
    const uint32_t ct_raw = blockIdx.x * CTS + ctl;
    const bool live = ct_raw < count;
    const uint32_t ct = live ? ct_raw : count - 1;
    double *tile = lds + TW_WORDS + wave * SCRATCH_WORDS;
    const double *ptile = lds + TW_WORDS + (wave ^ 1) * SCRATCH_WORDS;
    uint16_t *at = reinterpret_cast<uint16_t *>(lds + TW_WORDS + 2 * CTS * SCRATCH_WORDS + ctl * AT_WORDS);
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = lane + 64 * c; i <= n; i += 128) at[i] = (uint16_t)f49::modswitch(lwe[i], LOG_N + 1);
    __syncthreads();

    double acc[16];
    {
        const double *tv = luts + (size_t)lut_ids[ct] * N;
        const uint32_t bt = at[n];
        static_for<0, 16>([&](auto J) {
            const uint32_t e = (lane + 64 * J + bt) & (2 * N - 1);
            const double v = tv[e & (N - 1)];
            acc[J] = c ? ((e & N) ? -v : v) : 0.0;
        });
    }

    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];
        const double *bsk_i = bsk + (size_t)i * 12 * N;
        wave_sync();
        static_for<0, 16>([&](auto J) { tile[lane + 64 * J] = acc[J]; });
        wave_sync();
        double r[16];
        static_for<0, 16>([&](auto J) {
            const uint32_t e = (lane + 64 * J + 2 * N - a_t) & (2 * N - 1);
            double v = tile[e & (N - 1)];
            v = (e & N) ? -v : v;
            r[J] = __builtin_rint(f49::red(v - acc[J]) * 0x1p-4);
        });
        double accn[16];
        static_for<0, 3>([&](auto LEV) {
            constexpr int lev = LEV;
            double x[16];
            static_for<0, 16>([&](auto J) { x[J] = digit_of(r[J], lev); });
            forward(x, lane, lds, tile);
            wave_sync();
            static_for<0, 16>([&](auto V) { tile[eval_offset(lane, V)] = x[V]; });
            // bootstrap-key rows of this level are requested BEFORE the barrier: their L2 latency overlaps the wait
            const double *row_own = bsk_i + ((size_t)(c * 3 + lev) * 2 + c) * N;
            const double *row_par = bsk_i + ((size_t)((c ^ 1) * 3 + lev) * 2 + c) * N;
            double2 bo[8], bp[8];
            static_for<0, 8>([&](auto VP) {
                bo[VP] = reinterpret_cast<const double2 *>(row_own)[VP * 64 + lane];
                bp[VP] = reinterpret_cast<const double2 *>(row_par)[VP * 64 + lane];
            });
            __syncthreads();
            static_for<0, 8>([&](auto VP) {
                const double2 xp = reinterpret_cast<const double2 *>(ptile)[VP * 64 + lane];
                const double s0 = f49::mul(x[2 * VP], bo[VP].x) + f49::mul(xp.x, bp[VP].x);      // lazy: <= 1.6p per level
                const double s1 = f49::mul(x[2 * VP + 1], bo[VP].y) + f49::mul(xp.y, bp[VP].y);
                if constexpr (lev == 0) {
                    accn[2 * VP] = s0;
                    accn[2 * VP + 1] = s1;
                } else {
                    accn[2 * VP] += s0;
                    accn[2 * VP + 1] += s1;
                }
            });
            __syncthreads();
        });
        static_for<0, 16>([&](auto V) { accn[V] = f49::red(accn[V]); });
        inverse(accn, lane, lds, tile);
        static_for<0, 16>([&](auto J) { acc[J] = f49::red(acc[J] + accn[J]); });
    }

    if (!live) return;
    u64 *o = out + (size_t)ct * (N + 1);
    if (c == 0) {
        static_for<0, 16>([&](auto J) {
            const uint32_t m = lane + 64 * J;
            if (m == 0) o[0] = f49::to_u(acc[J]);
            else o[N - m] = f49::to_u(-acc[J]);
        });
    } else if (lane == 0) {
        o[N] = f49::to_u(acc[0]);
    }
}

constexpr int LAT_THREADS = 512;
constexpr int LAT_LDS_WORDS = TW_WORDS + 2 * N + 6 * SCRATCH_WORDS + 2 * N + 160;

__global__ void __launch_bounds__(LAT_THREADS)
    k_blind_rotate_lat49(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                         const double *__restrict__ luts, const double *__restrict__ bsk, const double *__restrict__ g_tw,
                         u64 *__restrict__ out, uint32_t count, uint32_t n) {
    extern __shared__ double lds[];
    double *acc = lds + TW_WORDS;          // [2][N], natural coefficient order, centred (<= p/2)
    double *tiles = acc + 2 * N;           // [6][SCRATCH_WORDS]
    double *Y = tiles + 6 * SCRATCH_WORDS; // [2][N], evaluation layout
    uint16_t *at = reinterpret_cast<uint16_t *>(Y + 2 * N);
    stage_twiddles(lds, g_tw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += LAT_THREADS) at[i] = (uint16_t)f49::modswitch(lwe[i], LOG_N + 1);
    __syncthreads();
    {
        const double *tv = luts + (size_t)lut_ids[ct] * N;
        const uint32_t bt = at[n];
        for (int m = tid; m < N; m += LAT_THREADS) {
            const uint32_t e = (m + bt) & (2 * N - 1);
            const double v = tv[e & (N - 1)];
            acc[m] = 0.0;
            acc[N + m] = (e & N) ? -v : v;
        }
    }
    __syncthreads();

    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];
        if (a_t == 0) continue;  // uniform over the workgroup
        const double *bsk_i = bsk + (size_t)i * 12 * N;
        double b[4][6];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int slot = tid + LAT_THREADS * m, oc = slot >> LOG_N, idx = slot & (N - 1);
#pragma unroll
            for (int r = 0; r < 6; r++) b[m][r] = bsk_i[(size_t)(r * 2 + oc) * N + idx];
        }
        if (wave < 6) {
            const int c = wave / 3, lev = wave - 3 * c;
            const double *a = acc + c * N;
            double x[16];
            static_for<0, 16>([&](auto J) {
                const uint32_t mm = lane + 64 * J;
                const uint32_t e = (mm + 2 * N - a_t) & (2 * N - 1);
                double v = a[e & (N - 1)];
                v = (e & N) ? -v : v;
                x[J] = digit_of(__builtin_rint(f49::red(v - a[mm]) * 0x1p-4), lev);
            });
            double *tile = tiles + wave * SCRATCH_WORDS;
            forward(x, lane, lds, tile);
            wave_sync();
            static_for<0, 16>([&](auto V) { tile[eval_offset(lane, V)] = x[V]; });
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int slot = tid + LAT_THREADS * m, idx = slot & (N - 1);
            double y = 0.0;  // lazy sum of six products (<= 4.8p), one reduction
#pragma unroll
            for (int r = 0; r < 6; r++) y += f49::mul(tiles[r * SCRATCH_WORDS + idx], b[m][r]);
            Y[slot] = f49::red(y);
        }
        __syncthreads();
        if (wave < 2) {
            double x[16];
            static_for<0, 16>([&](auto V) { x[V] = Y[wave * N + eval_offset(lane, V)]; });
            double *tile = tiles + wave * SCRATCH_WORDS;
            inverse(x, lane, lds, tile);
            double *a = acc + wave * N;
            static_for<0, 16>([&](auto J) { a[lane + 64 * J] = f49::red(a[lane + 64 * J] + x[J]); });
        }
        __syncthreads();
    }
    u64 *o = out + (size_t)ct * (N + 1);
    for (int m = tid; m < N; m += LAT_THREADS) {
        if (m == 0) {
            o[0] = f49::to_u(acc[0]);
            o[N] = f49::to_u(acc[N]);
        } else {
            o[N - m] = f49::to_u(-acc[m]);
        }
    }
}

struct Field49 {
    static __device__ __forceinline__ void digits(u64 a, uint32_t levels, uint32_t base_log, unsigned char *d) {
        // centred lift, every rounding round-half-to-even, digits in [-B/2, B/2]
        const i64 half = (i64)1 << (base_log - 1);
        i64 r = f49::rne_shift(f49::centered(a), f49::QBITS - levels * base_log);
        for (int lev = (int)levels - 1; lev >= 1; lev--) {
            const i64 rn = f49::rne_shift(r, base_log);
            d[lev] = (unsigned char)(r - (rn << base_log) + half);
            r = rn;
        }
        d[0] = (unsigned char)(r + half);
    }
    static __device__ __forceinline__ u64 add(u64 a, u64 b) { return f49::addq(a, b); }
    static __device__ __forceinline__ u64 sub(u64 a, u64 b) { return f49::subq(a, b); }
    static __device__ __forceinline__ u64 neg(u64 a) { return f49::negq(a); }
    static __device__ __forceinline__ u64 mul_small(i64 cf, u64 v) { return f49::mulq(f49::from_i64(cf), v); }
    static __device__ __forceinline__ u64 reduce96(uint32_t hi, u64 lo) {
        return (u64)((((unsigned __int128)hi << 64) | lo) % f49::Q);
    }
    static __device__ __forceinline__ u64 reduce128(u64 hi, u64 lo) {
        return (u64)((((unsigned __int128)hi << 64) | lo) % f49::Q);
    }
};

}  // namespace

namespace bmi49 {

#define BMI49_LAUNCH_CHECK()                    \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int launch_bsk_to_ntt(const u64 *std_polys, double *ntt_polys, const double *g_tw, uint32_t n_polys, hipStream_t s) {
    hipLaunchKernelGGL(k_bsk_to_ntt49, dim3((n_polys + 3) / 4), dim3(256), 0, s, std_polys, ntt_polys, g_tw, n_polys);
    BMI49_LAUNCH_CHECK();
    return 0;
}

int launch_negacyclic_mul(const u64 *a, const u64 *b, u64 *c, const double *g_tw, uint32_t count, hipStream_t s) {
    hipLaunchKernelGGL(k_negacyclic_mul49, dim3((count + 3) / 4), dim3(256), 0, s, a, b, c, g_tw, count);
    BMI49_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_tp(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk,
                           const double *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    if (count == 0) return 0;
    constexpr int CTS = BMI_TP49_CTS;
    hipLaunchKernelGGL((k_blind_rotate_tp49<CTS>), dim3((count + CTS - 1) / CTS), dim3(128 * CTS), 0, s, small_cts, lut_ids,
                       luts, bsk, g_tw, out, count, n);
    BMI49_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_lat(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk,
                            const double *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    if (count == 0) return 0;
    static bool attr_set = false;
    const size_t lds = (size_t)LAT_LDS_WORDS * sizeof(double);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_blind_rotate_lat49),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_blind_rotate_lat49, dim3(count), dim3(LAT_THREADS), lds, s, small_cts, lut_ids, luts, bsk, g_tw,
                       out, count, n);
    BMI49_LAUNCH_CHECK();
    return 0;
}

int launch_keyswitch(const u64 *in, const u64 *ksk, const u64 *ks_bias, u64 *out, void *partial, uint32_t slices,
                     uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels, uint32_t base_log, uint32_t ks_stride,
                     hipStream_t s) {
    return ksl::launch_keyswitch<Field49>(in, ksk, ks_bias, out, partial, slices, count, n, big_n, levels, base_log,
                                          ks_stride, s);
}

int launch_lincomb(const u64 *store, const uint32_t *row_ptr, const uint32_t *idx, const i64 *coef,
                   const u64 *const_body, u64 *out, uint32_t count, uint32_t width, hipStream_t s) {
    return ksl::launch_lincomb<Field49>(store, row_ptr, idx, coef, const_body, out, count, width, s);
}

}  // namespace bmi49
