// HIP kernel (gfx950) of the UNROLLED blind rotation on the 49-bit field (bmi_set_bsk_unroll(ctx, 2)).  A translation unit
// of its own because it is compiled with the compiler's default instruction scheduler: the rest of the f64 kernels are built
// with -amdgpu-sched-strategy=max-ilp, which costs this kernel 3 % (2.28 against 2.23 ms per bootstrap, A/B in one call).
#include <hip/hip_runtime.h>

#include <atomic>

#include "bmi_internal.hpp"
#include "dec49.hpp"
#include "ntt_half_f64.hpp"
#include "ntt_wave_f64.hpp"

using f49::i64;
using f49::u64;
using namespace nttf;
using dec49::Dec;
using dec49::round_half_up;

namespace {

// Phase timing (debug build only: make prof; tools/phase_prof.py with unroll = 2)
#ifdef BMI_PHASE_PROF
__device__ unsigned long long g_phase_u[128];
__device__ unsigned long long g_wg_times_u[1024][2];   // [workgroup][start, end] in ticks of the 100 MHz real-time counter
#define PH_DECL() unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = clock64()
#define PH_MARK(k)                               \
    do {                                         \
        const unsigned long long t_ = clock64(); \
        ph_[k] += t_ - tl_;                      \
        tl_ = t_;                                \
    } while (0)
#else
#define PH_DECL()
#define PH_MARK(k)
#endif

// ------------------------------------------------------------------------------------------------------------------
// LATENCY kernel with the UNROLLED bootstrap key (two LWE coefficients per step; oracle/tfhe_oracle.c
// ora_blind_rotate_extract_unrolled):   ACC <- ACC + sum_{j<3} (X^(c_j) - 1) (K3[i][j] [.] ACC),  c = (a + a', a, a').
// The structure of k_blind_rotate_lat2_49 with half the steps: phase A decomposes ACC itself (no rotation) and runs the
// twelve forward half transforms ONCE per pair of coefficients; phase B multiplies the six digit transforms with the three
// GGSW keys of the pair and scales each product by  psi^(e c_j) - 1,  the value of X^(c_j) - 1 at the slot's root psi^e
// (e = 2 kk + 1 for A_lo, e + N for A_hi: a look-up in a 2N-entry table of root powers), so the rotation never touches the
// coefficient domain; phase C is unchanged.  Per pair: 12 + 4 half transforms and 42 modular multiplications per thread,
// against 24 + 8 and 24 of two plain steps.  The price is noise: the key-noise term of the output variance triples.
constexpr int L2_THREADS = 1024;
// twiddles, accumulator, twelve tiles, sums / differences, mod-switched ciphertext (the layout of k_blind_rotate_lat2_49), root powers
constexpr int L2U_LDS_WORDS = ntth::HT_WORDS + 2 * N + 12 * ntth::HSCRATCH + 2 * N + BMI_AT_WORDS + 2 * N;
static_assert(L2U_LDS_WORDS <= BMI_LDS_WORDS_MAX, "L2U_LDS_WORDS exceeds the 160 KB of LDS");

template <int L = 3, int BG = 15>
__global__ void __launch_bounds__(L2_THREADS)
    k_blind_rotate_lat2u_49(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                            const double *__restrict__ luts, const double *__restrict__ bsk3_lat,
                            const double *__restrict__ g_tw_h, const double *__restrict__ g_root_pow,
                            u64 *__restrict__ out, uint32_t count, uint32_t n) {
#ifdef BMI_PHASE_PROF
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_times_u[blockIdx.x][0] = wall_clock64();
#endif
    extern __shared__ double lds[];
    double *acc = lds + ntth::HT_WORDS;              // [2 components][2 parities][512], centred (<= q/2 + 2)
    double *tiles = acc + 2 * N;                     // [12][HSCRATCH]
    double *SD = tiles + 12 * ntth::HSCRATCH;        // [2 outputs][sum, difference][512]
    uint16_t *at = reinterpret_cast<uint16_t *>(SD + 2 * N);
    double *RP = SD + 2 * N + BMI_AT_WORDS;          // psi^x, x in [0, 2N)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < ntth::HT_WORDS; i += L2_THREADS) lds[i] = g_tw_h[i];
    for (int i = tid; i < 2 * N; i += L2_THREADS) RP[i ^ ((i >> 5) & 31)] = g_root_pow[i];
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += L2_THREADS) at[i] = (uint16_t)f49::modswitch(lwe[i], LOG_N + 1);
    __syncthreads();
    {
        const double *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at[n];
        const uint32_t nn = tid;  // coefficient index
        const uint32_t e = (nn + bt) & (2 * N - 1);
        const double v = tv[e & (N - 1)];
        acc[(nn & 1) * ntth::HALF + (nn >> 1)] = 0.0;
        acc[N + (nn & 1) * ntth::HALF + (nn >> 1)] = (e & N) ? -v : v;
    }
    __syncthreads();
    const int mo = tid >> 9, mp = tid & 511;  // phase B: output polynomial, slot
    const uint32_t root_e = 2 * ntth::kk_of(mp & 63, mp >> 6) + 1;   // A_lo[mp] is the value at psi^root_e, A_hi[mp] at -psi^root_e
    const uint32_t pairs = (n + 1) >> 1;
    // this thread's 2 L x 2 words of GGSW key `key` (0..2) of pair `ip`: [2 L rows][2 outputs][512 slots][A_lo, A_hi], one
    // 16-byte request per row
    auto load_key = [&](double (&dst)[2 * L][2], uint32_t ip, int key) {
        const double *bj = bsk3_lat + ((size_t)ip * 3 + key) * 4 * L * N;
#pragma unroll
        for (int r = 0; r < 2 * L; r++) {
            const double2 w = reinterpret_cast<const double2 *>(bj + (size_t)(r * 2 + mo) * N)[mp];
            dst[r][0] = w.x;
            dst[r][1] = w.y;
        }
    };
    double b[2 * L][2];

    PH_DECL();
    for (uint32_t ip = 0; ip < pairs; ip++) {
        const uint32_t a1 = at[2 * ip], a2 = (2 * ip + 1 < n) ? at[2 * ip + 1] : 0u;
        if ((a1 | a2) == 0) continue;  // uniform over the workgroup: every factor X^0 - 1 vanishes
        const uint32_t cj[3] = {(a1 + a2) & (2 * N - 1), a1, a2};
        PH_MARK(7);
        load_key(b, ip, 0);
        if (wave < 4 * L) {
            const int c = wave / (2 * L), lev = (wave % (2 * L)) >> 1, h = wave & 1;
            const int pz = wave >> 1;
            const double *ac = acc + c * N + h * ntth::HALF;
            double x[8];
#if BMI_LAT2_PRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            static_for<0, 8>([&](auto J) {
                x[J] = Dec<L, BG>::digit(round_half_up(ac[lane + 64 * J], Dec<L, BG>::SC), lev);
            });
            double *tile = tiles + (2 * pz + h) * ntth::HSCRATCH;
            if (h) ntth::forward_half<true>(x, lane, lds, tile);
            else ntth::forward_half<false>(x, lane, lds, tile);
            wave_sync();
            static_for<0, 8>([&](auto R) { tile[R * 64 + lane] = x[R]; });
#if BMI_LAT2_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        PH_MARK(0);
        __syncthreads();
        PH_MARK(1);
        {
            double alo[2 * L], ahi[2 * L];
#pragma unroll
            for (int r = 0; r < 2 * L; r++) {
                const double e = tiles[(2 * r) * ntth::HSCRATCH + mp], od = tiles[(2 * r + 1) * ntth::HSCRATCH + mp];
                alo[r] = e + od;
                ahi[r] = e - od;
            }
            double slo = 0.0, shi = 0.0;   // sums of three reduced products (<= 1.6 q)
            auto one_key = [&](const double (&kw)[2 * L][2], uint32_t c) {
                double ylo = 0.0, yhi = 0.0;  // lazy sums of 2 L <= six reduced products (<= 3.1 q)
#pragma unroll
                for (int r = 0; r < 2 * L; r++) {
                    ylo += f49::mul(alo[r], kw[r][0]);
                    yhi += f49::mul(ahi[r], kw[r][1]);
                }
                // psi^(e c); at the root -psi^e: (-1)^c times it.  The table is stored at x ^ (bits 5..9 of x): the exponents of
                // a wavefront, odd multiples of c, share their low bits when c is even - without the fold they would meet in a
                // few LDS banks (SQ_LDS_BANK_CONFLICT: 30 % of this kernel's LDS cycles, rocprofv3 --pmc)
                const uint32_t xe = (root_e * c) & (2 * N - 1);
                const double w = RP[xe ^ ((xe >> 5) & 31)];
                const double wh = (c & 1) ? -w : w;
                slo += f49::mul(ylo, w - 1.0);   // the lazy sum goes into the product as it is (|.| <= 3.1 q: exact, like e + od above)
                shi += f49::mul(yhi, wh - 1.0);
            };
            // key 0 was requested before the forward phase; key 1 is requested now, key 2 as soon as key 0 has been consumed.
            // (Requesting more ahead was built and measured: the next pair's first key under the inverse phase or key 1 under
            // the forward phase need registers the forward tasks do not have - 76-80 bytes of spills per lane, 25 % slower;
            // key 1 right before the barrier, or keys 1 and 2 together: no gain / 5 % slower; a second copy of the step loop for
            // the four wavefronts without a forward task, which then hold all three keys from the start of a step - 124
            // registers, no spills, bit-exact: 1 % faster, not kept; those wavefronts touching every line of keys 1 and 2
            // under the forward phase so that the real requests hit the L2: 3 % SLOWER.)
            double bn[2 * L][2];
            load_key(bn, ip, 1);
            one_key(b, cj[0]);
            load_key(b, ip, 2);
            one_key(bn, cj[1]);
            one_key(b, cj[2]);
            slo = f49::red(slo);
            shi = f49::red(shi);
            SD[(mo * 2 + 0) * ntth::HALF + mp] = slo + shi;
            SD[(mo * 2 + 1) * ntth::HALF + mp] = slo - shi;
        }
        PH_MARK(2);
        __syncthreads();
        PH_MARK(3);
        if (wave < 4) {
            const int o = wave >> 1, h = wave & 1;
            double x[8];
            static_for<0, 8>([&](auto R) { x[R] = SD[(o * 2 + h) * ntth::HALF + R * 64 + lane]; });
            double *tile = tiles + wave * ntth::HSCRATCH;
            if (h) ntth::inverse_half<true>(x, lane, lds, tile);
            else ntth::inverse_half<false>(x, lane, lds, tile);
            double *ao = acc + o * N + h * ntth::HALF;
            static_for<0, 8>([&](auto J) { ao[lane + 64 * J] = f49::red(ao[lane + 64 * J] + x[J]); });
        }
        PH_MARK(4);
        __syncthreads();
        PH_MARK(5);
    }
#ifdef BMI_PHASE_PROF
    if (blockIdx.x == 0 && lane == 0)
        for (int k_ = 0; k_ < 8; k_++) g_phase_u[wave * 8 + k_] = ph_[k_];
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_times_u[blockIdx.x][1] = wall_clock64();
#endif
    u64 *o = out + (size_t)ct * (N + 1);
    {
        const uint32_t nn = tid;
        const double a0 = acc[(nn & 1) * ntth::HALF + (nn >> 1)];
        if (nn == 0) {
            o[0] = f49::to_u(a0);
            o[N] = f49::to_u(acc[N]);
        } else {
            o[N - nn] = f49::to_u(-a0);
        }
    }
}

#ifndef BMI_LAT2U_PIPE
#define BMI_LAT2U_PIPE 0   // 1: the flag-synchronised form of the kernel above (an experiment that measured slower: ab/bmi_kernels_f64u_pipe.inc)
#endif
#if BMI_LAT2U_PIPE
#include "ab/bmi_kernels_f64u_pipe.inc"
#endif

}  // namespace

#ifdef BMI_PHASE_PROF
extern "C" int bmi_debug_phase_prof_unrolled(unsigned long long *out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phase_u), sizeof(unsigned long long) * 128);
}
extern "C" int bmi_debug_wg_times_unrolled(unsigned long long *out2048) {
    return (int)hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_wg_times_u), sizeof(unsigned long long) * 2048);
}
#endif

namespace bmi49 {

#define BMI49_LAUNCH_CHECK()                    \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

template <int L, int BG>
struct LaunchLat2u {
    static int go(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk3_lat, const double *g_tw_h,
                  const double *g_root_pow, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
        static std::atomic<uint64_t> configured{0};
#if BMI_LAT2U_PIPE
        const size_t lds = (size_t)L2UP_LDS_WORDS * sizeof(double);
        auto kern = k_blind_rotate_lat2up_49<L, BG>;
#else
        const size_t lds = (size_t)L2U_LDS_WORDS * sizeof(double);
        auto kern = k_blind_rotate_lat2u_49<L, BG>;
#endif
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
        hipLaunchKernelGGL(kern, dim3(count), dim3(L2_THREADS), lds, s, small_cts, lut_ids, luts, bsk3_lat, g_tw_h, g_root_pow, out,
                           count, n);
        BMI49_LAUNCH_CHECK();
        return 0;
    }
};
typedef int (*launch10_t)(const u64 *, const uint32_t *, const double *, const double *, const double *, const double *, u64 *,
                          uint32_t, uint32_t, hipStream_t);
static launch10_t pick_lat2u(uint32_t levels, uint32_t base_log) {
    if (levels == 3 && base_log == 15) return LaunchLat2u<3, 15>::go;
    if (levels == 2 && base_log == 15) return LaunchLat2u<2, 15>::go;
    if (levels == 1 && base_log == 23) return LaunchLat2u<1, 23>::go;
    return nullptr;
}

int launch_blind_rotate_lat2u(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk3_lat,
                              const double *g_tw_h, const double *g_root_pow, u64 *out, uint32_t count, uint32_t n,
                              uint32_t levels, uint32_t base_log, hipStream_t s) {
    if (count == 0) return 0;
    launch10_t f = pick_lat2u(levels, base_log);
    return f ? f(small_cts, lut_ids, luts, bsk3_lat, g_tw_h, g_root_pow, out, count, n, s) : (int)hipErrorInvalidValue;
}

}  // namespace bmi49
