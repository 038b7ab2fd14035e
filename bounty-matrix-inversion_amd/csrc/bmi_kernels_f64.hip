// HIP kernels (gfx950) for the 49-bit ciphertext field (q = 2^49 - 720895): the same pipeline as bmi_kernels.hip
// with exact integer arithmetic carried in f64 (field49.hpp, ntt_wave_f64.hpp).  Ciphertexts, keyswitch key and
// linear combinations stay canonical 64-bit integers in HBM; the bootstrap key (NTT domain), the twiddles and
// the test polynomials are stored as centred doubles so that the hot loop does no conversions.
//
//   k_bsk_to_ntt49           standard-domain GGSW rows (u64) -> NTT domain (f64), lane layout
//   k_blind_rotate_tpx49     THROUGHPUT: one pair of wavefronts per ciphertext, one exchange per CMUX
//   k_blind_rotate_lat2_49   LATENCY: one workgroup of 16 wavefronts per ciphertext, two wavefronts per transform
//   k_blind_rotate_wide49 / wide49u / quad49   N = 2048 (plain, unrolled key) and N = 4096
//   keyswitch / lincomb      ks_lincomb.hpp with the Field49 policy
// (the unrolled N = 1024 kernel lives in bmi_kernels_f64u.hip; the predecessors k_blind_rotate_tp49 / lat49 - kernel variants
// 1 and 4 - are A/B builds only: ab/bmi_kernels_f64_ab.inc, `make ab`)
#include <hip/hip_runtime.h>

#include "bmi_internal.hpp"
#include "dec49.hpp"
#include "ks_lincomb.hpp"
#include "ks_mfma.hpp"
#include "ntt_half_f64.hpp"
#include "ntt_wave_f64.hpp"

using f49::i64;
using f49::u64;
using namespace nttf;
using dec49::Dec;
using dec49::digit_of;
using dec49::round_half_up;

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

// Phase timing (debug build only: make prof): wall-clock cycles per phase of one wavefront, see tools/phase_prof.py
#ifdef BMI_PHASE_PROF
__device__ unsigned long long g_phase[128];
#define PH_DECL() unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = clock64()
#define PH_MARK(k)                               \
    do {                                         \
        const unsigned long long t_ = clock64(); \
        ph_[k] += t_ - tl_;                      \
        tl_ = t_;                                \
    } while (0)
#define PH_FLUSH()                                                           \
    do {                                                                     \
        if (blockIdx.x == gridDim.x / 2 && lane == 0)                        \
            for (int k_ = 0; k_ < 8; k_++) g_phase[wave * 8 + k_] = ph_[k_]; \
    } while (0)
#else
#define PH_DECL()
#define PH_MARK(k)
#define PH_FLUSH()
#endif

// Peels the least significant remaining digit off r: returns it and leaves floor(r / 2^15 + 1/2) in r.
// Walking the levels 2, 1, 0 this way keeps ONE array live (digit_of() from r would keep r, r1 and r2).
__device__ __forceinline__ double peel_digit(double &r) {
    const double rn = round_half_up(r, 0x1p-15);
    const double d = __builtin_fma(-32768.0, rn, r);
    r = rn;
    return d;
}

#ifdef BMI_AB_KERNELS
#include "ab/bmi_kernels_f64_ab.inc"
#endif

// THROUGHPUT, second form: one exchange per CMUX instead of three.
// Wavefront c of a pair owns INPUT polynomial c of the CMUX: it decomposes rot(acc_c) - acc_c, transforms the three
// digit polynomials and multiplies each with BOTH output columns of its three GGSW rows, so it ends the levels with
// a partial sum for its own component and one for the partner's.  The partner's partial goes through the tile once,
// guarded by a pair of LDS counters (publish / consumed) that only the two wavefronts of the pair poll: the exchange
// needs no workgroup barrier (one every BMI_TPX49_RESYNC iterations only keeps the pairs on the same key rows, which
// they share through L1), so the workgroup can be as large as
// the CU (8 wavefronts share one copy of the twiddle tables, which leaves LDS room for the accumulators: the
// accumulator lives in LDS between iterations, not in registers).
constexpr int TPX_CTS = 4;
constexpr int TPX_AT_WORDS = BMI_AT_WORDS;
constexpr int TPX_LDS_WORDS = TW_WORDS + 2 * TPX_CTS * (SCRATCH_WORDS + N) + TPX_CTS * TPX_AT_WORDS + 4 * TPX_CTS;
static_assert(TPX_LDS_WORDS <= BMI_LDS_WORDS_MAX, "TPX_LDS_WORDS exceeds the 160 KB of LDS");

__device__ __forceinline__ void pair_post(uint32_t *flag, uint32_t v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The poll loop is one opaque asm block: as C++ control flow it splits the loop body into several blocks and the
// register allocator then spills ~180 dwords per lane (measured); all lanes read the same LDS word.
__device__ __forceinline__ void pair_wait(uint32_t *flag, uint32_t v) {
#if BMI_TPX49_SYNC == 1
    (void)flag;
    (void)v;
    __syncthreads();
#else
    const uint32_t addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)flag;
    uint32_t tmp;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %0, %1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32 vcc, %2, %0\n\t"
        "s_cbranch_vccnz 2f\n\t"
        "s_sleep 1\n\t"
        "s_branch 1b\n"
        "2:"
        : "=&v"(tmp)
        : "v"(addr), "s"(v)
        : "vcc", "memory");
#endif
}

// keeps memory operations and the machine scheduler from moving work across this point (register pressure control)
__device__ __forceinline__ void pin() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int PF, int L = 3, int BG = 15>
__global__ void __launch_bounds__(128 * TPX_CTS)
    k_blind_rotate_tpx49(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                         const double *__restrict__ luts, const double *__restrict__ bsk,
                         const double *__restrict__ g_tw, u64 *__restrict__ out, uint32_t count, uint32_t n) {
    constexpr int CTS = TPX_CTS;
    extern __shared__ double lds[];
    double *tiles = lds + TW_WORDS;
    double *accs = tiles + 2 * CTS * SCRATCH_WORDS;
    double *at_base = accs + 2 * CTS * N;
    uint32_t *flags = reinterpret_cast<uint32_t *>(at_base + CTS * TPX_AT_WORDS);  // [2 CTS] published, [2 CTS] consumed
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ctl = wave >> 1, c = wave & 1;
    if (threadIdx.x < 4 * CTS) flags[threadIdx.x] = 0;
    stage_twiddles(lds, g_tw);
    const uint32_t ct_raw = blockIdx.x * CTS + ctl;
    const bool live = ct_raw < count;
    const uint32_t ct = live ? ct_raw : count - 1;
    double *tile = tiles + wave * SCRATCH_WORDS;
    const double *ptile = tiles + (wave ^ 1) * SCRATCH_WORDS;
    double *accl = accs + wave * N;
    uint16_t *at = reinterpret_cast<uint16_t *>(at_base + ctl * TPX_AT_WORDS);
    uint32_t *f_pub = flags + wave, *f_pub_partner = flags + (wave ^ 1);
    uint32_t *f_ack = flags + 2 * CTS + wave, *f_ack_partner = flags + 2 * CTS + (wave ^ 1);
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = lane + 64 * c; i <= n; i += 128) at[i] = (uint16_t)f49::modswitch(lwe[i], LOG_N + 1);
    __syncthreads();
    {
        const double *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at[n];
        static_for<0, 16>([&](auto J) {
            const uint32_t e = (lane + 64 * J + bt) & (2 * N - 1);
            const double v = tv[e & (N - 1)];
            accl[lane + 64 * J] = c ? ((e & N) ? -v : v) : 0.0;
        });
    }

    for (uint32_t i = 0; i < n; i++) {
#if BMI_TPX49_RESYNC
        if (i % BMI_TPX49_RESYNC == 0) __syncthreads();  // keeps the four pairs on the same key rows (shared through L1)
#endif
        const uint32_t a_t = at[i];
        const double *bsk_c = bsk + ((size_t)i * 4 * L + c * 2 * L) * N;  // this wavefront's L GGSW rows (two columns each)
#if BMI_TPX49_PRIO == 1
        __builtin_amdgcn_s_setprio(1);
#elif BMI_TPX49_PRIO == 2 || BMI_TPX49_PRIO == 4
        __builtin_amdgcn_s_setprio(0);
#elif BMI_TPX49_PRIO == 3
        __builtin_amdgcn_s_setprio(3);
#endif
        wave_sync();
        double r[16];
        {
            double vr[16], vs[16];  // all 32 reads in flight before the first use
            static_for<0, 16>([&](auto J) {
                vr[J] = accl[(lane + 64 * J + 2 * N - a_t) & (N - 1)];
                vs[J] = accl[lane + 64 * J];
            });
            sched_fence();
            static_for<0, 16>([&](auto J) {
                const uint32_t e = (lane + 64 * J + 2 * N - a_t) & (2 * N - 1);
                const double v = (e & N) ? -vr[J] : vr[J];
                r[J] = round_half_up(f49::red(v - vs[J]), Dec<L, BG>::SC);
            });
        }
        double am[16], ao[16];  // partial sums: own component, partner's component
        static_for<0, L>([&](auto LEV) {
            constexpr int lev = L - 1 - LEV;  // least significant digit first
            const double2 *row_m = reinterpret_cast<const double2 *>(bsk_c + (size_t)(lev * 2 + c) * N);
            const double2 *row_o = reinterpret_cast<const double2 *>(bsk_c + (size_t)(lev * 2 + (c ^ 1)) * N);
            double2 bm[8], bo[8];
            // PF = 10 * (where the own-column row is requested) + (where the partner-column row is requested):
            // 0 before the transform, 1 before its second DFT16, 2 before its quad transpose, 3 after it
            constexpr int PM = PF / 10, PO = PF % 10;
            auto load_m = [&]() { static_for<0, 8>([&](auto VP) { bm[VP] = row_m[VP * 64 + lane]; }); };
            auto load_o = [&]() { static_for<0, 8>([&](auto VP) { bo[VP] = row_o[VP * 64 + lane]; }); };
            pin();
#if BMI_TPX49_PRIO == 3
            __builtin_amdgcn_s_setprio(lev + 1);
#endif
            if constexpr (PM == 0) load_m();
            if constexpr (PO == 0) load_o();
            double x[16];
            static_for<0, 16>([&](auto J) { x[J] = lev == 0 ? r[J] : Dec<L, BG>::peel(r[J]); });
            forward(
                x, lane, lds, tile,
                [&]() {
                    if constexpr (PM == 1) load_m();
                    if constexpr (PO == 1) load_o();
                },
                [&]() {
                    if constexpr (PM == 2) load_m();
                    if constexpr (PO == 2) load_o();
                });
            pin();
            if constexpr (PM == 3) load_m();
            if constexpr (PO == 3) load_o();
            static_for<0, 8>([&](auto VP) {
                const double m0 = f49::mul(x[2 * VP], bm[VP].x), m1 = f49::mul(x[2 * VP + 1], bm[VP].y);
                if constexpr (lev == L - 1) {
                    am[2 * VP] = m0;
                    am[2 * VP + 1] = m1;
                } else {
                    am[2 * VP] += m0;
                    am[2 * VP + 1] += m1;
                }
            });
            static_for<0, 8>([&](auto VP) {
                const double o0 = f49::mul(x[2 * VP], bo[VP].x), o1 = f49::mul(x[2 * VP + 1], bo[VP].y);
                if constexpr (lev == L - 1) {
                    ao[2 * VP] = o0;
                    ao[2 * VP + 1] = o1;
                } else {
                    ao[2 * VP] += o0;
                    ao[2 * VP + 1] += o1;
                }
            });
            pin();
        });
        // exchange: the partner's partial goes through this wavefront's tile (free since the last forward transform)
        wave_sync();
        static_for<0, 8>([&](auto VP) {
            reinterpret_cast<double2 *>(tile)[VP * 64 + lane] = double2{ao[2 * VP], ao[2 * VP + 1]};
        });
        pair_post(f_pub, i + 1);
#if BMI_TPX49_PRIO == 1 || BMI_TPX49_PRIO == 3
        __builtin_amdgcn_s_setprio(0);
#elif BMI_TPX49_PRIO == 2
        __builtin_amdgcn_s_setprio(1);
#endif
        pair_wait(f_pub_partner, i + 1);
        static_for<0, 8>([&](auto VP) {
            const double2 p = reinterpret_cast<const double2 *>(ptile)[VP * 64 + lane];
            am[2 * VP] = f49::red(am[2 * VP] + p.x);          // <= 2 * 3 * 1.3p before the reduction
            am[2 * VP + 1] = f49::red(am[2 * VP + 1] + p.y);
        });
        pair_post(f_ack, i + 1);          // release: the reads above have landed
        pair_wait(f_ack_partner, i + 1);  // the partner has read this tile: the inverse transform may overwrite it
#if BMI_TPX49_PRIO == 4
        __builtin_amdgcn_s_setprio(1);
#endif
        inverse(am, lane, lds, tile);
        static_for<0, 16>([&](auto J) { accl[lane + 64 * J] = f49::red(accl[lane + 64 * J] + am[J]); });
    }

    if (!live) return;
    wave_sync();
    u64 *o = out + (size_t)ct * (N + 1);
    if (c == 0) {
        static_for<0, 16>([&](auto J) {
            const uint32_t m = lane + 64 * J;
            const double v = accl[m];
            if (m == 0) o[0] = f49::to_u(v);
            else o[N - m] = f49::to_u(-v);
        });
    } else if (lane == 0) {
        o[N] = f49::to_u(accl[0]);
    }
}



// ------------------------------------------------------------------------------------------------------------------
// LATENCY kernel, second form: one workgroup of 16 wavefronts per ciphertext, every transform split over two
// wavefronts by parity (ntt_half_f64.hpp).  Per CMUX:
//   A  wavefronts 0..11  = (input polynomial c, level, parity): rotate/decompose 512 coefficients, half transform -> tile
//   B  all 1,024 threads = (output polynomial o, slot p): form A_lo = E + O', A_hi = E - O' of the six digit transforms
//      on the fly, multiply with the key (own copy in slot order, k_bsk_to_lat49), write S = Y_lo + Y_hi and Y_lo - Y_hi
//   C  wavefronts 0..3   = (o, parity): inverse half transform, accumulator update
// The accumulator is kept de-interleaved in LDS (acc[c][parity][512]) so that every access of a wavefront is contiguous.
constexpr int L2_THREADS = 1024;
constexpr int L2_LDS_WORDS = ntth::HT_WORDS + 2 * N + 12 * ntth::HSCRATCH + 2 * N + BMI_AT_WORDS;   // 12 = 4 x (L = 3) tiles
static_assert(L2_LDS_WORDS <= BMI_LDS_WORDS_MAX, "L2_LDS_WORDS exceeds the 160 KB of LDS");

// PAIRED: the two words of a slot, A_lo[p] and A_hi[p], are stored side by side ([512][2]) so that the consumer requests them
// as ONE 16-byte word: a compute unit takes in 47-50 B per cycle with 16-byte requests against 29 with 8-byte ones
// (tools/microbench/cu_intake.hip), which is what the unrolled kernel's 295 KB of key per step need.
template <bool PAIRED>
__global__ void __launch_bounds__(256) k_bsk_to_lat49(const u64 *__restrict__ std_polys, double *__restrict__ lat_polys,
                                                      const double *__restrict__ g_tw_h, uint32_t n_polys) {
    __shared__ double lds[ntth::HT_WORDS + 4 * ntth::HSCRATCH];
    for (int i = threadIdx.x; i < ntth::HT_WORDS; i += blockDim.x) lds[i] = g_tw_h[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = wave & 1;
    const uint32_t poly = blockIdx.x * 2 + (wave >> 1);   // two polynomials per workgroup, two wavefronts each
    double *tile = lds + ntth::HT_WORDS + wave * ntth::HSCRATCH;
    if (poly < n_polys) {
        double x[8];
        static_for<0, 8>([&](auto J) { x[J] = f49::to_f(std_polys[(size_t)poly * N + 2 * (lane + 64 * J) + h]); });
        if (h) ntth::forward_half<true>(x, lane, lds, tile);
        else ntth::forward_half<false>(x, lane, lds, tile);
        wave_sync();
        static_for<0, 8>([&](auto R) { tile[R * 64 + lane] = x[R]; });
    }
    __syncthreads();
    if (poly < n_polys) {
        const double *te = lds + ntth::HT_WORDS + (wave & ~1) * ntth::HSCRATCH, *to = te + ntth::HSCRATCH;
        double *o = lat_polys + (size_t)poly * N;
        static_for<0, 4>([&](auto Q4) {
            const int p = (h * 4 + Q4) * 64 + lane;   // the pair's 128 lanes cover the 512 slots in 4 steps
            const double e = te[p], od = to[p];
            if constexpr (PAIRED) {
                reinterpret_cast<double2 *>(o)[p] = make_double2(f49::red(e + od), f49::red(e - od));
            } else {
                o[p] = f49::red(e + od);
                o[ntth::HALF + p] = f49::red(e - od);
            }
        });
    }
}

template <int L = 3, int BG = 15>
__global__ void __launch_bounds__(L2_THREADS)
    k_blind_rotate_lat2_49(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                           const double *__restrict__ luts, const double *__restrict__ bsk_lat,
                           const double *__restrict__ g_tw_h, u64 *__restrict__ out, uint32_t count, uint32_t n) {
    extern __shared__ double lds[];
    double *acc = lds + ntth::HT_WORDS;              // [2 components][2 parities][512], centred (<= q/2 + 2)
    double *tiles = acc + 2 * N;                     // [12][HSCRATCH]
    double *SD = tiles + 12 * ntth::HSCRATCH;        // [2 outputs][sum, difference][512]
    uint16_t *at = reinterpret_cast<uint16_t *>(SD + 2 * N);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < ntth::HT_WORDS; i += L2_THREADS) lds[i] = g_tw_h[i];
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

    PH_DECL();
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];
        if (a_t == 0) continue;  // uniform over the workgroup
        const double *bi = bsk_lat + (size_t)i * 4 * L * N;
        PH_MARK(7);
        double b[2 * L][2];
#pragma unroll
        for (int r = 0; r < 2 * L; r++) {
            b[r][0] = bi[(size_t)(r * 2 + mo) * N + mp];
            b[r][1] = bi[(size_t)(r * 2 + mo) * N + ntth::HALF + mp];
        }
        if (wave < 4 * L) {
            // (tasks -> SIMDs so that no SIMD gets three of the heavier odd halves: measured 4 % slower)
            const int c = wave / (2 * L), lev = (wave % (2 * L)) >> 1, h = wave & 1;
            const int pz = wave >> 1;
            const double *ac = acc + c * N;
            double x[8];
#if BMI_LAT2_PRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            static_for<0, 8>([&](auto J) {
                const uint32_t m = lane + 64 * J;
                const uint32_t e = (2 * m + h + 2 * N - a_t) & (2 * N - 1);
                const uint32_t n2 = e & (N - 1);
                double v = ac[(n2 & 1) * ntth::HALF + (n2 >> 1)];
                v = (e & N) ? -v : v;
                x[J] = Dec<L, BG>::digit(round_half_up(f49::red(v - ac[h * ntth::HALF + m]), Dec<L, BG>::SC), lev);
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
            double ylo = 0.0, yhi = 0.0;  // lazy sums of 2 L <= six products (<= 10.2 q)
#pragma unroll
            for (int r = 0; r < 2 * L; r++) {
                const double e = tiles[(2 * r) * ntth::HSCRATCH + mp], od = tiles[(2 * r + 1) * ntth::HSCRATCH + mp];
                ylo += f49::mul(e + od, b[r][0]);
                yhi += f49::mul(e - od, b[r][1]);
            }
            ylo = f49::red(ylo);
            yhi = f49::red(yhi);
            SD[(mo * 2 + 0) * ntth::HALF + mp] = ylo + yhi;
            SD[(mo * 2 + 1) * ntth::HALF + mp] = ylo - yhi;
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
        for (int k_ = 0; k_ < 8; k_++) g_phase[wave * 8 + k_] = ph_[k_];
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


// ------------------------------------------------------------------------------------------------------------------
// Second parameter set, N = 2048 (k = 1, l = 3): a 2048-point negacyclic transform is two 1024-point wave transforms
// (nttf::forward / inverse on the even and the odd coefficients, root psi_4096^2 = the 2048th root they are built on)
// combined exactly like the two halves of ntt_half_f64.hpp:  A[kk] = E[kk] + T[kk] O[kk],  A[kk + 1024] = E - T O,
// T = psi_4096^(2 kk + 1) in the slot order of the wave transform.  One workgroup of 8 wavefronts per ciphertext;
// per CMUX the twelve half-transform tasks of the six digit polynomials run in two rounds, eight (GGSW rows 0-3: two
// per SIMD) and four (rows 4-5: one per SIMD) - three task times on the critical path where rounds of six took four;
// every thread multiply-accumulates four (output, slot) items against the key (own copy in slot order, already
// scaled by 1/2 for the split), four wavefronts run the inverse halves; the sums handed to them live in the first
// four tiles (rows padded like a tile, so each inverse task transposes in the row it has just read).
// Twice the polynomial at the same n doubles the look-up boxes: 4-bit look-ups sit at 12.5 sigma instead of 6.2.
constexpr int W_N = 2 * N;                 // 2048
constexpr int W_THREADS = 512;
constexpr int W_T = TW_WORDS;              // T table [reg][lane] (1024 words), then T^-1 (1024 words)
constexpr int W_LDS_WORDS = TW_WORDS + 2 * N + 2 * W_N + 8 * SCRATCH_WORDS + BMI_AT_WORDS;
static_assert(W_LDS_WORDS <= BMI_LDS_WORDS_MAX, "W_LDS_WORDS exceeds the 160 KB of LDS");

// Issue priority of a wavefront (0..3).  Tasks that share a SIMD start at 3 and step down as they advance: the one that
// is ahead yields issue slots to the others, so they finish together instead of the last one running its tail alone
// (measured on the N = 1024 latency kernel: 4.02 -> 3.57 ms per bootstrap).
template <int P>
__device__ __forceinline__ void prio() {
#if BMI_LAT2_PRIO
    __builtin_amdgcn_s_setprio(P);
#endif
}
struct PrioStep1 {
    __device__ __forceinline__ void operator()() const { prio<1>(); }
};
struct PrioStep0 {
    __device__ __forceinline__ void operator()() const { prio<0>(); }
};

template <bool PRIO = false>
__device__ __forceinline__ void wide_forward_task(double (&x)[16], int h, int lane, const double *lds, double *tile) {
    if constexpr (PRIO) {
        prio<2>();
        forward(x, lane, lds, tile, PrioStep1(), PrioStep0());
    } else {
        forward(x, lane, lds, tile);
    }
    if (h) static_for<0, 16>([&](auto V) { x[V] = f49::mul(x[V], lds[W_T + V * 64 + lane]); });
    wave_sync();
    static_for<0, 16>([&](auto V) { tile[V * 64 + lane] = f49::red(x[V]); });
}

__global__ void __launch_bounds__(128) k_bsk_to_wide49(const u64 *__restrict__ std_polys, double *__restrict__ wide_polys,
                                                       const double *__restrict__ g_tw, const double *__restrict__ g_tw_wide,
                                                       uint32_t n_polys) {
    __shared__ double lds[TW_WORDS + N + 2 * SCRATCH_WORDS];
    for (int i = threadIdx.x; i < TW_WORDS; i += blockDim.x) lds[i] = g_tw[i];
    for (int i = threadIdx.x; i < N; i += blockDim.x) lds[W_T + i] = g_tw_wide[i];
    __syncthreads();
    const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t poly = blockIdx.x;      // one polynomial (2048 coefficients) per workgroup of two wavefronts
    double *tile = lds + TW_WORDS + N + h * SCRATCH_WORDS;
    double x[16];
    static_for<0, 16>([&](auto J) { x[J] = f49::to_f(std_polys[(size_t)poly * W_N + 2 * (lane + 64 * J) + h]); });
    wide_forward_task(x, h, lane, lds, tile);
    __syncthreads();
    const double *te = lds + TW_WORDS + N, *to = te + SCRATCH_WORDS;
    constexpr double INV2 = f49::centred_c((f49::Q + 1) / 2);   // 1/2 mod q
    double *o = wide_polys + (size_t)poly * W_N;
    // A[p], A[p + 1024] side by side: the kernel requests them as ONE 16-byte word (47-50 B per cycle into a compute unit
    // against 29 with 8-byte requests, tools/microbench/cu_intake.hip)
    for (int p = threadIdx.x; p < N; p += blockDim.x)
        reinterpret_cast<double2 *>(o)[p] = make_double2(f49::red(f49::mul(te[p] + to[p], INV2)), f49::red(f49::mul(te[p] - to[p], INV2)));
}

template <int L = 3, int BG = 15>
__global__ void __launch_bounds__(W_THREADS)
    k_blind_rotate_wide49(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                          const double *__restrict__ luts, const double *__restrict__ bsk_wide,
                          const double *__restrict__ g_tw, const double *__restrict__ g_tw_wide, u64 *__restrict__ out,
                          uint32_t count, uint32_t n) {
    extern __shared__ double lds[];
    double *acc = lds + TW_WORDS + 2 * N;            // [2 components][2 parities][1024]
    double *tiles = acc + 2 * W_N;                   // [8][SCRATCH_WORDS]; rows 0-3 also carry the sums to the inverse
    uint16_t *at = reinterpret_cast<uint16_t *>(tiles + 8 * SCRATCH_WORDS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TW_WORDS; i += W_THREADS) lds[i] = g_tw[i];
    for (int i = tid; i < 2 * N; i += W_THREADS) lds[W_T + i] = g_tw_wide[i];
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += W_THREADS) at[i] = (uint16_t)f49::modswitch(lwe[i], LOG_N + 2);
    __syncthreads();
    {
        const double *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * W_N;
        const uint32_t bt = at[n];
        for (uint32_t nn = tid; nn < (uint32_t)W_N; nn += W_THREADS) {
            const uint32_t e = (nn + bt) & (2 * W_N - 1);
            const double v = tv[e & (W_N - 1)];
            acc[(nn & 1) * N + (nn >> 1)] = 0.0;
            acc[W_N + (nn & 1) * N + (nn >> 1)] = (e & W_N) ? -v : v;
        }
    }
    __syncthreads();

    PH_DECL();
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];
        if (a_t == 0) continue;  // uniform over the workgroup
        const double *bi = bsk_wide + (size_t)i * 4 * L * W_N;
        PH_MARK(7);
        double ylo[4], yhi[4];
#pragma unroll
        for (int q = 0; q < 4; q++) ylo[q] = yhi[q] = 0.0;
        // one round: GGSW rows R0 .. R0 + NR - 1 (row = 3 * input polynomial + level), 2 NR half-transform tasks
        auto round = [&](auto R0_, auto NR_) {
            constexpr int R0 = R0_, NR = NR_;
            double b[4][NR][2];            // key words of this thread's four (output, slot) items
            // One CU takes its key words in at ~29 B per cycle with 8-byte requests, ~48 with 16-byte ones (131 KB in the first
            // round; the (A[p], A[p + 1024]) pairs are stored side by side for that), and a wavefront cannot run
            // ahead of a load it has not been able to issue: the rows are requested in stages between the pieces of
            // the task instead of all up front (BMI_WIDE_STAGE: 0 = all first, 1 = two stages, 2 = four).
            auto load_rows = [&](auto RA_, auto RB_) {
                static_for<RA_, RB_>([&](auto R) {
                    constexpr int r = R;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int idx = tid + W_THREADS * q, o = idx >> 10, p = idx & (N - 1);
                        const double2 w = reinterpret_cast<const double2 *>(bi + (size_t)((R0 + r) * 2 + o) * W_N)[p];
                        b[q][r][0] = w.x;
                        b[q][r][1] = w.y;
                    }
                });
            };
            using IC0 = std::integral_constant<int, 0>;
            using ICN = std::integral_constant<int, NR>;
            constexpr int S1 = BMI_WIDE_STAGE == 0 ? NR : (BMI_WIDE_STAGE == 1 ? NR / 2 : NR / 4);   // rows requested first
            constexpr int S2 = BMI_WIDE_STAGE == 2 ? NR / 2 : NR;                                    // ... by the end of the decomposition
            constexpr int S3 = BMI_WIDE_STAGE == 2 ? (3 * NR + 3) / 4 : NR;                          // ... by the middle of the transform
            if (wave < 2 * NR) {
                load_rows(IC0(), std::integral_constant<int, S1>());
                pin();
                const int row = R0 + (wave >> 1), c = row / L, lev = row % L, h = wave & 1;
                const double *ac = acc + c * W_N;
                double x[16];
                prio<3>();
                static_for<0, 16>([&](auto J) {
                    const uint32_t m = lane + 64 * J;
                    const uint32_t e = (2 * m + h + 2 * W_N - a_t) & (2 * W_N - 1);
                    const uint32_t n2 = e & (W_N - 1);
                    double v = ac[(n2 & 1) * N + (n2 >> 1)];
                    v = (e & W_N) ? -v : v;
                    x[J] = Dec<L, BG>::digit(round_half_up(f49::red(v - ac[h * N + m]), Dec<L, BG>::SC), lev);
                });
                pin();
                load_rows(std::integral_constant<int, S1>(), std::integral_constant<int, S2>());
                pin();
                double *tile = tiles + wave * SCRATCH_WORDS;
                prio<2>();
                forward(
                    x, lane, lds, tile,
                    [&]() {
                        prio<1>();
                        load_rows(std::integral_constant<int, S2>(), std::integral_constant<int, S3>());
                    },
                    [&]() {
                        prio<0>();
                        load_rows(std::integral_constant<int, S3>(), ICN());
                    });
                if (h) static_for<0, 16>([&](auto V) { x[V] = f49::mul(x[V], lds[W_T + V * 64 + lane]); });
                wave_sync();
                static_for<0, 16>([&](auto V) { tile[V * 64 + lane] = f49::red(x[V]); });
            } else {
                load_rows(IC0(), ICN());
            }
            PH_MARK(0);
            __syncthreads();
            PH_MARK(1);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int p = (tid + W_THREADS * q) & (N - 1);
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const double e = tiles[(2 * r) * SCRATCH_WORDS + p], od = tiles[(2 * r + 1) * SCRATCH_WORDS + p];
                    ylo[q] += f49::mul(e + od, b[q][r][0]);   // e, od reduced (<= q/2): lazy sums of six products
                    yhi[q] += f49::mul(e - od, b[q][r][1]);
                }
            }
            PH_MARK(2);
            __syncthreads();   // the tiles are rewritten by the next round / the sums below
            PH_MARK(3);
        };
        // 2 L GGSW rows: rounds of at most four (eight half-transform tasks, one per wavefront)
        if constexpr (L == 3) {
            round(std::integral_constant<int, 0>(), std::integral_constant<int, 4>());
            round(std::integral_constant<int, 4>(), std::integral_constant<int, 2>());
        } else {
            round(std::integral_constant<int, 0>(), std::integral_constant<int, 2 * L>());
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = tid + W_THREADS * q, o = idx >> 10, p = idx & (N - 1);
            const double lo = f49::red(ylo[q]), hi = f49::red(yhi[q]);
            tiles[(o * 2 + 0) * SCRATCH_WORDS + p] = f49::red(lo + hi);   // inverse() takes |.| <= 0.51 q
            tiles[(o * 2 + 1) * SCRATCH_WORDS + p] = lo - hi;             // reduced by the multiplication with T^-1
        }
        __syncthreads();
        PH_MARK(4);
        if (wave < 4) {
            const int o = wave >> 1, h = wave & 1;
            double *tile = tiles + wave * SCRATCH_WORDS;   // row (o, h): read into registers, then the transpose scratch
            double x[16];
            static_for<0, 16>([&](auto V) { x[V] = tile[V * 64 + lane]; });
            if (h) static_for<0, 16>([&](auto V) { x[V] = f49::mul(x[V], lds[W_T + N + V * 64 + lane]); });
            wave_sync();
            inverse(x, lane, lds, tile);
            double *ao = acc + o * W_N + h * N;
            static_for<0, 16>([&](auto J) { ao[lane + 64 * J] = f49::red(ao[lane + 64 * J] + x[J]); });
        }
        PH_MARK(5);
        __syncthreads();
        PH_MARK(6);
    }
#ifdef BMI_PHASE_PROF
    if (blockIdx.x == 0 && lane == 0)
        for (int k_ = 0; k_ < 8; k_++) g_phase[wave * 8 + k_] = ph_[k_];
#endif
    u64 *o = out + (size_t)ct * (W_N + 1);
    for (uint32_t nn = tid; nn < (uint32_t)W_N; nn += W_THREADS) {
        const double a0 = acc[(nn & 1) * N + (nn >> 1)];
        if (nn == 0) {
            o[0] = f49::to_u(a0);
            o[W_N] = f49::to_u(acc[W_N]);
        } else {
            o[W_N - nn] = f49::to_u(-a0);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// N = 2048 with the UNROLLED bootstrap key (bmi_set_bsk_unroll(ctx, 2); scheme in bmi_kernels_f64u.hip): the structure of
// k_blind_rotate_wide49 with one step per PAIR of LWE coefficients.  The forward tasks decompose ACC itself (no rotation) and
// run once per pair; per round of GGSW rows the multiply phase walks the three keys of the pair (the next key's rows are
// requested while this one is multiplied), reduces each key's partial sum and scales it by  psi^(e c_j) - 1  (psi = psi_4096,
// e = 2 kk + 1 the root of the slot, + 2048 for the upper half: a sign; 2,048 root powers in LDS) before it joins the single
// accumulator pair of the item - three accumulator pairs per item would not fit the registers next to two keys' rows.
constexpr int WU_LDS_WORDS = W_LDS_WORDS + W_N;
static_assert(WU_LDS_WORDS <= BMI_LDS_WORDS_MAX, "WU_LDS_WORDS exceeds the 160 KB of LDS");

template <int L = 3, int BG = 15>
__global__ void __launch_bounds__(W_THREADS)
    k_blind_rotate_wide49u(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                           const double *__restrict__ luts, const double *__restrict__ bsk3_wide,
                           const double *__restrict__ g_tw, const double *__restrict__ g_tw_wide,
                           const double *__restrict__ g_root_pow, u64 *__restrict__ out, uint32_t count, uint32_t n) {
    extern __shared__ double lds[];
    double *acc = lds + TW_WORDS + 2 * N;            // [2 components][2 parities][1024]
    double *tiles = acc + 2 * W_N;                   // [8][SCRATCH_WORDS]; rows 0-3 also carry the sums to the inverse
    uint16_t *at = reinterpret_cast<uint16_t *>(tiles + 8 * SCRATCH_WORDS);
    double *RP = tiles + 8 * SCRATCH_WORDS + BMI_AT_WORDS;   // psi_4096^x, x in [0, 2048); psi^(x + 2048) = -psi^x
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TW_WORDS; i += W_THREADS) lds[i] = g_tw[i];
    for (int i = tid; i < 2 * N; i += W_THREADS) lds[W_T + i] = g_tw_wide[i];
    for (int i = tid; i < W_N; i += W_THREADS) RP[i ^ ((i >> 5) & 31)] = g_root_pow[i];   // folded against bank conflicts
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += W_THREADS) at[i] = (uint16_t)f49::modswitch(lwe[i], LOG_N + 2);
    __syncthreads();
    {
        const double *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * W_N;
        const uint32_t bt = at[n];
        for (uint32_t nn = tid; nn < (uint32_t)W_N; nn += W_THREADS) {
            const uint32_t e = (nn + bt) & (2 * W_N - 1);
            const double v = tv[e & (W_N - 1)];
            acc[(nn & 1) * N + (nn >> 1)] = 0.0;
            acc[W_N + (nn & 1) * N + (nn >> 1)] = (e & W_N) ? -v : v;
        }
    }
    __syncthreads();
    // root exponents of this thread's four (output, slot) items: slot p = V * 64 + l of the wave transform holds the value at
    // psi_4096^(2 kk + 1), kk = (l >> 2) + 16 (4 (V >> 2) + (l & 3)) + 256 (V & 3) (the table T of the even / odd combination)
    uint32_t root_e[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t p = (tid + W_THREADS * q) & (N - 1), l = p & 63, V = p >> 6;
        root_e[q] = 2 * ((l >> 2) + 16 * (4 * (V >> 2) + (l & 3)) + 256 * (V & 3)) + 1;
    }
    const uint32_t pairs = (n + 1) >> 1;

    for (uint32_t ip = 0; ip < pairs; ip++) {
        const uint32_t a1 = at[2 * ip], a2 = (2 * ip + 1 < n) ? at[2 * ip + 1] : 0u;
        if ((a1 | a2) == 0) continue;  // uniform over the workgroup
        const uint32_t cj[3] = {(a1 + a2) & (2 * W_N - 1), a1, a2};
        const double *bi = bsk3_wide + (size_t)ip * 3 * 4 * L * W_N;   // [3 keys][2 L rows][2 outputs][1024 slots][A, A + 1024]
        double ylo[4], yhi[4];
#pragma unroll
        for (int q = 0; q < 4; q++) ylo[q] = yhi[q] = 0.0;
        // one round: GGSW rows R0 .. R0 + NR - 1 (row = 3 * input polynomial + level), 2 NR half-transform tasks
        auto round = [&](auto R0_, auto NR_) {
            constexpr int R0 = R0_, NR = NR_;
            double b[4][NR][2];   // key 0's words of this thread's four (output, slot) items
            auto load_rows = [&](double (&dst)[4][NR][2], int key, auto RA_, auto RB_) {
                static_for<RA_, RB_>([&](auto R) {
                    constexpr int r = R;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int idx = tid + W_THREADS * q, o = idx >> 10, p = idx & (N - 1);
                        const double2 w = reinterpret_cast<const double2 *>(bi + ((size_t)key * 4 * L + (R0 + r) * 2 + o) * W_N)[p];
                        dst[q][r][0] = w.x;
                        dst[q][r][1] = w.y;
                    }
                });
            };
            using IC0 = std::integral_constant<int, 0>;
            using ICN = std::integral_constant<int, NR>;
            constexpr int S1 = NR / 4, S2 = NR / 2, S3 = (3 * NR + 3) / 4;   // key 0's rows are requested in four stages (BMI_WIDE_STAGE 2)
            if (wave < 2 * NR) {
                load_rows(b, 0, IC0(), std::integral_constant<int, S1>());
                pin();
                const int row = R0 + (wave >> 1), c = row / L, lev = row % L, h = wave & 1;
                const double *ac = acc + c * W_N + h * N;
                double x[16];
                prio<3>();
                static_for<0, 16>([&](auto J) { x[J] = Dec<L, BG>::digit(round_half_up(ac[lane + 64 * J], Dec<L, BG>::SC), lev); });
                pin();
                load_rows(b, 0, std::integral_constant<int, S1>(), std::integral_constant<int, S2>());
                pin();
                double *tile = tiles + wave * SCRATCH_WORDS;
                prio<2>();
                forward(
                    x, lane, lds, tile,
                    [&]() {
                        prio<1>();
                        load_rows(b, 0, std::integral_constant<int, S2>(), std::integral_constant<int, S3>());
                    },
                    [&]() {
                        prio<0>();
                        load_rows(b, 0, std::integral_constant<int, S3>(), ICN());
                    });
                if (h) static_for<0, 16>([&](auto V) { x[V] = f49::mul(x[V], lds[W_T + V * 64 + lane]); });
                wave_sync();
                static_for<0, 16>([&](auto V) { tile[V * 64 + lane] = f49::red(x[V]); });
            } else {
                load_rows(b, 0, IC0(), ICN());
            }
            __syncthreads();
            // key 0's rows are all here (b); the rows of keys 1 and 2 arrive in chunks of two, the next chunk requested while
            // this one is multiplied (two keys' worth of rows next to b do not fit the registers: 544 bytes of spills per lane)
            constexpr int CR = 2, NCH = NR / CR, NS = 2 * NCH;   // chunk s = (key 1 + s / NCH, rows of chunk s % NCH)
            double ck[2][4][CR][2];
            auto load_chunk = [&](auto S) {
                constexpr int sidx = S, key = 1 + sidx / NCH, ch = sidx % NCH;
#pragma unroll
                for (int r = 0; r < CR; r++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int idx = tid + W_THREADS * q, o = idx >> 10, p = idx & (N - 1);
                        const double2 w = reinterpret_cast<const double2 *>(bi + ((size_t)key * 4 * L + (R0 + ch * CR + r) * 2 + o) * W_N)[p];
                        ck[sidx & 1][q][r][0] = w.x;
                        ck[sidx & 1][q][r][1] = w.y;
                    }
            };
            auto scale = [&](double (&plo)[4], double (&phi)[4], uint32_t c) {   // key's partial sums * (psi^(e c) - 1) -> accumulators
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t xe = root_e[q] * c;                         // exponent mod 4096: bit 11 is a sign
                    const uint32_t xi = xe & (W_N - 1);
                    double w = RP[xi ^ ((xi >> 5) & 31)];
                    w = (xe & W_N) ? -w : w;
                    const double wh = (c & 1) ? -w : w;                        // the root of the upper half is -psi^e
                    ylo[q] += f49::mul(plo[q], w - 1.0);   // lazy sums of <= four reduced products go into the product as they are
                    yhi[q] += f49::mul(phi[q], wh - 1.0);
                }
            };
            load_chunk(std::integral_constant<int, 0>());
            {
                double plo[4], phi[4];   // lazy sums of NR <= four products
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int p = (tid + W_THREADS * q) & (N - 1);
                    plo[q] = phi[q] = 0.0;
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        const double e = tiles[(2 * r) * SCRATCH_WORDS + p], od = tiles[(2 * r + 1) * SCRATCH_WORDS + p];
                        plo[q] += f49::mul(e + od, b[q][r][0]);
                        phi[q] += f49::mul(e - od, b[q][r][1]);
                    }
                }
                scale(plo, phi, cj[0]);
            }
            double plo[4], phi[4];
            static_for<0, NS>([&](auto S) {
                constexpr int sidx = S, key = 1 + sidx / NCH, ch = sidx % NCH;
                if constexpr (sidx + 1 < NS) load_chunk(std::integral_constant<int, sidx + 1>());
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int p = (tid + W_THREADS * q) & (N - 1);
                    if constexpr (ch == 0) plo[q] = phi[q] = 0.0;
#pragma unroll
                    for (int r = 0; r < CR; r++) {
                        const int rr = ch * CR + r;
                        const double e = tiles[(2 * rr) * SCRATCH_WORDS + p], od = tiles[(2 * rr + 1) * SCRATCH_WORDS + p];
                        plo[q] += f49::mul(e + od, ck[sidx & 1][q][r][0]);
                        phi[q] += f49::mul(e - od, ck[sidx & 1][q][r][1]);
                    }
                }
                if constexpr (ch == NCH - 1) scale(plo, phi, cj[key]);
            });
            __syncthreads();   // the tiles are rewritten by the next round / the sums below
        };
        // 2 L <= four GGSW rows: one round (eight half-transform tasks, one per wavefront)
        static_assert(L <= 2, "see above");
        round(std::integral_constant<int, 0>(), std::integral_constant<int, 2 * L>());
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = tid + W_THREADS * q, o = idx >> 10, p = idx & (N - 1);
            const double lo = f49::red(ylo[q]), hi = f49::red(yhi[q]);
            tiles[(o * 2 + 0) * SCRATCH_WORDS + p] = f49::red(lo + hi);   // inverse() takes |.| <= 0.51 q
            tiles[(o * 2 + 1) * SCRATCH_WORDS + p] = lo - hi;             // reduced by the multiplication with T^-1
        }
        __syncthreads();
        if (wave < 4) {
            const int o = wave >> 1, h = wave & 1;
            double *tile = tiles + wave * SCRATCH_WORDS;   // row (o, h): read into registers, then the transpose scratch
            double x[16];
            static_for<0, 16>([&](auto V) { x[V] = tile[V * 64 + lane]; });
            if (h) static_for<0, 16>([&](auto V) { x[V] = f49::mul(x[V], lds[W_T + N + V * 64 + lane]); });
            wave_sync();
            inverse(x, lane, lds, tile);
            double *ao = acc + o * W_N + h * N;
            static_for<0, 16>([&](auto J) { ao[lane + 64 * J] = f49::red(ao[lane + 64 * J] + x[J]); });
        }
        __syncthreads();
    }
    u64 *o = out + (size_t)ct * (W_N + 1);
    for (uint32_t nn = tid; nn < (uint32_t)W_N; nn += W_THREADS) {
        const double a0 = acc[(nn & 1) * N + (nn >> 1)];
        if (nn == 0) {
            o[0] = f49::to_u(a0);
            o[W_N] = f49::to_u(acc[W_N]);
        } else {
            o[W_N - nn] = f49::to_u(-a0);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Third parameter set, N = 4096: the same construction one level deeper.  a is split by index mod 4 into four
// 1024-coefficient parts a_j; with psi = psi_8192 (psi^4 is the 2048th root of the wave transform) and S_j = NTT1024(a_j):
//     A[kk + 1024 t] = sum_j i^(t j) T_j[kk] S_j[kk],   T_j = psi^((2 kk + 1) j),   i = psi^2048 (i^2 = -1),
// i.e. a 4-point DFT (one multiplication by i) of the twisted parts, formed where the values are consumed.  Per CMUX
// three rounds of two GGSW rows (8 part-transform tasks = one per wavefront, 8 LDS tiles); every thread owns four
// (output, slot) items with four accumulators each; the inverse 4-point DFT is applied by the same threads, the sums go
// through the tile region to eight inverse part-transform tasks.  The key copy is in slot order, scaled by 1/4.
constexpr int Q_N = 4 * N;                 // 4096
constexpr int Q_THREADS = 512;
constexpr int Q_LDS_WORDS = TW_WORDS + 2 * Q_N + 8 * SCRATCH_WORDS + BMI_AT_WORDS;
static_assert(Q_LDS_WORDS <= BMI_LDS_WORDS_MAX, "Q_LDS_WORDS exceeds the 160 KB of LDS");
constexpr double I4 = f49::centred_c(f49::powmod_c(f49::GEN, (f49::Q - 1) / 4));          // psi_8192^2048
constexpr double I4_INV = f49::centred_c(f49::powmod_c(f49::GEN, 3 * ((f49::Q - 1) / 4)));  // -i
static_assert(f49::mulmod_c(f49::powmod_c(f49::GEN, (f49::Q - 1) / 4), f49::powmod_c(f49::GEN, (f49::Q - 1) / 4)) == f49::Q - 1, "i^2 = -1");

// part transform of part j (0..3): forward, twist by T_j (j >= 1; g_t = tables [3][1024] in (reg, lane) order), store reduced
template <bool PRIO = false>
__device__ __forceinline__ void quad_forward_task(double (&x)[16], int j, int lane, const double *lds, const double *g_t,
                                                  double *tile) {
    double t[16];   // requested before the transform that hides the latency (part 0 reads T_1 and ignores it)
    const double *tp = g_t + (j ? j - 1 : 0) * N;
    static_for<0, 16>([&](auto V) { t[V] = tp[V * 64 + lane]; });
    sched_fence();
    if constexpr (PRIO) {
        prio<2>();
        forward(x, lane, lds, tile, PrioStep1(), PrioStep0());
    } else {
        forward(x, lane, lds, tile);
    }
    if (j) static_for<0, 16>([&](auto V) { x[V] = f49::mul(x[V], t[V]); });
    wave_sync();
    static_for<0, 16>([&](auto V) { tile[V * 64 + lane] = f49::red(x[V]); });
}

// the four values A_t (t = 0..3) of one slot from the four twisted parts (each <= q/2): |A_t| <= 2.6 q
__device__ __forceinline__ void dft4_parts(double s0, double s1, double s2, double s3, double (&a)[4]) {
    const double u = s0 + s2, v = s0 - s2, w = s1 + s3, z = f49::mul(s1 - s3, I4);
    a[0] = u + w;
    a[1] = v + z;
    a[2] = u - w;
    a[3] = v - z;
}

__global__ void __launch_bounds__(256) k_bsk_to_quad49(const u64 *__restrict__ std_polys, double *__restrict__ quad_polys,
                                                       const double *__restrict__ g_tw, const double *__restrict__ g_t,
                                                       uint32_t n_polys) {
    __shared__ double lds[TW_WORDS + 4 * SCRATCH_WORDS];
    for (int i = threadIdx.x; i < TW_WORDS; i += blockDim.x) lds[i] = g_tw[i];
    __syncthreads();
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t poly = blockIdx.x;      // one polynomial (4096 coefficients) per workgroup of four wavefronts
    double *tiles = lds + TW_WORDS;
    double x[16];
    static_for<0, 16>([&](auto J) { x[J] = f49::to_f(std_polys[(size_t)poly * Q_N + 4 * (lane + 64 * J) + j]); });
    quad_forward_task(x, j, lane, lds, g_t, tiles + j * SCRATCH_WORDS);
    __syncthreads();
    constexpr double INV4 = f49::centred_c(f49::powmod_c(4, f49::Q - 2));
    double *o = quad_polys + (size_t)poly * Q_N;
    for (int p = threadIdx.x; p < N; p += blockDim.x) {
        double a[4];
        dft4_parts(tiles[p], tiles[SCRATCH_WORDS + p], tiles[2 * SCRATCH_WORDS + p], tiles[3 * SCRATCH_WORDS + p], a);
#pragma unroll
        for (int t = 0; t < 4; t++) o[t * N + p] = f49::red(f49::mul(a[t], INV4));
    }
}

__global__ void __launch_bounds__(Q_THREADS)
    k_blind_rotate_quad49(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids,
                          const double *__restrict__ luts, const double *__restrict__ bsk_quad,
                          const double *__restrict__ g_tw, const double *__restrict__ g_t, u64 *__restrict__ out,
                          uint32_t count, uint32_t n) {
    extern __shared__ double lds[];
    double *acc = lds + TW_WORDS;                    // [2 components][4 parts][1024]
    double *tiles = acc + 2 * Q_N;                   // [8][SCRATCH_WORDS]; also takes the 2 x 4 x 1024 sums before the inverses
    uint16_t *at = reinterpret_cast<uint16_t *>(tiles + 8 * SCRATCH_WORDS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TW_WORDS; i += Q_THREADS) lds[i] = g_tw[i];
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += Q_THREADS) at[i] = (uint16_t)f49::modswitch(lwe[i], LOG_N + 3);
    __syncthreads();
    {
        const double *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * Q_N;
        const uint32_t bt = at[n];
        for (uint32_t nn = tid; nn < (uint32_t)Q_N; nn += Q_THREADS) {
            const uint32_t e = (nn + bt) & (2 * Q_N - 1);
            const double v = tv[e & (Q_N - 1)];
            acc[(nn & 3) * N + (nn >> 2)] = 0.0;
            acc[Q_N + (nn & 3) * N + (nn >> 2)] = (e & Q_N) ? -v : v;
        }
    }
    __syncthreads();

    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];
        if (a_t == 0) continue;  // uniform over the workgroup
        const double *bi = bsk_quad + (size_t)i * 12 * Q_N;
        double y[4][4];          // [item][t]
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int t = 0; t < 4; t++) y[q][t] = 0.0;
#pragma unroll 1
        for (int rnd = 0; rnd < 3; rnd++) {     // GGSW rows 2 rnd, 2 rnd + 1
            {
                const int row = 2 * rnd + (wave >> 2), c = row / 3, lev = row % 3, j = wave & 3;
                const double *ac = acc + c * Q_N;
                double x[16];
                prio<3>();
                static_for<0, 16>([&](auto J) {
                    const uint32_t m = lane + 64 * J;
                    const uint32_t e = (4 * m + j + 2 * Q_N - a_t) & (2 * Q_N - 1);
                    const uint32_t n2 = e & (Q_N - 1);
                    double v = ac[(n2 & 3) * N + (n2 >> 2)];
                    v = (e & Q_N) ? -v : v;
                    x[J] = digit_of(round_half_up(f49::red(v - ac[j * N + m]), 0x1p-4), lev);
                });
                quad_forward_task<true>(x, j, lane, lds, g_t, tiles + wave * SCRATCH_WORDS);
            }
            __syncthreads();
            {
                // key words of item q + 1 are requested while item q is multiplied (16 words in flight per thread)
                auto key_ptr = [&](int q, int rr) {
                    const int idx = tid + Q_THREADS * q, o = idx >> 10, p = idx & (N - 1);
                    return bi + (size_t)((2 * rnd + rr) * 2 + o) * Q_N + p;
                };
                double bc[2][4], bn[2][4];
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int t = 0; t < 4; t++) bc[rr][t] = key_ptr(0, rr)[t * N];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int p = (tid + Q_THREADS * q) & (N - 1);
                    if (q < 3) {
#pragma unroll
                        for (int rr = 0; rr < 2; rr++)
#pragma unroll
                            for (int t = 0; t < 4; t++) bn[rr][t] = key_ptr(q + 1, rr)[t * N];
                    }
                    sched_fence();
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) {
                        const double *tb = tiles + (4 * rr) * SCRATCH_WORDS + p;
                        double a[4];
                        dft4_parts(tb[0], tb[SCRATCH_WORDS], tb[2 * SCRATCH_WORDS], tb[3 * SCRATCH_WORDS], a);
#pragma unroll
                        for (int t = 0; t < 4; t++) y[q][t] += f49::mul(a[t], bc[rr][t]);   // six products of <= 0.83 q each
                    }
#pragma unroll
                    for (int rr = 0; rr < 2; rr++)
#pragma unroll
                        for (int t = 0; t < 4; t++) bc[rr][t] = bn[rr][t];
                }
            }
            __syncthreads();   // the tiles are rewritten by the next round
        }
        double ti[16];   // this wavefront's T_j^-1 row, requested ahead of the barrier (part 0 ignores it)
        {
            const double *tp = g_t + (3 + ((wave & 3) ? (wave & 3) - 1 : 0)) * N;
            static_for<0, 16>([&](auto V) { ti[V] = tp[V * 64 + lane]; });
        }
        // inverse 4-point DFT of the (reduced) sums; the results go through the tile region
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = tid + Q_THREADS * q, o = idx >> 10, p = idx & (N - 1);
            const double y0 = f49::red(y[q][0]), y1 = f49::red(y[q][1]), y2 = f49::red(y[q][2]), y3 = f49::red(y[q][3]);
            const double u = y0 + y2, v = y0 - y2, w = y1 + y3, z = f49::mul(y1 - y3, I4_INV);
            double *sd = tiles + (size_t)(o * 4) * N + p;    // packed [2][4][1024] inside the tile region
            sd[0] = f49::red(u + w);          // inverse() takes |.| <= 0.57 q
            sd[N] = f49::red(v + z);
            sd[2 * N] = f49::red(u - w);
            sd[3 * N] = f49::red(v - z);
        }
        __syncthreads();
        double x[16];
        const int o = wave >> 2, j = wave & 3;
        static_for<0, 16>([&](auto V) { x[V] = tiles[(size_t)(o * 4 + j) * N + V * 64 + lane]; });
        if (j) static_for<0, 16>([&](auto V) { x[V] = f49::mul(x[V], ti[V]); });
        __syncthreads();       // every wavefront holds its sums: the tile region is scratch again
        inverse(x, lane, lds, tiles + wave * SCRATCH_WORDS);
        {
            double *ao = acc + o * Q_N + j * N;
            static_for<0, 16>([&](auto J) { ao[lane + 64 * J] = f49::red(ao[lane + 64 * J] + x[J]); });
        }
        __syncthreads();
    }
    u64 *o = out + (size_t)ct * (Q_N + 1);
    for (uint32_t nn = tid; nn < (uint32_t)Q_N; nn += Q_THREADS) {
        const double a0 = acc[(nn & 3) * N + (nn >> 2)];
        if (nn == 0) {
            o[0] = f49::to_u(a0);
            o[Q_N] = f49::to_u(acc[Q_N]);
        } else {
            o[Q_N - nn] = f49::to_u(-a0);
        }
    }
}

struct Field49 {
    static __device__ __forceinline__ void digits(u64 a, uint32_t levels, uint32_t base_log, unsigned char *d) {
        // centred lift, round half up to the top levels * base_log bits, signed digits in [-B/2, B/2), top absorbs the carry
        const uint32_t shift = f49::QBITS - levels * base_log;
        const i64 B = (i64)1 << base_log, half = B >> 1;
        const i64 c = f49::centered(a);
        i64 r = (c >> shift) + ((c >> (shift - 1)) & 1);
        for (int lev = (int)levels - 1; lev >= 1; lev--) {
            i64 v = r & (B - 1);
            r >>= base_log;
            if (v >= half) { v -= B; r += 1; }
            d[lev] = (unsigned char)(v + half);
        }
        d[0] = (unsigned char)(r + half);
    }
    static __device__ __forceinline__ i64 centered(u64 a) { return f49::centered(a); }
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

#ifdef BMI_PHASE_PROF
extern "C" int bmi_debug_phase_prof(unsigned long long *out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 128);
}
#endif

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

#ifdef BMI_AB_KERNELS
int launch_blind_rotate_tp(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk,
                           const double *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    if (count == 0) return 0;
    constexpr int CTS = BMI_TP49_CTS;
    hipLaunchKernelGGL((k_blind_rotate_tp49<CTS>), dim3((count + CTS - 1) / CTS), dim3(128 * CTS), 0, s, small_cts, lut_ids,
                       luts, bsk, g_tw, out, count, n);
    BMI49_LAUNCH_CHECK();
    return 0;
}
#else
// variants 1 and 4 are A/B kernels (make ab): not in the product build
int launch_blind_rotate_tp(const u64 *, const uint32_t *, const double *, const double *, const double *, u64 *, uint32_t, uint32_t, hipStream_t) {
    return (int)hipErrorNotSupported;
}
#endif


// (levels, base log) pairs the templated kernels are instantiated for
#define BMI49_FOR_LB(levels, base_log, F)                            \
    do {                                                             \
        if ((levels) == 3 && (base_log) == 15) return F<3, 15>::go;  \
        if ((levels) == 2 && (base_log) == 15) return F<2, 15>::go;  \
        if ((levels) == 1 && (base_log) == 23) return F<1, 23>::go;  \
        return nullptr;                                              \
    } while (0)

template <int L, int BG>
struct LaunchTpx {
    static int go(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk, const double *g_tw,
                  u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
        static std::atomic<uint64_t> configured{0};
        const size_t lds = (size_t)TPX_LDS_WORDS * sizeof(double);
        auto kern = k_blind_rotate_tpx49<BMI_TPX49_PF, L, BG>;
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
        hipLaunchKernelGGL(kern, dim3((count + TPX_CTS - 1) / TPX_CTS), dim3(128 * TPX_CTS), lds, s, small_cts, lut_ids, luts,
                           bsk, g_tw, out, count, n);
        BMI49_LAUNCH_CHECK();
        return 0;
    }
};
typedef int (*launch9_t)(const u64 *, const uint32_t *, const double *, const double *, const double *, u64 *, uint32_t, uint32_t,
                         hipStream_t);
static launch9_t pick_tpx(uint32_t levels, uint32_t base_log) { BMI49_FOR_LB(levels, base_log, LaunchTpx); }

int launch_blind_rotate_tpx(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk,
                            const double *g_tw, u64 *out, uint32_t count, uint32_t n, uint32_t levels, uint32_t base_log,
                            hipStream_t s) {
    if (count == 0) return 0;
    launch9_t f = pick_tpx(levels, base_log);
    return f ? f(small_cts, lut_ids, luts, bsk, g_tw, out, count, n, s) : (int)hipErrorInvalidValue;
}




int launch_bsk_to_quad(const u64 *std_polys, double *quad_polys, const double *g_tw, const double *g_t, uint32_t n_polys,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_bsk_to_quad49, dim3(n_polys), dim3(256), 0, s, std_polys, quad_polys, g_tw, g_t, n_polys);
    BMI49_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_quad(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk_quad,
                             const double *g_tw, const double *g_t, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    if (count == 0) return 0;
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)Q_LDS_WORDS * sizeof(double);
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(k_blind_rotate_quad49), lds, configured)) return rc;
    hipLaunchKernelGGL(k_blind_rotate_quad49, dim3(count), dim3(Q_THREADS), lds, s, small_cts, lut_ids, luts, bsk_quad, g_tw,
                       g_t, out, count, n);
    BMI49_LAUNCH_CHECK();
    return 0;
}

int launch_bsk_to_wide(const u64 *std_polys, double *wide_polys, const double *g_tw, const double *g_tw_wide,
                       uint32_t n_polys, hipStream_t s) {
    hipLaunchKernelGGL(k_bsk_to_wide49, dim3(n_polys), dim3(128), 0, s, std_polys, wide_polys, g_tw, g_tw_wide, n_polys);
    BMI49_LAUNCH_CHECK();
    return 0;
}

template <int L, int BG>
struct LaunchWide {
    static int go(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk_wide, const double *g_tw,
                  const double *g_tw_wide, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
        static std::atomic<uint64_t> configured{0};
        const size_t lds = (size_t)W_LDS_WORDS * sizeof(double);
        auto kern = k_blind_rotate_wide49<L, BG>;
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
        hipLaunchKernelGGL(kern, dim3(count), dim3(W_THREADS), lds, s, small_cts, lut_ids, luts, bsk_wide, g_tw, g_tw_wide, out,
                           count, n);
        BMI49_LAUNCH_CHECK();
        return 0;
    }
};
typedef int (*launch10_t)(const u64 *, const uint32_t *, const double *, const double *, const double *, const double *, u64 *,
                          uint32_t, uint32_t, hipStream_t);
static launch10_t pick_wide(uint32_t levels, uint32_t base_log) { BMI49_FOR_LB(levels, base_log, LaunchWide); }

int launch_blind_rotate_wide(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk_wide,
                             const double *g_tw, const double *g_tw_wide, u64 *out, uint32_t count, uint32_t n,
                             uint32_t levels, uint32_t base_log, hipStream_t s) {
    if (count == 0) return 0;
    launch10_t f = pick_wide(levels, base_log);
    return f ? f(small_cts, lut_ids, luts, bsk_wide, g_tw, g_tw_wide, out, count, n, s) : (int)hipErrorInvalidValue;
}

template <int L, int BG>
struct LaunchWideU {
    static int go(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk3_wide, const double *g_tw,
                  const double *g_tw_wide, const double *g_root_pow, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
        static std::atomic<uint64_t> configured{0};
        const size_t lds = (size_t)WU_LDS_WORDS * sizeof(double);
        auto kern = k_blind_rotate_wide49u<L, BG>;
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
        hipLaunchKernelGGL(kern, dim3(count), dim3(W_THREADS), lds, s, small_cts, lut_ids, luts, bsk3_wide, g_tw, g_tw_wide,
                           g_root_pow, out, count, n);
        BMI49_LAUNCH_CHECK();
        return 0;
    }
};
typedef int (*launch11_t)(const u64 *, const uint32_t *, const double *, const double *, const double *, const double *,
                          const double *, u64 *, uint32_t, uint32_t, hipStream_t);
static launch11_t pick_wide_u(uint32_t levels, uint32_t base_log) {
    if (levels == 2 && base_log == 15) return LaunchWideU<2, 15>::go;
    if (levels == 1 && base_log == 23) return LaunchWideU<1, 23>::go;
    return nullptr;
}

int launch_blind_rotate_wide_u(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk3_wide,
                               const double *g_tw, const double *g_tw_wide, const double *g_root_pow, u64 *out, uint32_t count,
                               uint32_t n, uint32_t levels, uint32_t base_log, hipStream_t s) {
    if (count == 0) return 0;
    launch11_t f = pick_wide_u(levels, base_log);
    return f ? f(small_cts, lut_ids, luts, bsk3_wide, g_tw, g_tw_wide, g_root_pow, out, count, n, s) : (int)hipErrorInvalidValue;
}

int launch_bsk_to_lat(const u64 *std_polys, double *lat_polys, const double *g_tw_h, uint32_t n_polys, bool paired, hipStream_t s) {
    if (paired) hipLaunchKernelGGL(k_bsk_to_lat49<true>, dim3((n_polys + 1) / 2), dim3(256), 0, s, std_polys, lat_polys, g_tw_h, n_polys);
    else hipLaunchKernelGGL(k_bsk_to_lat49<false>, dim3((n_polys + 1) / 2), dim3(256), 0, s, std_polys, lat_polys, g_tw_h, n_polys);
    BMI49_LAUNCH_CHECK();
    return 0;
}

template <int L, int BG>
struct LaunchLat2 {
    static int go(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk_lat, const double *g_tw_h,
                  u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
        static std::atomic<uint64_t> configured{0};
        const size_t lds = (size_t)L2_LDS_WORDS * sizeof(double);
        auto kern = k_blind_rotate_lat2_49<L, BG>;
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
        hipLaunchKernelGGL(kern, dim3(count), dim3(L2_THREADS), lds, s, small_cts, lut_ids, luts, bsk_lat, g_tw_h, out, count, n);
        BMI49_LAUNCH_CHECK();
        return 0;
    }
};
static launch9_t pick_lat2(uint32_t levels, uint32_t base_log) { BMI49_FOR_LB(levels, base_log, LaunchLat2); }

int launch_blind_rotate_lat2(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk_lat,
                             const double *g_tw_h, u64 *out, uint32_t count, uint32_t n, uint32_t levels, uint32_t base_log,
                             hipStream_t s) {
    if (count == 0) return 0;
    launch9_t f = pick_lat2(levels, base_log);
    return f ? f(small_cts, lut_ids, luts, bsk_lat, g_tw_h, out, count, n, s) : (int)hipErrorInvalidValue;
}

#ifdef BMI_AB_KERNELS
int launch_blind_rotate_lat(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk,
                            const double *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    if (count == 0) return 0;
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)LAT_LDS_WORDS * sizeof(double);
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(k_blind_rotate_lat49), lds, configured)) return rc;
    hipLaunchKernelGGL(k_blind_rotate_lat49, dim3(count), dim3(LAT_THREADS), lds, s, small_cts, lut_ids, luts, bsk, g_tw,
                       out, count, n);
    BMI49_LAUNCH_CHECK();
    return 0;
}
#else
int launch_blind_rotate_lat(const u64 *, const uint32_t *, const double *, const double *, const double *, u64 *, uint32_t, uint32_t, hipStream_t) {
    return (int)hipErrorNotSupported;
}
#endif

int launch_keyswitch(const u64 *in, const u64 *ksk, const u64 *ks_bias, u64 *out, void *partial, uint32_t slices,
                     uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels, uint32_t base_log, uint32_t ks_stride,
                     hipStream_t s) {
    return ksl::launch_keyswitch<Field49>(in, ksk, ks_bias, out, partial, slices, count, n, big_n, levels, base_log,
                                          ks_stride, s);
}

int launch_ksk_to_limbs(const u64 *ksk, signed char *limbs, uint32_t rows, uint32_t n, uint32_t ks_stride, hipStream_t s) {
    return ksm::launch_ksk_to_limbs<Field49>(ksk, limbs, rows, n, ks_stride, KS_LIMBS, s);
}

int launch_keyswitch_mfma(const u64 *in, const signed char *limbs, signed char *digits, int *sums, u64 *out,
                          uint32_t slices, uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels,
                          uint32_t base_log, hipStream_t s) {
    return ksm::launch_keyswitch<Field49, KS_LIMBS>(in, limbs, digits, sums, out, slices, count, n, big_n, levels, base_log, s);
}

int launch_lincomb(const u64 *store, const uint32_t *row_ptr, const uint32_t *idx, const i64 *coef,
                   const u64 *const_body, u64 *out, uint32_t count, uint32_t width, hipStream_t s) {
    return ksl::launch_lincomb<Field49>(store, row_ptr, idx, coef, const_body, out, count, width, s);
}

}  // namespace bmi49
