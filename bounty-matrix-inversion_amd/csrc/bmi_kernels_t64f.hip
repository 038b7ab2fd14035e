// 2^64 TORUS, throughput blind rotation with the exact products carried by a floating-point transform (gfx950).
//
// Same function as k_blind_rotate_t64<48, L, 10> (bmi_kernels_t64.hip) - bootstrap key stored at 48 bits of precision as two
// balanced 24-bit limbs, digits in base 2^10, accumulator as the exact integer word / 2^16 in a double - and the same results
// bit for bit: per limb the sum over the 2 L digit x limb polynomial products is an integer below 2^45, and it is computed here
// through the folded 512-point complex FFT of fft_wave_f64.hpp and ROUNDED TO THE NEAREST INTEGER, which returns that
// integer exactly (the transform's error on these operands stays below 2^-11, see the header; tests/test_gpu_parity.py and
// tests/test_gpu_torus_fft.py hold the kernel to the oracle's integer arithmetic).  What changes is the cost: a transform of
// 1,024 real coefficients is ~300 f64 instructions per lane instead of ~700 for the exact one mod 2^49 - 720895, and a
// multiply-accumulate is 4 fused multiply-adds per complex point (2 coefficients) instead of 6 instructions per coefficient.
//
// Structure = k_blind_rotate_t64: a pair of wavefronts per ciphertext, four ciphertexts per workgroup; wavefront c owns input
// polynomial c (decomposes it, transforms its L digit polynomials once, multiplies them with both key columns per limb),
// publishes the partner's partial sum through its LDS tile, adds the partner's, and runs the inverse transform of output c.
#include <hip/hip_runtime.h>

#include "bmi_internal.hpp"
#include "fft_half_f64.hpp"
#include "fft_wave_f64.hpp"
#include "pair_sync.hpp"
#include "t64_common.hpp"

using t64::i64;
using t64::u64;
using namespace fftw;

namespace {

#ifndef BMI_T64F_RESYNC
#define BMI_T64F_RESYNC 1   // workgroup barrier every so many CMUXes: keeps the four pairs on the same key rows, which they share through L1 (0: 83.2 ms, 4: 70.1, 1: 68.9 per 8,192)
#endif
#ifndef BMI_T64F_PRIO
#define BMI_T64F_PRIO 1     // issue priority steps down through the forward transforms (3, 2, 1), 0 in the limb loop
#endif

#ifdef BMI_PHASE_PROF   // make -C csrc prof; tools/phase_prof_t64f.py
__device__ unsigned long long g_phase_f[128];
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
using t64::f64_to_word;
using t64::Scheme;
#ifndef BMI_T64F_CTS
#define BMI_T64F_CTS 4      // ciphertexts (wavefront pairs) per workgroup: 4 fill the CU's LDS; 3 and 2 measured slower per ciphertext
#endif
#ifndef BMI_T64F_LDSKEY
#define BMI_T64F_LDSKEY 0   // TIMING-ONLY A/B (wrong results on purpose; EXPERIMENTS A15): key rows staged once per workgroup in an LDS ring
                            // by LDS-DMA (each wavefront of a column group requests a third of a row) and read back from LDS, with NO
                            // hand-off between the wavefronts - the lower bound of what any correct staging scheme costs.  Needs BMI_T64F_CTS=3.
#endif
constexpr int TF_CTS = BMI_T64F_CTS;
constexpr int TF_RING_WORDS = BMI_T64F_LDSKEY ? 2 * 2 * N + 128 : 0;   // [column group 2][slot 2][512 complex] + 1 KiB of slack
constexpr int TF_LDS_WORDS = TW_WORDS + 2 * TF_CTS * (SCRATCH_WORDS + N) + TF_CTS * BMI_AT_WORDS + 4 * TF_CTS + TF_RING_WORDS;
static_assert(TF_LDS_WORDS <= BMI_LDS_WORDS_MAX, "TF_LDS_WORDS exceeds the 160 KB of LDS");
static_assert(SCRATCH_WORDS >= N, "a tile carries 512 complex partial sums to the partner");

// standard-domain GGSW polynomials (u64 torus words, already rounded to the key precision) -> per polynomial the two limb
// polynomials in the evaluation layout of fft_wave_f64.hpp ([poly][limb][register c][lane] complex words)
__global__ void __launch_bounds__(256) k_bsk_to_fft_t64(const u64 *__restrict__ std_polys, double *__restrict__ limb_polys,
                                                        const double *__restrict__ g_tw, uint32_t n_polys, int prec) {
    const int limbs = t64::limbs_of(prec);
    __shared__ double lds[TW_WORDS + 4 * SCRATCH_WORDS];
    for (int i = threadIdx.x; i < TW_WORDS; i += blockDim.x) lds[i] = g_tw[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + wave;            // (polynomial, limb)
    if (item >= n_polys * (uint32_t)limbs) return;
    const uint32_t poly = item / limbs;
    const int j = (int)(item % limbs);
    double *scratch = lds + TW_WORDS + wave * SCRATCH_WORDS;
    double x[16];
    static_for<0, 16>([&](auto J) { x[J] = (double)t64::limb_of((i64)std_polys[(size_t)poly * N + lane + 64 * J], j, prec); });
    forward(x, lane, lds, scratch);
    double2 *dst = reinterpret_cast<double2 *>(limb_polys + (size_t)item * N);
    static_for<0, 8>([&](auto Cc) { dst[Cc * 64 + lane] = double2{x[Cc], x[Cc + 8]}; });
}

// STATS (the test hook bmi_fft_margin_host): also records the largest distance of a limb sum from the integer it is rounded to
template <int L, int BG, bool STATS = false>
__global__ void __launch_bounds__(128 * TF_CTS)
    k_blind_rotate_t64f(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                        const double *__restrict__ bsk, const double *__restrict__ g_tw, u64 *__restrict__ out,
                        uint32_t count, uint32_t n, unsigned long long *__restrict__ stat) {
    constexpr int CTS = TF_CTS;
    constexpr int LIMBS = Scheme<48>::LIMBS, LB = Scheme<48>::BITS, PRE = Scheme<48>::PRE, AB = 64 - PRE;
    // a limb's sum: 2 L N terms of |digit| <= 2^(BG-1) times |limb| <= 2^(LB-1) - the transform's error bound is stated for this size
    static_assert(2.0 * L * N * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) <= 0x1p45, "limb sums must stay below 2^45");
    static_assert(LIMBS == 2 && L * BG < AB, "two limbs; the decomposed bits lie inside the key precision");
    extern __shared__ double lds[];
    double *tiles = lds + TW_WORDS;
    double *accs = tiles + 2 * CTS * SCRATCH_WORDS;          // accumulators: exact integers word / 2^PRE, centred mod 2^AB
    double *at_base = accs + 2 * CTS * N;
    uint32_t *flags = reinterpret_cast<uint32_t *>(at_base + CTS * BMI_AT_WORDS);  // [2 CTS] published, [2 CTS] consumed
#if BMI_T64F_LDSKEY
    double2 *ring = reinterpret_cast<double2 *>(at_base + CTS * BMI_AT_WORDS + 4 * CTS);
#endif
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ctl = wave >> 1, c = wave & 1;
    if (threadIdx.x < 4 * CTS) flags[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < TW_WORDS; i += blockDim.x) lds[i] = g_tw[i];
    const uint32_t ct_raw = blockIdx.x * CTS + ctl;
    const bool live = ct_raw < count;
    const uint32_t ct = live ? ct_raw : count - 1;
    double *tile = tiles + wave * SCRATCH_WORDS;
    const double *ptile = tiles + (wave ^ 1) * SCRATCH_WORDS;
    double *accf = accs + wave * N;
    auto mod_ab = [](double t) {   // centred residue mod 2^AB of an exact integer |t| < 2^53
        // (ties go to the negative end, like the two's complement reading of the u64 word: + 2^(AB-1) is - 2^(AB-1))
        return __builtin_fma(-(double)(1ull << AB), __builtin_floor(__builtin_fma(t, 1.0 / (double)(1ull << AB), 0.5)), t);
    };
    uint16_t *at = reinterpret_cast<uint16_t *>(at_base + ctl * BMI_AT_WORDS);
    uint32_t *f_pub = flags + wave, *f_pub_partner = flags + (wave ^ 1);
    uint32_t *f_ack = flags + 2 * CTS + wave, *f_ack_partner = flags + 2 * CTS + (wave ^ 1);
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = lane + 64 * c; i <= n; i += 128) at[i] = (uint16_t)t64::modswitch<LOG_N + 1>(lwe[i]);
    __syncthreads();
    {
        const u64 *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at[n];
        static_for<0, 16>([&](auto J) {
            const uint32_t e = (lane + 64 * J + bt) & (2 * N - 1);
            const u64 v = tv[e & (N - 1)];
            const u64 w0 = c ? ((e & N) ? (u64)0 - v : v) : (u64)0;
            accf[lane + 64 * J] = (double)((i64)w0 >> PRE);     // test polynomials are multiples of 2^(59 or so)
        });
    }

    uint32_t hand = 0;   // handshake counter of the pair (one per inverse transform)
    double dev = 0.0;    // STATS: largest |value - nearest integer| this lane has rounded away
    PH_DECL();
    for (uint32_t i = 0; i < n; i++) {
        PH_MARK(7);
#if BMI_T64F_RESYNC
        if (i % BMI_T64F_RESYNC == 0) __syncthreads();
#endif
        PH_MARK(0);   // resync barrier
        const uint32_t a_t = at[i];
        // this wavefront's L GGSW rows: [row = L c + lev][column][limb][N]
        const double *bsk_c = bsk + ((size_t)i * 4 * L + c * 2 * L) * LIMBS * N;
#if BMI_T64F_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        wave_sync();
        double r[16];
        {
            double vr[16], vs[16];  // all 32 reads in flight before the first use
            static_for<0, 16>([&](auto J) {
                vr[J] = accf[(lane + 64 * J + 2 * N - a_t) & (N - 1)];
                vs[J] = accf[lane + 64 * J];
            });
            sched_fence();
            static_for<0, 16>([&](auto J) {
                const uint32_t e = (lane + 64 * J + 2 * N - a_t) & (2 * N - 1);
                const double d = mod_ab(((e & N) ? -vr[J] : vr[J]) - vs[J]);            // the centred lift of the u64 difference, / 2^PRE
                r[J] = __builtin_floor(__builtin_fma(d, 1.0 / (double)(1ull << (AB - L * BG)), 0.5));   // round half up to L BG bits
            });
        }
        PH_MARK(1);   // accumulator reads, difference, rounding
        double X[L][16];   // the L digit polynomials, evaluation layout, live across the limb loop
        // key rows in the order they are multiplied: t = limb * 2 L + q; q < L: (level q, partner's column c^1) - its partial sum is
        // published while the own column (q >= L: level q - L, column c) is still being multiplied.  A row is 8 complex words per
        // lane; the NEXT row is requested before the current one is multiplied (two register sets), the first one before the last
        // forward transform: a row's 32 multiply-adds are far shorter than the latency of its loads.
        auto row_ptr = [&](int t) {
            const int j = t / (2 * L), q = t % (2 * L), lev = q % L, col = q < L ? (c ^ 1) : c;
            return reinterpret_cast<const double2 *>(bsk_c + ((size_t)(lev * 2 + col) * LIMBS + j) * N);
        };
        double2 kb[2][8];
#if BMI_T64F_LDSKEY
        // a row: this wavefront's share of the LDS-DMA requests (slices ctl, ctl + 3, ctl + 6 of the eight 1-KiB slices), then the whole
        // row read back from the ring slot (no hand-off: timing only)
        auto fetch = [&](double2 (&dst)[8], int t) {
            double2 *slot = ring + ((size_t)c * 2 + (t & 1)) * (N / 2);
            // this wavefront's three consecutive 1-KiB slices, starting at slice 0 / 3 / 5 (slice 5 is requested twice: every request stays
            // inside the row); the instruction offset moves the global and the LDS address alike
            const int first = ctl * 3 - (ctl >> 1);
            const double2 *g = row_ptr(t) + first * 64 + lane;
            double2 *l = slot + first * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 2048, 0);
            static_for<0, 8>([&](auto P) { dst[P] = slot[P * 64 + lane]; });
        };
#else
        auto fetch = [&](double2 (&dst)[8], int t) { static_for<0, 8>([&](auto P) { dst[P] = row_ptr(t)[P * 64 + lane]; }); };
#endif
        static_for<0, L>([&](auto LEV) {
            constexpr int lev = L - 1 - LEV;  // least significant digit first
            pin();
#if BMI_T64F_PRIO
            __builtin_amdgcn_s_setprio(lev + 1);
#endif
            static_for<0, 16>([&](auto J) {
                if constexpr (lev == 0) {
                    X[0][J] = r[J];
                } else {
                    const double rn = __builtin_floor(__builtin_fma(r[J], 1.0 / (double)(1ull << BG), 0.5));
                    X[lev][J] = __builtin_fma(-(double)(1ull << BG), rn, r[J]);          // digit in [-2^(BG-1), 2^(BG-1))
                    r[J] = rn;
                }
            });
            if constexpr (lev == 0) {
                fetch(kb[0], 0);
                pin();
            }
            forward(X[lev], lane, lds, tile);
        });
#if BMI_T64F_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        PH_MARK(2);   // digits + L forward transforms
        double acc[16];
        static_for<0, LIMBS * 2 * L>([&](auto T) {
            constexpr int t = T, j = t / (2 * L), q = t % (2 * L), lev = q % L, cur = t & 1;
            if constexpr (q == 0) hand++;
            // (the first row of the next limb is requested after the inverse transform, which needs the registers)
            if constexpr (t + 1 < LIMBS * 2 * L && q != 2 * L - 1) fetch(kb[cur ^ 1], t + 1);
            sched_fence();
            static_for<0, 8>([&](auto P) {
                const double xr = X[lev][P], xi = X[lev][P + 8];
                if constexpr (lev == 0) {
                    acc[P] = __builtin_fma(xr, kb[cur][P].x, -(xi * kb[cur][P].y));
                    acc[P + 8] = __builtin_fma(xr, kb[cur][P].y, xi * kb[cur][P].x);
                } else {
                    acc[P] = __builtin_fma(xr, kb[cur][P].x, __builtin_fma(-xi, kb[cur][P].y, acc[P]));
                    acc[P + 8] = __builtin_fma(xr, kb[cur][P].y, __builtin_fma(xi, kb[cur][P].x, acc[P + 8]));
                }
            });
            if constexpr (q == L - 1) {
                // the partner's partial goes through this wavefront's tile (free since the last transform)
                wave_sync();
                static_for<0, 8>([&](auto P) {
                    reinterpret_cast<double2 *>(tile)[P * 64 + lane] = double2{acc[P], acc[P + 8]};
                });
                pair_post(f_pub, hand);
            }
            pin();
            if constexpr (q == 2 * L - 1) {
                PH_MARK(3);   // 2L rows of products (+ publishing the partner's partial)
                pair_wait(f_pub_partner, hand);
                PH_MARK(4);   // waiting for the partner's partial
                static_for<0, 8>([&](auto P) {
                    const double2 p = reinterpret_cast<const double2 *>(ptile)[P * 64 + lane];
                    acc[P] += p.x;
                    acc[P + 8] += p.y;
                });
                pair_post(f_ack, hand);          // release: the reads above have landed
                pair_wait(f_ack_partner, hand);  // the partner has read this tile: the inverse transform may overwrite it
                PH_MARK(5);   // adding the partner's partial, acknowledging
                inverse(acc, lane, lds, tile);
                PH_MARK(6);   // inverse transform
                if constexpr (t + 1 < LIMBS * 2 * L) {
                    fetch(kb[cur ^ 1], t + 1);
                    pin();
                }
                // the limb's exact integer result (|.| < 2^45: nearest integer of the transform's output), shifted into place
                static_for<0, 16>([&](auto J) {
                    double x = __builtin_rint(acc[J]);
                    if constexpr (STATS) dev = __builtin_fmax(dev, __builtin_fabs(acc[J] - x));
                    if constexpr (j > 0) {
                        // x 2^(LB j) mod 2^AB: only the low AB - LB j bits of the limb's integer survive the shift
                        constexpr double W = (double)(1ull << (AB - LB * j));
                        x = __builtin_fma(-W, __builtin_rint(x * (1.0 / W)), x);
                        accf[lane + 64 * J] = mod_ab(__builtin_fma(x, (double)(1ull << (LB * j)), accf[lane + 64 * J]));
                    } else {
                        accf[lane + 64 * J] += x;   // (reduced mod 2^AB with the last limb: 2^47 + 2^45 + 2^47 stays exact)
                    }
                });
                pin();
            }
        });
    }

#ifdef BMI_PHASE_PROF
    PH_MARK(7);
    if (blockIdx.x == 0 && lane == 0)
        for (int k_ = 0; k_ < 8; k_++) g_phase_f[wave * 8 + k_] = ph_[k_];
#endif
    if constexpr (STATS) atomicMax(stat, (unsigned long long)__double_as_longlong(dev));   // non-negative doubles order like their bit patterns
    if (!live) return;
    wave_sync();
    u64 *o = out + (size_t)ct * (N + 1);
    if (c == 0) {
        static_for<0, 16>([&](auto J) {
            const uint32_t m = lane + 64 * J;
            const u64 v = f64_to_word(accf[m]) << PRE;
            if (m == 0) o[0] = v;
            else o[N - m] = (u64)0 - v;
        });
    } else if (lane == 0) {
        o[N] = f64_to_word(accf[0]) << PRE;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LATENCY form: one workgroup of 16 wavefronts per ciphertext, every transform split over TWO wavefronts by parity
// (fft_half_f64.hpp: 256-point halves whose lane exchanges are register swaps and DPP moves - no LDS round trip inside a
// transform); three phases per CMUX separated by workgroup barriers
//   A  wavefronts 0 .. 4L-1 = (input polynomial c, level, parity h): rotate / decompose 512 coefficients of the u64 accumulator
//      (integer rule of the oracle), forward half -> tile (slot order)
//   B  all 1,024 threads = (limb j, output polynomial o, slot q): E + O' and E - O' of the 2L digit transforms, the two complex
//      multiply-accumulates per row against this thread's key words (own key copy: per slot the pair F_k, F_{k+256} side by
//      side; requested before phase A, landing under it), then the sum and the twisted difference for the inverse halves
//   C  wavefronts 0 .. 7 = (limb, o, parity): inverse half, nearest integer, shift into place and ONE LDS atomic add (f64) per
//      coefficient into the accumulator (the two limbs of a coefficient meet there)
// The accumulator is the exact integer word / 2^16 in a double, as in the wave-pair kernel, but only re-centred mod 2^48 every
// LF_RECENTRE steps (the atomic adds cannot reduce): 2^47 + 8 (2^45 + 2^47) < 2^51 keeps every sum exact.
#ifndef BMI_LATF_ATOMIC
#define BMI_LATF_ATOMIC 1   // phase C of the latency form: 1 = eight tasks (limb, output, parity) meeting by LDS f64 atomics, 0 = four tasks (output, parity) on wavefronts 12-15 running both limbs' inverse halves with plain read-modify-writes (A/B r04: 3.80 / 4.01 ms for 1 / 256 against 3.60 / 3.95 - four wavefronts leave the SIMDs idle; kept as an option)
#endif
constexpr int LF_THREADS = 1024;
constexpr int LF_MAX_L = 3;
constexpr int LF_HALF = N / 2;
constexpr int LF_RECENTRE = 8;
constexpr int LF_LDS_WORDS = ffth::HT_WORDS + 2 * N + 2 * LF_MAX_L * N + 2 * 2 * N + BMI_AT_WORDS;
static_assert(LF_LDS_WORDS <= BMI_LDS_WORDS_MAX, "LF_LDS_WORDS exceeds the 160 KB of LDS");

// accumulator words are kept split by parity (a lane's four points are 128 coefficients apart and of one parity)
__device__ __forceinline__ uint32_t acc_slot(uint32_t n) { return (n & 1) * LF_HALF + (n >> 1); }

// standard-domain GGSW polynomials -> per (polynomial, limb) 256 slots of [F_k, F_{k+256}] (k = ffth::slot_freq of the slot)
__global__ void __launch_bounds__(256) k_bsk_to_latf_t64(const u64 *__restrict__ std_polys, double *__restrict__ lat_polys,
                                                         const double *__restrict__ g_tw_h, uint32_t n_polys, int prec) {
    const int limbs = t64::limbs_of(prec);
    __shared__ double lds[ffth::HT_WORDS + 4 * LF_HALF];
    for (int i = threadIdx.x; i < ffth::HT_WORDS; i += blockDim.x) lds[i] = g_tw_h[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = wave & 1;
    const uint32_t item = blockIdx.x * 2 + (wave >> 1);   // (polynomial, limb): two per workgroup, two wavefronts each
    const bool ok = item < n_polys * (uint32_t)limbs;
    const uint32_t poly = ok ? item / limbs : 0;
    const int j = ok ? (int)(item % limbs) : 0;
    double2 *tile = reinterpret_cast<double2 *>(lds + ffth::HT_WORDS + wave * LF_HALF);   // 256 complex per wavefront
    if (ok) {
        double re[4], im[4];
        static_for<0, 4>([&](auto R) {
            const uint32_t m = 2 * (lane + 64 * R) + h;
            re[R] = (double)t64::limb_of((i64)std_polys[(size_t)poly * N + m], j, prec);
            im[R] = (double)t64::limb_of((i64)std_polys[(size_t)poly * N + m + 512], j, prec);
        });
        ffth::C v[4];
        if (h) ffth::forward_half<1>(re, im, v, lane, lds);
        else ffth::forward_half<0>(re, im, v, lane, lds);
        static_for<0, 4>([&](auto R) { tile[R * 64 + lane] = double2{v[R].r, v[R].i}; });
    }
    __syncthreads();
    if (ok) {
        const double2 *te = reinterpret_cast<const double2 *>(lds + ffth::HT_WORDS + (wave & ~1) * LF_HALF), *to = te + LF_HALF / 2;
        double2 *o = reinterpret_cast<double2 *>(lat_polys + (size_t)item * N);
        static_for<0, 2>([&](auto Q2) {
            const int q = (h * 2 + Q2) * 64 + lane;
            const double2 e = te[q], od = to[q];
            o[2 * q] = double2{e.x + od.x, e.y + od.y};
            o[2 * q + 1] = double2{e.x - od.x, e.y - od.y};
        });
    }
}

// STATS (the test hook bmi_fft_margin_host with the latency form selected): also records the largest distance of a limb sum from the
// integer it is rounded to
template <int L, int BG, bool STATS = false>
__global__ void __launch_bounds__(LF_THREADS)
    k_blind_rotate_lat_t64f(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                            const double *__restrict__ bsk_latf, const double *__restrict__ g_tw_h, u64 *__restrict__ out,
                            uint32_t count, uint32_t n, unsigned long long *__restrict__ stat) {
    constexpr int LIMBS = Scheme<48>::LIMBS, LB = Scheme<48>::BITS, PRE = Scheme<48>::PRE, AB = 64 - PRE;
    static_assert(2.0 * L * N * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) <= 0x1p45, "limb sums must stay below 2^45");
    static_assert(LIMBS == 2 && L <= LF_MAX_L && L * BG < AB, "two limbs, at most three levels");
    extern __shared__ double lds[];
    double *acc = lds + ffth::HT_WORDS;                                     // [2 components][2 parities][512]: word / 2^16, exact, |.| < 2^51
    auto mod_ab = [](double t) {   // centred residue mod 2^AB of an exact integer |t| < 2^53 (ties to the negative end, like the u64 word)
        return __builtin_fma(-(double)(1ull << AB), __builtin_floor(__builtin_fma(t, 1.0 / (double)(1ull << AB), 0.5)), t);
    };
    double2 *tiles = reinterpret_cast<double2 *>(lds + ffth::HT_WORDS + 2 * N);   // [2L rows][2 halves][256 slots] complex
    double2 *SD = tiles + LF_MAX_L * N;                                     // [limb][output][S, D][256 slots] complex
    uint16_t *at = reinterpret_cast<uint16_t *>(SD + 2 * N);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < ffth::HT_WORDS; i += LF_THREADS) lds[i] = g_tw_h[i];
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += LF_THREADS) at[i] = (uint16_t)t64::modswitch<LOG_N + 1>(lwe[i]);
    __syncthreads();
    {
        const u64 *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at[n];
        const uint32_t nn = tid;  // coefficient index
        const uint32_t e = (nn + bt) & (2 * N - 1);
        const u64 v = tv[e & (N - 1)];
        acc[acc_slot(nn)] = 0.0;
        acc[N + acc_slot(nn)] = (double)((i64)((e & N) ? (u64)0 - v : v) >> PRE);     // test polynomials are multiples of 2^(59 or so)
    }
    __syncthreads();
    const int mj = tid >> 9, mo = (tid >> 8) & 1, mq = tid & 255;   // phase B: limb, output polynomial, slot
    uint32_t since_centred = 0;   // steps taken since the accumulator was last reduced mod 2^48
    double dev = 0.0;             // STATS: largest |value - nearest integer| this lane has rounded away

    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];
        if (a_t == 0) continue;  // uniform over the workgroup
        // key words of this thread: [row 2L][output 2][limb][256 slots][F_k, F_{k+256}]
        const double *bi = bsk_latf + (size_t)i * 4 * L * LIMBS * N;
        double2 klo[2 * L], khi[2 * L];
        static_for<0, 2 * L>([&](auto R) {
            const double2 *row = reinterpret_cast<const double2 *>(bi + ((size_t)(R * 2 + mo) * LIMBS + mj) * N);
            klo[R] = row[2 * mq];
            khi[R] = row[2 * mq + 1];
        });
        if (wave < 4 * L) {
            const int c = wave / (2 * L), lev = (wave % (2 * L)) >> 1, h = wave & 1;
            const double *ac = acc + c * N;
            double x[8];   // re[r] = x[r], im[r] = x[r + 4]
            double vr[8], vs[8];
            // coefficient m_J = 2 (lane + 64 (J & 3)) + h + 512 (J >> 2); its rotated source e_J = m_J - a_t mod 2N: half of it is
            // t0 + 64 (J & 3) + 256 (J >> 2) - the low 9 bits are the slot inside the parity block, bit 9 is the sign
            const uint32_t e0 = (2 * lane + h + 2 * N - a_t) & (2 * N - 1);
            const uint32_t t0 = e0 >> 1, pbase = (e0 & 1) * LF_HALF;
            static_for<0, 8>([&](auto J) {
                const uint32_t t = t0 + 64 * (J & 3) + 256 * (J >> 2);
                vr[J] = ac[pbase + (t & (LF_HALF - 1))];
                vs[J] = ac[h * LF_HALF + lane + 64 * (J & 3) + 256 * (J >> 2)];
            });
            static_for<0, 8>([&](auto J) {
                const uint32_t t = t0 + 64 * (J & 3) + 256 * (J >> 2);
                const double dd = mod_ab(((t >> 9) & 1) ? -vr[J] - vs[J] : vr[J] - vs[J]);   // the centred lift of the u64 difference, / 2^PRE
                double r = __builtin_floor(__builtin_fma(dd, 1.0 / (double)(1ull << (AB - L * BG)), 0.5));   // round half up to L BG bits
                double d = r;                                                          // digit `lev`, balanced [-2^(BG-1), 2^(BG-1))
#pragma unroll
                for (int s = L - 1; s > 0; s--) {
                    const double rn = __builtin_floor(__builtin_fma(r, 1.0 / (double)(1ull << BG), 0.5));
                    if (s == lev) d = __builtin_fma(-(double)(1ull << BG), rn, r);
                    r = rn;
                }
                x[J] = lev == 0 ? r : d;
            });
            const double re[4] = {x[0], x[1], x[2], x[3]}, im[4] = {x[4], x[5], x[6], x[7]};
            ffth::C v[4];
            if (h) ffth::forward_half<1>(re, im, v, lane, lds);
            else ffth::forward_half<0>(re, im, v, lane, lds);
            double2 *tile = tiles + (size_t)(wave >> 1) * LF_HALF + h * (LF_HALF / 2);
            static_for<0, 4>([&](auto R) { tile[R * 64 + lane] = double2{v[R].r, v[R].i}; });
        }
        __syncthreads();
        {
            ffth::C ylo{0.0, 0.0}, yhi{0.0, 0.0};
            static_for<0, 2 * L>([&](auto R) {
                const double2 e = tiles[(size_t)R * LF_HALF + mq], od = tiles[(size_t)R * LF_HALF + LF_HALF / 2 + mq];
                const double lr = e.x + od.x, li = e.y + od.y, hr = e.x - od.x, hi = e.y - od.y;
                ylo.r = __builtin_fma(lr, klo[R].x, __builtin_fma(-li, klo[R].y, ylo.r));
                ylo.i = __builtin_fma(lr, klo[R].y, __builtin_fma(li, klo[R].x, ylo.i));
                yhi.r = __builtin_fma(hr, khi[R].x, __builtin_fma(-hi, khi[R].y, yhi.r));
                yhi.i = __builtin_fma(hr, khi[R].y, __builtin_fma(hi, khi[R].x, yhi.i));
            });
            const double2 w = reinterpret_cast<const double2 *>(lds + ffth::HT_W)[mq];
            const ffth::C d = ffth::cmul<true>(ffth::C{ylo.r - yhi.r, ylo.i - yhi.i}, w.x, w.y);
            double2 *sd = SD + (size_t)(mj * 2 + mo) * LF_HALF;
            sd[mq] = double2{ylo.r + yhi.r, ylo.i + yhi.i};
            sd[LF_HALF / 2 + mq] = double2{d.r, d.i};
        }
        __syncthreads();
#if BMI_LATF_ATOMIC
        if (wave < 4 * LIMBS) {
            const int j = wave >> 2, o = (wave >> 1) & 1, h = wave & 1;
            const double2 *sd = SD + (size_t)(j * 2 + o) * LF_HALF + h * (LF_HALF / 2);
            ffth::C v[4];
            static_for<0, 4>([&](auto R) {
                const double2 t = sd[R * 64 + lane];
                v[R] = ffth::C{t.x, t.y};
            });
            double re[4], im[4];
            if (h) ffth::inverse_half<1>(v, re, im, lane, lds);
            else ffth::inverse_half<0>(v, re, im, lane, lds);
            double *ao = acc + o * N + h * LF_HALF + lane;
            auto place = [&](double v) {   // the limb's exact integer (|.| < 2^45: nearest integer of the transform's output), shifted into place
                double xr = __builtin_rint(v);
                if constexpr (STATS) dev = __builtin_fmax(dev, __builtin_fabs(v - xr));
                if (j == 0) return xr;
                constexpr double W = (double)(1ull << (AB - LB));   // x 2^LB mod 2^AB: only the low AB - LB bits survive the shift
                xr = __builtin_fma(-W, __builtin_rint(xr * (1.0 / W)), xr);
                return xr * (double)(1ull << LB);
            };
            static_for<0, 4>([&](auto R) {
                atomicAdd(ao + 64 * R, place(re[R]));          // coefficient 2 (lane + 64 R) + h
                atomicAdd(ao + 64 * R + 256, place(im[R]));    // ... + 512
            });
        }
#else
        if (wave >= 12) {   // (wavefronts 12 .. 15 had no forward task) = (output polynomial, parity): the inverse halves of BOTH limbs
            const int o = (wave >> 1) & 1, h = wave & 1;
            double re[LIMBS][4], im[LIMBS][4];
            static_for<0, LIMBS>([&](auto J) {
                const double2 *sd = SD + (size_t)(J * 2 + o) * LF_HALF + h * (LF_HALF / 2);
                ffth::C v[4];
                static_for<0, 4>([&](auto R) {
                    const double2 t = sd[R * 64 + lane];
                    v[R] = ffth::C{t.x, t.y};
                });
                if (h) ffth::inverse_half<1>(v, re[J], im[J], lane, lds);
                else ffth::inverse_half<0>(v, re[J], im[J], lane, lds);
            });
            double *ao = acc + o * N + h * LF_HALF + lane;
            // a limb's exact integer (|.| < 2^45: nearest integer of the transform's output); limb 1 shifted into place: x 2^LB mod 2^AB,
            // of which only the low AB - LB bits survive
            auto place = [&](double v0, double v1) {
                const double x0 = __builtin_rint(v0);
                double x1 = __builtin_rint(v1);
                if constexpr (STATS) dev = __builtin_fmax(dev, __builtin_fmax(__builtin_fabs(v0 - x0), __builtin_fabs(v1 - x1)));
                constexpr double W = (double)(1ull << (AB - LB));
                x1 = __builtin_fma(-W, __builtin_rint(x1 * (1.0 / W)), x1);
                return __builtin_fma(x1, (double)(1ull << LB), x0);
            };
            static_for<0, 4>([&](auto R) {
                ao[64 * R] += place(re[0][R], re[1][R]);              // coefficient 2 (lane + 64 R) + h
                ao[64 * R + 256] += place(im[0][R], im[1][R]);        // ... + 512
            });
        }
#endif
        __syncthreads();
        if (++since_centred == LF_RECENTRE) {   // (uniform: counts the steps actually taken) keep the accumulator's magnitude below 2^51
            since_centred = 0;
            acc[tid] = mod_ab(acc[tid]);
            acc[N + tid] = mod_ab(acc[N + tid]);
            __syncthreads();
        }
    }
    if constexpr (STATS) atomicMax(stat, (unsigned long long)__double_as_longlong(dev));   // non-negative doubles order like their bit patterns
    u64 *o = out + (size_t)ct * (N + 1);
    {
        const uint32_t nn = tid;
        const u64 a0 = f64_to_word(mod_ab(acc[acc_slot(nn)])) << PRE;
        if (nn == 0) {
            o[0] = a0;
            o[N] = f64_to_word(mod_ab(acc[N + acc_slot(0)])) << PRE;
        } else {
            o[N - nn] = (u64)0 - a0;
        }
    }
}

}  // namespace

#ifdef BMI_PHASE_PROF
extern "C" int bmi_debug_phase_prof_t64f(unsigned long long *out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phase_f), sizeof(unsigned long long) * 128);
}
#endif

namespace bmit {

#define BMITF_LAUNCH_CHECK()                    \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// (precision, levels, base log) combinations the transform's error bound was established for
bool shape_supported_fft(int prec, uint32_t levels, uint32_t base_log) {
    return prec == 48 && base_log == 10 && (levels == 3 || levels == 2);
}

int launch_bsk_to_fft(const u64 *std_polys, double *limb_polys, const double *g_tw_fft, uint32_t n_polys, int prec, hipStream_t s) {
    if (prec != 48) return (int)hipErrorInvalidValue;
    const uint32_t items = n_polys * (uint32_t)t64::limbs_of(prec);
    hipLaunchKernelGGL(k_bsk_to_fft_t64, dim3((items + 3) / 4), dim3(256), 0, s, std_polys, limb_polys, g_tw_fft, n_polys, prec);
    BMITF_LAUNCH_CHECK();
    return 0;
}

template <int L, int BG, bool STATS>
static int launch_t64f(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_fft,
                       const double *g_tw_fft, u64 *out, uint32_t count, uint32_t n, unsigned long long *stat, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)TF_LDS_WORDS * sizeof(double);
    auto kern = k_blind_rotate_t64f<L, BG, STATS>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3((count + TF_CTS - 1) / TF_CTS), dim3(128 * TF_CTS), lds, s, small_cts, lut_ids, luts, bsk_fft,
                       g_tw_fft, out, count, n, stat);
    BMITF_LAUNCH_CHECK();
    return 0;
}

int launch_bsk_to_latf(const u64 *std_polys, double *lat_polys, const double *g_tw_h, uint32_t n_polys, int prec, hipStream_t s) {
    if (prec != 48 && prec != 42) return (int)hipErrorInvalidValue;   // (42: the unrolled key of bmi_kernels_t64fu.hip)
    const uint32_t items = n_polys * (uint32_t)t64::limbs_of(prec);
    hipLaunchKernelGGL(k_bsk_to_latf_t64, dim3((items + 1) / 2), dim3(256), 0, s, std_polys, lat_polys, g_tw_h, n_polys, prec);
    BMITF_LAUNCH_CHECK();
    return 0;
}

template <int L, int BG, bool STATS>
static int launch_lat_t64f(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_fft,
                           const double *g_tw_fft, u64 *out, uint32_t count, uint32_t n, unsigned long long *stat, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)LF_LDS_WORDS * sizeof(double);
    auto kern = k_blind_rotate_lat_t64f<L, BG, STATS>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3(count), dim3(LF_THREADS), lds, s, small_cts, lut_ids, luts, bsk_fft, g_tw_fft, out, count, n, stat);
    BMITF_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_lat_fft(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_fft,
                                const double *g_tw_fft, u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels,
                                uint32_t base_log, unsigned long long *stat, hipStream_t s) {
    if (count == 0) return 0;
    if (!shape_supported_fft(prec, levels, base_log)) return (int)hipErrorInvalidValue;
    if (stat) {
        if (levels == 3) return launch_lat_t64f<3, 10, true>(small_cts, lut_ids, luts, bsk_fft, g_tw_fft, out, count, n, stat, s);
        return launch_lat_t64f<2, 10, true>(small_cts, lut_ids, luts, bsk_fft, g_tw_fft, out, count, n, stat, s);
    }
    if (levels == 3) return launch_lat_t64f<3, 10, false>(small_cts, lut_ids, luts, bsk_fft, g_tw_fft, out, count, n, nullptr, s);
    return launch_lat_t64f<2, 10, false>(small_cts, lut_ids, luts, bsk_fft, g_tw_fft, out, count, n, nullptr, s);
}

int launch_blind_rotate_fft(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_fft,
                            const double *g_tw_fft, u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels,
                            uint32_t base_log, unsigned long long *stat, hipStream_t s) {
    if (count == 0) return 0;
    if (!shape_supported_fft(prec, levels, base_log)) return (int)hipErrorInvalidValue;
    if (stat) {
        if (levels == 3) return launch_t64f<3, 10, true>(small_cts, lut_ids, luts, bsk_fft, g_tw_fft, out, count, n, stat, s);
        return launch_t64f<2, 10, true>(small_cts, lut_ids, luts, bsk_fft, g_tw_fft, out, count, n, stat, s);
    }
    if (levels == 3) return launch_t64f<3, 10, false>(small_cts, lut_ids, luts, bsk_fft, g_tw_fft, out, count, n, nullptr, s);
    return launch_t64f<2, 10, false>(small_cts, lut_ids, luts, bsk_fft, g_tw_fft, out, count, n, nullptr, s);
}

}  // namespace bmit
