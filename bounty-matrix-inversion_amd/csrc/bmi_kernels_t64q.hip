// 2^64 TORUS at N = 4096 (6-bit look-ups; 5-bit look-ups at 128-bit-secure noise: preset secure128_torus_wide): blind rotation with
// the exact limb products carried by the floating-point transform of fft_eighth_f64.hpp (gfx950).
//
// Same scheme as bmi_kernels_t64w.hip one size up: bootstrap key stored at 44 bits of precision (words rounded to multiples of 2^20;
// the rounded key IS the key) as two balanced 22-bit limbs, digits in base 2^10, accumulator the exact integer word / 2^20 in a
// double; per limb the sum over the 2 l digit x limb polynomial products is an integer below 2^45, computed through the folded
// 2,048-point complex FFT and ROUNDED TO THE NEAREST INTEGER (a-priori bound 0.45 < 1/2 in the header) - the kernel's words equal
// the oracle's integer arithmetic bit for bit (tests/test_gpu_torus_quad.py).
//
// One workgroup of 16 wavefronts per ciphertext, every batch size; a transform is split over EIGHT wavefronts (eighths of 256 points,
// no LDS inside an eighth).  The forward transforms of ONE decomposition level (2 polynomials x 8 eighths = 16 tasks = 64 KB of
// tiles) are all the LDS holds beside the accumulator, so a CMUX walks the levels:
//   for each level:  A  16 forward tasks (component c, eighth h): rotate / decompose 512 coefficients, digit `level`, forward eighth
//                       -> tile (slot order, times W_h)
//                    B  all 1,024 threads = (slot, output polynomial, limb): per row the radix-8 butterfly over the eight tiles
//                       (fftw::dft8) and eight complex multiply-accumulates with this thread's key words, into eight running sums
//   then the inverse butterfly (dft8), and per limb: the sums to LDS (over the tiles), 16 inverse tasks (output, eighth): conj W_h,
//   inverse eighth, nearest integer, shift into place, plain read-modify-write of the accumulator (the limbs take turns).
#include <hip/hip_runtime.h>

#include <atomic>

#include "bmi_internal.hpp"
#include "fft_eighth_f64.hpp"
#include "pair_sync.hpp"
#include "t64_common.hpp"

using t64::i64;
using t64::u64;

namespace {

using ffte::C;
using ffte::static_for;
using t64::f64_to_word;
using t64::Scheme;

#ifndef BMI_T64Q_BOTH_ROWS
#define BMI_T64Q_BOTH_ROWS 0   // 1: both rows' key words of a level requested before the level's barrier (A/B)
#endif
constexpr int QN = 4096, QLOG = 12;
constexpr int QS = ffte::EIGHTH;                 // slots per eighth
constexpr int QF_THREADS = 1024;
constexpr int QF_MAX_L = 3;
constexpr int QF_RECENTRE = 8;
constexpr int QF_RES = QN / 8;                   // accumulator words per residue class mod 8
constexpr int QF_TILE_CPLX = 2 * 8 * QS;         // complex words of one level's tiles [2 components][8 eighths][256]; one limb's sums [2 outputs][8][256] overlay them
// LDS (doubles): tables | accumulator [2 components][8 residues][512] | tiles | mod-switched LWE words
constexpr int QF_LDS_WORDS = ffte::ET_WORDS + 2 * QN + 2 * QF_TILE_CPLX + BMI_AT_WORDS;
static_assert(QF_LDS_WORDS <= BMI_LDS_WORDS_MAX, "QF_LDS_WORDS exceeds the 160 KB of LDS");

__device__ __forceinline__ uint32_t acc_slot(uint32_t n) { return (n & 7) * QF_RES + (n >> 3); }

// an empty statement that reads and writes the eight sums: they must be in registers here
__device__ __forceinline__ void keep(fftw::C (&y)[8]) {
#pragma unroll
    for (int t = 0; t < 8; t++) asm volatile("" : "+v"(y[t].r), "+v"(y[t].i));
}

// standard-domain GGSW polynomials (u64 torus words, already rounded to the key precision) -> per (polynomial, limb) 2,048 complex
// words A_k / 4 as [t 8][slot 256]: frequency kappa(slot) + 256 t.  One workgroup of eight wavefronts (the eighths) per item.
__global__ void __launch_bounds__(512) k_bsk_to_q_t64(const u64 *__restrict__ std_polys, double *__restrict__ q_polys,
                                                      const double *__restrict__ g_tw, uint32_t n_polys, int prec) {
    const int limbs = t64::limbs_of(prec);
    __shared__ double lds[ffte::ET_WORDS + 8 * QS * 2];
    for (int i = threadIdx.x; i < ffte::ET_WORDS; i += blockDim.x) lds[i] = g_tw[i];
    __syncthreads();
    const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x;
    const uint32_t poly = item / limbs;
    const int j = (int)(item % limbs);
    double2 *tile = reinterpret_cast<double2 *>(lds + ffte::ET_WORDS);
    {
        double re[4], im[4];
        static_for<0, 4>([&](auto R) {
            const uint32_t m = 8 * (lane + 64 * R) + h;
            re[R] = (double)t64::limb_of((i64)std_polys[(size_t)poly * QN + m], j, prec);
            im[R] = (double)t64::limb_of((i64)std_polys[(size_t)poly * QN + m + QN / 2], j, prec);
        });
        C v[4];
        ffte::forward_eighth(h, re, im, v, lane, lds);
        static_for<0, 4>([&](auto R) { tile[h * QS + R * 64 + lane] = double2{v[R].r, v[R].i}; });
    }
    __syncthreads();
    if (threadIdx.x < QS) {
        const int p = threadIdx.x;
        fftw::C q[8];
        static_for<0, 8>([&](auto H) {
            const double2 t = tile[H * QS + p];
            q[H] = fftw::C{t.x, t.y};
        });
        fftw::dft8<false>(q);
        double2 *o = reinterpret_cast<double2 *>(q_polys + (size_t)item * QN);
        static_for<0, 8>([&](auto T) { o[T * QS + p] = double2{0.25 * q[T].r, 0.25 * q[T].i}; });
    }
}

// STATS (the test hook bmi_fft_margin_host): also records the largest distance of a limb sum from the integer it is rounded to
template <int L, int BG, int PREC, bool STATS>
__global__ void __launch_bounds__(QF_THREADS)
    k_blind_rotate_q_t64f(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                          const double *__restrict__ bsk_q, const double *__restrict__ g_tw, u64 *__restrict__ out, uint32_t count,
                          uint32_t n, unsigned long long *__restrict__ stat) {
    constexpr int LIMBS = Scheme<PREC>::LIMBS, LB = Scheme<PREC>::BITS, PRE = Scheme<PREC>::PRE, AB = 64 - PRE;
    static_assert(2.0 * L * QN * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) <= 0x1p45, "limb sums must stay below 2^45");
    static_assert(LIMBS == 2 && L <= QF_MAX_L && L * BG < AB, "two limbs, at most three levels");
    extern __shared__ double lds[];
    double *acc = lds + ffte::ET_WORDS;                                     // [2 components][8 residues][512]: word / 2^PRE, exact, |.| < 2^51
    auto mod_ab = [](double t) {   // centred residue mod 2^AB of an exact integer |t| < 2^53 (ties to the negative end, like the u64 word)
        return __builtin_fma(-(double)(1ull << AB), __builtin_floor(__builtin_fma(t, 1.0 / (double)(1ull << AB), 0.5)), t);
    };
    double2 *tiles = reinterpret_cast<double2 *>(acc + 2 * QN);             // [2 components][8 eighths][256 slots] of the current level
    double2 *SD = tiles;                                                    // one limb's sums [2 outputs][8 eighths][256 slots]
    uint16_t *at = reinterpret_cast<uint16_t *>(tiles + QF_TILE_CPLX);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < ffte::ET_WORDS; i += QF_THREADS) lds[i] = g_tw[i];
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += QF_THREADS) at[i] = (uint16_t)t64::modswitch<QLOG + 1>(lwe[i]);
    __syncthreads();
    {
        const u64 *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * QN;
        const uint32_t bt = at[n];
        static_for<0, 4>([&](auto Q) {
            const uint32_t nn = tid + QF_THREADS * Q;  // coefficient index
            const uint32_t e = (nn + bt) & (2 * QN - 1);
            const u64 v = tv[e & (QN - 1)];
            acc[acc_slot(nn)] = 0.0;
            acc[QN + acc_slot(nn)] = (double)((i64)((e & QN) ? (u64)0 - v : v) >> PRE);     // test polynomials are multiples of 2^PRE (host-checked)
        });
    }
    __syncthreads();
    // phase B: slot, output polynomial, limb (the four combinations of a slot sit 16 lanes apart: their tile reads coincide)
    const int slot = wave * 16 + (lane & 15), mo = lane >> 5, mj = (lane >> 4) & 1;
    uint32_t since_centred = 0;
    double dev = 0.0;             // STATS: largest |value - nearest integer| this lane has rounded away

    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];
        if (a_t == 0) continue;  // uniform over the workgroup
        // key words of this thread: [row 2L][output 2][limb][t 8][256 slots] complex
        const double2 *kth = reinterpret_cast<const double2 *>(bsk_q + (size_t)i * 4 * L * LIMBS * QN) + ((size_t)mo * LIMBS + mj) * (QN / 2) + slot;
        auto row_ptr = [&](int R) { return kth + (size_t)R * 2 * LIMBS * (QN / 2); };
        fftw::C y[8];
        static_for<0, 8>([&](auto T) { y[T] = fftw::C{0.0, 0.0}; });
        static_for<0, L>([&](auto LEV) {
            constexpr int lev = LEV;
            {   // phase A: wavefront = (component c, eighth h)
                const int c = wave >> 3, h = wave & 7;
                const double *ac = acc + c * QN;
                double x[8];   // re[r] = x[r], im[r] = x[r + 4]
                // coefficient m_J = 8 (lane + 64 (J & 3)) + h + 2048 (J >> 2); its rotated source e_J = m_J - a_t mod 2N: an eighth of it is
                // t0 + 64 (J & 3) + 256 (J >> 2) - the low 9 bits are the slot inside the residue block, bit 9 is the sign
                const uint32_t e0 = (8 * lane + h + 2 * QN - a_t) & (2 * QN - 1);
                const uint32_t t0 = e0 >> 3, pbase = (e0 & 7) * QF_RES;
                static_for<0, 2>([&](auto G) {
                    double vr[4], vs[4];
                    static_for<0, 4>([&](auto J4) {
                        constexpr int J = G * 4 + J4;
                        const uint32_t t = t0 + 64 * (J & 3) + 256 * (J >> 2);
                        vr[J4] = ac[pbase + (t & (QF_RES - 1))];
                        vs[J4] = ac[h * QF_RES + lane + 64 * (J & 3) + 256 * (J >> 2)];
                    });
                    static_for<0, 4>([&](auto J4) {
                        constexpr int J = G * 4 + J4;
                        const uint32_t t = t0 + 64 * (J & 3) + 256 * (J >> 2);
                        const double dd = mod_ab(((t >> 9) & 1) ? -vr[J4] - vs[J4] : vr[J4] - vs[J4]);   // the centred lift of the u64 difference, / 2^PRE
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
                    pin();
                });
                const double re[4] = {x[0], x[1], x[2], x[3]}, im[4] = {x[4], x[5], x[6], x[7]};
                C v[4];
                ffte::forward_eighth(h, re, im, v, lane, lds);
                double2 *tile = tiles + (size_t)wave * QS;
                static_for<0, 4>([&](auto R4) { tile[R4 * 64 + lane] = double2{v[R4].r, v[R4].i}; });
            }
            // this thread's key words of the level's first row (component 0): they land under the barrier
            double2 kw[8];
            static_for<0, 8>([&](auto T) { kw[T] = row_ptr(0 * L + lev)[T * QS]; });
#if BMI_T64Q_BOTH_ROWS
            double2 kw1[8];
            static_for<0, 8>([&](auto T) { kw1[T] = row_ptr(1 * L + lev)[T * QS]; });
#endif
            pin();
            __syncthreads();
            static_for<0, 2>([&](auto CC) {   // rows (component CC, this level)
                fftw::C q[8];
                static_for<0, 8>([&](auto H) {
                    const double2 t = tiles[(size_t)(CC * 8 + H) * QS + slot];
                    q[H] = fftw::C{t.x, t.y};
                });
                fftw::dft8<false>(q);   // frequency kappa + 256 t in q[t]
                static_for<0, 8>([&](auto T) {
#if BMI_T64Q_BOTH_ROWS
                    const double2 k = CC == 0 ? kw[T] : kw1[T];
#else
                    const double2 k = kw[T];
#endif
                    y[T].r = __builtin_fma(q[T].r, k.x, __builtin_fma(-q[T].i, k.y, y[T].r));
                    y[T].i = __builtin_fma(q[T].r, k.y, __builtin_fma(q[T].i, k.x, y[T].i));
                });
                // the sums are materialised HERE: the branches of the next task (times_w) would otherwise let the compiler sink these
                // multiply-adds below them, keeping every level's tile and key words alive (scratch)
                keep(y);
#if !BMI_T64Q_BOTH_ROWS
                if constexpr (CC == 0) {
                    static_for<0, 8>([&](auto T) { kw[T] = row_ptr(1 * L + lev)[T * QS]; });
                }
#endif
                pin();
            });
            __syncthreads();   // every thread has read this level's tiles: the next level (or the sums) may overwrite them
        });
        fftw::dft8<true>(y);   // sum_t e8^(-h t) Y_t in y[h]; conj W_h is applied by the inverse task
        static_for<0, LIMBS>([&](auto J) {
            if (mj == J) {
                static_for<0, 8>([&](auto H) { SD[(size_t)(mo * 8 + H) * QS + slot] = double2{y[H].r, y[H].i}; });
            }
            __syncthreads();
            {
                const int o = wave >> 3, h = wave & 7;
                const double2 *sd = SD + (size_t)wave * QS;
                C v[4];
                static_for<0, 4>([&](auto R) {
                    const double2 t = sd[R * 64 + lane];
                    v[R] = C{t.x, t.y};
                });
                double re[4], im[4];
                ffte::inverse_eighth(h, v, re, im, lane, lds);
                double *ao = acc + o * QN + h * QF_RES + lane;
                // the limb's exact integer (|.| < 2^45: nearest integer of the transform's output), shifted into place: limb 1 x 2^LB mod
                // 2^AB, of which only the low AB - LB bits survive
                constexpr int j = J;
                auto place = [&](double v0) {
                    double xr = __builtin_rint(v0);
                    if constexpr (STATS) dev = __builtin_fmax(dev, __builtin_fabs(v0 - xr));
                    if constexpr (j > 0) {
                        constexpr double W = (double)(1ull << (AB - LB));
                        xr = __builtin_fma(-W, __builtin_rint(xr * (1.0 / W)), xr) * (double)(1ull << LB);
                    }
                    return xr;
                };
                static_for<0, 4>([&](auto R) {
                    ao[64 * R] += place(re[R]);              // coefficient 8 (lane + 64 R) + h
                    ao[64 * R + 256] += place(im[R]);        // ... + 2048
                });
            }
            __syncthreads();
        });
        if (++since_centred == QF_RECENTRE) {   // (uniform: counts the steps actually taken) keep the accumulator's magnitude below 2^51
            since_centred = 0;
            static_for<0, 8>([&](auto Q) { acc[tid + QF_THREADS * Q] = mod_ab(acc[tid + QF_THREADS * Q]); });
            __syncthreads();
        }
    }
    if constexpr (STATS) atomicMax(stat, (unsigned long long)__double_as_longlong(dev));   // non-negative doubles order like their bit patterns
    u64 *o = out + (size_t)ct * (QN + 1);
    static_for<0, 4>([&](auto Q) {
        const uint32_t nn = tid + QF_THREADS * Q;
        const u64 a0 = f64_to_word(mod_ab(acc[acc_slot(nn)])) << PRE;
        if (nn == 0) {
            o[0] = a0;
            o[QN] = f64_to_word(mod_ab(acc[QN + acc_slot(0)])) << PRE;
        } else {
            o[QN - nn] = (u64)0 - a0;
        }
    });
}

}  // namespace

namespace bmit {

#define BMITQ_LAUNCH_CHECK()                    \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// (precision, levels, base log) combinations the N = 4096 transform's error bound was established for
bool shape_supported_quad(int prec, uint32_t levels, uint32_t base_log) {
    return prec == 44 && base_log == 10 && (levels == 3 || levels == 2);
}

int launch_bsk_to_quad(const u64 *std_polys, double *q_polys, const double *g_tw_e, uint32_t n_polys, int prec, hipStream_t s) {
    if (prec != 44) return (int)hipErrorInvalidValue;
    const uint32_t items = n_polys * (uint32_t)t64::limbs_of(prec);
    hipLaunchKernelGGL(k_bsk_to_q_t64, dim3(items), dim3(512), 0, s, std_polys, q_polys, g_tw_e, n_polys, prec);
    BMITQ_LAUNCH_CHECK();
    return 0;
}

template <int L, int BG, int PREC, bool STATS>
static int launch_q(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_q, const double *g_tw_e, u64 *out,
                    uint32_t count, uint32_t n, unsigned long long *stat, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)QF_LDS_WORDS * sizeof(double);
    auto kern = k_blind_rotate_q_t64f<L, BG, PREC, STATS>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3(count), dim3(QF_THREADS), lds, s, small_cts, lut_ids, luts, bsk_q, g_tw_e, out, count, n, stat);
    BMITQ_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_quad(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_q, const double *g_tw_e,
                             u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels, uint32_t base_log, unsigned long long *stat,
                             hipStream_t s) {
    if (count == 0) return 0;
    if (!shape_supported_quad(prec, levels, base_log)) return (int)hipErrorInvalidValue;
    if (stat) {
        if (levels == 3) return launch_q<3, 10, 44, true>(small_cts, lut_ids, luts, bsk_q, g_tw_e, out, count, n, stat, s);
        return launch_q<2, 10, 44, true>(small_cts, lut_ids, luts, bsk_q, g_tw_e, out, count, n, stat, s);
    }
    if (levels == 3) return launch_q<3, 10, 44, false>(small_cts, lut_ids, luts, bsk_q, g_tw_e, out, count, n, nullptr, s);
    return launch_q<2, 10, 44, false>(small_cts, lut_ids, luts, bsk_q, g_tw_e, out, count, n, nullptr, s);
}

}  // namespace bmit
