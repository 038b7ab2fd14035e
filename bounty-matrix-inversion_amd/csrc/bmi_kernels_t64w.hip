// 2^64 TORUS at N = 2048 (the secure128_torus set: n 742, k 1, l 3, Bg 2^10): blind rotation with the exact limb products
// carried by the floating-point transform of fft_quarter_f64.hpp (gfx950).
//
// Same scheme as bmi_kernels_t64f.hip one size up: bootstrap key stored at 46 bits of precision (words rounded to multiples of
// 2^18; the rounded key IS the key: exported, given to the oracle) as two balanced 23-bit limbs, digits in base 2^10,
// accumulator the exact integer word / 2^18 in a double.  Per limb the sum over the 2 l digit x limb polynomial products is an
// integer below 2^45; it is computed through the folded 1,024-point complex FFT and ROUNDED TO THE NEAREST INTEGER, which
// returns it exactly (a-priori bound 0.41 < 1/2 in the header; measured distance ~2^-12, bmi_fft_margin_host) - so the kernel's
// words equal the oracle's integer arithmetic bit for bit (tests/test_gpu_torus_wide.py).
//
// One workgroup of 16 wavefronts per ciphertext (what auto dispatch runs up to 256 ciphertexts; beyond: bmi_kernels_t64w2.hip); a transform is split over FOUR wavefronts by the folded index
// mod 4 (quarters of 256 points, 4 complex points per lane, no LDS inside a quarter).  Per CMUX:
//   A  8 l forward tasks (input polynomial c, level, quarter h) over the 16 wavefronts: rotate / decompose 512 coefficients of the
//      accumulator (the oracle's integer rule), forward quarter -> tile (slot order, times W_h)
//   B  all 1,024 threads = (output polynomial o, slot, tp): a pair of lanes 32 apart shares a slot - lane tp reads quarters tp and
//      tp + 2 of each row, half a radix-4 butterfly each, ONE 2 x 2 transpose on lane bit 5 (v_permlane32_swap) completes it: two
//      frequencies A_{kappa + 256 tp}, A_{kappa + 256 (tp + 2)} per thread and row, multiplied with both limbs' key words (32
//      KiB per wavefront, row and limb; the first rows of a step are requested during the step before, row r + KD when row r is done);
//      the inverse butterfly the same way back, times conj W_h -> the sums S_h (over the tiles, after a barrier)
//   C  8 inverse tasks (o, quarter) on wavefronts 8 .. 15: the inverse quarters of BOTH limbs (two independent transforms in one
//      wavefront), nearest integers, limb 1 shifted into place, one plain read-modify-write per coefficient of the accumulator
//      (an LDS f64 atomic costs ~32 cycles per wavefront instruction: 128 of them per CMUX were 4.4 k cycles); re-centred mod 2^46
//      every 8 steps
#include <hip/hip_runtime.h>

#include "bmi_internal.hpp"
#include "fft_quarter_f64.hpp"
#include "pair_sync.hpp"
#include "t64_common.hpp"

using t64::i64;
using t64::u64;

namespace {

using fftq::C;
using fftq::static_for;
using t64::f64_to_word;
using t64::Scheme;

#ifndef BMI_T64W_PRIO
#define BMI_T64W_PRIO 0   // issue priority in the forward phase: 0 = none, 1 = a task steps down 3 (decomposition), 1 (transform), 0 (done): a wavefront that is ahead yields to the ones sharing its SIMD, 2 = the eight wavefronts that run TWO forward tasks (L = 3) keep priority 2 through the phase
#endif
#ifndef BMI_T64W_KEY_ROWS_AHEAD
#define BMI_T64W_KEY_ROWS_AHEAD 2   // key rows (of 2 l) a thread holds in registers: requested before phase A, then row r + this many when row r is done
#endif
#ifdef BMI_PHASE_PROF   // make -C csrc prof; tools/phase_prof_t64w.py
__device__ unsigned long long g_phase_w[128];
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
constexpr int WN = 2048, WLOG = 11;
constexpr int WQ = fftq::QUARTER;
constexpr int WF_THREADS = 1024;
constexpr int WF_MAX_L = 3;
constexpr int WF_RECENTRE = 8;
constexpr int WF_RES = WN / 4;                                // accumulator words per residue class mod 4
constexpr int WF_TILE_CPLX = 2 * WF_MAX_L * 4 * WQ;           // complex words of the forward tiles (the sums S overlay them)
// LDS (doubles): tables | accumulator [2 components][4 residues][512] | tiles | mod-switched LWE words
constexpr int WF_LDS_WORDS = fftq::QT_WORDS + 2 * WN + 2 * WF_TILE_CPLX + BMI_AT_WORDS;
static_assert(WF_LDS_WORDS <= BMI_LDS_WORDS_MAX, "WF_LDS_WORDS exceeds the 160 KB of LDS");
static_assert(2 * 2 * 4 * WQ <= WF_TILE_CPLX, "the sums S of both limbs and outputs fit over the tiles");

// accumulator words are kept split by residue mod 4 (a lane's points of a quarter are 256 coefficients apart and of one residue)
__device__ __forceinline__ uint32_t acc_slot(uint32_t n) { return (n & 3) * WF_RES + (n >> 2); }

// standard-domain GGSW polynomials (u64 torus words, already rounded to the key precision) -> per (polynomial, limb) 1,024
// complex words A_k / 2 in the order the multiplying threads read them: thread (w8 = slot / 32, lane) of phase B owns slot
// p = 32 w8 + (lane & 31) and, with tp = lane >> 5, the frequencies kappa(p) + 256 (tp + 2 f), f = 0, 1 - complex word
// 512 f + 64 w8 + lane (a wavefront's request is 1 KiB contiguous).  One workgroup of four wavefronts (the four quarters) per item.
__global__ void __launch_bounds__(256) k_bsk_to_w_t64(const u64 *__restrict__ std_polys, double *__restrict__ w_polys,
                                                      const double *__restrict__ g_tw, uint32_t n_polys, int prec) {
    const int limbs = t64::limbs_of(prec);
    __shared__ double lds[fftq::QT_WORDS + 4 * WQ * 2];
    for (int i = threadIdx.x; i < fftq::QT_WORDS; i += blockDim.x) lds[i] = g_tw[i];
    __syncthreads();
    const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x;   // (polynomial, limb)
    const uint32_t poly = item / limbs;
    const int j = (int)(item % limbs);
    double2 *tile = reinterpret_cast<double2 *>(lds + fftq::QT_WORDS);
    {
        double re[4], im[4];
        static_for<0, 4>([&](auto R) {
            const uint32_t m = 4 * (lane + 64 * R) + h;
            re[R] = (double)t64::limb_of((i64)std_polys[(size_t)poly * WN + m], j, prec);
            im[R] = (double)t64::limb_of((i64)std_polys[(size_t)poly * WN + m + WN / 2], j, prec);
        });
        C v[4];
        fftq::forward_quarter(h, re, im, v, lane, lds);
        static_for<0, 4>([&](auto R) { tile[h * WQ + R * 64 + lane] = double2{v[R].r, v[R].i}; });
    }
    __syncthreads();
    {
        const int p = threadIdx.x;
        const double2 q0 = tile[p], q1 = tile[WQ + p], q2 = tile[2 * WQ + p], q3 = tile[3 * WQ + p];
        const C a{q0.x + q2.x, q0.y + q2.y}, b{q0.x - q2.x, q0.y - q2.y};
        const C cc{q1.x + q3.x, q1.y + q3.y}, d{-(q1.y - q3.y), q1.x - q3.x};   // i (q1 - q3)
        const C A[4] = {a + cc, b + d, a - cc, b - d};
        double2 *o = reinterpret_cast<double2 *>(w_polys + (size_t)item * WN);
        const int w8 = p >> 5, l5 = p & 31;
        static_for<0, 4>([&](auto T) {
            constexpr int tp = T & 1, f = T >> 1;
            o[f * (WN / 4) + w8 * 64 + tp * 32 + l5] = double2{0.5 * A[T].r, 0.5 * A[T].i};
        });
    }
}

// STATS (the test hook bmi_fft_margin_host): also records the largest distance of a limb sum from the integer it is rounded to
template <int L, int BG, int PREC, bool STATS>
__global__ void __launch_bounds__(WF_THREADS)
    k_blind_rotate_w_t64f(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                          const double *__restrict__ bsk_w, const double *__restrict__ g_tw, u64 *__restrict__ out, uint32_t count,
                          uint32_t n, unsigned long long *__restrict__ stat) {
    constexpr int LIMBS = Scheme<PREC>::LIMBS, LB = Scheme<PREC>::BITS, PRE = Scheme<PREC>::PRE, AB = 64 - PRE;
    // a limb's sum: 2 L N terms of |digit| <= 2^(BG-1) times |limb| <= 2^(LB-1) - the transform's error bound is stated for this size
    static_assert(2.0 * L * WN * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) <= 0x1p45, "limb sums must stay below 2^45");
    static_assert(LIMBS == 2 && L <= WF_MAX_L && L * BG < AB, "two limbs, at most three levels");
    constexpr int KD = BMI_T64W_KEY_ROWS_AHEAD < 2 * L ? BMI_T64W_KEY_ROWS_AHEAD : 2 * L;
    extern __shared__ double lds[];
    double *acc = lds + fftq::QT_WORDS;                                     // [2 components][4 residues][512]: word / 2^PRE, exact, |.| < 2^51
    auto mod_ab = [](double t) {   // centred residue mod 2^AB of an exact integer |t| < 2^53 (ties to the negative end, like the u64 word)
        return __builtin_fma(-(double)(1ull << AB), __builtin_floor(__builtin_fma(t, 1.0 / (double)(1ull << AB), 0.5)), t);
    };
    double2 *tiles = reinterpret_cast<double2 *>(acc + 2 * WN);             // [2L rows][4 quarters][256 slots] complex
    double2 *SD = tiles;                                                    // [limb][output][4 quarters][256 slots], once the tiles are read
    uint16_t *at = reinterpret_cast<uint16_t *>(tiles + WF_TILE_CPLX);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < fftq::QT_WORDS; i += WF_THREADS) lds[i] = g_tw[i];
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += WF_THREADS) at[i] = (uint16_t)t64::modswitch<WLOG + 1>(lwe[i]);
    __syncthreads();
    {
        const u64 *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * WN;
        const uint32_t bt = at[n];
        static_for<0, 2>([&](auto Q) {
            const uint32_t nn = tid + WF_THREADS * Q;  // coefficient index
            const uint32_t e = (nn + bt) & (2 * WN - 1);
            const u64 v = tv[e & (WN - 1)];
            acc[acc_slot(nn)] = 0.0;
            acc[WN + acc_slot(nn)] = (double)((i64)((e & WN) ? (u64)0 - v : v) >> PRE);     // test polynomials are multiples of 2^PRE (host-checked)
        });
    }
    __syncthreads();
    // phase B: output polynomial, slot, and which half of the radix-4 butterfly this lane starts from
    const int mo = wave >> 3, w8 = wave & 7, tp = lane >> 5, mq = w8 * 32 + (lane & 31);
    uint32_t since_centred = 0;   // steps taken since the accumulator was last reduced mod 2^AB
    double dev = 0.0;             // STATS: largest |value - nearest integer| this lane has rounded away
    PH_DECL();

    // steps are the LWE coefficients that switch to a non-zero rotation (X^0 ACC - ACC = 0: the oracle skips those too)
    auto next_step = [&](uint32_t i) {   // uniform over the workgroup
        while (i < n && at[i] == 0) i++;
        return i;
    };
    // key words of this thread at step i: [row 2L][output 2][limb][f][64 w8 + lane] complex
    auto key_ptr = [&](uint32_t i) {
        return reinterpret_cast<const double2 *>(bsk_w + (size_t)i * 4 * L * LIMBS * WN) + (size_t)mo * LIMBS * (WN / 2) + (w8 * 64 + lane);
    };
    // rows 0 .. KD-1 of a step are requested during the step BEFORE (after its products: they land under its inverse quarters and
    // the next forward tasks), row r + KD when row r has been multiplied
    double2 kk[KD][LIMBS][2];
    auto request_first_rows = [&](const double2 *kth) {
        static_for<0, KD>([&](auto R) {
            static_for<0, LIMBS>([&](auto J) {
                kk[R][J][0] = kth[((size_t)R * 2 * LIMBS + J) * (WN / 2)];
                kk[R][J][1] = kth[((size_t)R * 2 * LIMBS + J) * (WN / 2) + WN / 4];
            });
        });
    };
    uint32_t i = next_step(0);
    if (i < n) request_first_rows(key_ptr(i));
    while (i < n) {
        const uint32_t a_t = at[i];
        const double2 *kth = key_ptr(i);
        const uint32_t i_next = next_step(i + 1);
        PH_MARK(0);   // loop head, key requests
        auto forward_task = [&](const int T) {
            const int R = T >> 2, h = T & 3, c = R / L, lev = R % L;
            const double *ac = acc + c * WN;
            double x[8];   // re[r] = x[r], im[r] = x[r + 4]
            // coefficient m_J = 4 (lane + 64 (J & 3)) + h + 1024 (J >> 2); its rotated source e_J = m_J - a_t mod 2N: a quarter of it is
            // t0 + 64 (J & 3) + 256 (J >> 2) - the low 9 bits are the slot inside the residue block, bit 9 is the sign
            const uint32_t e0 = (4 * lane + h + 2 * WN - a_t) & (2 * WN - 1);
            const uint32_t t0 = e0 >> 2, pbase = (e0 & 3) * WF_RES;
#if BMI_T64W_PRIO == 1
            __builtin_amdgcn_s_setprio(3);
#endif
            // (four coefficients at a time, fenced: the 32 key registers in flight leave this task ~90)
            static_for<0, 2>([&](auto G) {
                double vr[4], vs[4];
                static_for<0, 4>([&](auto J4) {
                    constexpr int J = G * 4 + J4;
                    const uint32_t t = t0 + 64 * (J & 3) + 256 * (J >> 2);
                    vr[J4] = ac[pbase + (t & (WF_RES - 1))];
                    vs[J4] = ac[h * WF_RES + lane + 64 * (J & 3) + 256 * (J >> 2)];
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
#if BMI_T64W_PRIO == 1
            __builtin_amdgcn_s_setprio(1);
#endif
            fftq::forward_quarter(h, re, im, v, lane, lds);
            double2 *tile = tiles + (size_t)T * WQ;
            static_for<0, 4>([&](auto R4) { tile[R4 * 64 + lane] = double2{v[R4].r, v[R4].i}; });
#if BMI_T64W_PRIO == 1
            __builtin_amdgcn_s_setprio(0);
#endif
        };
#if BMI_T64W_PRIO == 2
        if (8 * L > 16 && wave < 8 * L - 16) __builtin_amdgcn_s_setprio(2);
#endif
        forward_task(wave);
        PH_MARK(1);   // first forward task
        if constexpr (8 * L > 16) {
            pin();
            if (wave < 8 * L - 16) forward_task(16 + wave);
        }
        PH_MARK(2);   // second forward task (wavefronts 0 .. 8 L - 17)
#if BMI_T64W_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();
        PH_MARK(3);   // barrier A -> B
        C s_lo[LIMBS], s_hi[LIMBS];
        {
            C y[LIMBS][2];
            static_for<0, LIMBS>([&](auto J) { y[J][0] = y[J][1] = C{0.0, 0.0}; });
            static_for<0, 2 * L>([&](auto R) {
                const double2 lo = tiles[(size_t)(R * 4 + tp) * WQ + mq], hi = tiles[(size_t)(R * 4 + tp + 2) * WQ + mq];
                // tp = 0: (q0 + q2, q0 - q2);  tp = 1: (q1 + q3, i (q1 - q3));  after the transpose  tp = 0: (a, cc),  tp = 1: (b, d)
                C u{lo.x + hi.x, lo.y + hi.y};
                const C dl{lo.x - hi.x, lo.y - hi.y};
                C w = tp ? C{-dl.i, dl.r} : dl;
                lanetr::tr_double<5>(u.r, w.r, lane);
                lanetr::tr_double<5>(u.i, w.i, lane);
                const C a0 = u + w, a1 = u - w;   // frequencies kappa + 256 tp and kappa + 256 (tp + 2)
                static_for<0, LIMBS>([&](auto J) {
                    const double2 k0 = kk[R % KD][J][0], k1 = kk[R % KD][J][1];
                    y[J][0].r = __builtin_fma(a0.r, k0.x, __builtin_fma(-a0.i, k0.y, y[J][0].r));
                    y[J][0].i = __builtin_fma(a0.r, k0.y, __builtin_fma(a0.i, k0.x, y[J][0].i));
                    y[J][1].r = __builtin_fma(a1.r, k1.x, __builtin_fma(-a1.i, k1.y, y[J][1].r));
                    y[J][1].i = __builtin_fma(a1.r, k1.y, __builtin_fma(a1.i, k1.x, y[J][1].i));
                });
                if constexpr (R + KD < 2 * L) {
                    static_for<0, LIMBS>([&](auto J) {
                        kk[R % KD][J][0] = kth[((size_t)(R + KD) * 2 * LIMBS + J) * (WN / 2)];
                        kk[R % KD][J][1] = kth[((size_t)(R + KD) * 2 * LIMBS + J) * (WN / 2) + WN / 4];
                    });
                }
                pin();   // one row at a time: neither the next rows' tile reads nor their key requests move up (registers)
            });
            // inverse butterfly: tp = 0: (Y0 + Y2, Y0 - Y2);  tp = 1: (Y1 + Y3, -i (Y1 - Y3));  transpose;  S_tp = u + w, S_{tp+2} = u - w
            // (branch-free: a conditional read would split the block and let the compiler sink every product below it. W_0 = 1 is the
            // first word of the omega_64 table)
            const double2 wl = reinterpret_cast<const double2 *>(lds)[tp ? fftq::QT_W1 / 2 + mq : ffth::HT_T2 / 2];
            const double2 wh = reinterpret_cast<const double2 *>(lds)[(tp ? fftq::QT_W3 : fftq::QT_W2) / 2 + mq];
            static_for<0, LIMBS>([&](auto J) {
                C u = y[J][0] + y[J][1];
                const C dl = y[J][0] - y[J][1];
                C w = tp ? C{dl.i, -dl.r} : dl;
                lanetr::tr_double<5>(u.r, w.r, lane);
                lanetr::tr_double<5>(u.i, w.i, lane);
                s_lo[J] = fftq::cmul<true>(u + w, wl.x, wl.y);
                s_hi[J] = fftq::cmul<true>(u - w, wh.x, wh.y);
            });
        }
        if (i_next < n) request_first_rows(key_ptr(i_next));   // (uniform) the next step's first rows
        pin();
        PH_MARK(4);   // products of 2 L rows, inverse butterfly, next key requests
        __syncthreads();   // every thread has read the tiles: the sums may overwrite them
        static_for<0, LIMBS>([&](auto J) {
            double2 *sd = SD + (size_t)((J * 2 + mo) * 4) * WQ + mq;
            sd[tp * WQ] = double2{s_lo[J].r, s_lo[J].i};
            sd[(tp + 2) * WQ] = double2{s_hi[J].r, s_hi[J].i};
        });
        __syncthreads();
        PH_MARK(5);   // barrier, the sums to LDS, barrier
        if (wave >= 8) {   // (wavefronts 0 .. 7 ran two forward tasks where L = 3)
            const int o = (wave >> 2) & 1, h = wave & 3;
            double re[LIMBS][4], im[LIMBS][4];
            static_for<0, LIMBS>([&](auto J) {
                const double2 *sd = SD + (size_t)((J * 2 + o) * 4 + h) * WQ;
                C v[4];
                static_for<0, 4>([&](auto R) {
                    const double2 t = sd[R * 64 + lane];
                    v[R] = C{t.x, t.y};
                });
                fftq::inverse_quarter(v, re[J], im[J], lane, lds);
            });
            double *ao = acc + o * WN + h * WF_RES + lane;
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
                ao[64 * R] += place(re[0][R], re[1][R]);              // coefficient 4 (lane + 64 R) + h
                ao[64 * R + 256] += place(im[0][R], im[1][R]);        // ... + 1024
            });
        }
        PH_MARK(6);   // inverse quarters of both limbs, rounding, accumulation
        __syncthreads();
        if (++since_centred == WF_RECENTRE) {   // (uniform: counts the steps actually taken) keep the accumulator's magnitude below 2^51
            since_centred = 0;
            static_for<0, 4>([&](auto Q) { acc[tid + WF_THREADS * Q] = mod_ab(acc[tid + WF_THREADS * Q]); });
            __syncthreads();
        }
        PH_MARK(7);   // closing barrier (+ re-centring every eighth step)
        i = i_next;
    }
#ifdef BMI_PHASE_PROF
    if (blockIdx.x == 0 && lane == 0)
        for (int k_ = 0; k_ < 8; k_++) g_phase_w[wave * 8 + k_] = ph_[k_];
#endif
    if constexpr (STATS) atomicMax(stat, (unsigned long long)__double_as_longlong(dev));   // non-negative doubles order like their bit patterns
    u64 *o = out + (size_t)ct * (WN + 1);
    static_for<0, 2>([&](auto Q) {
        const uint32_t nn = tid + WF_THREADS * Q;
        const u64 a0 = f64_to_word(mod_ab(acc[acc_slot(nn)])) << PRE;
        if (nn == 0) {
            o[0] = a0;
            o[WN] = f64_to_word(mod_ab(acc[WN + acc_slot(0)])) << PRE;
        } else {
            o[WN - nn] = (u64)0 - a0;
        }
    });
}

}  // namespace

#ifdef BMI_PHASE_PROF
extern "C" int bmi_debug_phase_prof_t64w(unsigned long long *out128) {
    return (int)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_phase_w), sizeof(unsigned long long) * 128);
}
#endif

namespace bmit {

#define BMITW_LAUNCH_CHECK()                    \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// (precision, levels, base log) combinations the N = 2048 transform's error bound was established for
bool shape_supported_wide(int prec, uint32_t levels, uint32_t base_log) {
    return prec == 46 && base_log == 10 && (levels == 3 || levels == 2);
}

int launch_bsk_to_wide(const u64 *std_polys, double *w_polys, const double *g_tw_q, uint32_t n_polys, int prec, hipStream_t s) {
    if (prec != 46) return (int)hipErrorInvalidValue;
    const uint32_t items = n_polys * (uint32_t)t64::limbs_of(prec);
    hipLaunchKernelGGL(k_bsk_to_w_t64, dim3(items), dim3(256), 0, s, std_polys, w_polys, g_tw_q, n_polys, prec);
    BMITW_LAUNCH_CHECK();
    return 0;
}

template <int L, int BG, int PREC, bool STATS>
static int launch_w(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_w, const double *g_tw_q, u64 *out,
                    uint32_t count, uint32_t n, unsigned long long *stat, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)WF_LDS_WORDS * sizeof(double);
    auto kern = k_blind_rotate_w_t64f<L, BG, PREC, STATS>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3(count), dim3(WF_THREADS), lds, s, small_cts, lut_ids, luts, bsk_w, g_tw_q, out, count, n, stat);
    BMITW_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_wide(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_w, const double *g_tw_q,
                             u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels, uint32_t base_log, unsigned long long *stat,
                             hipStream_t s) {
    if (count == 0) return 0;
    if (!shape_supported_wide(prec, levels, base_log)) return (int)hipErrorInvalidValue;
    if (stat) {
        if (levels == 3) return launch_w<3, 10, 46, true>(small_cts, lut_ids, luts, bsk_w, g_tw_q, out, count, n, stat, s);
        return launch_w<2, 10, 46, true>(small_cts, lut_ids, luts, bsk_w, g_tw_q, out, count, n, stat, s);
    }
    if (levels == 3) return launch_w<3, 10, 46, false>(small_cts, lut_ids, luts, bsk_w, g_tw_q, out, count, n, nullptr, s);
    return launch_w<2, 10, 46, false>(small_cts, lut_ids, luts, bsk_w, g_tw_q, out, count, n, nullptr, s);
}

}  // namespace bmit
