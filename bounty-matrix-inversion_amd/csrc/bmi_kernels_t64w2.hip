// 2^64 TORUS at N = 2048, THROUGHPUT form: two ciphertexts per workgroup sharing every key word a thread loads (gfx950).
//
// Same function and the same words as k_blind_rotate_w_t64f (bmi_kernels_t64w.hip: key at 46 bits of precision = two balanced
// 23-bit limbs, digits in base 2^10, accumulator the exact integer word / 2^18 in a double, limb sums through the folded
// 1,024-point complex FFT of fft_quarter_f64.hpp and ROUNDED TO THE NEAREST INTEGER).  What changes is the arrangement: that kernel
// keeps all 2 l digit transforms of a CMUX in LDS (96 KB of tiles: one ciphertext per compute unit) and a CMUX takes in 393 KB of
// key for ONE ciphertext.  Here a CMUX walks the decomposition levels as the N = 4096 kernel does (bmi_kernels_t64q.hip) - one
// level's tiles in LDS, the running sums in registers across the levels - which leaves room for a SECOND ciphertext, and every
// key word is multiplied with both (half the key bytes per bootstrap: the batches a server runs, bmi dispatch for count > 256).
//   for each level:  A  16 forward tasks (ciphertext, component c, quarter h): rotate / decompose 512 coefficients, digit `level`,
//                       forward quarter -> tile (slot order, times W_h)
//                    B  all 1,024 threads = (slot, output polynomial, limb): per row and ciphertext the radix-4 butterfly over the
//                       four tiles and four complex multiply-accumulates with this thread's key words (loaded once)
//   then per ciphertext the inverse butterfly, and per limb: the sums to LDS (over the tiles), 16 inverse tasks (ciphertext, output,
//   quarter): conj W_h, inverse quarter, nearest integer, shift into place, plain read-modify-write of the accumulator.
#include <hip/hip_runtime.h>

#include <atomic>

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

#ifndef BMI_T64W2_LEVEL_AHEAD
#define BMI_T64W2_LEVEL_AHEAD 0   // 1: a level's key words requested during the level before (A/B)
#endif
constexpr int WN = 2048, WLOG = 11;
constexpr int WS = fftq::QUARTER;                // slots per quarter
constexpr int W2_THREADS = 1024;
constexpr int W2_MAX_L = 3;
constexpr int W2_RECENTRE = 8;
constexpr int W2_RES = WN / 4;                   // accumulator words per residue class mod 4
constexpr int W2_TILE_CPLX = 2 * 2 * 4 * WS;     // one level's tiles [ciphertext 2][component 2][quarter 4][256]; one limb's sums overlay them
// LDS (doubles): tables | accumulators [ciphertext 2][component 2][residue 4][512] | tiles | mod-switched LWE words of both
constexpr int W2_LDS_WORDS = fftq::QT_WORDS + 2 * 2 * WN + 2 * W2_TILE_CPLX + 2 * BMI_AT_WORDS;
static_assert(W2_LDS_WORDS <= BMI_LDS_WORDS_MAX, "W2_LDS_WORDS exceeds the 160 KB of LDS");

__device__ __forceinline__ uint32_t acc_slot(uint32_t n) { return (n & 3) * W2_RES + (n >> 2); }

// an empty statement that reads and writes the sums: they must be in registers here (the branches of the next task would otherwise
// let the compiler sink the multiply-adds below them: bmi_kernels_t64q.hip)
__device__ __forceinline__ void keep(C (&y)[2][4]) {
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int t = 0; t < 4; t++) asm volatile("" : "+v"(y[b][t].r), "+v"(y[b][t].i));
}

// sum_h i^(h t) q_h in q[t]  (INV: i^(-h t))
template <bool INV>
__device__ __forceinline__ void dft4(C (&q)[4]) {
    const C a0{q[0].r + q[2].r, q[0].i + q[2].i}, a1{q[0].r - q[2].r, q[0].i - q[2].i};
    const C b0{q[1].r + q[3].r, q[1].i + q[3].i}, b1{q[1].r - q[3].r, q[1].i - q[3].i};
    q[0] = C{a0.r + b0.r, a0.i + b0.i};
    q[2] = C{a0.r - b0.r, a0.i - b0.i};
    // i b1 = (-b1.i, b1.r)
    const C p{a1.r - b1.i, a1.i + b1.r}, m{a1.r + b1.i, a1.i - b1.r};
    q[1] = INV ? m : p;
    q[3] = INV ? p : m;
}

// standard-domain GGSW polynomials (u64 torus words, already rounded to the key precision) -> per (polynomial, limb) 1,024 complex
// words A_k / 2 as [t 4][slot 256]: frequency kappa(slot) + 256 t.  One workgroup of four wavefronts (the quarters) per item.
__global__ void __launch_bounds__(256) k_bsk_to_w2_t64(const u64 *__restrict__ std_polys, double *__restrict__ w2_polys,
                                                       const double *__restrict__ g_tw, uint32_t n_polys, int prec) {
    const int limbs = t64::limbs_of(prec);
    __shared__ double lds[fftq::QT_WORDS + 4 * WS * 2];
    for (int i = threadIdx.x; i < fftq::QT_WORDS; i += blockDim.x) lds[i] = g_tw[i];
    __syncthreads();
    const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x;
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
        static_for<0, 4>([&](auto R) { tile[h * WS + R * 64 + lane] = double2{v[R].r, v[R].i}; });
    }
    __syncthreads();
    {
        const int p = threadIdx.x;
        C q[4];
        static_for<0, 4>([&](auto H) {
            const double2 t = tile[H * WS + p];
            q[H] = C{t.x, t.y};
        });
        dft4<false>(q);
        double2 *o = reinterpret_cast<double2 *>(w2_polys + (size_t)item * WN);
        static_for<0, 4>([&](auto T) { o[T * WS + p] = double2{0.5 * q[T].r, 0.5 * q[T].i}; });
    }
}

template <int L, int BG, int PREC>
__global__ void __launch_bounds__(W2_THREADS)
    k_blind_rotate_w2_t64f(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                           const double *__restrict__ bsk_w2, const double *__restrict__ g_tw, u64 *__restrict__ out, uint32_t count,
                           uint32_t n) {
    constexpr int LIMBS = Scheme<PREC>::LIMBS, LB = Scheme<PREC>::BITS, PRE = Scheme<PREC>::PRE, AB = 64 - PRE;
    static_assert(2.0 * L * WN * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) <= 0x1p45, "limb sums must stay below 2^45");
    static_assert(LIMBS == 2 && L <= W2_MAX_L && L * BG < AB, "two limbs, at most three levels");
    extern __shared__ double lds[];
    double *accs = lds + fftq::QT_WORDS;                                    // [ciphertext][2 components][4 residues][512]: word / 2^PRE, exact
    auto mod_ab = [](double t) {   // centred residue mod 2^AB of an exact integer |t| < 2^53 (ties to the negative end, like the u64 word)
        return __builtin_fma(-(double)(1ull << AB), __builtin_floor(__builtin_fma(t, 1.0 / (double)(1ull << AB), 0.5)), t);
    };
    double2 *tiles = reinterpret_cast<double2 *>(accs + 2 * 2 * WN);        // [ciphertext 2][component 2][quarter 4][256 slots] of the current level
    double2 *SD = tiles;                                                    // one limb's sums [ciphertext 2][output 2][quarter 4][256 slots]
    uint16_t *at = reinterpret_cast<uint16_t *>(tiles + W2_TILE_CPLX);      // [ciphertext 2][BMI_AT_WORDS * 4]
    constexpr int AT_STRIDE = BMI_AT_WORDS * 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < fftq::QT_WORDS; i += W2_THREADS) lds[i] = g_tw[i];
    // the two ciphertexts of this workgroup (an odd batch: the last workgroup runs its one ciphertext twice and writes it once)
    const uint32_t ct0 = blockIdx.x * 2;
    const bool live1 = ct0 + 1 < count;
    const uint32_t cts[2] = {ct0, live1 ? ct0 + 1 : ct0};
    for (int b = 0; b < 2; b++) {
        const u64 *lwe = small_cts + (size_t)cts[b] * (n + 1);
        for (uint32_t i = tid; i <= n; i += W2_THREADS) at[b * AT_STRIDE + i] = (uint16_t)t64::modswitch<WLOG + 1>(lwe[i]);
    }
    __syncthreads();
    for (int b = 0; b < 2; b++) {
        const u64 *tv = luts + (size_t)(lut_ids[cts[b]] & (BMI_LUT_CAP - 1)) * WN;
        const uint32_t bt = at[b * AT_STRIDE + n];
        double *acc = accs + b * 2 * WN;
        static_for<0, 2>([&](auto Q) {
            const uint32_t nn = tid + W2_THREADS * Q;  // coefficient index
            const uint32_t e = (nn + bt) & (2 * WN - 1);
            const u64 v = tv[e & (WN - 1)];
            acc[acc_slot(nn)] = 0.0;
            acc[WN + acc_slot(nn)] = (double)((i64)((e & WN) ? (u64)0 - v : v) >> PRE);     // test polynomials are multiples of 2^PRE (host-checked)
        });
    }
    __syncthreads();
    // phase B: slot, output polynomial, limb (the four combinations of a slot sit 16 lanes apart: their tile reads coincide)
    const int slot = wave * 16 + (lane & 15), mo = lane >> 5, mj = (lane >> 4) & 1;
    // phases A and C: wavefront = (ciphertext, component / output, quarter)
    const int wb = wave >> 3, wc = (wave >> 2) & 1, wh = wave & 3;
    uint32_t since_centred = 0;

    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_own = at[wb * AT_STRIDE + i];                       // this wavefront's ciphertext (phase A)
        if ((at[i] | at[AT_STRIDE + i]) == 0) continue;                      // uniform over the workgroup: neither ciphertext rotates
        // key words of this thread: [row 2L][output 2][limb][t 4][256 slots] complex
        const double2 *kth = reinterpret_cast<const double2 *>(bsk_w2 + (size_t)i * 4 * L * LIMBS * WN) + ((size_t)mo * LIMBS + mj) * (WN / 2) + slot;
        auto row_ptr = [&](int R) { return kth + (size_t)R * 2 * LIMBS * (WN / 2); };
        C y[2][4];
        static_for<0, 2>([&](auto B) { static_for<0, 4>([&](auto T) { y[B][T] = C{0.0, 0.0}; }); });
        double2 kw[2][4];
#if BMI_T64W2_LEVEL_AHEAD
        static_for<0, 2>([&](auto CC) { static_for<0, 4>([&](auto T) { kw[CC][T] = row_ptr(CC * L)[T * WS]; }); });
#endif
        static_for<0, L>([&](auto LEV) {
            constexpr int lev = LEV;
            {   // phase A
                const double *ac = accs + wb * 2 * WN + wc * WN;
                double x[8];   // re[r] = x[r], im[r] = x[r + 4]
                // coefficient m_J = 4 (lane + 64 (J & 3)) + h + 1024 (J >> 2); its rotated source e_J = m_J - a_t mod 2N: a quarter of it is
                // t0 + 64 (J & 3) + 256 (J >> 2) - the low 9 bits are the slot inside the residue block, bit 9 is the sign
                const uint32_t e0 = (4 * lane + wh + 2 * WN - a_own) & (2 * WN - 1);
                const uint32_t t0 = e0 >> 2, pbase = (e0 & 3) * W2_RES;
                static_for<0, 2>([&](auto G) {
                    double vr[4], vs[4];
                    static_for<0, 4>([&](auto J4) {
                        constexpr int J = G * 4 + J4;
                        const uint32_t t = t0 + 64 * (J & 3) + 256 * (J >> 2);
                        vr[J4] = ac[pbase + (t & (W2_RES - 1))];
                        vs[J4] = ac[wh * W2_RES + lane + 64 * (J & 3) + 256 * (J >> 2)];
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
                fftq::forward_quarter(wh, re, im, v, lane, lds);
                double2 *tile = tiles + (size_t)wave * WS;     // wave = (ciphertext 2, component 2, quarter 4)
                static_for<0, 4>([&](auto R4) { tile[R4 * 64 + lane] = double2{v[R4].r, v[R4].i}; });
            }
#if !BMI_T64W2_LEVEL_AHEAD
            // this thread's key words of the level's two rows (component 0 / 1): they land under the barrier
            static_for<0, 2>([&](auto CC) { static_for<0, 4>([&](auto T) { kw[CC][T] = row_ptr(CC * L + lev)[T * WS]; }); });
#endif
            pin();
            __syncthreads();
            static_for<0, 2>([&](auto CC) {   // rows (component CC, this level)
                static_for<0, 2>([&](auto B) {
                    C q[4];
                    static_for<0, 4>([&](auto H) {
                        const double2 t = tiles[(size_t)((B * 2 + CC) * 4 + H) * WS + slot];
                        q[H] = C{t.x, t.y};
                    });
                    dft4<false>(q);   // frequency kappa + 256 t in q[t]
                    static_for<0, 4>([&](auto T) {
                        y[B][T].r = __builtin_fma(q[T].r, kw[CC][T].x, __builtin_fma(-q[T].i, kw[CC][T].y, y[B][T].r));
                        y[B][T].i = __builtin_fma(q[T].r, kw[CC][T].y, __builtin_fma(q[T].i, kw[CC][T].x, y[B][T].i));
                    });
                });
                keep(y);
                pin();
            });
#if BMI_T64W2_LEVEL_AHEAD
            if constexpr (lev + 1 < L)
                static_for<0, 2>([&](auto CC) { static_for<0, 4>([&](auto T) { kw[CC][T] = row_ptr(CC * L + lev + 1)[T * WS]; }); });
#endif
            __syncthreads();   // every thread has read this level's tiles: the next level (or the sums) may overwrite them
        });
        static_for<0, 2>([&](auto B) { dft4<true>(y[B]); });   // sum_t i^(-h t) Y_t in y[b][h]; conj W_h is applied by the inverse task
        static_for<0, LIMBS>([&](auto J) {
            if (mj == J) {
                static_for<0, 2>([&](auto B) {
                    static_for<0, 4>([&](auto H) { SD[(size_t)((B * 2 + mo) * 4 + H) * WS + slot] = double2{y[B][H].r, y[B][H].i}; });
                });
            }
            __syncthreads();
            {
                const double2 *sd = SD + (size_t)wave * WS;     // wave = (ciphertext, output, quarter)
                C v[4];
                static_for<0, 4>([&](auto R) {
                    const double2 t = sd[R * 64 + lane];
                    v[R] = C{t.x, t.y};
                });
                if (wh != 0) {   // (uniform over the wavefront) conj W_h
                    const double2 *w = reinterpret_cast<const double2 *>(lds + fftq::w_offset(wh));
                    static_for<0, 4>([&](auto R) {
                        const double2 t = w[R * 64 + lane];
                        v[R] = fftq::cmul<true>(v[R], t.x, t.y);
                    });
                }
                double re[4], im[4];
                fftq::inverse_quarter(v, re, im, lane, lds);
                double *ao = accs + wb * 2 * WN + wc * WN + wh * W2_RES + lane;
                // the limb's exact integer (|.| < 2^45: nearest integer of the transform's output), shifted into place: limb 1 x 2^LB mod
                // 2^AB, of which only the low AB - LB bits survive
                constexpr int j = J;
                auto place = [&](double v0) {
                    double xr = __builtin_rint(v0);
                    if constexpr (j > 0) {
                        constexpr double W = (double)(1ull << (AB - LB));
                        xr = __builtin_fma(-W, __builtin_rint(xr * (1.0 / W)), xr) * (double)(1ull << LB);
                    }
                    return xr;
                };
                static_for<0, 4>([&](auto R) {
                    ao[64 * R] += place(re[R]);              // coefficient 4 (lane + 64 R) + h
                    ao[64 * R + 256] += place(im[R]);        // ... + 1024
                });
            }
            __syncthreads();
        });
        if (++since_centred == W2_RECENTRE) {   // (uniform: counts the steps actually taken) keep the accumulators' magnitude below 2^51
            since_centred = 0;
            static_for<0, 8>([&](auto Q) { accs[tid + W2_THREADS * Q] = mod_ab(accs[tid + W2_THREADS * Q]); });
            __syncthreads();
        }
    }
    for (int b = 0; b < (live1 ? 2 : 1); b++) {
        const double *acc = accs + b * 2 * WN;
        u64 *o = out + (size_t)cts[b] * (WN + 1);
        static_for<0, 2>([&](auto Q) {
            const uint32_t nn = tid + W2_THREADS * Q;
            const u64 a0 = f64_to_word(mod_ab(acc[acc_slot(nn)])) << PRE;
            if (nn == 0) {
                o[0] = a0;
                o[WN] = f64_to_word(mod_ab(acc[WN + acc_slot(0)])) << PRE;
            } else {
                o[WN - nn] = (u64)0 - a0;
            }
        });
    }
}

}  // namespace

namespace bmit {

#define BMITW2_LAUNCH_CHECK()                   \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int launch_bsk_to_wide2(const u64 *std_polys, double *w2_polys, const double *g_tw_q, uint32_t n_polys, int prec, hipStream_t s) {
    if (prec != 46) return (int)hipErrorInvalidValue;
    const uint32_t items = n_polys * (uint32_t)t64::limbs_of(prec);
    hipLaunchKernelGGL(k_bsk_to_w2_t64, dim3(items), dim3(256), 0, s, std_polys, w2_polys, g_tw_q, n_polys, prec);
    BMITW2_LAUNCH_CHECK();
    return 0;
}

template <int L, int BG, int PREC>
static int launch_w2(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_w2, const double *g_tw_q, u64 *out,
                     uint32_t count, uint32_t n, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)W2_LDS_WORDS * sizeof(double);
    auto kern = k_blind_rotate_w2_t64f<L, BG, PREC>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3((count + 1) / 2), dim3(W2_THREADS), lds, s, small_cts, lut_ids, luts, bsk_w2, g_tw_q, out, count, n);
    BMITW2_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_wide2(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_w2, const double *g_tw_q,
                              u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels, uint32_t base_log, hipStream_t s) {
    if (count == 0) return 0;
    if (!shape_supported_wide(prec, levels, base_log)) return (int)hipErrorInvalidValue;
    if (levels == 3) return launch_w2<3, 10, 46>(small_cts, lut_ids, luts, bsk_w2, g_tw_q, out, count, n, s);
    return launch_w2<2, 10, 46>(small_cts, lut_ids, luts, bsk_w2, g_tw_q, out, count, n, s);
}

}  // namespace bmit
