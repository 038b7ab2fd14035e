// 2^64 TORUS, UNROLLED blind rotation (two LWE coefficients per step) with the exact limb products carried by the floating-point
// transform (gfx950): the latency form of bmi_kernels_t64f.hip (one workgroup of 16 wavefronts per ciphertext, half transforms of
// fft_half_f64.hpp, f64 accumulator) running the step of bmi_kernels_t64u.hip (oracle/tfhe_oracle.c ora_blind_rotate_extract_unrolled):
//
//     ACC <- ACC + sum_{j<3} (X^(c_j) - 1) (K3[i][j] [.] ACC),   c = (a + a', a, a'),
//     K3[i] = GGSW(s s'), GGSW(s (1 - s')), GGSW((1 - s) s')   of the key bits (s, s') = (s_2i, s_2i+1),
//
// one decomposition of ACC itself and one set of forward transforms per PAIR of coefficients; the factors X^c - 1 are applied in the
// transform domain (X^c at the root zeta^(4k+1) of slot frequency k is zeta^((4k+1) c), from a table of 1,024 powers of zeta).
//
// Exactness.  A limb's inverse transform returns  sum_j (X^(c_j) - 1) sum_rows digit x limb : three keys, each scaled by a factor
// of magnitude <= 2 - six times the plain step's sum.  The bootstrap key is therefore stored at 42 bits of precision (two balanced
// 21-bit limbs, words rounded to multiples of 2^22; the rounded key IS the key): |sum| < 6 * 2 l N 2^9 2^20 = 2^44.2 < 2^45, and the
// a-priori bound of the transform's error (fft_wave_f64.hpp, tools/fft_bound.py: 0.38 for six products of base-2^10 digits with
// 24-bit limbs) becomes 0.38 * 6 / 8 = 0.29 < 1/2, so the nearest integer of the inverse transform is the exact sum and the words
// equal the oracle's integer arithmetic on the same key (tests/test_gpu_torus_unrolled.py).  Price: output noise 2^-19.4 (the 48-bit
// plain key: 2^-23.1) - still far below the keyswitch noise it feeds (error_budget.py).
//
// Per step:
//   A  wavefronts 0 .. 4L-1 = (component c, level, parity h): decompose 512 coefficients of the accumulator (no rotation), forward
//      half -> tile (slot order)
//   B  all 1,024 threads = (limb, output polynomial, slot): for each of the three keys the 2L complex multiply-accumulates of
//      E + O' / E - O' with this thread's key words (chunks of three rows, the next chunk requested while this one is multiplied;
//      the first before phase A), the key's sum scaled by zeta^((4k+1) c_j) - 1 at the slot's two roots and added; then the sum
//      and the twisted difference for the inverse halves
//   C  wavefronts 0 .. 7 = (limb, output, parity): inverse half, nearest integer, shift into place, one LDS atomic add (f64) per
//      coefficient; the accumulator is re-centred mod 2^42 every eight steps
#include <hip/hip_runtime.h>

#include <atomic>

#include "bmi_internal.hpp"
#include "fft_half_f64.hpp"
#include "pair_sync.hpp"
#include "t64_common.hpp"

using t64::i64;
using t64::u64;

namespace {

using fftw::static_for;
using t64::f64_to_word;
using t64::Scheme;

constexpr int N = ffth::N;
constexpr int LOG_N = 10;
constexpr int UF_THREADS = 1024;
constexpr int UF_MAX_L = 3;
constexpr int UF_HALF = N / 2;
constexpr int UF_RECENTRE = 8;
constexpr int UF_ZP_WORDS = 2 * N;   // zeta^x for x in [0, 1024) as (re, im); zeta^(x + 1024) = -zeta^x
// tables | accumulator | tiles [2L rows][2 halves][256] complex | sums / differences [limb][output][S, D][256] complex | LWE words | zeta powers
constexpr int UF_LDS_WORDS = ffth::HT_WORDS + 2 * N + 2 * UF_MAX_L * N + 2 * 2 * N + BMI_AT_WORDS + UF_ZP_WORDS;
static_assert(UF_LDS_WORDS <= BMI_LDS_WORDS_MAX, "UF_LDS_WORDS exceeds the 160 KB of LDS");

__device__ __forceinline__ uint32_t acc_slot(uint32_t n) { return (n & 1) * UF_HALF + (n >> 1); }

template <int L, int BG, int PREC, bool STATS>
__global__ void __launch_bounds__(UF_THREADS)
    k_blind_rotate_lat2u_t64f(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                              const double *__restrict__ bsk3_latf, const double *__restrict__ g_tw_h, const double *__restrict__ g_zeta_pow,
                              u64 *__restrict__ out, uint32_t count, uint32_t n, unsigned long long *__restrict__ stat) {
    constexpr int LIMBS = Scheme<PREC>::LIMBS, LB = Scheme<PREC>::BITS, PRE = Scheme<PREC>::PRE, AB = 64 - PRE;
    // three keys, each scaled by |X^c - 1| <= 2: the limb sums are six times the plain step's
    static_assert(6.0 * 2.0 * L * N * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) <= 0x1p45, "limb sums must stay below 2^45");
    static_assert(LIMBS == 2 && L <= UF_MAX_L && L * BG < AB, "two limbs, at most three levels");
    extern __shared__ double lds[];
    double *acc = lds + ffth::HT_WORDS;                                     // [2 components][2 parities][512]: word / 2^PRE, exact, |.| < 2^51
    auto mod_ab = [](double t) {   // centred residue mod 2^AB of an exact integer |t| < 2^53 (ties to the negative end, like the u64 word)
        return __builtin_fma(-(double)(1ull << AB), __builtin_floor(__builtin_fma(t, 1.0 / (double)(1ull << AB), 0.5)), t);
    };
    double2 *tiles = reinterpret_cast<double2 *>(lds + ffth::HT_WORDS + 2 * N);   // [2L rows][2 halves][256 slots] complex
    double2 *SD = tiles + UF_MAX_L * N;                                     // [limb][output][S, D][256 slots] complex
    uint16_t *at = reinterpret_cast<uint16_t *>(SD + 2 * N);
    const double2 *ZP = reinterpret_cast<const double2 *>(reinterpret_cast<double *>(SD + 2 * N) + BMI_AT_WORDS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < ffth::HT_WORDS; i += UF_THREADS) lds[i] = g_tw_h[i];
    {
        double *zp = reinterpret_cast<double *>(SD + 2 * N) + BMI_AT_WORDS;
        for (int i = tid; i < UF_ZP_WORDS; i += UF_THREADS) zp[i] = g_zeta_pow[i];
    }
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += UF_THREADS) at[i] = (uint16_t)t64::modswitch<LOG_N + 1>(lwe[i]);
    __syncthreads();
    {
        const u64 *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at[n];
        const uint32_t nn = tid;  // coefficient index
        const uint32_t e = (nn + bt) & (2 * N - 1);
        const u64 v = tv[e & (N - 1)];
        acc[acc_slot(nn)] = 0.0;
        acc[N + acc_slot(nn)] = (double)((i64)((e & N) ? (u64)0 - v : v) >> PRE);     // test polynomials are multiples of 2^PRE (host-checked)
    }
    __syncthreads();
    const int mj = tid >> 9, mo = (tid >> 8) & 1, mq = tid & 255;   // phase B: limb, output polynomial, slot
    // X^c at the roots of slot mq: frequency k = slot_freq -> zeta^((4k+1) c) for F_k, its negative (c odd) for F_{k+256}
    const uint32_t root_e = 4 * (uint32_t)ffth::slot_freq(mq >> 6, mq & 63) + 1;
    const uint32_t pairs = (n + 1) >> 1;
    uint32_t since_centred = 0;
    double dev = 0.0;             // STATS: largest |value - nearest integer| this lane has rounded away
    constexpr int CH = L;         // rows per chunk of key words (half a key): 2 CH double2 per chunk, two chunks in flight
    constexpr int NCH = 3 * 2;    // chunks per step
    // this thread's key words: [pair][key 3][row 2L][output 2][limb][256 slots][F_k, F_{k+256}]
    auto chunk_ptr = [&](uint32_t ip, int t) {   // chunk t = (key t / 2, rows (t & 1) CH .. + CH)
        const int key = t >> 1, r0 = (t & 1) * CH;
        return reinterpret_cast<const double2 *>(bsk3_latf + (((size_t)ip * 3 + key) * 2 * L + r0) * 2 * LIMBS * N) +
               ((size_t)mo * LIMBS + mj) * (N / 2) + 2 * mq;
    };
    double2 kb[2][CH][2];
    auto request = [&](double2 (&dst)[CH][2], const double2 *p) {
        static_for<0, CH>([&](auto R) {
            dst[R][0] = p[(size_t)R * 2 * LIMBS * (N / 2)];
            dst[R][1] = p[(size_t)R * 2 * LIMBS * (N / 2) + 1];
        });
    };

    for (uint32_t ip = 0; ip < pairs; ip++) {
        const uint32_t a1 = at[2 * ip], a2 = (2 * ip + 1 < n) ? at[2 * ip + 1] : 0u;
        if ((a1 | a2) == 0) continue;  // uniform over the workgroup: every factor X^0 - 1 vanishes
        const uint32_t cj[3] = {(a1 + a2) & (2 * N - 1), a1, a2};
        request(kb[0], chunk_ptr(ip, 0));
        if (wave < 4 * L) {
            const int c = wave / (2 * L), lev = (wave % (2 * L)) >> 1, h = wave & 1;
            const double *ac = acc + c * N + h * UF_HALF;
            double x[8];   // re[r] = x[r], im[r] = x[r + 4]
            static_for<0, 8>([&](auto J) {
                const double dd = mod_ab(ac[lane + 64 * (J & 3) + 256 * (J >> 2)]);    // coefficient 2 (lane + 64 (J & 3)) + h + 512 (J >> 2)
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
            double2 *tile = tiles + (size_t)(wave >> 1) * UF_HALF + h * (UF_HALF / 2);
            static_for<0, 4>([&](auto R) { tile[R * 64 + lane] = double2{v[R].r, v[R].i}; });
        }
        __syncthreads();
        {
            ffth::C slo{0.0, 0.0}, shi{0.0, 0.0};   // sums over the three keys, scaled
            ffth::C ylo{0.0, 0.0}, yhi{0.0, 0.0};   // one key's sums over its 2L rows
            static_for<0, NCH>([&](auto T) {
                constexpr int t = T, key = t >> 1, r0 = (t & 1) * CH, cur = t & 1;
                if constexpr (t + 1 < NCH) request(kb[cur ^ 1], chunk_ptr(ip, t + 1));
                static_for<0, CH>([&](auto R) {
                    const double2 e = tiles[(size_t)(r0 + R) * UF_HALF + mq], od = tiles[(size_t)(r0 + R) * UF_HALF + UF_HALF / 2 + mq];
                    const double lr = e.x + od.x, li = e.y + od.y, hr = e.x - od.x, hi = e.y - od.y;
                    const double2 klo = kb[cur][R][0], khi = kb[cur][R][1];
                    ylo.r = __builtin_fma(lr, klo.x, __builtin_fma(-li, klo.y, ylo.r));
                    ylo.i = __builtin_fma(lr, klo.y, __builtin_fma(li, klo.x, ylo.i));
                    yhi.r = __builtin_fma(hr, khi.x, __builtin_fma(-hi, khi.y, yhi.r));
                    yhi.i = __builtin_fma(hr, khi.y, __builtin_fma(hi, khi.x, yhi.i));
                });
                if constexpr (t & 1) {   // the key is complete: scale by X^c - 1 at the slot's two roots, add
                    const uint32_t xe = (root_e * cj[key]) & (2 * N - 1);
                    double2 w = ZP[xe & (N - 1)];
                    if (xe & N) w = double2{-w.x, -w.y};
                    const double wlr = w.x - 1.0, wli = w.y;
                    const double whr = ((cj[key] & 1) ? -w.x : w.x) - 1.0, whi = (cj[key] & 1) ? -w.y : w.y;
                    slo.r += __builtin_fma(ylo.r, wlr, -(ylo.i * wli));
                    slo.i += __builtin_fma(ylo.r, wli, ylo.i * wlr);
                    shi.r += __builtin_fma(yhi.r, whr, -(yhi.i * whi));
                    shi.i += __builtin_fma(yhi.r, whi, yhi.i * whr);
                    ylo = yhi = ffth::C{0.0, 0.0};
                }
                pin();   // one chunk at a time: neither the next chunks' tile reads nor their key requests move up (registers)
            });
            const double2 w = reinterpret_cast<const double2 *>(lds + ffth::HT_W)[mq];
            const ffth::C d = ffth::cmul<true>(ffth::C{slo.r - shi.r, slo.i - shi.i}, w.x, w.y);
            double2 *sd = SD + (size_t)(mj * 2 + mo) * UF_HALF;
            sd[mq] = double2{slo.r + shi.r, slo.i + shi.i};
            sd[UF_HALF / 2 + mq] = double2{d.r, d.i};
        }
        __syncthreads();
        if (wave < 4 * LIMBS) {
            const int j = wave >> 2, o = (wave >> 1) & 1, h = wave & 1;
            const double2 *sd = SD + (size_t)(j * 2 + o) * UF_HALF + h * (UF_HALF / 2);
            ffth::C v[4];
            static_for<0, 4>([&](auto R) {
                const double2 t = sd[R * 64 + lane];
                v[R] = ffth::C{t.x, t.y};
            });
            double re[4], im[4];
            if (h) ffth::inverse_half<1>(v, re, im, lane, lds);
            else ffth::inverse_half<0>(v, re, im, lane, lds);
            double *ao = acc + o * N + h * UF_HALF + lane;
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
        __syncthreads();
        if (++since_centred == UF_RECENTRE) {   // (uniform: counts the steps actually taken) keep the accumulator's magnitude below 2^51
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


// ------------------------------------------------------------------------------------------------------------------
// THROUGHPUT form: TWO ciphertexts per workgroup sharing every key word in registers.  The one-ciphertext kernel above is bound
// by the key words a compute unit can take in (590 KB per step at ~40 B per cycle: 14.7 k of its 21 k cycles); here a thread
// multiplies each key word it loaded with the transforms of BOTH ciphertexts, so the key bytes per bootstrap halve and the step
// becomes compute-bound.  Same arithmetic, same words (every limb sum is exact): the host picks this kernel for batches wider
// than one round of the one-ciphertext kernel.
//   A  8 L forward tasks (ciphertext, component, level, parity) over the 16 wavefronts (wavefronts 0 .. 8L-17 run two)
//   B  all 1,024 threads = (limb, output polynomial, slot): per chunk of three key rows the multiply-accumulates for both
//      ciphertexts; each key's sums scaled by that ciphertext's zeta^((4k+1) c_j) - 1; then (barrier: the sums overlay the tiles)
//      the sum and the twisted difference for the inverse halves
//   C  16 inverse tasks (ciphertext, limb, output, parity), LDS f64 atomics into the accumulators
constexpr int U2_ZQ_WORDS = N;   // zeta^x for x in [0, 512) as (re, im); zeta^(x + 512) = i zeta^x
constexpr int U2_LDS_WORDS = ffth::HT_WORDS + U2_ZQ_WORDS + 2 * (2 * N) + 2 * (2 * UF_MAX_L * N) + 2 * BMI_AT_WORDS;
static_assert(U2_LDS_WORDS <= BMI_LDS_WORDS_MAX, "U2_LDS_WORDS exceeds the 160 KB of LDS");
static_assert(2 * 2 * N <= 2 * UF_MAX_L * N, "a ciphertext's sums / differences fit over its tiles");

template <int L, int BG, int PREC>
__global__ void __launch_bounds__(UF_THREADS)
    k_blind_rotate_tp2u_t64f(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                             const double *__restrict__ bsk3_latf, const double *__restrict__ g_tw_h, const double *__restrict__ g_zeta_pow,
                             u64 *__restrict__ out, uint32_t count, uint32_t n) {
    constexpr int LIMBS = Scheme<PREC>::LIMBS, LB = Scheme<PREC>::BITS, PRE = Scheme<PREC>::PRE, AB = 64 - PRE;
    static_assert(6.0 * 2.0 * L * N * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) <= 0x1p45, "limb sums must stay below 2^45");
    static_assert(LIMBS == 2 && L <= UF_MAX_L && L * BG < AB, "two limbs, at most three levels");
    constexpr int TILE_CPLX = UF_MAX_L * N;   // complex words of one ciphertext's tiles: [2L rows][2 halves][256]
    extern __shared__ double lds[];
    const double2 *ZQ = reinterpret_cast<const double2 *>(lds + ffth::HT_WORDS);
    double *acc_all = lds + ffth::HT_WORDS + U2_ZQ_WORDS;                 // [2 ciphertexts][2 components][2 parities][512]
    double2 *tiles_all = reinterpret_cast<double2 *>(acc_all + 2 * 2 * N);   // [2 ciphertexts][TILE_CPLX]; the sums overlay them
    uint16_t *at_all = reinterpret_cast<uint16_t *>(tiles_all + 2 * TILE_CPLX);   // [2][BMI_AT_WORDS * 4]
    auto mod_ab = [](double t) {
        return __builtin_fma(-(double)(1ull << AB), __builtin_floor(__builtin_fma(t, 1.0 / (double)(1ull << AB), 0.5)), t);
    };
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < ffth::HT_WORDS; i += UF_THREADS) lds[i] = g_tw_h[i];
    for (int i = tid; i < U2_ZQ_WORDS; i += UF_THREADS) lds[ffth::HT_WORDS + i] = g_zeta_pow[i];   // the first 512 powers
    const uint32_t ct0 = 2 * blockIdx.x;
    const uint32_t cts[2] = {ct0, ct0 + 1 < count ? ct0 + 1 : ct0};      // an odd batch: the last workgroup runs its ciphertext twice
    static_for<0, 2>([&](auto Z) {
        const u64 *lwe = small_cts + (size_t)cts[Z] * (n + 1);
        uint16_t *at = at_all + Z * BMI_AT_WORDS * 4;
        for (uint32_t i = tid; i <= n; i += UF_THREADS) at[i] = (uint16_t)t64::modswitch<LOG_N + 1>(lwe[i]);
    });
    __syncthreads();
    static_for<0, 2>([&](auto Z) {
        const u64 *tv = luts + (size_t)(lut_ids[cts[Z]] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at_all[Z * BMI_AT_WORDS * 4 + n];
        const uint32_t nn = tid;
        const uint32_t e = (nn + bt) & (2 * N - 1);
        const u64 v = tv[e & (N - 1)];
        double *acc = acc_all + Z * 2 * N;
        acc[acc_slot(nn)] = 0.0;
        acc[N + acc_slot(nn)] = (double)((i64)((e & N) ? (u64)0 - v : v) >> PRE);
    });
    __syncthreads();
    const int mj = tid >> 9, mo = (tid >> 8) & 1, mq = tid & 255;
    const uint32_t root_e = 4 * (uint32_t)ffth::slot_freq(mq >> 6, mq & 63) + 1;
    const uint32_t pairs = (n + 1) >> 1;
    uint32_t since_centred = 0;
    constexpr int CH = L, NCH = 3 * 2;
    auto chunk_ptr = [&](uint32_t ip, int t) {
        const int key = t >> 1, r0 = (t & 1) * CH;
        return reinterpret_cast<const double2 *>(bsk3_latf + (((size_t)ip * 3 + key) * 2 * L + r0) * 2 * LIMBS * N) +
               ((size_t)mo * LIMBS + mj) * (N / 2) + 2 * mq;
    };
    double2 kb[2][CH][2];
    auto request = [&](double2 (&dst)[CH][2], const double2 *p) {
        static_for<0, CH>([&](auto R) {
            dst[R][0] = p[(size_t)R * 2 * LIMBS * (N / 2)];
            dst[R][1] = p[(size_t)R * 2 * LIMBS * (N / 2) + 1];
        });
    };
    // zeta^x, x in [0, 2N): the table holds the first quadrant, the others are its multiples by i
    auto zeta_pow = [&](uint32_t xe) {
        const double2 w = ZQ[xe & 511];
        const uint32_t quad = (xe >> 9) & 3;
        const double sr = (quad & 1) ? -w.y : w.x, si = (quad & 1) ? w.x : w.y;     // times i for odd quadrants
        return double2{(quad & 2) ? -sr : sr, (quad & 2) ? -si : si};               // times -1 for quadrants 2, 3
    };

    for (uint32_t ip = 0; ip < pairs; ip++) {
        uint32_t cj[2][3];
        static_for<0, 2>([&](auto Z) {
            const uint16_t *at = at_all + Z * BMI_AT_WORDS * 4;
            const uint32_t a1 = at[2 * ip], a2 = (2 * ip + 1 < n) ? at[2 * ip + 1] : 0u;
            cj[Z][0] = (a1 + a2) & (2 * N - 1);
            cj[Z][1] = a1;
            cj[Z][2] = a2;
        });
        if ((cj[0][1] | cj[0][2] | cj[1][1] | cj[1][2]) == 0) continue;  // uniform: every factor X^0 - 1 of both ciphertexts vanishes
        request(kb[0], chunk_ptr(ip, 0));
        auto forward_task = [&](const int T) {
            const int z = T / (4 * L), w12 = T % (4 * L);
            const int c = w12 / (2 * L), lev = (w12 % (2 * L)) >> 1, h = w12 & 1;
            const double *ac = acc_all + z * 2 * N + c * N + h * UF_HALF;
            double x[8];
            static_for<0, 8>([&](auto J) {
                const double dd = mod_ab(ac[lane + 64 * (J & 3) + 256 * (J >> 2)]);
                double r = __builtin_floor(__builtin_fma(dd, 1.0 / (double)(1ull << (AB - L * BG)), 0.5));
                double d = r;
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
            double2 *tile = tiles_all + (size_t)z * TILE_CPLX + (size_t)(w12 >> 1) * UF_HALF + h * (UF_HALF / 2);
            static_for<0, 4>([&](auto R) { tile[R * 64 + lane] = double2{v[R].r, v[R].i}; });
        };
        if (wave < 8 * L) forward_task(wave);
        if constexpr (8 * L > 16) {
            pin();
            if (wave < 8 * L - 16) forward_task(16 + wave);
        }
        __syncthreads();
        ffth::C sS[2], sD[2];
        {
            ffth::C slo[2], shi[2], ylo[2], yhi[2];
            static_for<0, 2>([&](auto Z) { slo[Z] = shi[Z] = ylo[Z] = yhi[Z] = ffth::C{0.0, 0.0}; });
            static_for<0, NCH>([&](auto T) {
                constexpr int t = T, key = t >> 1, r0 = (t & 1) * CH, cur = t & 1;
                if constexpr (t + 1 < NCH) request(kb[cur ^ 1], chunk_ptr(ip, t + 1));
                static_for<0, 2>([&](auto Z) {
                    const double2 *tl = tiles_all + (size_t)Z * TILE_CPLX;
                    static_for<0, CH>([&](auto R) {
                        const double2 e = tl[(size_t)(r0 + R) * UF_HALF + mq], od = tl[(size_t)(r0 + R) * UF_HALF + UF_HALF / 2 + mq];
                        const double lr = e.x + od.x, li = e.y + od.y, hr = e.x - od.x, hi = e.y - od.y;
                        const double2 klo = kb[cur][R][0], khi = kb[cur][R][1];
                        ylo[Z].r = __builtin_fma(lr, klo.x, __builtin_fma(-li, klo.y, ylo[Z].r));
                        ylo[Z].i = __builtin_fma(lr, klo.y, __builtin_fma(li, klo.x, ylo[Z].i));
                        yhi[Z].r = __builtin_fma(hr, khi.x, __builtin_fma(-hi, khi.y, yhi[Z].r));
                        yhi[Z].i = __builtin_fma(hr, khi.y, __builtin_fma(hi, khi.x, yhi[Z].i));
                    });
                    if constexpr (t & 1) {   // the key is complete: scale by this ciphertext's X^c - 1 at the slot's two roots, add
                        const uint32_t c = cj[Z][key];
                        const double2 w = zeta_pow((root_e * c) & (2 * N - 1));
                        const double wlr = w.x - 1.0, wli = w.y;
                        const double whr = ((c & 1) ? -w.x : w.x) - 1.0, whi = (c & 1) ? -w.y : w.y;
                        slo[Z].r += __builtin_fma(ylo[Z].r, wlr, -(ylo[Z].i * wli));
                        slo[Z].i += __builtin_fma(ylo[Z].r, wli, ylo[Z].i * wlr);
                        shi[Z].r += __builtin_fma(yhi[Z].r, whr, -(yhi[Z].i * whi));
                        shi[Z].i += __builtin_fma(yhi[Z].r, whi, yhi[Z].i * whr);
                        ylo[Z] = yhi[Z] = ffth::C{0.0, 0.0};
                    }
                });
                pin();
            });
            const double2 w = reinterpret_cast<const double2 *>(lds + ffth::HT_W)[mq];
            static_for<0, 2>([&](auto Z) {
                sS[Z] = ffth::C{slo[Z].r + shi[Z].r, slo[Z].i + shi[Z].i};
                sD[Z] = ffth::cmul<true>(ffth::C{slo[Z].r - shi[Z].r, slo[Z].i - shi[Z].i}, w.x, w.y);
            });
        }
        __syncthreads();   // every thread has read the tiles: the sums may overwrite them
        static_for<0, 2>([&](auto Z) {
            double2 *sd = tiles_all + (size_t)Z * TILE_CPLX + (size_t)(mj * 2 + mo) * UF_HALF;
            sd[mq] = double2{sS[Z].r, sS[Z].i};
            sd[UF_HALF / 2 + mq] = double2{sD[Z].r, sD[Z].i};
        });
        __syncthreads();
        {
            const int z = wave >> 3, j = (wave >> 2) & 1, o = (wave >> 1) & 1, h = wave & 1;
            const double2 *sd = tiles_all + (size_t)z * TILE_CPLX + (size_t)(j * 2 + o) * UF_HALF + h * (UF_HALF / 2);
            ffth::C v[4];
            static_for<0, 4>([&](auto R) {
                const double2 t = sd[R * 64 + lane];
                v[R] = ffth::C{t.x, t.y};
            });
            double re[4], im[4];
            if (h) ffth::inverse_half<1>(v, re, im, lane, lds);
            else ffth::inverse_half<0>(v, re, im, lane, lds);
            double *ao = acc_all + z * 2 * N + o * N + h * UF_HALF + lane;
            auto place = [&](double v) {
                double xr = __builtin_rint(v);
                if (j == 0) return xr;
                constexpr double W = (double)(1ull << (AB - LB));
                xr = __builtin_fma(-W, __builtin_rint(xr * (1.0 / W)), xr);
                return xr * (double)(1ull << LB);
            };
            static_for<0, 4>([&](auto R) {
                atomicAdd(ao + 64 * R, place(re[R]));
                atomicAdd(ao + 64 * R + 256, place(im[R]));
            });
        }
        __syncthreads();
        if (++since_centred == UF_RECENTRE) {
            since_centred = 0;
            static_for<0, 4>([&](auto Q) { acc_all[tid + UF_THREADS * Q] = mod_ab(acc_all[tid + UF_THREADS * Q]); });
            __syncthreads();
        }
    }
    static_for<0, 2>([&](auto Z) {
        if (Z == 1 && cts[1] == cts[0]) return;   // the padding copy of an odd batch
        const double *acc = acc_all + Z * 2 * N;
        u64 *o = out + (size_t)cts[Z] * (N + 1);
        const uint32_t nn = tid;
        const u64 a0 = f64_to_word(mod_ab(acc[acc_slot(nn)])) << PRE;
        if (nn == 0) {
            o[0] = a0;
            o[N] = f64_to_word(mod_ab(acc[N + acc_slot(0)])) << PRE;
        } else {
            o[N - nn] = (u64)0 - a0;
        }
    });
}

}  // namespace

namespace bmit {

// (precision, levels, base log) of the unrolled floating-point-transform kernel: the 42-bit key (two 21-bit limbs) in base 2^10
bool shape_supported_unrolled_fft(int prec, uint32_t levels, uint32_t base_log) {
    return prec == 42 && base_log == 10 && (levels == 3 || levels == 2);
}

template <int L, int BG, int PREC, bool STATS>
static int launch_lat2u_t64f(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk3_latf, const double *g_tw_h,
                             const double *g_zeta_pow, u64 *out, uint32_t count, uint32_t n, unsigned long long *stat, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)UF_LDS_WORDS * sizeof(double);
    auto kern = k_blind_rotate_lat2u_t64f<L, BG, PREC, STATS>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3(count), dim3(UF_THREADS), lds, s, small_cts, lut_ids, luts, bsk3_latf, g_tw_h, g_zeta_pow, out, count, n, stat);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

template <int L, int BG, int PREC>
static int launch_tp2u_t64f(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk3_latf, const double *g_tw_h,
                            const double *g_zeta_pow, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)U2_LDS_WORDS * sizeof(double);
    auto kern = k_blind_rotate_tp2u_t64f<L, BG, PREC>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3((count + 1) / 2), dim3(UF_THREADS), lds, s, small_cts, lut_ids, luts, bsk3_latf, g_tw_h, g_zeta_pow, out, count, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// two ciphertexts per workgroup (key words shared in registers): the throughput form, same words as launch_blind_rotate_lat2u_fft
int launch_blind_rotate_tp2u_fft(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk3_latf,
                                 const double *g_tw_h, const double *g_zeta_pow, u64 *out, uint32_t count, uint32_t n, int prec,
                                 uint32_t levels, uint32_t base_log, hipStream_t s) {
    if (count == 0) return 0;
    if (!shape_supported_unrolled_fft(prec, levels, base_log)) return (int)hipErrorInvalidValue;
    if (levels == 3) return launch_tp2u_t64f<3, 10, 42>(small_cts, lut_ids, luts, bsk3_latf, g_tw_h, g_zeta_pow, out, count, n, s);
    return launch_tp2u_t64f<2, 10, 42>(small_cts, lut_ids, luts, bsk3_latf, g_tw_h, g_zeta_pow, out, count, n, s);
}

int launch_blind_rotate_lat2u_fft(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk3_latf,
                                  const double *g_tw_h, const double *g_zeta_pow, u64 *out, uint32_t count, uint32_t n, int prec,
                                  uint32_t levels, uint32_t base_log, unsigned long long *stat, hipStream_t s) {
    if (count == 0) return 0;
    if (!shape_supported_unrolled_fft(prec, levels, base_log)) return (int)hipErrorInvalidValue;
    if (stat) {
        if (levels == 3) return launch_lat2u_t64f<3, 10, 42, true>(small_cts, lut_ids, luts, bsk3_latf, g_tw_h, g_zeta_pow, out, count, n, stat, s);
        return launch_lat2u_t64f<2, 10, 42, true>(small_cts, lut_ids, luts, bsk3_latf, g_tw_h, g_zeta_pow, out, count, n, stat, s);
    }
    if (levels == 3) return launch_lat2u_t64f<3, 10, 42, false>(small_cts, lut_ids, luts, bsk3_latf, g_tw_h, g_zeta_pow, out, count, n, nullptr, s);
    return launch_lat2u_t64f<2, 10, 42, false>(small_cts, lut_ids, luts, bsk3_latf, g_tw_h, g_zeta_pow, out, count, n, nullptr, s);
}

}  // namespace bmit
