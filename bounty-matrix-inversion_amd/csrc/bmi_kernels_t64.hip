// HIP kernels (gfx950) for ciphertexts on the 2^64 TORUS (q = 2^64 exactly: the modulus Concrete computes on; SURVEY.md
// section 7 hard part 1, option A).  No transform exists mod 2^64, so the external product is computed EXACTLY over the
// integers and only then reduced:
//
//   * the six digit polynomials of a CMUX are small (|d| <= 2^14) and are transformed once, mod p = 2^49 - 720895, with
//     the f64 wave transform of ntt_wave_f64.hpp (the machinery of the 49-bit field kernels);
//   * every 64-bit bootstrap-key word k (read as a signed integer) is split once, at key upload, into LIMBS balanced
//     limbs, k = 2^PRE sum_j k_j 2^(BITS j) (t64_common.hpp: the scheme follows from the precision the key is stored at),
//     and each limb polynomial is stored in the transform domain;
//   * per limb, the sum over the six GGSW rows of digit x limb is an integer of magnitude
//     < 6 * 1024 * 2^(Bg-1) * 2^(BITS-1) < p / 2, so its centred residue mod p IS that integer (exact, no rounding);
//   * the limb results are recombined with shifts and added to the accumulator mod 2^64.
//
// Cost against the 49-bit field kernel: the same 6 forward transforms, LIMBS x the multiply-accumulates, key bytes and
// inverse transforms.  Structure of k_blind_rotate_t64 = k_blind_rotate_tpx49 (a pair of wavefronts per ciphertext,
// wavefront c owns INPUT polynomial c, one exchange per inverse transform through the pair's LDS flags), with the three
// transformed digit polynomials held in registers across the limb loop and the accumulator kept as u64 in LDS.
//
// Decomposition (identical to the oracle's 64-bit rule, oracle/tfhe_oracle.c ora_decompose): centred lift = the word as
// int64, rounded half-up to its top 45 bits, signed digits in [-2^14, 2^14), the top digit absorbing the last carry.
#include <hip/hip_runtime.h>

#include "bmi_internal.hpp"
#include "ks_lincomb.hpp"
#include "ks_mfma.hpp"
#include "ntt_half_f64.hpp"
#include "ntt_wave_f64.hpp"
#include "pair_sync.hpp"
#include "t64_common.hpp"

using f49::i64;
using f49::u64;
using namespace nttf;

namespace {

#ifndef BMI_T64_DB
#define BMI_T64_DB 0      // key rows: 1 = one row requested ahead of the one being multiplied (12 spilled registers), 0 = request, then multiply (no spills); measured equal (156.5 vs 156.7 ms): the kernel is VALU-issue bound
#endif
#ifndef BMI_T64_RESYNC
#define BMI_T64_RESYNC BMI_TPX49_RESYNC   // workgroup barrier every so many CMUXes (keeps the four pairs on the same key rows)
#endif
#ifndef BMI_T64_ACCF
#define BMI_T64_ACCF 1    // wave-pair kernel, 48-bit key: accumulator as the exact integer word / 2^16 in a double (0: u64 words)
#endif
#ifndef BMI_T64_PRIO
#define BMI_T64_PRIO 1    // 1 = issue priority steps down through the forward transforms (3, 2, 1), 0 in the limb loop; 0 = none
#endif
using t64::f64_to_word;
using t64::Scheme;
using t64::word_to_f64;
constexpr int T64_CTS = 4;
constexpr int T64_AT_WORDS = BMI_AT_WORDS;
constexpr int T64_LDS_WORDS = TW_WORDS + 2 * T64_CTS * (SCRATCH_WORDS + N) + T64_CTS * T64_AT_WORDS + 4 * T64_CTS;
static_assert(T64_LDS_WORDS <= BMI_LDS_WORDS_MAX, "T64_LDS_WORDS exceeds the 160 KB of LDS");

__device__ __forceinline__ uint32_t modswitch_t64(u64 a) { return t64::modswitch<LOG_N + 1>(a); }

// standard-domain GGSW polynomials (u64 torus words) -> LIMBS transform-domain limb polynomials each, lane layout
__global__ void __launch_bounds__(256) k_bsk_to_limbs_t64(const u64 *__restrict__ std_polys, double *__restrict__ limb_polys,
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
    static_for<0, 16>([&](auto V) { limb_polys[(size_t)item * N + eval_offset(lane, V)] = f49::red(x[V]); });
}

template <int PREC, int L = 3, int BG = 15>
__global__ void __launch_bounds__(128 * T64_CTS)
    k_blind_rotate_t64(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                       const double *__restrict__ bsk, const double *__restrict__ g_tw, u64 *__restrict__ out,
                       uint32_t count, uint32_t n) {
    constexpr int CTS = T64_CTS;
    // exactness of a limb's sum: 2 L N terms of |digit| <= 2^(BG-1) times |limb| <= 2^(T64_LIMB_BITS-1) must stay below p/2
    constexpr int LIMBS = Scheme<PREC>::LIMBS, LB = Scheme<PREC>::BITS, PRE = Scheme<PREC>::PRE;
    static_assert(2.0 * L * N * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) < f49::P / 2, "limb sums must stay below p/2");
    static_assert(PRE + LIMBS * LB >= 64 && L * BG < 63, "limbs must cover the 64-bit word");
    extern __shared__ double lds[];
    double *tiles = lds + TW_WORDS;
    // A key stored at fewer than 53 bits of precision (PRE >= 12) makes every accumulator word a multiple of 2^PRE: the
    // accumulator is then kept as the exact integer word / 2^PRE, centred mod 2^AB (AB = 64 - PRE <= 52), in a DOUBLE - rotation,
    // difference and limb accumulation are a handful of f64 instructions instead of half-rate 64-bit integer ones, and the result
    // is the same word (the reduction mod 2^AB is the wrap of the u64 arithmetic).  The exact key (PRE = 0) keeps u64 words.
    constexpr bool ACCF = BMI_T64_ACCF && PREC == 48;   // (the 42-bit option would spill registers in this form: it keeps u64 words)
    constexpr int AB = ACCF ? 64 - PRE : 52;   // (unused without ACCF)
    u64 *accs = reinterpret_cast<u64 *>(tiles + 2 * CTS * SCRATCH_WORDS);
    double *at_base = reinterpret_cast<double *>(accs + 2 * CTS * N);
    uint32_t *flags = reinterpret_cast<uint32_t *>(at_base + CTS * T64_AT_WORDS);  // [2 CTS] published, [2 CTS] consumed
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ctl = wave >> 1, c = wave & 1;
    if (threadIdx.x < 4 * CTS) flags[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < TW_WORDS; i += blockDim.x) lds[i] = g_tw[i];
    const uint32_t ct_raw = blockIdx.x * CTS + ctl;
    const bool live = ct_raw < count;
    const uint32_t ct = live ? ct_raw : count - 1;
    double *tile = tiles + wave * SCRATCH_WORDS;
    const double *ptile = tiles + (wave ^ 1) * SCRATCH_WORDS;
    u64 *accl = accs + wave * N;
    double *accf = reinterpret_cast<double *>(accl);     // the same LDS words, read as doubles when ACCF
    auto mod_ab = [](double t) {   // centred residue mod 2^AB of an exact integer |t| < 2^53
        // (ties go to the negative end, like the two's complement reading of the u64 word: + 2^(AB-1) is - 2^(AB-1))
        return __builtin_fma(-(double)(1ull << AB), __builtin_floor(__builtin_fma(t, 1.0 / (double)(1ull << AB), 0.5)), t);
    };
    uint16_t *at = reinterpret_cast<uint16_t *>(at_base + ctl * T64_AT_WORDS);
    uint32_t *f_pub = flags + wave, *f_pub_partner = flags + (wave ^ 1);
    uint32_t *f_ack = flags + 2 * CTS + wave, *f_ack_partner = flags + 2 * CTS + (wave ^ 1);
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = lane + 64 * c; i <= n; i += 128) at[i] = (uint16_t)modswitch_t64(lwe[i]);
    __syncthreads();
    {
        const u64 *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at[n];
        static_for<0, 16>([&](auto J) {
            const uint32_t e = (lane + 64 * J + bt) & (2 * N - 1);
            const u64 v = tv[e & (N - 1)];
            const u64 w0 = c ? ((e & N) ? (u64)0 - v : v) : (u64)0;
            if constexpr (ACCF) accf[lane + 64 * J] = (double)((i64)w0 >> PRE);     // test polynomials are multiples of 2^(59 or so)
            else accl[lane + 64 * J] = w0;
        });
    }

    uint32_t hand = 0;   // handshake counter of the pair (one per inverse transform)
    for (uint32_t i = 0; i < n; i++) {
#if BMI_T64_RESYNC
        if (i % BMI_T64_RESYNC == 0) __syncthreads();  // keeps the four pairs on the same key rows (shared through L1)
#endif
        const uint32_t a_t = at[i];
        // this wavefront's three GGSW rows: [row = 3 c + lev][column][limb][N]
        const double *bsk_c = bsk + ((size_t)i * 4 * L + c * 2 * L) * LIMBS * N;
#if BMI_T64_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        wave_sync();
        double r[16];
        if constexpr (ACCF) {
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
        } else {
            u64 vr[16], vs[16];  // all 32 reads in flight before the first use
            static_for<0, 16>([&](auto J) {
                vr[J] = accl[(lane + 64 * J + 2 * N - a_t) & (N - 1)];
                vs[J] = accl[lane + 64 * J];
            });
            sched_fence();
            static_for<0, 16>([&](auto J) {
                const uint32_t e = (lane + 64 * J + 2 * N - a_t) & (2 * N - 1);
                const u64 v = (e & N) ? (u64)0 - vr[J] : vr[J];
                r[J] = t64::rounded_top<L, BG>(v - vs[J]);                               // round half up to L BG bits
            });
        }
        double X[L][16];   // the L digit polynomials, transform domain, live across the limb loop
        static_for<0, L>([&](auto LEV) {
            constexpr int lev = L - 1 - LEV;  // least significant digit first
            pin();
#if BMI_T64_PRIO
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
            forward(X[lev], lane, lds, tile);
        });
#if BMI_T64_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        static_for<0, LIMBS>([&](auto JL) {
            constexpr int j = JL;
            // 2 L key rows of this limb, partner's column first (its partial sum is published while the own column is
            // still being multiplied): rows 0..L-1 = (lev, column c^1), rows L..2L-1 = (lev, column c)
            auto row_ptr = [&](int q) {
                const int lev = q % L, col = q < L ? (c ^ 1) : c;
                return reinterpret_cast<const double2 *>(bsk_c + ((size_t)(lev * 2 + col) * LIMBS + j) * N);
            };
            double2 kb[BMI_T64_DB ? 2 : 1][8];
            if constexpr (BMI_T64_DB) static_for<0, 8>([&](auto VP) { kb[0][VP] = row_ptr(0)[VP * 64 + lane]; });
            double acc[16];
            hand++;
            static_for<0, 2 * L>([&](auto Q) {
                constexpr int q = Q, lev = q % L, cur = BMI_T64_DB ? (q & 1) : 0;
                if constexpr (BMI_T64_DB) {
                    if constexpr (q < 2 * L - 1) static_for<0, 8>([&](auto VP) { kb[cur ^ 1][VP] = row_ptr(q + 1)[VP * 64 + lane]; });
                } else {
                    static_for<0, 8>([&](auto VP) { kb[0][VP] = row_ptr(q)[VP * 64 + lane]; });
                }
                sched_fence();
                static_for<0, 8>([&](auto VP) {
                    const double m0 = f49::mul(X[lev][2 * VP], kb[cur][VP].x), m1 = f49::mul(X[lev][2 * VP + 1], kb[cur][VP].y);
                    if constexpr (lev == 0) {
                        acc[2 * VP] = m0;
                        acc[2 * VP + 1] = m1;
                    } else {
                        acc[2 * VP] += m0;
                        acc[2 * VP + 1] += m1;
                    }
                });
                if constexpr (q == L - 1) {
                    // the partner's partial goes through this wavefront's tile (free since the last transform)
                    wave_sync();
                    static_for<0, 8>([&](auto VP) {
                        reinterpret_cast<double2 *>(tile)[VP * 64 + lane] = double2{acc[2 * VP], acc[2 * VP + 1]};
                    });
                    pair_post(f_pub, hand);
                }
                pin();
            });
            pair_wait(f_pub_partner, hand);
            static_for<0, 8>([&](auto VP) {
                const double2 p = reinterpret_cast<const double2 *>(ptile)[VP * 64 + lane];
                acc[2 * VP] = f49::red(acc[2 * VP] + p.x);          // <= 2 * 3 * 1.4p before the reduction
                acc[2 * VP + 1] = f49::red(acc[2 * VP + 1] + p.y);
            });
            pair_post(f_ack, hand);          // release: the reads above have landed
            pair_wait(f_ack_partner, hand);  // the partner has read this tile: the inverse transform may overwrite it
            inverse(acc, lane, lds, tile);
            // the limb's exact integer result (|.| < 2^47.6 < p/2: the centred residue is the integer), shifted into place
            static_for<0, 16>([&](auto J) {
                if constexpr (ACCF) {
                    // x 2^(LB j) mod 2^AB: only the low AB - LB j bits of the limb's integer survive the shift
                    double x = f49::red(acc[J]);
                    if constexpr (j > 0) {
                        constexpr double W = (double)(1ull << (AB - LB * j));
                        x = __builtin_fma(-W, __builtin_rint(x * (1.0 / W)), x);
                        accf[lane + 64 * J] = mod_ab(__builtin_fma(x, (double)(1ull << (LB * j)), accf[lane + 64 * J]));
                    } else {
                        accf[lane + 64 * J] = mod_ab(accf[lane + 64 * J] + x);
                    }
                } else {
                    accl[lane + 64 * J] += f64_to_word(f49::red(acc[J])) << (PRE + LB * j);
                }
            });
            pin();
        });
    }

    if (!live) return;
    wave_sync();
    u64 *o = out + (size_t)ct * (N + 1);
    if (c == 0) {
        static_for<0, 16>([&](auto J) {
            const uint32_t m = lane + 64 * J;
            const u64 v = ACCF ? f64_to_word(accf[m]) << PRE : accl[m];
            if (m == 0) o[0] = v;
            else o[N - m] = (u64)0 - v;
        });
    } else if (lane == 0) {
        o[N] = ACCF ? f64_to_word(accf[0]) << PRE : accl[0];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LATENCY form for the torus: one workgroup of 16 wavefronts per ciphertext, every transform split over two wavefronts
// by parity (ntt_half_f64.hpp) - the structure of k_blind_rotate_lat2_49 with the limb dimension added:
//   A  wavefronts 0 .. 4L-1 = (input polynomial c, level, parity): rotate / decompose 512 coefficients of the u64
//      accumulator (integer rule of the oracle), forward half transform -> tile
//   B  all 1,024 threads = (output polynomial o, slot p): A_lo = E + O', A_hi = E - O' of the 2L digit transforms once,
//      then per limb the multiply-accumulate against the key (own copy in slot order, k_bsk_to_lat_t64) and the sums /
//      differences for the inverse halves; the next limb's key words are requested while this one is multiplied
//   C  wavefronts 0 .. 11 = (limb, o, parity): inverse half transform, conversion of the exact integers to words,
//      shift into place and ONE LDS atomic add per coefficient into the accumulator (the limbs of a coefficient meet in a slot)
constexpr int LT_THREADS = 1024;
constexpr int LT_LDS_WORDS = ntth::HT_WORDS + 2 * N + 12 * ntth::HSCRATCH + 3 * 2 * N + BMI_AT_WORDS;   // sized for 3 limbs
static_assert(LT_LDS_WORDS <= BMI_LDS_WORDS_MAX, "LT_LDS_WORDS exceeds the 160 KB of LDS");

// standard-domain GGSW polynomials -> per limb, the slot-order pair (A_lo, A_hi) of the two-wave half transform:
// out[((poly * limbs) + j) * N + p] = E + O', out[... + 512 + p] = E - O'
__global__ void __launch_bounds__(256) k_bsk_to_lat_t64(const u64 *__restrict__ std_polys, double *__restrict__ lat_polys,
                                                        const double *__restrict__ g_tw_h, uint32_t n_polys, int prec) {
    const int limbs = t64::limbs_of(prec);
    __shared__ double lds[ntth::HT_WORDS + 4 * ntth::HSCRATCH];
    for (int i = threadIdx.x; i < ntth::HT_WORDS; i += blockDim.x) lds[i] = g_tw_h[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = wave & 1;
    const uint32_t item = blockIdx.x * 2 + (wave >> 1);   // (polynomial, limb): two per workgroup, two wavefronts each
    const bool ok = item < n_polys * (uint32_t)limbs;
    const uint32_t poly = ok ? item / limbs : 0;
    const int j = ok ? (int)(item % limbs) : 0;
    double *tile = lds + ntth::HT_WORDS + wave * ntth::HSCRATCH;
    if (ok) {
        double x[8];
        static_for<0, 8>([&](auto J) {
            x[J] = (double)t64::limb_of((i64)std_polys[(size_t)poly * N + 2 * (lane + 64 * J) + h], j, prec);
        });
        if (h) ntth::forward_half<true>(x, lane, lds, tile);
        else ntth::forward_half<false>(x, lane, lds, tile);
        wave_sync();
        static_for<0, 8>([&](auto R) { tile[R * 64 + lane] = x[R]; });
    }
    __syncthreads();
    if (ok) {
        const double *te = lds + ntth::HT_WORDS + (wave & ~1) * ntth::HSCRATCH, *to = te + ntth::HSCRATCH;
        double *o = lat_polys + (size_t)item * N;
        static_for<0, 4>([&](auto Q4) {
            const int p = (h * 4 + Q4) * 64 + lane;
            const double e = te[p], od = to[p];
            // A_lo[p], A_hi[p] side by side: the kernel requests them as ONE 16-byte word (a compute unit takes in 47-50 B
            // per cycle with 16-byte requests against 29 with 8-byte ones, tools/microbench/cu_intake.hip)
            reinterpret_cast<double2 *>(o)[p] = make_double2(f49::red(e + od), f49::red(e - od));
        });
    }
}

template <int L, int PREC = 64, int BG = 15>
__global__ void __launch_bounds__(LT_THREADS)
    k_blind_rotate_lat_t64(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                           const double *__restrict__ bsk_lat, const double *__restrict__ g_tw_h, u64 *__restrict__ out,
                           uint32_t count, uint32_t n) {
    constexpr int LIMBS = Scheme<PREC>::LIMBS, LB = Scheme<PREC>::BITS, PRE = Scheme<PREC>::PRE;
    static_assert(2.0 * L * N * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) < f49::P / 2, "limb sums must stay below p/2");
    extern __shared__ double lds[];
    u64 *acc = reinterpret_cast<u64 *>(lds + ntth::HT_WORDS);   // [2 components][2 parities][512] words mod 2^64
    double *tiles = lds + ntth::HT_WORDS + 2 * N;               // [12][HSCRATCH]
    double *SD = tiles + 12 * ntth::HSCRATCH;                   // [limb][2 outputs][sum, difference][512]
    uint16_t *at = reinterpret_cast<uint16_t *>(SD + LIMBS * 2 * N);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < ntth::HT_WORDS; i += LT_THREADS) lds[i] = g_tw_h[i];
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += LT_THREADS) at[i] = (uint16_t)modswitch_t64(lwe[i]);
    __syncthreads();
    {
        const u64 *tv = luts + (size_t)(lut_ids[ct] & (BMI_LUT_CAP - 1)) * N;
        const uint32_t bt = at[n];
        const uint32_t nn = tid;  // coefficient index
        const uint32_t e = (nn + bt) & (2 * N - 1);
        const u64 v = tv[e & (N - 1)];
        acc[(nn & 1) * ntth::HALF + (nn >> 1)] = 0;
        acc[N + (nn & 1) * ntth::HALF + (nn >> 1)] = (e & N) ? (u64)0 - v : v;
    }
    __syncthreads();
    const int mo = tid >> 9, mp = tid & 511;  // phase B: output polynomial, slot

    for (uint32_t i = 0; i < n; i++) {
        const uint32_t a_t = at[i];
        if (a_t == 0) continue;  // uniform over the workgroup
        // key words of this thread's (output, slot): [row 2L][output 2][limb][512 slots][A_lo, A_hi], one 16-byte request per
        // (row, limb)
        const double *bi = bsk_lat + (size_t)i * 4 * L * LIMBS * N;
        auto load_limb = [&](double (&dst)[2 * L][2], int j) {
#pragma unroll
            for (int r = 0; r < 2 * L; r++) {
                const double2 w = reinterpret_cast<const double2 *>(bi + (((size_t)(r * 2 + mo)) * LIMBS + j) * N)[mp];
                dst[r][0] = w.x;
                dst[r][1] = w.y;
            }
        };
        double b0[2 * L][2];
        load_limb(b0, 0);
        if (wave < 4 * L) {
            const int c = wave / (2 * L), lev = (wave % (2 * L)) >> 1, h = wave & 1;
            const int pz = wave >> 1;
            const u64 *ac = acc + c * N;
            double x[8];
            __builtin_amdgcn_s_setprio(3);
            static_for<0, 8>([&](auto J) {
                const uint32_t m = lane + 64 * J;
                const uint32_t e = (2 * m + h + 2 * N - a_t) & (2 * N - 1);
                const uint32_t n2 = e & (N - 1);
                u64 v = ac[(n2 & 1) * ntth::HALF + (n2 >> 1)];
                v = (e & N) ? (u64)0 - v : v;
                double r = t64::rounded_top<L, BG>(v - ac[h * ntth::HALF + m]);        // round half up to L BG bits
                double d = r;                                                          // digit `lev`, balanced [-2^14, 2^14)
#pragma unroll
                for (int s = L - 1; s > 0; s--) {
                    const double rn = __builtin_floor(__builtin_fma(r, 1.0 / (double)(1ull << BG), 0.5));
                    if (s == lev) d = __builtin_fma(-(double)(1ull << BG), rn, r);
                    r = rn;
                }
                x[J] = lev == 0 ? r : d;
            });
            double *tile = tiles + (2 * pz + h) * ntth::HSCRATCH;
            if (h) ntth::forward_half<true>(x, lane, lds, tile);
            else ntth::forward_half<false>(x, lane, lds, tile);
            wave_sync();
            static_for<0, 8>([&](auto R) { tile[R * 64 + lane] = x[R]; });
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        {
            double alo[2 * L], ahi[2 * L];
#pragma unroll
            for (int r = 0; r < 2 * L; r++) {
                const double e = tiles[(2 * r) * ntth::HSCRATCH + mp], od = tiles[(2 * r + 1) * ntth::HSCRATCH + mp];
                alo[r] = e + od;
                ahi[r] = e - od;
            }
            // limb 0 was requested before the forward phase; limb 1 is requested now, limb 2 as soon as limb 0 has been consumed
            // (its registers are reused): a request is in flight during every multiplication
            auto one_limb = [&](const double (&kw)[2 * L][2], int j) {
                double ylo = 0.0, yhi = 0.0;  // lazy sums of 2 L products (<= 10.3 p)
#pragma unroll
                for (int r = 0; r < 2 * L; r++) {
                    ylo += f49::mul(alo[r], kw[r][0]);
                    yhi += f49::mul(ahi[r], kw[r][1]);
                }
                ylo = f49::red(ylo);
                yhi = f49::red(yhi);
                double *sd = SD + (size_t)j * 2 * N;
                sd[(mo * 2 + 0) * ntth::HALF + mp] = ylo + yhi;
                sd[(mo * 2 + 1) * ntth::HALF + mp] = ylo - yhi;
            };
            double bn[2 * L][2];
            load_limb(bn, 1);
            sched_fence();
            one_limb(b0, 0);
            if constexpr (LIMBS == 3) {
                load_limb(b0, 2);
                sched_fence();
            }
            one_limb(bn, 1);
            if constexpr (LIMBS == 3) one_limb(b0, 2);
        }
        __syncthreads();
        if (wave < 4 * LIMBS) {
            const int j = wave >> 2, o = (wave >> 1) & 1, h = wave & 1;
            double x[8];
            const double *sd = SD + (size_t)j * 2 * N + (o * 2 + h) * ntth::HALF;
            static_for<0, 8>([&](auto R) { x[R] = sd[R * 64 + lane]; });
            double *tile = tiles + wave * ntth::HSCRATCH;
            __builtin_amdgcn_s_setprio(2);
            if (h) ntth::inverse_half<true>(x, lane, lds, tile);
            else ntth::inverse_half<false>(x, lane, lds, tile);
            __builtin_amdgcn_s_setprio(0);
            unsigned long long *ao = reinterpret_cast<unsigned long long *>(acc + o * N + h * ntth::HALF);
            const int sh = PRE + LB * j;
            static_for<0, 8>([&](auto J) {
                // the limb's exact integer (|.| < 2^47.6 < p/2), shifted into place; the three limbs of a slot add atomically
                atomicAdd(ao + lane + 64 * J, (unsigned long long)(f64_to_word(f49::red(x[J])) << sh));
            });
        }
        __syncthreads();
    }
    u64 *o = out + (size_t)ct * (N + 1);
    {
        const uint32_t nn = tid;
        const u64 a0 = acc[(nn & 1) * ntth::HALF + (nn >> 1)];
        if (nn == 0) {
            o[0] = a0;
            o[N] = acc[N];
        } else {
            o[N - nn] = (u64)0 - a0;
        }
    }
}

// keyswitch / linear combinations: wrap-around arithmetic, the oracle's 64-bit decomposition rule
struct FieldT {
    static __device__ __forceinline__ void digits(u64 a, uint32_t levels, uint32_t base_log, unsigned char *d) {
        const uint32_t shift = 64 - levels * base_log;
        const i64 B = (i64)1 << base_log, half = B >> 1;
        const i64 c = (i64)a;
        i64 r = (c >> shift) + ((c >> (shift - 1)) & 1);
        for (int lev = (int)levels - 1; lev >= 1; lev--) {
            i64 v = r & (B - 1);
            r >>= base_log;
            if (v >= half) { v -= B; r += 1; }
            d[lev] = (unsigned char)(v + half);
        }
        d[0] = (unsigned char)(r + half);
    }
    static __device__ __forceinline__ i64 centered(u64 a) { return (i64)a; }
    static __device__ __forceinline__ u64 add(u64 a, u64 b) { return a + b; }
    static __device__ __forceinline__ u64 sub(u64 a, u64 b) { return a - b; }
    static __device__ __forceinline__ u64 neg(u64 a) { return (u64)0 - a; }
    static __device__ __forceinline__ u64 mul_small(i64 cf, u64 v) { return (u64)cf * v; }
    static __device__ __forceinline__ u64 reduce96(uint32_t, u64 lo) { return lo; }
    static __device__ __forceinline__ u64 reduce128(u64, u64 lo) { return lo; }
};

}  // namespace

namespace bmit {

#define BMIT_LAUNCH_CHECK()                     \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int launch_bsk_to_limbs(const u64 *std_polys, double *limb_polys, const double *g_tw, uint32_t n_polys, int prec,
                        hipStream_t s) {
    if (!t64::precision_ok(prec)) return (int)hipErrorInvalidValue;
    const uint32_t items = n_polys * (uint32_t)t64::limbs_of(prec);
    hipLaunchKernelGGL(k_bsk_to_limbs_t64, dim3((items + 3) / 4), dim3(256), 0, s, std_polys, limb_polys, g_tw, n_polys, prec);
    BMIT_LAUNCH_CHECK();
    return 0;
}

template <int PREC, int L, int BG>
static int launch_t64(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_limbs,
                      const double *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)T64_LDS_WORDS * sizeof(double);
    auto kern = k_blind_rotate_t64<PREC, L, BG>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3((count + T64_CTS - 1) / T64_CTS), dim3(128 * T64_CTS), lds, s, small_cts, lut_ids, luts,
                       bsk_limbs, g_tw, out, count, n);
    BMIT_LAUNCH_CHECK();
    return 0;
}

// the instantiated (precision, levels, base log) combinations; bmi_host.cpp params_supported / bmi_set_bsk_precision admit
// exactly these
#define BMIT_FOR_EACH_SHAPE(X) X(64, 3, 15) X(64, 2, 15) X(42, 3, 15) X(42, 2, 15) X(48, 3, 10) X(48, 2, 10) X(64, 3, 10)

bool shape_supported(int prec, uint32_t levels, uint32_t base_log) {
#define BMIT_SHAPE_OK(P, L, B) if (prec == P && levels == L && base_log == B) return true;
    BMIT_FOR_EACH_SHAPE(BMIT_SHAPE_OK)
#undef BMIT_SHAPE_OK
    return false;
}

int launch_blind_rotate(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_limbs,
                        const double *g_tw, u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels,
                        uint32_t base_log, hipStream_t s) {
    if (count == 0) return 0;
#define BMIT_GO(P, L, B) \
    if (prec == P && levels == L && base_log == B) return launch_t64<P, L, B>(small_cts, lut_ids, luts, bsk_limbs, g_tw, out, count, n, s);
    BMIT_FOR_EACH_SHAPE(BMIT_GO)
#undef BMIT_GO
    return (int)hipErrorInvalidValue;
}

int launch_bsk_to_lat(const u64 *std_polys, double *lat_polys, const double *g_tw_h, uint32_t n_polys, int prec, hipStream_t s) {
    if (!t64::precision_ok(prec)) return (int)hipErrorInvalidValue;
    const uint32_t items = n_polys * (uint32_t)t64::limbs_of(prec);
    hipLaunchKernelGGL(k_bsk_to_lat_t64, dim3((items + 1) / 2), dim3(256), 0, s, std_polys, lat_polys, g_tw_h, n_polys, prec);
    BMIT_LAUNCH_CHECK();
    return 0;
}

template <int PREC, int L, int BG>
static int launch_lat_t64(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_lat,
                          const double *g_tw_h, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)LT_LDS_WORDS * sizeof(double);
    auto kern = k_blind_rotate_lat_t64<L, PREC, BG>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3(count), dim3(LT_THREADS), lds, s, small_cts, lut_ids, luts, bsk_lat, g_tw_h, out, count, n);
    BMIT_LAUNCH_CHECK();
    return 0;
}

int launch_blind_rotate_lat(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_lat,
                            const double *g_tw_h, u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels,
                            uint32_t base_log, hipStream_t s) {
    if (count == 0) return 0;
#define BMIT_GO(P, L, B) \
    if (prec == P && levels == L && base_log == B) return launch_lat_t64<P, L, B>(small_cts, lut_ids, luts, bsk_lat, g_tw_h, out, count, n, s);
    BMIT_FOR_EACH_SHAPE(BMIT_GO)
#undef BMIT_GO
    return (int)hipErrorInvalidValue;
}

int launch_keyswitch(const u64 *in, const u64 *ksk, const u64 *ks_bias, u64 *out, void *partial, uint32_t slices,
                     uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels, uint32_t base_log, uint32_t ks_stride,
                     hipStream_t s) {
    return ksl::launch_keyswitch<FieldT>(in, ksk, ks_bias, out, partial, slices, count, n, big_n, levels, base_log,
                                         ks_stride, s);
}

int launch_ksk_to_limbs(const u64 *ksk, signed char *limbs, uint32_t rows, uint32_t n, uint32_t ks_stride, hipStream_t s) {
    return ksm::launch_ksk_to_limbs<FieldT>(ksk, limbs, rows, n, ks_stride, KS_LIMBS, s);
}

int launch_keyswitch_mfma(const u64 *in, const signed char *limbs, signed char *digits, int *sums, u64 *out,
                          uint32_t slices, uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels,
                          uint32_t base_log, hipStream_t s) {
    return ksm::launch_keyswitch<FieldT, KS_LIMBS>(in, limbs, digits, sums, out, slices, count, n, big_n, levels, base_log, s);
}

int launch_lincomb(const u64 *store, const uint32_t *row_ptr, const uint32_t *idx, const i64 *coef,
                   const u64 *const_body, u64 *out, uint32_t count, uint32_t width, hipStream_t s) {
    return ksl::launch_lincomb<FieldT>(store, row_ptr, idx, coef, const_body, out, count, width, s);
}

}  // namespace bmit
