// HIP kernel (gfx950) of the UNROLLED blind rotation on the 2^64 TORUS (q = 2^64 exactly, Concrete's ciphertext modulus;
// bmi_set_bsk_unroll(ctx, 2) on a torus context).  One step absorbs two LWE coefficients (oracle/tfhe_oracle.c
// ora_blind_rotate_extract_unrolled):
//
//     ACC <- ACC + sum_{j<3} (X^(c_j) - 1) (K3[i][j] [.] ACC),   c = (a + a', a, a'),
//     K3[i] = GGSW(s s'), GGSW(s (1 - s')), GGSW((1 - s) s')   of the key bits (s, s') = (s_2i, s_2i+1).
//
// Structure = k_blind_rotate_lat2u_49 (bmi_kernels_f64u.hip: one workgroup of 16 wavefronts per ciphertext, every transform
// split over two wavefronts by parity, the factors X^c - 1 applied in the TRANSFORM domain as psi^(e c) - 1) with the limb
// dimension of the torus kernels (bmi_kernels_t64.hip, t64_common.hpp): the products are exact integers, formed per limb of
// the key mod p = 2^49 - 720895 and recombined mod 2^64.
//   A  wavefronts 0 .. 4L-1 = (component c, level, parity): decompose 512 coefficients of the u64 accumulator itself (the
//      oracle's integer rule; no rotation), forward half transform -> tile            [once per PAIR of LWE coefficients]
//   B  all 1,024 threads = (output polynomial o, slot p): A_lo = E + O', A_hi = E - O' of the 2L digit transforms once, then
//      for each of the three keys and each limb the multiply-accumulate over the 2L rows, scaled by psi^(e c_j) - 1 and summed
//      over the keys into ONE pair of sums per limb; the (key, limb) blocks of key words are requested one block ahead
//   C  wavefronts 0 .. 4 LIMBS - 1 = (limb, o, parity): inverse half transform, conversion of the exact integer to a word,
//      shift into place and one LDS atomic add per coefficient (the limbs of a coefficient meet in a slot)
// Exactness: the inverse transform of a limb returns  sum_j (X^(c_j) - 1) sum_rows digit x limb,  an integer of magnitude
// < 3 * 2 * 2 L N 2^(BG-1) 2^(BITS-1) = 2^47.2 at (L, BG, BITS) = (3, 10, 24) < p/2: its centred residue IS the integer.
#include <hip/hip_runtime.h>

#include <atomic>

#include "bmi_internal.hpp"
#include "ntt_half_f64.hpp"
#include "ntt_wave_f64.hpp"
#include "t64_common.hpp"

using f49::i64;
using f49::u64;
using namespace nttf;
using t64::f64_to_word;
using t64::Scheme;

namespace {

#ifdef BMI_PHASE_PROF
__device__ unsigned long long g_phase_tu[128];
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

constexpr int LU_THREADS = 1024;
// twiddles, accumulator (u64), twelve tiles, sums / differences per limb, mod-switched ciphertext, root powers
template <int LIMBS>
constexpr int lu_lds_words() { return ntth::HT_WORDS + 2 * N + 12 * ntth::HSCRATCH + LIMBS * 2 * N + BMI_AT_WORDS + 2 * N; }

template <int L, int BG, int PREC>
__global__ void __launch_bounds__(LU_THREADS)
    k_blind_rotate_lat2u_t64(const u64 *__restrict__ small_cts, const uint32_t *__restrict__ lut_ids, const u64 *__restrict__ luts,
                             const double *__restrict__ bsk3_lat, const double *__restrict__ g_tw_h,
                             const double *__restrict__ g_root_pow, u64 *__restrict__ out, uint32_t count, uint32_t n) {
    constexpr int LIMBS = Scheme<PREC>::LIMBS, LB = Scheme<PREC>::BITS, PRE = Scheme<PREC>::PRE;
    static_assert(6.0 * 2.0 * L * N * (double)(1ull << (BG - 1)) * (double)(1ull << (LB - 1)) < f49::P / 2,
                  "the unrolled step's limb sums (three keys, each scaled by X^c - 1) must stay below p/2");
    static_assert(PRE + LIMBS * LB >= 64 && L * BG < 63 && 4 * LIMBS <= 12, "limbs must cover the 64-bit word");
    static_assert(lu_lds_words<LIMBS>() <= BMI_LDS_WORDS_MAX, "exceeds the 160 KB of LDS");
    extern __shared__ double lds[];
    u64 *acc = reinterpret_cast<u64 *>(lds + ntth::HT_WORDS);   // [2 components][2 parities][512] words mod 2^64
    double *tiles = lds + ntth::HT_WORDS + 2 * N;               // [12][HSCRATCH]
    double *SD = tiles + 12 * ntth::HSCRATCH;                   // [limb][2 outputs][sum, difference][512]
    uint16_t *at = reinterpret_cast<uint16_t *>(SD + LIMBS * 2 * N);
    double *RP = SD + LIMBS * 2 * N + BMI_AT_WORDS;             // psi^x, x in [0, 2N), folded (see below)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < ntth::HT_WORDS; i += LU_THREADS) lds[i] = g_tw_h[i];
    for (int i = tid; i < 2 * N; i += LU_THREADS) RP[i ^ ((i >> 5) & 31)] = g_root_pow[i];
    const uint32_t ct = blockIdx.x;
    const u64 *lwe = small_cts + (size_t)ct * (n + 1);
    for (uint32_t i = tid; i <= n; i += LU_THREADS) at[i] = (uint16_t)t64::modswitch<LOG_N + 1>(lwe[i]);
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
    const uint32_t root_e = 2 * ntth::kk_of(mp & 63, mp >> 6) + 1;   // A_lo[mp] is the value at psi^root_e, A_hi[mp] at -psi^root_e
    const uint32_t pairs = (n + 1) >> 1;
    // this thread's 2 L x 2 words of limb `j` of GGSW key `key` (0..2) of pair `ip`:
    // [2 L rows][2 outputs][limb][512 slots][A_lo, A_hi], one 16-byte request per row
    auto load_block = [&](double (&dst)[2 * L][2], uint32_t ip, int key, int j) {
        const double *bj = bsk3_lat + ((size_t)ip * 3 + key) * 4 * L * LIMBS * N;
#pragma unroll
        for (int r = 0; r < 2 * L; r++) {
            const double2 w = reinterpret_cast<const double2 *>(bj + ((size_t)(r * 2 + mo) * LIMBS + j) * N)[mp];
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
        load_block(b, ip, 0, 0);
        if (wave < 4 * L) {
            const int c = wave / (2 * L), lev = (wave % (2 * L)) >> 1, h = wave & 1;
            const int pz = wave >> 1;
            const u64 *ac = acc + c * N + h * ntth::HALF;
            double x[8];
#if BMI_LAT2_PRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            static_for<0, 8>([&](auto J) {
                double r = t64::rounded_top<L, BG>(ac[lane + 64 * J]);   // round half up to L BG bits
                double d = r;                                           // digit `lev`, balanced [-2^(BG-1), 2^(BG-1))
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
            double slo[LIMBS], shi[LIMBS];   // per limb: sums over the three keys of reduced products (<= 1.6 q)
#pragma unroll
            for (int j = 0; j < LIMBS; j++) slo[j] = shi[j] = 0.0;
            double wlo = 0.0, whi = 0.0;     // psi^(e c_j) - 1 at the slot's two roots (psi^e and -psi^e), for the key being multiplied
            auto root_factor = [&](uint32_t c) {
                // the table is stored at x ^ (bits 5..9 of x): the exponents of a wavefront, odd multiples of c, share their low
                // bits when c is even - without the fold they would meet in a few LDS banks (bmi_kernels_f64u.hip)
                const uint32_t xe = (root_e * c) & (2 * N - 1);
                const double w = RP[xe ^ ((xe >> 5) & 31)];
                wlo = w - 1.0;
                whi = ((c & 1) ? -w : w) - 1.0;
            };
            auto one_block = [&](const double (&kw)[2 * L][2], int key, int j) {
                double ylo = 0.0, yhi = 0.0;  // lazy sums of 2 L <= six reduced products (<= 3.1 q)
#pragma unroll
                for (int r = 0; r < 2 * L; r++) {
                    ylo += f49::mul(alo[r], kw[r][0]);
                    yhi += f49::mul(ahi[r], kw[r][1]);
                }
                slo[j] += f49::mul(ylo, wlo);   // the lazy sum goes into the product as it is (exact: |.| < 2^53)
                shi[j] += f49::mul(yhi, whi);
            };
            // blocks in the order (key 0, limb 0), (key 0, limb 1), (key 1, limb 0) ...; block t + 1 is requested before block t
            // is multiplied (two register sets of 4 L doubles); block 0 was requested before the forward phase
            double bn[2 * L][2];
            static_for<0, 3 * LIMBS>([&](auto T) {
                constexpr int t = T, key = t / LIMBS, j = t % LIMBS;
                if constexpr (t + 1 < 3 * LIMBS) {
                    constexpr int key1 = (t + 1) / LIMBS, j1 = (t + 1) % LIMBS;
                    if constexpr (t & 1) load_block(b, ip, key1, j1);
                    else load_block(bn, ip, key1, j1);
                }
                if constexpr (j == 0) root_factor(cj[key]);
                if constexpr (t & 1) one_block(bn, key, j);
                else one_block(b, key, j);
            });
#pragma unroll
            for (int j = 0; j < LIMBS; j++) {
                const double lo = f49::red(slo[j]), hi = f49::red(shi[j]);
                double *sd = SD + (size_t)j * 2 * N;
                sd[(mo * 2 + 0) * ntth::HALF + mp] = lo + hi;
                sd[(mo * 2 + 1) * ntth::HALF + mp] = lo - hi;
            }
        }
        PH_MARK(2);
        __syncthreads();
        PH_MARK(3);
        if (wave < 4 * LIMBS) {
            const int j = wave >> 2, o = (wave >> 1) & 1, h = wave & 1;
            double x[8];
            const double *sd = SD + (size_t)j * 2 * N + (o * 2 + h) * ntth::HALF;
            static_for<0, 8>([&](auto R) { x[R] = sd[R * 64 + lane]; });
            double *tile = tiles + wave * ntth::HSCRATCH;
            if (h) ntth::inverse_half<true>(x, lane, lds, tile);
            else ntth::inverse_half<false>(x, lane, lds, tile);
            unsigned long long *ao = reinterpret_cast<unsigned long long *>(acc + o * N + h * ntth::HALF);
            const int sh = PRE + LB * j;
            static_for<0, 8>([&](auto J) {
                // the limb's exact integer (|.| < p/2), shifted into place; the limbs of a slot add atomically
                atomicAdd(ao + lane + 64 * J, (unsigned long long)(f64_to_word(f49::red(x[J])) << sh));
            });
        }
        PH_MARK(4);
        __syncthreads();
        PH_MARK(5);
    }
#ifdef BMI_PHASE_PROF
    if (blockIdx.x == 0 && lane == 0)
        for (int k_ = 0; k_ < 8; k_++) g_phase_tu[wave * 8 + k_] = ph_[k_];
#endif
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

}  // namespace

#ifdef BMI_PHASE_PROF
extern "C" int bmi_debug_phase_prof_unrolled_t64(unsigned long long *out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phase_tu), sizeof(unsigned long long) * 128);
}
#endif

namespace bmit {

#define BMIT_FOR_EACH_SHAPE_U(X) X(48, 3, 10) X(48, 2, 10)

bool shape_supported_unrolled(int prec, uint32_t levels, uint32_t base_log) {
#define BMIT_SHAPE_OK(P, L, B) if (prec == P && levels == L && base_log == B) return true;
    BMIT_FOR_EACH_SHAPE_U(BMIT_SHAPE_OK)
#undef BMIT_SHAPE_OK
    return false;
}

template <int PREC, int L, int BG>
static int launch_lat2u_t64(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk3_lat,
                            const double *g_tw_h, const double *g_root_pow, u64 *out, uint32_t count, uint32_t n, hipStream_t s) {
    static std::atomic<uint64_t> configured{0};
    const size_t lds = (size_t)lu_lds_words<Scheme<PREC>::LIMBS>() * sizeof(double);
    auto kern = k_blind_rotate_lat2u_t64<L, BG, PREC>;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds, configured)) return rc;
    hipLaunchKernelGGL(kern, dim3(count), dim3(LU_THREADS), lds, s, small_cts, lut_ids, luts, bsk3_lat, g_tw_h, g_root_pow, out,
                       count, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_blind_rotate_lat2u(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk3_lat,
                              const double *g_tw_h, const double *g_root_pow, u64 *out, uint32_t count, uint32_t n, int prec,
                              uint32_t levels, uint32_t base_log, hipStream_t s) {
    if (count == 0) return 0;
#define BMIT_GO(P, L, B) \
    if (prec == P && levels == L && base_log == B)  \
        return launch_lat2u_t64<P, L, B>(small_cts, lut_ids, luts, bsk3_lat, g_tw_h, g_root_pow, out, count, n, s);
    BMIT_FOR_EACH_SHAPE_U(BMIT_GO)
#undef BMIT_GO
    return (int)hipErrorInvalidValue;
}

}  // namespace bmit
