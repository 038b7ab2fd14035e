// Internal declarations shared by the kernel TU (bmi_kernels.hip) and the host TU (bmi_host.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goldilocks.hpp"

#ifndef BMI_DEFAULT_Q_BITS
#define BMI_DEFAULT_Q_BITS 65  // modulus of bmi_default_params(): 65 = BMI_Q_TORUS64 (q = 2^64, Concrete's own: the default since round 4), 49 (f64 kernels mod 2^49 - 720895) or 64 (Goldilocks)
#endif

#ifndef BMI_TP49_CTS
#define BMI_TP49_CTS 2              // ciphertexts per workgroup, f64 throughput kernel (3 with a 168-VGPR budget spills: measured slower)
#define BMI_TP49_WAVES_PER_SIMD 2
#endif

#ifndef BMI_TPX49_PF
#define BMI_TPX49_PF 13  // exchange-once kernel: where the two GGSW rows of a level are requested (see the kernel)
#endif

#ifndef BMI_TPX49_RESYNC
#define BMI_TPX49_RESYNC 4  // workgroup barrier every so many CMUX iterations (0: never): keeps the four pairs on the same key rows, which they share through L1 (83.5 -> 80.6 ms)
#endif

#ifndef BMI_TPX49_SYNC
#define BMI_TPX49_SYNC 0  // pair synchronisation of the exchange-once kernel: 0 = LDS counters (pairs only), 1 = workgroup barrier
#endif

#ifndef BMI_TPX49_PRIO
#define BMI_TPX49_PRIO 3  // s_setprio inside a CMUX of the exchange-once kernel: 0 = none, 1 = raised until the partial sums are published, 2 = raised after, 3 = stepping down 3,3,2,1 over decomposition and the three levels, 0 from the exchange on (the wavefront that is behind on a SIMD gets the issue slots: 80.0 -> 78.1 ms), 4 = raised for the inverse only (no gain)
#endif
#ifndef BMI_LAT2_PRIO
#define BMI_LAT2_PRIO 2  // s_setprio in the forward tasks of the latency kernels (N = 1024 two-wave transforms, N = 2048, N = 4096): tasks sharing a SIMD step down 3,2,1,0 as they advance, so they finish together instead of the last one running its tail alone (4.02 -> 3.57 ms per bootstrap at N = 1024; 1: three steps, 3.60 ms; 0: none)
#endif
#ifndef BMI_WIDE_STAGE
#define BMI_WIDE_STAGE 2  // key-word requests of the N = 2048 kernel: 0 = all before the task, 1 = half before / half after the decomposition, 2 = a quarter each before the task, after the decomposition, mid-transform and before its last transpose (8.95 -> 8.55 ms per bootstrap in one session)
#endif
// Capacity of a context's look-up table buffer (tables of N words, allocated at creation).  A power of two: the kernels
// mask the ids they are handed with it, so that a wrong id in a device-resident id array reads a wrong (possibly
// unregistered) table instead of faulting; the host-buffer entry points refuse unknown ids outright.
#define BMI_LUT_CAP 1024
// Largest small-LWE dimension n the blind-rotation kernels take: every kernel stages the n + 1 mod-switched words of a
// ciphertext in LDS as uint16 (BMI_AT_WORDS 8-byte words per ciphertext).
#define BMI_MAX_LWE_N 1024
constexpr int BMI_AT_WORDS = 264;   // 1,056 uint16 slots >= BMI_MAX_LWE_N + 1
constexpr int BMI_LDS_WORDS_MAX = 160 * 1024 / 8;   // 160 KB of LDS per workgroup on gfx950
#ifndef BMI_KS_MFMA_MIN
#define BMI_KS_MFMA_MIN 1  // smallest batch that takes the matrix-core keyswitch (0.05 ms against 0.14 ms scalar even at one ciphertext)
#endif

#ifndef BMI_TP_CTS
#define BMI_TP_CTS 2  // ciphertexts (= wavefront pairs) per workgroup in the throughput blind rotation
#endif

// hipFuncSetAttribute acts on the current device only: remember per kernel which devices have been configured
// (several contexts on several GPUs may live in one process).
#include <atomic>
inline int set_max_dynamic_lds(const void *kernel, size_t bytes, std::atomic<uint64_t> &done_devices) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 64 && ((done_devices.load() >> dev) & 1)) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    if (dev < 64) done_devices.fetch_or((uint64_t)1 << dev);
    return 0;
}

namespace bmi {
using gl::i64;
using gl::u64;

int launch_bsk_to_ntt(const u64 *std_polys, u64 *ntt_polys, const u64 *g_tw, uint32_t n_polys, hipStream_t s);
int launch_negacyclic_mul(const u64 *a, const u64 *b, u64 *c, const u64 *g_tw, uint32_t count, hipStream_t s);
int launch_blind_rotate_tp(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const u64 *bsk,
                           const u64 *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s);
int launch_blind_rotate_lat(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const u64 *bsk,
                            const u64 *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s);
// slices > 1 (with a partial buffer of slices * count * ks_stride 16-byte words): latency form for small batches
// ks_bias[col] = (B/2) * sum over all rows of ksk[row][col] mod q (the digits are staged unsigned, d + B/2)
int launch_keyswitch(const u64 *in, const u64 *ksk, const u64 *ks_bias, u64 *out, void *partial, uint32_t slices,
                     uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels, uint32_t base_log, uint32_t ks_stride,
                     hipStream_t s);
// Keyswitch on the matrix cores (ks_mfma.hpp): the key as KS_LIMBS balanced base-256 limbs in MFMA operand order,
// digits = count * big_n * levels bytes, sums = slices * count * KS_LIMBS * ceil((n+1)/32)*32 int32 words.
constexpr uint32_t KS_LIMBS = 9;
int launch_ksk_to_limbs(const u64 *ksk, signed char *limbs, uint32_t rows, uint32_t n, uint32_t ks_stride, hipStream_t s);
int launch_keyswitch_mfma(const u64 *in, const signed char *limbs, signed char *digits, int *sums, u64 *out,
                          uint32_t slices, uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels,
                          uint32_t base_log, hipStream_t s);
int launch_lincomb(const u64 *store, const uint32_t *row_ptr, const uint32_t *idx, const i64 *coef,
                   const u64 *const_body, u64 *out, uint32_t count, uint32_t width, hipStream_t s);
// store[rows[i]] = src[i] (rows of `width` words); field independent
int launch_scatter_rows(const u64 *src, u64 *store, const uint32_t *rows, uint32_t count, uint32_t width, hipStream_t s);
}  // namespace bmi

// The same launchers for the 49-bit field (bmi_kernels_f64.hip): NTT-domain key, twiddles and test polynomials are f64.
namespace bmi49 {
using gl::i64;
using gl::u64;
int launch_bsk_to_ntt(const u64 *std_polys, double *ntt_polys, const double *g_tw, uint32_t n_polys, hipStream_t s);
int launch_negacyclic_mul(const u64 *a, const u64 *b, u64 *c, const double *g_tw, uint32_t count, hipStream_t s);
int launch_blind_rotate_tp(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk,
                           const double *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s);
// third parameter set N = 4096: key copy in slot order (scaled by 1/4); g_t = T_1, T_2, T_3, then their inverses ([6][1024])
int launch_bsk_to_quad(const u64 *std_polys, double *quad_polys, const double *g_tw, const double *g_t, uint32_t n_polys,
                       hipStream_t s);
int launch_blind_rotate_quad(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk_quad,
                             const double *g_tw, const double *g_t, u64 *out, uint32_t count, uint32_t n, hipStream_t s);
// second parameter set N = 2048: key copy in slot order (scaled by 1/2), table T / T^-1 of the even/odd combination
int launch_bsk_to_wide(const u64 *std_polys, double *wide_polys, const double *g_tw, const double *g_tw_wide,
                       uint32_t n_polys, hipStream_t s);
int launch_blind_rotate_wide(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk_wide,
                             const double *g_tw, const double *g_tw_wide, u64 *out, uint32_t count, uint32_t n,
                             uint32_t levels, uint32_t base_log, hipStream_t s);
// N = 2048 with the unrolled key: bsk3_wide = launch_bsk_to_wide of the unrolled key, g_root_pow = psi_4096^x for x in [0, 2048)
int launch_blind_rotate_wide_u(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk3_wide,
                               const double *g_tw, const double *g_tw_wide, const double *g_root_pow, u64 *out, uint32_t count,
                               uint32_t n, uint32_t levels, uint32_t base_log, hipStream_t s);
// latency kernel, two wavefronts per transform (ntt_half_f64.hpp): own key copy in slot order, own twiddle tables
// paired: A_lo[p], A_hi[p] side by side (one 16-byte request per slot; the unrolled kernel's layout), else [A_lo 512][A_hi 512]
int launch_bsk_to_lat(const u64 *std_polys, double *lat_polys, const double *g_tw_h, uint32_t n_polys, bool paired, hipStream_t s);
// (levels, base log) of the bootstrap decomposition: (3, 15), (2, 15) and (1, 23) are instantiated in the three kernels
// below; the other 49-bit kernels (variants 1 and 4, N = 4096) and the Goldilocks ones take (3, 15) only
int launch_blind_rotate_lat2(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk_lat,
                             const double *g_tw_h, u64 *out, uint32_t count, uint32_t n, uint32_t levels, uint32_t base_log,
                             hipStream_t s);
// the same with the unrolled key (two LWE coefficients per step): bsk3_lat = [ceil(n/2)][3 keys] GGSW copies in the PAIRED slot
// order of launch_bsk_to_lat, g_root_pow = psi^x for x in [0, 2N) as centred doubles
int launch_blind_rotate_lat2u(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk3_lat,
                              const double *g_tw_h, const double *g_root_pow, u64 *out, uint32_t count, uint32_t n,
                              uint32_t levels, uint32_t base_log, hipStream_t s);
int launch_blind_rotate_tpx(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk,
                            const double *g_tw, u64 *out, uint32_t count, uint32_t n, uint32_t levels, uint32_t base_log,
                            hipStream_t s);
int launch_blind_rotate_lat(const u64 *small_cts, const uint32_t *lut_ids, const double *luts, const double *bsk,
                            const double *g_tw, u64 *out, uint32_t count, uint32_t n, hipStream_t s);
int launch_keyswitch(const u64 *in, const u64 *ksk, const u64 *ks_bias, u64 *out, void *partial, uint32_t slices,
                     uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels, uint32_t base_log, uint32_t ks_stride,
                     hipStream_t s);
// Keyswitch on the matrix cores (ks_mfma.hpp): the key as KS_LIMBS balanced base-256 limbs in MFMA operand order,
// digits = count * big_n * levels bytes, sums = slices * count * KS_LIMBS * ceil((n+1)/32)*32 int32 words.
constexpr uint32_t KS_LIMBS = 7;
int launch_ksk_to_limbs(const u64 *ksk, signed char *limbs, uint32_t rows, uint32_t n, uint32_t ks_stride, hipStream_t s);
int launch_keyswitch_mfma(const u64 *in, const signed char *limbs, signed char *digits, int *sums, u64 *out,
                          uint32_t slices, uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels,
                          uint32_t base_log, hipStream_t s);
int launch_lincomb(const u64 *store, const uint32_t *row_ptr, const uint32_t *idx, const i64 *coef,
                   const u64 *const_body, u64 *out, uint32_t count, uint32_t width, hipStream_t s);
}  // namespace bmi49

// The 2^64 torus (bmi_kernels_t64.hip, bmi_kernels_t64u.hip): ciphertexts, test polynomials and keyswitch key are plain u64
// words; the bootstrap key is LIMBS transform-domain f64 limb polynomials per key polynomial ([poly][limb][N]); the limb
// scheme follows from the precision `prec` (64, 48 or 42 bits) the key is stored at (t64_common.hpp).
namespace bmit {
using gl::i64;
using gl::u64;
constexpr uint32_t KS_LIMBS = 9;    // balanced base-256 limbs of a keyswitch-key word
// (precision, levels, base log) combinations with instantiated kernels
bool shape_supported(int prec, uint32_t levels, uint32_t base_log);
bool shape_supported_unrolled(int prec, uint32_t levels, uint32_t base_log);
int launch_bsk_to_limbs(const u64 *std_polys, double *limb_polys, const double *g_tw, uint32_t n_polys, int prec,
                        hipStream_t s);
int launch_blind_rotate(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_limbs,
                        const double *g_tw, u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels,
                        uint32_t base_log, hipStream_t s);
// latency form (one workgroup of 16 wavefronts per ciphertext): own key copy, per limb in the slot order of the two-wave
// half transform ([poly][limb][512 slots][A_lo, A_hi]); tables of ntt_half_f64.hpp
int launch_bsk_to_lat(const u64 *std_polys, double *lat_polys, const double *g_tw_h, uint32_t n_polys, int prec, hipStream_t s);
int launch_blind_rotate_lat(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_lat,
                            const double *g_tw_h, u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels,
                            uint32_t base_log, hipStream_t s);
// the same with the unrolled key (two LWE coefficients per step; bmi_kernels_t64u.hip): bsk3_lat = launch_bsk_to_lat of the
// unrolled key [ceil(n/2)][3 keys] GGSW copies, g_root_pow = psi^x (mod 2^49 - 720895) for x in [0, 2N) as centred doubles
int launch_blind_rotate_lat2u(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk3_lat,
                              const double *g_tw_h, const double *g_root_pow, u64 *out, uint32_t count, uint32_t n, int prec,
                              uint32_t levels, uint32_t base_log, hipStream_t s);
// wave pairs with the exact limb products carried by the folded complex FFT (bmi_kernels_t64f.hip, fft_wave_f64.hpp): key copy
// [poly][limb][8 registers][64 lanes] complex words; tables fftw::TW_WORDS doubles; same results as launch_blind_rotate
bool shape_supported_fft(int prec, uint32_t levels, uint32_t base_log);
int launch_bsk_to_fft(const u64 *std_polys, double *limb_polys, const double *g_tw_fft, uint32_t n_polys, int prec, hipStream_t s);
// stat (may be null): receives, as the bit pattern of a double, the largest distance of a limb sum from the integer it was rounded to
int launch_blind_rotate_fft(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_fft,
                            const double *g_tw_fft, u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels,
                            uint32_t base_log, unsigned long long *stat, hipStream_t s);
// latency form (one workgroup of 16 wavefronts per ciphertext, half transforms: fft_half_f64.hpp): own key copy, per (polynomial,
// limb) 256 slots of [F_k, F_{k+256}]; tables ffth::HT_WORDS doubles
int launch_bsk_to_latf(const u64 *std_polys, double *lat_polys, const double *g_tw_h, uint32_t n_polys, int prec, hipStream_t s);
int launch_blind_rotate_lat_fft(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_fft,
                                const double *g_tw_fft, u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels,
                                uint32_t base_log, unsigned long long *stat, hipStream_t s);
// the unrolled step (two LWE coefficients) through the floating-point transform (bmi_kernels_t64fu.hip): key at 42 bits of precision
// (two 21-bit limbs: the six-times larger limb sums stay inside the transform's certified range); bsk3_latf = launch_bsk_to_latf of
// the unrolled key, g_zeta_pow = exp(i pi x / 1024) for x in [0, 1024) as (re, im); stat as in launch_blind_rotate_fft
bool shape_supported_unrolled_fft(int prec, uint32_t levels, uint32_t base_log);
int launch_blind_rotate_lat2u_fft(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk3_latf,
                                  const double *g_tw_h, const double *g_zeta_pow, u64 *out, uint32_t count, uint32_t n, int prec,
                                  uint32_t levels, uint32_t base_log, unsigned long long *stat, hipStream_t s);
// ... and its throughput form: two ciphertexts per workgroup, every key word a thread loads multiplied with both (same words)
int launch_blind_rotate_tp2u_fft(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk3_latf,
                                 const double *g_tw_h, const double *g_zeta_pow, u64 *out, uint32_t count, uint32_t n, int prec,
                                 uint32_t levels, uint32_t base_log, hipStream_t s);
// N = 4096 (bmi_kernels_t64q.hip, fft_eighth_f64.hpp): key at 44 bits of precision (two 22-bit limbs), one workgroup per ciphertext,
// one decomposition level of tiles in LDS at a time; key copy per (polynomial, limb) [t 8][256 slots] complex words A_k / 4; tables
// ffte::ET_WORDS doubles
bool shape_supported_quad(int prec, uint32_t levels, uint32_t base_log);
int launch_bsk_to_quad(const u64 *std_polys, double *q_polys, const double *g_tw_e, uint32_t n_polys, int prec, hipStream_t s);
int launch_blind_rotate_quad(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_q, const double *g_tw_e,
                             u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels, uint32_t base_log, unsigned long long *stat,
                             hipStream_t s);
// N = 2048, throughput form (bmi_kernels_t64w2.hip): two ciphertexts per workgroup, every key word multiplied with both; its own key copy,
// per (polynomial, limb) [t 4][256 slots] complex words A_k / 2; tables and results as launch_blind_rotate_wide
int launch_bsk_to_wide2(const u64 *std_polys, double *w2_polys, const double *g_tw_q, uint32_t n_polys, int prec, hipStream_t s);
int launch_blind_rotate_wide2(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_w2, const double *g_tw_q,
                              u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels, uint32_t base_log, hipStream_t s);
// N = 2048 (bmi_kernels_t64w.hip, fft_quarter_f64.hpp): key at 46 bits of precision (two 23-bit limbs), one workgroup of 16
// wavefronts per ciphertext (auto dispatch: up to 256 ciphertexts; launch_blind_rotate_wide2 beyond); key copy per (polynomial, limb) 1,024 complex words A_k / 2 in the order of the
// multiplying threads; tables fftq::QT_WORDS doubles; stat as in launch_blind_rotate_fft
bool shape_supported_wide(int prec, uint32_t levels, uint32_t base_log);
int launch_bsk_to_wide(const u64 *std_polys, double *w_polys, const double *g_tw_q, uint32_t n_polys, int prec, hipStream_t s);
int launch_blind_rotate_wide(const u64 *small_cts, const uint32_t *lut_ids, const u64 *luts, const double *bsk_w, const double *g_tw_q,
                             u64 *out, uint32_t count, uint32_t n, int prec, uint32_t levels, uint32_t base_log, unsigned long long *stat,
                             hipStream_t s);
int launch_keyswitch(const u64 *in, const u64 *ksk, const u64 *ks_bias, u64 *out, void *partial, uint32_t slices,
                     uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels, uint32_t base_log, uint32_t ks_stride,
                     hipStream_t s);
int launch_ksk_to_limbs(const u64 *ksk, signed char *limbs, uint32_t rows, uint32_t n, uint32_t ks_stride, hipStream_t s);
int launch_keyswitch_mfma(const u64 *in, const signed char *limbs, signed char *digits, int *sums, u64 *out,
                          uint32_t slices, uint32_t count, uint32_t n, uint32_t big_n, uint32_t levels,
                          uint32_t base_log, hipStream_t s);
int launch_lincomb(const u64 *store, const uint32_t *row_ptr, const uint32_t *idx, const i64 *coef,
                   const u64 *const_body, u64 *out, uint32_t count, uint32_t width, hipStream_t s);
}  // namespace bmit
