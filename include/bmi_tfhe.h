/*
 * bmi_tfhe.h — C ABI of the MI355X-native TFHE programmable-bootstrap engine
 * (libbmi_tfhe.so) that sits behind the QFloat / qfloat_matrix_inverse API of
 * zama-ai/bounty-matrix-inversion.
 *
 * The reference reaches its FHE runtime only through concrete-python's Circuit object
 * (file:line relative to /root/reference):
 *      fhe.Compiler(...).compile(inputset)   matrix_inversion/main.py:53-66
 *      circuit.keygen()                      matrix_inversion/main.py:177
 *      circuit.encrypt(arrays, signs)        matrix_inversion/main.py:73-76
 *      circuit.run(args)                     matrix_inversion/main.py:78-81   <- every PBS executes here
 *      circuit.decrypt(result)               matrix_inversion/main.py:83-86
 * and, inside the traced circuit body, through table look-ups on encrypted scalars
 * (every non-linear operator on a Tracer; the one explicit LUT is
 * matrix_inversion/base_p_arrays.py:359-365).  Each entry point below names the call it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; bmi_last_error() gives the text.
 *   - plain pointers and sizes only.  Pointers named d_* are DEVICE pointers (HIP), all
 *     others are host pointers.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - ciphertext modulus q (bmi_params.q_bits): 2^64 - 2^32 + 1, 2^49 - 720895 or 2^64; words are canonical (< q)
 *     64-bit integers at this boundary for all three.
 *   - a "big" LWE ciphertext has k*N+1 words (mask, then body); a "small" one n+1 words.
 *   - messages are signed integers m encoded as m * 2^delta_log (mod q); a p-bit signed message space uses
 *     delta_log = q_bits - 1 - p (59 resp. 44 for p = 4).
 *   - a context is single-caller (no internal locking); work on a stream is asynchronous
 *     until bmi_sync() or a call that returns data to the host.
 */
#ifndef BMI_TFHE_H
#define BMI_TFHE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bmi_ctx bmi_ctx;

/* bmi_params.q_bits value selecting q = 2^64 EXACTLY - the torus concrete-python computes on: ciphertext and key words
 * are Concrete's own u64 representation (no modulus-switch hop), the external product is the exact negacyclic product
 * mod 2^64.  Message scaling as for q_bits = 64 (delta_log = 63 - msg_bits). */
#define BMI_Q_TORUS64 65u

typedef struct {
    uint32_t n;           /* small LWE dimension (630; up to 1024) */
    uint32_t log_N;       /* log2 polynomial size (10; 11 and 12 on the 2^64 torus and the 49-bit field) */
    uint32_t k;           /* GLWE dimension (1) */
    uint32_t bs_levels;   /* l: bootstrap decomposition levels (3; (2, 15) and (1, 23) on the 49-bit field at N <= 2048, */
    uint32_t bs_base_log; /* bootstrap decomposition base log (15; 10 on the 2^64 torus)   (3 | 2, 10 | 15) on the torus) */
    uint32_t ks_levels;   /* keyswitch levels (8; up to 16 through the matrix-core kernel) */
    uint32_t ks_base_log; /* keyswitch base log (4) */
    uint32_t q_bits;      /* ciphertext modulus: 64 -> q = 2^64 - 2^32 + 1 (Goldilocks, integer kernels);
                             49 -> q = 2^49 - 720895 (exact integers carried in f64: the fastest kernels);
                             BMI_Q_TORUS64 -> q = 2^64 (Concrete's torus; exact products through limb-split f64
                             transforms); 0 = 64 */
    double lwe_noise;     /* std-dev / q of keyswitch-key encryptions */
    double glwe_noise;    /* std-dev / q of bootstrap-key rows and fresh big-key encryptions */
} bmi_params;

/* The north-star parameter set of BASELINE.json (n=630, N=1024, k=1, l=3) with the build's default modulus: q = 2^64
 * (BMI_Q_TORUS64, the modulus concrete-python computes on; Bg = 2^10, bootstrap key at 48 bits of precision).  One default across
 * the library, EncryptedMatrixInversion, smoke() and bench.py. */
int bmi_default_params(bmi_params *out);
/* ... and with an explicit choice of the ciphertext modulus (q_bits = BMI_Q_TORUS64, 49 or 64). */
int bmi_default_params_for(uint32_t q_bits, bmi_params *out);

/* Named parameter sets:
 *   "north_star_torus64"    BASELINE.json's shape (n 630, N 1024, k 1, l 3) on the 2^64 torus, Concrete's own modulus, with
 *                           Bg = 2^10 (two-limb key at 48 bits of precision, see bmi_set_bsk_precision) - bmi_default_params
 *   "north_star"            the same shape on the 49-bit field q = 2^49 - 720895, Bg = 2^15 (rounds 1-2's default);  "north_star_goldilocks": (l 3, Bg 2^15) on 2^64 - 2^32 + 1
 *   "secure128"             n 742, N 2048, k 1, l 2 x 15 bits, keyswitch 8 x 2 bits, 49-bit field, LWE noise 7.07e-6
 *                           (2^-17.1), GLWE noise 2^-44.  Security: the (dimension, noise / q) pairs are those of TFHE-rs'
 *                           published 128-bit set PARAM_MESSAGE_2_CARRY_2_KS_PBS (lwe_dimension 742, lwe std 7.07e-6;
 *                           glwe k N = 2048 with std 2.94e-16: here the GLWE noise is LARGER, 5.7e-14, hence at least as
 *                           hard), binary keys as there; the hardness of LWE depends on (n, sigma / q), not on q itself.
 *                           The decompositions are this build's (security does not depend on them).  4-bit look-ups
 *                           sit at ~8.7 sigma of keyswitch + mod-switch noise (measured, tests/test_gpu_parity.py: a
 *                           failure probability of ~2^-58 each; that set's own target is 2^-40).  What compiler.compile's parameter
 *                           optimiser guarantees for the reference (main.py:66) - here a fixed, documented set.
 *   "secure128_torus"       the same security-relevant pairs (n 742 at 2^-17.1; GLWE 2048 at 2^-44) on Concrete's own modulus
 *                           q = 2^64: N 2048, k 1, l 3 x 10 bits, keyswitch 8 x 2 bits, bootstrap key stored at 46 bits of
 *                           precision (two 23-bit limbs; exact limb sums through the floating-point transform,
 *                           k_blind_rotate_w_t64f).  Output noise 2^-22.9 (the 49-bit preset: 2^-19.5).
 *   "secure128_torus_wide"  the same LWE pair under N 4096 on q = 2^64, for 5-bit look-ups (the reference's unmodified circuits,
 *                           bases other than 2): l 3 x 10 bits, bootstrap key at 44 bits of precision (two 22-bit limbs,
 *                           k_blind_rotate_q_t64f), keyswitch 16 x 1 bit - the keyswitch noise bounds the margin: a 5-bit look-up
 *                           sits at 5.4 sigma (4.6 with 8 x 2 bits; 4.4 at N 2048).  error_budget.choose_params decides per circuit.
 * The north-star sets keep n = 630 as BASELINE.json prescribes; their noise is sized for correctness, NOT for 128-bit
 * security. */
int bmi_preset_params(const char *name, bmi_params *out);

/* replaces fhe.Compiler(...).compile(...) (main.py:53-66): fixes the crypto parameters, binds a GPU. */
int bmi_ctx_create(const bmi_params *params, int device, bmi_ctx **out);
void bmi_ctx_destroy(bmi_ctx *ctx);
/* ctx may be NULL: returns the last error of a failed bmi_ctx_create on this thread. */
const char *bmi_last_error(const bmi_ctx *ctx);
int bmi_get_params(const bmi_ctx *ctx, bmi_params *out);

/* replaces circuit.keygen() (main.py:177).  Generates both secret keys, the bootstrap key (uploaded and transformed
 * to the NTT domain on the GPU) and the keyswitch key.  All randomness comes from a CSPRNG: ChaCha20 keyed from the
 * operating system (getrandom), with one key for secret material (key bits, noise) and an independent one for the
 * public masks, so the evaluation keys reveal nothing about the secret streams.  bmi_encrypt then draws fresh
 * mask + noise for every ciphertext from the same generators (one nonce per ciphertext, never reused). */
int bmi_keygen(bmi_ctx *ctx);
/* TEST ONLY - NOT CRYPTOGRAPHIC.  The same key set, deterministic in `seed`: every word is a counter-indexed
 * splitmix64 output (an invertible map: public key words reveal the seed and with it the secret keys), and
 * bmi_encrypt under such a key set is deterministic too (ciphertext i of a key set repeats its mask and noise).  It
 * exists so that oracle/tfhe_oracle.c can reproduce keys and ciphertexts bit for bit in the parity tests and so that
 * every rank of a benchmark holds the same keys without an exchange (the reference's analogue: Concrete's
 * use_insecure_key_cache, qfloat_matrix_inversion.py:997-998).  Never use it for data that matters. */
int bmi_keygen_insecure_deterministic(bmi_ctx *ctx, uint64_t seed);
/* Test hook: copies out the secret keys and the standard-domain evaluation keys.
 * Sizes: sk_small[n], sk_big[k*N], bsk[n*(k+1)*l*(k+1)*N], ksk[k*N*ks_levels*(n+1)]; any may be NULL. */
int bmi_export_keys(const bmi_ctx *ctx, uint64_t *sk_small, uint64_t *sk_big, uint64_t *bsk, uint64_t *ksk);
/* Loads a key set generated elsewhere (same sizes and standard-domain layout as bmi_export_keys; words reduced mod q)
 * and uploads the evaluation keys.  sk_small and sk_big may both be NULL: the context is then evaluation-only (PBS,
 * keyswitch, linear combinations work; bmi_encrypt / bmi_decrypt / bmi_phase fail) - the server half of the
 * client/server split the reference reaches through Concrete's key objects (circuit.keys, the ".keys" cache of
 * qfloat_matrix_inversion.py:997-998). */
int bmi_import_keys(bmi_ctx *ctx, const uint64_t *sk_small, const uint64_t *sk_big, const uint64_t *bsk, const uint64_t *ksk);

/* Evaluation keys for secret keys made elsewhere (binary, sizes as in bmi_export_keys): the trusted-benchmark half of
 * interoperating with a Concrete client, whose LWE secret keys are binary vectors too.
 * seed = 0: masks and noise from the CSPRNG (production).  seed != 0: deterministic test vectors, NOT cryptographic
 * (same caveats as bmi_keygen_insecure_deterministic). */
int bmi_keygen_from_secret(bmi_ctx *ctx, const uint64_t *sk_small, const uint64_t *sk_big, uint64_t seed);
/* 2^64-torus interop (context-free, host): ciphertext words as Concrete stores them (u64, the torus scaled by 2^64)
 * to words mod q and back, by the modulus switch round(x * q / 2^64) resp. round(x * 2^64 / q).  A message m * 2^(63-p)
 * on the torus becomes m * 2^(q_bits-1-p) up to a relative 2^-30; the rounding adds < sqrt(kN/12) units of 1/q to
 * the phase.  `words` counts u64 words (count * (k*N+1) for a batch of big-key ciphertexts). */
int bmi_torus64_to_field(uint32_t q_bits, const uint64_t *in, uint64_t words, uint64_t *out);
int bmi_field_to_torus64(uint32_t q_bits, const uint64_t *in, uint64_t words, uint64_t *out);

/* replaces circuit.encrypt (main.py:76): big-key LWE encryptions of msgs[i] * 2^delta_log.  Fresh CSPRNG mask and noise
 * per ciphertext, except under bmi_keygen_insecure_deterministic keys (deterministic, test only).  Key sets loaded
 * with bmi_import_keys always encrypt from the CSPRNG. */
int bmi_encrypt(bmi_ctx *ctx, const int64_t *msgs, uint32_t count, uint32_t delta_log, uint64_t *ct_out);
/* replaces circuit.decrypt (main.py:86): msgs[i] = round(phase / 2^delta_log), signed. */
int bmi_decrypt(const bmi_ctx *ctx, const uint64_t *ct_in, uint32_t count, uint32_t delta_log, int64_t *msgs);
/* raw phases (for noise measurements) */
int bmi_phase(const bmi_ctx *ctx, const uint64_t *ct_in, uint32_t count, uint64_t *phase);

/* replaces a table look-up definition (fhe.univariate, base_p_arrays.py:365, and every non-linear
 * Tracer operator).  table[m + 2^(msg_bits-1)] = f(m) for signed m in [-2^(msg_bits-1), 2^(msg_bits-1));
 * the looked-up value is returned encoded as f(m) * 2^out_delta_log.  The input ciphertext of a PBS
 * using this LUT must encode m * 2^(q_bits-1-msg_bits).  Returns the id used by the batch calls.
 * On the 2^64 torus out_delta_log >= 22 (the kernels on a rounded bootstrap key keep accumulators as multiples of 2^16 / 2^22). */
int bmi_lut_register(bmi_ctx *ctx, const int64_t *table, uint32_t msg_bits, uint32_t out_delta_log, uint32_t *lut_id);
/* the N-coefficient test polynomial built for a LUT (host copy; test hook) */
int bmi_lut_get(const bmi_ctx *ctx, uint32_t lut_id, uint64_t *test_vector);

/* ---- the hot path: replaces circuit.run (main.py:81) -------------------------------------------- */
/* Programmable bootstrap of `count` big-key ciphertexts, Concrete order:
 *   keyswitch (big -> small) -> modulus switch to 2N -> blind rotation -> sample extraction.
 * d_in/d_out: [count][k*N+1] device words (may alias); d_lut_ids: [count] device uint32, every id one returned by
 * bmi_lut_register (the device-pointer entry points cannot inspect their ids; the *_host forms refuse unknown ones). */
int bmi_pbs_batch(bmi_ctx *ctx, const uint64_t *d_in, const uint32_t *d_lut_ids, uint32_t count, uint64_t *d_out,
                  void *stream);
/* the two stages, separately (d_small: [count][n+1]) */
int bmi_keyswitch_batch(bmi_ctx *ctx, const uint64_t *d_in, uint32_t count, uint64_t *d_small, void *stream);
int bmi_blind_rotate_batch(bmi_ctx *ctx, const uint64_t *d_small, const uint32_t *d_lut_ids, uint32_t count,
                           uint64_t *d_out, void *stream);
/* Leveled (PBS-free) linear ops on big-key ciphertexts, CSR form:
 *   out[i] = const_body[i] + sum_{e in [row_ptr[i], row_ptr[i+1])} coef[e] * store[idx[e]]
 * replaces + - and constant * on Tracers and np.sum / slicing (qfloat.py:811-826, 901, 1015). */
int bmi_lincomb_batch(bmi_ctx *ctx, const uint64_t *d_store, const uint32_t *d_row_ptr, const uint32_t *d_idx,
                      const int64_t *d_coef, const uint64_t *d_const_body, uint32_t count, uint64_t *d_out,
                      void *stream);
/* d_store[d_rows[i]] = d_src[i] for i < count (big-key ciphertext rows).  Lets a scheduler keep a level's outputs
 * contiguous for the batch call and still place them in recycled rows of its ciphertext store (the reference leaves
 * value lifetimes to Concrete's runtime inside circuit.run, main.py:81). */
int bmi_scatter_rows(bmi_ctx *ctx, const uint64_t *d_src, uint32_t count, uint64_t *d_store, const uint32_t *d_rows,
                     void *stream);
/* Host-buffer convenience forms (what a ctypes/cgo binding over numpy buffers would call):
 * copy in, run on the context's own stream, copy out, synchronise. */
int bmi_pbs_batch_host(bmi_ctx *ctx, const uint64_t *in, const uint32_t *lut_ids, uint32_t count, uint64_t *out);
int bmi_keyswitch_batch_host(bmi_ctx *ctx, const uint64_t *in, uint32_t count, uint64_t *small_out);
int bmi_blind_rotate_batch_host(bmi_ctx *ctx, const uint64_t *small_in, const uint32_t *lut_ids, uint32_t count,
                                uint64_t *out);
/* Test hook of the floating-point-transform kernels (2^64 torus; N = 1024: key at 48 bits, kernel variants 5 / 6; N = 2048: key at
 * 46 bits; N = 4096: key at 44 bits, k_blind_rotate_q_t64f): blind rotation of `count` small-key ciphertexts by the wave-pair kernel - by the latency form when kernel variant 6 or 2
 * is selected, by k_blind_rotate_w_t64f at N = 2048 -, which here also records how far every limb sum was from the integer it was
 * rounded to.  *max_distance must stay far below 1/2 (measured: below 2^-11) - that is what makes the
 * rounded results the exact integer sums, whatever the order of the floating-point operations. */
int bmi_fft_margin_host(bmi_ctx *ctx, const uint64_t *small_in, const uint32_t *lut_ids, uint32_t count, uint64_t *out,
                        double *max_distance);
/* negacyclic product of two polynomials through the GPU NTT (test hook for the transform) */
int bmi_negacyclic_mul_host(bmi_ctx *ctx, const uint64_t *a, const uint64_t *b, uint32_t count, uint64_t *c);

/* Pre-sizes the context's internal scratch (small-key ciphertexts of bmi_pbs_batch, digit matrix and limb sums of the
 * matrix-core keyswitch) for batches up to max_count, so that no allocation happens on the hot path afterwards. */
int bmi_reserve(bmi_ctx *ctx, uint32_t max_count);

int bmi_sync(bmi_ctx *ctx, void *stream);

/* Selects the blind-rotation kernel: 0 = auto (by batch size), 1 = throughput, a pair of wavefronts per
 * ciphertext exchanging every level, 2 = latency (one workgroup of 8 wavefronts per ciphertext), 3 = throughput,
 * a pair of wavefronts per ciphertext exchanging once per CMUX (49-bit field; what auto picks for large batches),
 * 4 = the one-wavefront-per-transform latency kernel (2 is the two-wavefronts-per-transform one on the 49-bit field),
 * 5 = 2^64 torus only, bootstrap key at 48 bits in base 2^10 (the torus default): the wave-pair kernel whose exact limb
 *     products are carried by a folded 512-point complex FFT in f64 and rounded to the nearest integer (same words as 1 / 3,
 *     which pin the exact transform mod 2^49 - 720895; what auto picks for large batches on that key),
 * 6 = the same for the latency form (one workgroup per ciphertext; what auto and 2 pick on that key; 4 pins the exact transform).
 * 2^64 torus at N = 2048: auto = one workgroup per ciphertext up to 256 ciphertexts (2 pins it), two ciphertexts per workgroup sharing
 *     the key words beyond (1 / 3 pin it); the same words.  N = 4096: one kernel. */
int bmi_set_kernel_variant(bmi_ctx *ctx, int variant);

/* 2^64 torus only, before keygen / import: precision the bootstrap key is stored at.  No transform exists mod 2^64, so the
 * external product is computed exactly over the integers from limbs of every key word (csrc/t64_common.hpp); the number of
 * limbs follows from the precision:
 *   64  exact key, three 22-bit limbs.  Default at Bg = 2^15 (51 k PBS/s, 5.5 ms per bootstrap).
 *   48  every key word (generated here or imported) rounded to a multiple of 2^16, two 24-bit limbs: 2/3 of the
 *       multiply-accumulates and inverse transforms.  Needs Bg <= 2^10 (a limb's sum must stay below p/2).  DEFAULT of the
 *       torus parameter set ("north_star_torus64": l = 3, Bg = 2^10): the rounding errors of a row (2^15 / sqrt 3 per word,
 *       summed over the ~N/2 set bits of the GLWE key: 2^18.7) stay under the key noise 2^20, and the finer base more than
 *       pays for them - bootstrap output noise 2^-22.6 against 2^-19.85 of (Bg = 2^15, exact key), measured on the formula
 *       (tests/test_gpu_parity.py).  With bmi_set_bsk_unroll: the exact-transform unrolled kernel (k_blind_rotate_lat2u_t64).
 *   46  N = 2048 only (its default and only precision; "secure128_torus"): multiples of 2^18, two 23-bit limbs - the limb width at
 *       which the a-priori error bound of the 1,024-point floating-point transform still certifies the rounding of a limb sum
 *       (0.41 < 1/2, csrc/fft_quarter_f64.hpp; 24-bit limbs: 0.83).
 *   44  N = 4096 only (its default and only precision; 6-bit look-ups, "secure128_torus_wide"): multiples of 2^20, two 22-bit limbs
 *       (2,048-point transform: bound 0.45, csrc/fft_eighth_f64.hpp; 23-bit limbs: 0.90).
 *   42  multiples of 2^22, two 21-bit limbs.  At Bg = 2^10 in UNROLLED mode only - the precision bmi_set_bsk_unroll(ctx, 2) selects
 *       by itself when none was set: the three scaled products of an unrolled step make the limb sums six times larger, and 21-bit
 *       limbs keep them inside the range the floating-point transform is certified for (k_blind_rotate_lat2u_t64f; output noise
 *       2^-19.3).  At Bg = 2^15: round 2's throughput option for flat PBS batches (output noise 2^-15.15: too noisy for the
 *       encrypted inverse).
 * The rounded key is the context's key from then on (bmi_export_keys / bmi_export_bsk_unrolled return it; results are
 * bit-exact against an oracle given that key; oracle/tfhe_oracle.c ora_round_key states the rule).  The rounding only ever
 * adds noise to valid LWE samples; generating the key on the grid directly would instead round its own noise away. */
int bmi_set_bsk_precision(bmi_ctx *ctx, uint32_t bits);
/* the precision in force (64 on the prime fields) */
int bmi_get_bsk_precision(const bmi_ctx *ctx, uint32_t *bits);

/* 49-bit field at N = 1024, and at N = 2048 with l <= 2 (the secure128 shape), and the 2^64 torus at its default set (Bg = 2^10;
 * key at 42 bits: k_blind_rotate_lat2u_t64f, the floating-point-transform route, 2.9 ms per bootstrap - what this call selects when
 * no precision was set -, or pinned at 48 bits: k_blind_rotate_lat2u_t64, the exact transform, 3.3 ms): bootstrap-key unrolling (Zhou et al. 2018, Bourse et al. 2018; unrolling factor 2).
 * factor 1 (default): the blind rotation of CGGI, one LWE coefficient per step.  factor 2: a step absorbs two coefficients,
 *   ACC <- ACC + sum_{j<3} (X^(c_j) - 1) (K_j [.] ACC),  c = (a + a', a, a'),  K = GGSW(s s'), GGSW(s (1 - s')), GGSW((1 - s) s'),
 * with one decomposition and one set of forward transforms per step (half as many as the plain rotation; the factors X^c - 1
 * are applied in the transform domain).  Every batch size then runs k_blind_rotate_lat2u_49 (49-bit field, N = 1024),
 * k_blind_rotate_wide49u (N = 2048) or the torus kernel above, one workgroup per ciphertext; on the torus's floating-point route,
 * batches beyond 256 ciphertexts run k_blind_rotate_tp2u_t64f: two ciphertexts per workgroup sharing every key word in registers
 * (half the key bytes per bootstrap; the same words).
 * The unrolled key (1.5 x the plain key's size) is generated by the next keygen call, or at once when the context already
 * holds secret keys; an evaluation-only context receives it through bmi_import_bsk_unrolled.  Layout:
 * [ceil(n/2)][3][(k+1) l][(k+1)][N] words, standard domain (an odd n is completed by a zero key bit).  Price: the key-noise
 * term of the output variance triples (three products per pair of coefficients, each scaled by X^c - 1) - size the GLWE noise
 * for it: 2^-41 keeps the margin of the default (3, 2^15) set; a one-level (1, 2^23) decomposition needs about 2^-42 or less for
 * 4-bit look-ups.  Results are
 * bit-exact against oracle/tfhe_oracle.c ora_blind_rotate_extract_unrolled on the same keys. */
int bmi_set_bsk_unroll(bmi_ctx *ctx, uint32_t factor);
int bmi_import_bsk_unrolled(bmi_ctx *ctx, const uint64_t *bsk3);
int bmi_export_bsk_unrolled(const bmi_ctx *ctx, uint64_t *bsk3);

/* Selects the keyswitch kernel: 0 = auto (int8 matrix-core product when the parameter set allows it),
 * 1 = scalar 96-bit multiply-accumulate kernel. */
int bmi_set_keyswitch_variant(bmi_ctx *ctx, int variant);

/* ---- compiler passes on a traced circuit in flat (CSR) form; context-free, CPU only -----------------------------
 * What fhe.Compiler(...).compile does inside Concrete (main.py:53-66) for this path.  A circuit has n_in input leaves
 * and n_nodes look-ups in creation (= topological) order; look-up i reads the leaves term_leaf[node_ptr[i] ..
 * node_ptr[i+1]) and produces leaf n_in + i.
 * bmi_circuit_prune: live_node[i] = 1 iff an output (out_leaf[0 .. n_out_terms)) depends on look-up i.
 * bmi_circuit_schedule: level (1 .. depth) of every look-up; same depth as the ASAP schedule, but nodes with slack
 * are moved out of levels that would otherwise need one more kernel round (`round_` ciphertexts per latency-kernel
 * round up to two rounds, `wide_round` per throughput-kernel round beyond). */
int bmi_circuit_prune(uint32_t n_in, uint32_t n_nodes, const int64_t *node_ptr, const int32_t *term_leaf,
                      const int32_t *out_leaf, uint64_t n_out_terms, uint8_t *live_node);
int bmi_circuit_schedule(uint32_t n_in, uint32_t n_nodes, const int64_t *node_ptr, const int32_t *term_leaf,
                         uint32_t round_, uint32_t wide_round, int32_t *level_out, int32_t *depth_out);

/* bytes of device memory held by the keys: every resident transform-domain copy of the bootstrap key (one per selectable kernel
 * family, plus the unrolled key when present), and the keyswitch key in word form */
int bmi_key_bytes(const bmi_ctx *ctx, uint64_t *bsk_bytes, uint64_t *ksk_bytes);

#ifdef __cplusplus
}
#endif
#endif /* BMI_TFHE_H */
