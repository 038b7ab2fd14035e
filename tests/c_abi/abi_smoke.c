/* Plain-C consumer of include/bmi_tfhe.h: what a cgo / FFI binding of the reference would call.
 * Build (see tests/test_gpu_c_abi.py):  gcc abi_smoke.c -I include -L <lib> -lbmi_tfhe -L/opt/rocm/lib -lamdhip64 -lm
 * Checks error behaviour (bad arguments, use before keygen) and one PBS batch end to end. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bmi_tfhe.h"

#define CHECK(cond, msg)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fprintf(stderr, "FAIL: %s (line %d)\n", msg, __LINE__); \
            return 1;                                     \
        }                                                 \
    } while (0)

int main(void) {
    bmi_params P;
    CHECK(bmi_default_params(&P) == 0, "default params");
    CHECK(P.n == 630 && P.log_N == 10 && P.k == 1 && P.bs_levels == 3, "north-star parameter set");
    CHECK(bmi_default_params(NULL) < 0, "NULL out pointer rejected");

    bmi_ctx *ctx = NULL;
    bmi_params bad = P;
    bad.log_N = 13; /* no HIP kernel for N = 8192 in this build (N = 2048 and 4096 exist on the 49-bit field) */
    CHECK(bmi_ctx_create(&bad, 0, &ctx) < 0 && ctx == NULL, "unsupported parameters rejected");
    CHECK(strlen(bmi_last_error(NULL)) > 0, "create error text available");
    CHECK(bmi_ctx_create(&P, 99, &ctx) < 0, "bad device index rejected");
    CHECK(bmi_ctx_create(&P, 0, &ctx) == 0 && ctx != NULL, "context created");

    const uint32_t big = P.k * (1u << P.log_N) + 1;
    int64_t msgs[5] = {-8, -1, 0, 3, 7};
    uint64_t *ct = (uint64_t *)malloc(5 * big * 8), *out = (uint64_t *)malloc(5 * big * 8);
    const uint32_t DL = (P.q_bits == 49 ? 49 : 64) - 1 - 4; /* scaling exponent of a 4-bit signed message space */
    CHECK(bmi_encrypt(ctx, msgs, 5, DL, ct) < 0, "encrypt before keygen fails");
    CHECK(strstr(bmi_last_error(ctx), "keygen") != NULL, "error text mentions keygen");
    CHECK(bmi_keygen(ctx) == 0, "keygen (CSPRNG)");

    int64_t table[16];
    for (int m = -8; m < 8; m++) table[m + 8] = (m * m) % 16 - 8;
    uint32_t lut = 0, ids[5];
    CHECK(bmi_lut_register(ctx, table, 11, DL, &lut) < 0, "msg_bits too large for N rejected");
    CHECK(bmi_lut_register(ctx, table, 4, DL, &lut) == 0, "lut registered");
    for (int i = 0; i < 5; i++) ids[i] = lut;
    CHECK(bmi_encrypt(ctx, msgs, 5, DL, ct) == 0, "encrypt");
    CHECK(bmi_pbs_batch_host(ctx, ct, ids, 5, out) == 0, "pbs batch");
    int64_t dec[5];
    CHECK(bmi_decrypt(ctx, out, 5, DL, dec) == 0, "decrypt");
    for (int i = 0; i < 5; i++) CHECK(dec[i] == table[msgs[i] + 8], "LUT value");
    CHECK(bmi_pbs_batch_host(ctx, ct, ids, 0, out) == 0, "empty batch is a no-op");
    uint64_t bsk_b = 0, ksk_b = 0;
    CHECK(bmi_key_bytes(ctx, &bsk_b, &ksk_b) == 0 && bsk_b >= 61931520ull && bsk_b % 61931520ull == 0,
          "bootstrap key bytes = resident copies x limbs x 61,931,520");
    /* fresh CSPRNG randomness: two encryptions of the same messages differ, both decrypt */
    uint64_t *ct2 = (uint64_t *)malloc(5 * big * 8);
    CHECK(bmi_encrypt(ctx, msgs, 5, DL, ct2) == 0 && memcmp(ct, ct2, 5 * big * 8) != 0, "encryptions are randomised");
    CHECK(bmi_decrypt(ctx, ct2, 5, DL, dec) == 0 && dec[0] == -8 && dec[4] == 7, "second encryption decrypts");
    /* the test-only seeded key generation is reproducible (keys and ciphertexts) */
    CHECK(bmi_keygen_insecure_deterministic(ctx, 42) == 0 && bmi_encrypt(ctx, msgs, 5, DL, ct) == 0, "seeded keygen");
    CHECK(bmi_keygen_insecure_deterministic(ctx, 42) == 0 && bmi_encrypt(ctx, msgs, 5, DL, ct2) == 0, "seeded keygen again");
    CHECK(memcmp(ct, ct2, 5 * big * 8) == 0, "seeded path is deterministic");
    {
        /* bootstrap-key unrolling: the unrolled key is derived from the secret keys the context holds; same look-ups, other
         * ciphertext bits; export / import round trip; refused where no kernel exists */
        CHECK(bmi_set_bsk_unroll(ctx, 3) < 0, "unrolling factor 3 rejected");
        CHECK(bmi_export_bsk_unrolled(ctx, out) < 0, "no unrolled key before bmi_set_bsk_unroll");
        CHECK(bmi_set_bsk_unroll(ctx, 2) == 0, "unroll 2");
        CHECK(bmi_lut_register(ctx, table, 4, DL, &lut) == 0, "lut");
        for (int i = 0; i < 5; i++) ids[i] = lut;
        CHECK(bmi_pbs_batch_host(ctx, ct, ids, 5, ct2) == 0 && bmi_decrypt(ctx, ct2, 5, DL, dec) == 0, "unrolled pbs");
        for (int i = 0; i < 5; i++) CHECK(dec[i] == table[msgs[i] + 8], "LUT value (unrolled key)");
        const size_t words3 = (size_t)((P.n + 1) / 2) * 3 * (P.k + 1) * P.bs_levels * (P.k + 1) * (1u << P.log_N);
        uint64_t *bsk3 = (uint64_t *)malloc(words3 * 8);
        CHECK(bsk3 && bmi_export_bsk_unrolled(ctx, bsk3) == 0 && bmi_import_bsk_unrolled(ctx, bsk3) == 0, "unrolled key export / import");
        CHECK(bmi_pbs_batch_host(ctx, ct, ids, 5, out) == 0 && memcmp(out, ct2, 5 * big * 8) == 0, "same key, same ciphertexts");
        CHECK(bmi_set_bsk_unroll(ctx, 1) == 0 && bmi_pbs_batch_host(ctx, ct, ids, 5, out) == 0 && memcmp(out, ct2, 5 * big * 8) != 0,
              "back to the plain blind rotation");
        free(bsk3);
    }
    bmi_ctx_destroy(ctx);

    /* named presets; the 2^64 torus (Concrete's modulus) end to end; the compiler passes (context-free, CPU) */
    bmi_params S;
    CHECK(bmi_preset_params("secure128", &S) == 0 && S.n == 742 && S.log_N == 11, "secure128 preset");
    CHECK(bmi_preset_params("secure128_torus", &S) == 0 && S.n == 742 && S.log_N == 11 && S.q_bits == BMI_Q_TORUS64, "secure128_torus preset");
    CHECK(bmi_preset_params("secure128_torus_wide", &S) == 0 && S.n == 742 && S.log_N == 12 && S.ks_levels == 16 && S.ks_base_log == 1,
          "secure128_torus_wide preset");
    CHECK(bmi_preset_params("no such set", &S) < 0, "unknown preset rejected");
    CHECK(bmi_preset_params("north_star_torus64", &S) == 0 && S.q_bits == BMI_Q_TORUS64, "torus preset");
    CHECK(bmi_ctx_create(&S, 0, &ctx) == 0, "torus context");
    {
        /* the torus set: Bg = 2^10, bootstrap key stored at 48 bits of precision (two limbs); 42 bits belongs to Bg = 2^15 - and, at
         * this base, to the unrolled mode only (bmi_set_bsk_unroll first) */
        uint32_t prec = 0;
        CHECK(S.bs_base_log == 10 && bmi_get_bsk_precision(ctx, &prec) == 0 && prec == 48, "torus set: base 2^10, 48-bit key");
        CHECK(bmi_set_bsk_precision(ctx, 42) < 0, "42-bit key refused at base 2^10");
    }
    CHECK(bmi_keygen(ctx) == 0, "torus keygen");
    CHECK(bmi_lut_register(ctx, table, 4, 59, &lut) == 0, "torus lut");
    for (int i = 0; i < 5; i++) ids[i] = lut;
    CHECK(bmi_encrypt(ctx, msgs, 5, 59, ct) == 0 && bmi_pbs_batch_host(ctx, ct, ids, 5, out) == 0, "torus pbs");
    CHECK(bmi_decrypt(ctx, out, 5, 59, dec) == 0, "torus decrypt");
    for (int i = 0; i < 5; i++) CHECK(dec[i] == table[msgs[i] + 8], "torus LUT value");
    {
        /* the kernels whose exact limb products go through the f64 FFT (variants 5 / 6; what auto ran above) against the exact
         * transform mod 2^49 - 720895 (variant 1 / 4): the same words; the test hook reports how far the FFT's limb sums were from
         * the integers they were rounded to (far below 1/2) */
        uint64_t *sm = (uint64_t *)malloc(5 * (S.n + 1) * 8), *r5 = (uint64_t *)malloc(5 * big * 8), *r1 = (uint64_t *)malloc(5 * big * 8);
        double margin = -1.0;
        CHECK(bmi_keyswitch_batch_host(ctx, ct, 5, sm) == 0, "torus keyswitch");
        CHECK(bmi_set_kernel_variant(ctx, 5) == 0 && bmi_blind_rotate_batch_host(ctx, sm, ids, 5, r5) == 0, "fft wave-pair kernel");
        CHECK(bmi_set_kernel_variant(ctx, 1) == 0 && bmi_blind_rotate_batch_host(ctx, sm, ids, 5, r1) == 0, "exact-transform wave-pair kernel");
        CHECK(memcmp(r5, r1, 5 * big * 8) == 0 && memcmp(r5, out, 5 * big * 8) == 0, "fft route == exact-transform route == auto");
        CHECK(bmi_set_kernel_variant(ctx, 6) == 0 && bmi_blind_rotate_batch_host(ctx, sm, ids, 5, r5) == 0 && memcmp(r5, r1, 5 * big * 8) == 0,
              "fft latency kernel");
        CHECK(bmi_set_kernel_variant(ctx, 0) == 0, "auto");
        CHECK(bmi_fft_margin_host(ctx, sm, ids, 5, r5, &margin) == 0 && memcmp(r5, r1, 5 * big * 8) == 0 && margin > 0.0 && margin < 1.0 / 512,
              "fft rounding margin");
        CHECK(bmi_lut_register(ctx, table, 4, 15, &lut) < 0, "tables below 2^22 refused on the torus");
        free(sm); free(r5); free(r1);
    }
    /* the unrolled torus kernel: the unrolled key is derived from the secret keys held; other ciphertext bits, same messages */
    CHECK(bmi_set_bsk_unroll(ctx, 2) == 0 && bmi_pbs_batch_host(ctx, ct, ids, 5, ct2) == 0, "unrolled torus pbs");
    CHECK(bmi_decrypt(ctx, ct2, 5, 59, dec) == 0 && memcmp(out, ct2, 5 * big * 8) != 0, "unrolled torus decrypt");
    for (int i = 0; i < 5; i++) CHECK(dec[i] == table[msgs[i] + 8], "torus LUT value (unrolled key)");
    bmi_ctx_destroy(ctx);
    {
        bmi_params T15 = S;     /* round 2's torus set (Bg = 2^15, exact key): no unrolled kernel (its limb sums would not fit) */
        T15.bs_base_log = 15;
        CHECK(bmi_ctx_create(&T15, 0, &ctx) == 0 && bmi_set_bsk_unroll(ctx, 2) < 0, "no unrolled torus kernel at base 2^15");
        bmi_ctx_destroy(ctx);
    }
    {
        /* x0, x1 inputs; node 0 reads x0; node 1 reads node 0 and x1; node 2 reads x1 and feeds nothing; output = node 1 */
        const int64_t ptr[4] = {0, 1, 3, 4};
        const int32_t leaf[4] = {0, 2, 1, 1}, outl[1] = {3};
        uint8_t live[3];
        int32_t level[3], depth = 0;
        CHECK(bmi_circuit_prune(2, 3, ptr, leaf, outl, 1, live) == 0 && live[0] && live[1] && !live[2], "prune");
        CHECK(bmi_circuit_schedule(2, 3, ptr, leaf, 256, 1024, level, &depth) == 0 && depth == 2 && level[0] == 1 && level[1] == 2,
              "schedule");
        const int32_t bad_leaf[4] = {0, 4, 1, 1};   /* node 1 reading its own output: refused */
        CHECK(bmi_circuit_schedule(2, 3, ptr, bad_leaf, 256, 1024, level, &depth) < 0, "malformed circuit rejected");
    }
    free(ct);
    free(ct2);
    free(out);
    printf("abi_smoke OK\n");
    return 0;
}
