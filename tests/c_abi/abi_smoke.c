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
    CHECK(bmi_key_bytes(ctx, &bsk_b, &ksk_b) == 0 && bsk_b == 61931520ull, "bootstrap key bytes = 61,931,520");
    bmi_ctx_destroy(ctx);
    free(ct);
    free(out);
    printf("abi_smoke OK\n");
    return 0;
}
