/*
 * ORACLE (test infrastructure, NOT product code) — exact CPU restatement of the
 * TFHE programmable bootstrap (PBS) this repository accelerates.
 *
 * PARITY STATUS: "parity unpinned" at the ciphertext level.  The reference
 * (zama-ai/bounty-matrix-inversion) contains no PBS code: every table lookup is
 * executed by the third-party dependency `concrete-python == 2.1.0`
 * (/root/reference/pyproject.toml:13), which is neither vendored nor installed
 * here and whose CPU bootstrap uses an f64 FFT (not exactly reproducible).  The
 * reference's own call sites for this path are main.py:76,81,86 and
 * qfloat_matrix_inversion.py:1031-1040 (encrypt / run / decrypt) and its tests
 * (tests/test_qfloat_fhe.py:186-335) assert only on DECRYPTED values.  This file
 * therefore restates the published CGGI/TFHE algorithm (keyswitch -> modulus
 * switch -> blind rotation by GGSW x GLWE external products -> sample extraction;
 * SURVEY.md Appendix A) in exact integer arithmetic over the Goldilocks prime
 * q = 2^64 - 2^32 + 1, and is pinned by:
 *   (1) NTT product == schoolbook negacyclic product,
 *   (2) decrypt(PBS(encrypt(m), LUT)) == LUT[m] for every m,
 *   (3) decrypted encrypted-inverse digits == the reference's plaintext QFloat
 *       output (tests/golden/inverse.json, produced by the reference itself).
 * The GPU library must match this oracle BIT FOR BIT on identical keys/inputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef int64_t i64;
typedef unsigned __int128 u128;

#define GOLD 0xFFFFFFFF00000001ULL /* 2^64 - 2^32 + 1 */
#define P49 562949952700417ULL       /* 2^49 - 720895, = 1 mod 2^16 */

/* The ciphertext modulus is a run-time choice between two NTT-friendly primes (set by ora_set_field, which every
 * entry point taking parameters calls): q_bits = 64 -> Goldilocks, q_bits = 49 -> P49 (the set whose GPU kernels
 * carry exact integers in f64).  "2^q_bits" plays the role of the torus size: gadget elements and message
 * scalings are powers of two below it. */
static u64 Q = GOLD;
static uint32_t QBITS = 64;
static u64 GEN = 7; /* generator of Z_q^* */
/* q_bits = 65 selects q = 2^64 EXACTLY, the torus Concrete computes on (SURVEY.md section 7 hard part 1, option A):
 * every Z_q operation below is then plain wrap-around u64 arithmetic, decomposition / message scaling use 64 bits
 * like the Goldilocks set, and the external product is the exact negacyclic product mod 2^64 (computed through
 * Goldilocks transforms of the two 32-bit halves of the key, see the torus section below; pinned against the
 * wrap-around schoolbook product). */
static int TORUS = 0;
#define Q_TORUS64 65u

/* Must mirror include/bmi_tfhe.h : bmi_params (same field order). */
typedef struct {
    uint32_t n;           /* small LWE dimension */
    uint32_t log_N;       /* log2 of the polynomial size */
    uint32_t k;           /* GLWE dimension */
    uint32_t bs_levels;   /* l  : bootstrap decomposition levels */
    uint32_t bs_base_log; /* Bg : bootstrap decomposition base log */
    uint32_t ks_levels;   /* keyswitch levels */
    uint32_t ks_base_log; /* keyswitch base log */
    uint32_t q_bits;      /* 64: q = 2^64 - 2^32 + 1 ; 49: q = 2^49 - 720895 */
    double lwe_noise;  /* std-dev (fraction of q) of keyswitch-key encryptions */
    double glwe_noise; /* std-dev (fraction of q) of GLWE / fresh big-key encryptions */
} ora_params;

int ora_set_field(uint32_t q_bits) {
    if (q_bits == 64) { Q = GOLD; QBITS = 64; GEN = 7; TORUS = 0; return 0; }
    if (q_bits == 49) { Q = P49; QBITS = 49; GEN = 5; TORUS = 0; return 0; }
    if (q_bits == Q_TORUS64) { Q = 0; QBITS = 64; GEN = 0; TORUS = 1; return 0; }
    return -1;
}
u64 ora_modulus(void) { return Q; } /* 0 stands for 2^64 */

/* ------------------------------------------------------------------ Z_q ---- */
static inline u64 addq(u64 a, u64 b) { if (TORUS) return a + b; u64 s = a + b; return (s < a || s >= Q) ? s - Q : s; }
static inline u64 subq(u64 a, u64 b) { if (TORUS) return a - b; return a >= b ? a - b : a + (Q - b); }
static inline u64 negq(u64 a) { if (TORUS) return (u64)0 - a; return a ? Q - a : 0; }
/* 128-bit product folded with 2^64 = 2^32 - 1 and 2^96 = -1 (mod q); checked against the plain `% Q` form in
 * ora_selftest_mulq (the baseline should not be handicapped by a 128-bit division per multiply). */
static inline u64 mulq(u64 a, u64 b) {
    if (TORUS) return a * b;
    if (Q != GOLD) return (u64)(((u128)a * b) % Q);
    u128 p = (u128)a * b;
    u64 lo = (u64)p, hi = (u64)(p >> 64), hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= 0xFFFFFFFFull;
    u64 t1 = hl * 0xFFFFFFFFull;
    u64 r = t0 + t1;
    if (r < t1) r += 0xFFFFFFFFull;
    return r >= Q ? r - Q : r;
}
int ora_selftest_mulq(u64 seed, uint32_t iters) {
    u64 x = seed | 1;
    for (uint32_t i = 0; i < iters; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        u64 a = x % Q;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        u64 b = (i & 7) == 0 ? Q - 1 - (x & 3) : x % Q;
        if (mulq(a, b) != (u64)(((u128)a * b) % Q)) return 0; /* call with the Goldilocks field selected */
    }
    return 1;
}
static u64 powq(u64 b, u64 e) { u64 r = 1; while (e) { if (e & 1) r = mulq(r, b); b = mulq(b, b); e >>= 1; } return r; }
static inline u64 from_i64(i64 v) { if (TORUS) return (u64)v; return v >= 0 ? (u64)v % Q : Q - ((u64)(-v) % Q); }
static inline i64 centered(u64 a) { if (TORUS) return (i64)a; return a > (Q >> 1) ? (i64)(a - Q) : (i64)a; } /* (-q/2, q/2] */

/* ------------------------------------------------------- deterministic RNG -- */
/* splitmix64 used as a counter-based generator: value = mix(stream_key + idx*GOLDEN). */
static inline u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline u64 stream_key(u64 seed, u64 stream) { return mix64(seed ^ (stream * 0xD6E8FEB86659FD93ULL)); }
static inline u64 rnd_u64(u64 key, u64 idx) { return mix64(key + (idx + 1) * 0x9E3779B97F4A7C15ULL); }
static inline u64 rnd_modq(u64 key, u64 idx) { u64 u = rnd_u64(key, idx); if (TORUS) return u; return QBITS == 64 ? (u >= Q ? u - Q : u) : u % Q; }
static inline u64 rnd_gauss(u64 key, u64 idx, double sigma) { /* element of Z_q */
    double u1 = ((double)((rnd_u64(key, 2 * idx) >> 11) + 1)) * (1.0 / 9007199254740992.0);
    double u2 = ((double)(rnd_u64(key, 2 * idx + 1) >> 11)) * (1.0 / 9007199254740992.0);
    double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
    return from_i64(llround(g * sigma * (QBITS == 64 ? 18446744073709551616.0 : (double)Q)));
}
enum { ST_SK_SMALL = 1, ST_SK_BIG = 2, ST_BSK_MASK = 3, ST_BSK_NOISE = 4, ST_KSK_MASK = 5, ST_KSK_NOISE = 6,
       ST_ENC_MASK = 7, ST_ENC_NOISE = 8, ST_BSK3_MASK = 9, ST_BSK3_NOISE = 10 };

/* ------------------------------------------------------------ negacyclic NTT */
typedef struct {
    uint32_t logN, N;
    u64 *psi_br;  /* psi^bitrev(i) */
    u64 *ipsi_br; /* psi^-bitrev(i) */
    u64 inv_N;
} ntt_tab;

static uint32_t bitrev(uint32_t x, uint32_t bits) { uint32_t r = 0; for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

static ntt_tab *ntt_make(uint32_t logN) {
    ntt_tab *t = (ntt_tab *)malloc(sizeof *t);
    t->logN = logN; t->N = 1u << logN;
    u64 psi = powq(GEN, (Q - 1) / (2ull * t->N)); /* GEN generates Z_q^* */
    u64 ipsi = powq(psi, Q - 2);
    t->psi_br = (u64 *)malloc(t->N * sizeof(u64));
    t->ipsi_br = (u64 *)malloc(t->N * sizeof(u64));
    for (uint32_t i = 0; i < t->N; i++) { t->psi_br[i] = powq(psi, bitrev(i, logN)); t->ipsi_br[i] = powq(ipsi, bitrev(i, logN)); }
    t->inv_N = powq(t->N, Q - 2);
    return t;
}
static void ntt_free(ntt_tab *t) { free(t->psi_br); free(t->ipsi_br); free(t); }

/* forward: natural order in, bit-reversed evaluation order out (merged psi twist) */
static void ntt_fwd(const ntt_tab *t, u64 *a) {
    uint32_t N = t->N;
    for (uint32_t m = 1, len = N >> 1; m < N; m <<= 1, len >>= 1)
        for (uint32_t i = 0; i < m; i++) {
            u64 w = t->psi_br[m + i];
            u64 *x = a + 2 * i * len, *y = x + len;
            for (uint32_t j = 0; j < len; j++) { u64 u = x[j], v = mulq(y[j], w); x[j] = addq(u, v); y[j] = subq(u, v); }
        }
}
static void ntt_inv(const ntt_tab *t, u64 *a) {
    uint32_t N = t->N;
    for (uint32_t m = N >> 1, len = 1; m >= 1; m >>= 1, len <<= 1)
        for (uint32_t i = 0; i < m; i++) {
            u64 w = t->ipsi_br[m + i];
            u64 *x = a + 2 * i * len, *y = x + len;
            for (uint32_t j = 0; j < len; j++) { u64 u = x[j], v = y[j]; x[j] = addq(u, v); y[j] = mulq(subq(u, v), w); }
        }
    for (uint32_t i = 0; i < N; i++) a[i] = mulq(a[i], t->inv_N);
}

/* ---- Goldilocks arithmetic with the modulus fixed (the torus path computes its exact products through it while the
 * selected "field" is the torus) */
static inline u64 g_add(u64 a, u64 b) { u64 s = a + b; return (s < a || s >= GOLD) ? s - GOLD : s; }
static inline u64 g_sub(u64 a, u64 b) { return a >= b ? a - b : a + (GOLD - b); }
static inline u64 g_mul(u64 a, u64 b) { return (u64)(((u128)a * b) % GOLD); }
static u64 g_pow(u64 b, u64 e) { u64 r = 1; while (e) { if (e & 1) r = g_mul(r, b); b = g_mul(b, b); e >>= 1; } return r; }
static inline u64 g_from_i64(i64 v) { return v >= 0 ? (u64)v : GOLD - (u64)(-v); }              /* |v| < 2^63 */
static inline i64 g_centered(u64 a) { return a > (GOLD >> 1) ? (i64)(a - GOLD) : (i64)a; }
static ntt_tab *g_ntt_make(uint32_t logN) {
    ntt_tab *t = (ntt_tab *)malloc(sizeof *t);
    t->logN = logN; t->N = 1u << logN;
    u64 psi = g_pow(7, (GOLD - 1) / (2ull * t->N)), ipsi = g_pow(psi, GOLD - 2);
    t->psi_br = (u64 *)malloc(t->N * sizeof(u64));
    t->ipsi_br = (u64 *)malloc(t->N * sizeof(u64));
    for (uint32_t i = 0; i < t->N; i++) { t->psi_br[i] = g_pow(psi, bitrev(i, logN)); t->ipsi_br[i] = g_pow(ipsi, bitrev(i, logN)); }
    t->inv_N = g_pow(t->N, GOLD - 2);
    return t;
}
static void g_ntt_fwd(const ntt_tab *t, u64 *a) {
    uint32_t N = t->N;
    for (uint32_t m = 1, len = N >> 1; m < N; m <<= 1, len >>= 1)
        for (uint32_t i = 0; i < m; i++) {
            u64 w = t->psi_br[m + i];
            u64 *x = a + 2 * i * len, *y = x + len;
            for (uint32_t j = 0; j < len; j++) { u64 u = x[j], v = g_mul(y[j], w); x[j] = g_add(u, v); y[j] = g_sub(u, v); }
        }
}
static void g_ntt_inv(const ntt_tab *t, u64 *a) {
    uint32_t N = t->N;
    for (uint32_t m = N >> 1, len = 1; m >= 1; m >>= 1, len <<= 1)
        for (uint32_t i = 0; i < m; i++) {
            u64 w = t->ipsi_br[m + i];
            u64 *x = a + 2 * i * len, *y = x + len;
            for (uint32_t j = 0; j < len; j++) { u64 u = x[j], v = y[j]; x[j] = g_add(u, v); y[j] = g_mul(g_sub(u, v), w); }
        }
    for (uint32_t i = 0; i < N; i++) a[i] = g_mul(a[i], t->inv_N);
}

/* c = a * b mod (X^N + 1, 2^64), wrap-around schoolbook: the definition the torus path is held to */
void ora_torus_negacyclic_schoolbook(uint32_t logN, const u64 *a, const u64 *b, u64 *c) {
    uint32_t N = 1u << logN;
    memset(c, 0, N * sizeof(u64));
    for (uint32_t i = 0; i < N; i++)
        for (uint32_t j = 0; j < N; j++) {
            u64 p = a[i] * b[j];
            uint32_t k = i + j;
            if (k < N) c[k] += p; else c[k - N] -= p;
        }
}
/* The same product for a SMALL signed first operand (|d_i| <= 2^bound_log, i.e. decomposition digits) through two
 * Goldilocks transforms of the 32-bit halves of b: each half-product is an integer of magnitude
 * < N * 2^bound_log * 2^32, which must stay below q/2 (checked), so its centred residue IS the integer. */
int ora_torus_negacyclic_split(uint32_t logN, const i64 *d, uint32_t bound_log, const u64 *b, u64 *c) {
    uint32_t N = 1u << logN;
    if (logN + bound_log + 32 >= 63) return -1;
    ntt_tab *t = g_ntt_make(logN);
    u64 *x = (u64 *)malloc(N * 8), *lo = (u64 *)malloc(N * 8), *hi = (u64 *)malloc(N * 8);
    for (uint32_t i = 0; i < N; i++) { x[i] = g_from_i64(d[i]); lo[i] = b[i] & 0xFFFFFFFFull; hi[i] = b[i] >> 32; }
    g_ntt_fwd(t, x); g_ntt_fwd(t, lo); g_ntt_fwd(t, hi);
    for (uint32_t i = 0; i < N; i++) { lo[i] = g_mul(lo[i], x[i]); hi[i] = g_mul(hi[i], x[i]); }
    g_ntt_inv(t, lo); g_ntt_inv(t, hi);
    for (uint32_t i = 0; i < N; i++) c[i] = (u64)g_centered(lo[i]) + ((u64)g_centered(hi[i]) << 32);
    free(x); free(lo); free(hi); ntt_free(t);
    return 0;
}

/* c = a * b mod (X^N + 1, q): schoolbook (slow, obviously correct) */
void ora_negacyclic_schoolbook(uint32_t logN, const u64 *a, const u64 *b, u64 *c) {
    uint32_t N = 1u << logN;
    memset(c, 0, N * sizeof(u64));
    for (uint32_t i = 0; i < N; i++)
        for (uint32_t j = 0; j < N; j++) {
            u64 p = mulq(a[i], b[j]);
            uint32_t k = i + j;
            if (k < N) c[k] = addq(c[k], p); else c[k - N] = subq(c[k - N], p);
        }
}
void ora_negacyclic_ntt(uint32_t logN, const u64 *a, const u64 *b, u64 *c) {
    uint32_t N = 1u << logN;
    ntt_tab *t = ntt_make(logN);
    u64 *x = (u64 *)malloc(N * 8), *y = (u64 *)malloc(N * 8);
    memcpy(x, a, N * 8); memcpy(y, b, N * 8);
    ntt_fwd(t, x); ntt_fwd(t, y);
    for (uint32_t i = 0; i < N; i++) c[i] = mulq(x[i], y[i]);
    ntt_inv(t, c);
    free(x); free(y); ntt_free(t);
}

/* --------------------------------------------------- signed decomposition --- */
/* Closest-representative signed digits of the centred lift of a in Z_q, keeping the
 * top levels*base_log bits of the 64-bit range; digit[0] is the MOST significant
 * (gadget element 2^(64 - base_log*(lev+1))).  Lower digits lie in [-B/2, B/2); the top
 * digit absorbs the last carry and lies in [-B/2, B/2], so sum_lev digit*gadget equals the
 * centred lift rounded to a multiple of 2^(64 - levels*base_log), exactly (no wrap). */
void ora_decompose(u64 a, uint32_t levels, uint32_t base_log, i64 *digits) {
    /* one rule for every modulus: round half up to the top levels*base_log bits, then balanced digits from the least
     * significant one, each step r <- floor(r / B + 1/2) (the signed decomposition of CGGI, Alg. 1 of the TFHE paper) */
    i64 c = centered(a);
    uint32_t shift = QBITS - levels * base_log;
    i64 r = (c >> shift) + ((c >> (shift - 1)) & 1); /* round half up, no 64-bit overflow */
    i64 B = (i64)1 << base_log, half = B >> 1;
    for (int lev = (int)levels - 1; lev >= 1; lev--) {
        i64 d = r & (B - 1);
        r >>= base_log;
        if (d >= half) { d -= B; r += 1; }
        digits[lev] = d;
    }
    digits[0] = r;
}

/* round(a * 2N / q) mod 2N, exact */
uint32_t ora_modswitch(u64 a, uint32_t log2N) {
    if (TORUS) return (uint32_t)(((a >> (63 - log2N)) + 1) >> 1) & ((1u << log2N) - 1); /* round(a 2N / 2^64), ties up */
    u128 t = ((u128)a << log2N) + (Q >> 1);
    return (uint32_t)(t / Q) & ((1u << log2N) - 1);
}

/* out = X^e * in  (mod X^N + 1), 0 <= e < 2N */
static void poly_rot(uint32_t N, const u64 *in, uint32_t e, u64 *out) {
    for (uint32_t j = 0; j < N; j++) {
        uint32_t pos = (j + e) & (2 * N - 1);
        if (pos < N) out[pos] = in[j]; else out[pos - N] = negq(in[j]);
    }
}

/* ------------------------------------------------------------------ keygen -- */
/* Layouts (all row-major):
 *   sk_small[n]            bits
 *   sk_big[k*N]            bits, polynomial j at [j*N, (j+1)*N)
 *   bsk[n][(k+1)*l][(k+1)][N]   standard (coefficient) domain GGSW rows;
 *                               row r = comp*l + lev carries s_i * 2^(64-Bg*(lev+1)) on component comp
 *   ksk[k*N][l_ks][n+1]    LWE_small( sk_big[j] * 2^(64-Bks*(lev+1)) ), body last
 *   bsk3[ceil(n/2)][3][(k+1)*l][(k+1)][N]   the UNROLLED bootstrap key (two LWE coefficients per blind-rotation step,
 *                               ora_keygen_bsk_unrolled): GGSW(s s'), GGSW(s (1 - s')), GGSW((1 - s) s') of the pair
 *                               (s, s') = (s_2i, s_2i+1); an odd n is completed by s_n = 0
 */
/* count GGSW encryptions under the GLWE key sk_big, standard (coefficient) domain, of the bits msg[0..count): row ir =
 * (g * (k+1) l + comp * l + lev) draws its masks from indices (ir (k+1) + j) N + x of stream km and its noise from indices
 * ir N + x of stream ke, and carries msg[g] * 2^(QBITS - Bg (lev+1)) on component comp. */
static void ggsw_rows(const ora_params *P, u64 km, u64 ke, const u64 *sk_big, const u64 *msg, uint32_t count, u64 *out) {
    uint32_t N = 1u << P->log_N, k = P->k, l = P->bs_levels;
    ntt_tab *t = TORUS ? NULL : ntt_make(P->log_N);
    u64 *S = (u64 *)malloc((size_t)k * N * 8); /* NTT of the GLWE secret polynomials */
    memcpy(S, sk_big, (size_t)k * N * 8);
    for (uint32_t j = 0; j < k && !TORUS; j++) ntt_fwd(t, S + (size_t)j * N);
    uint32_t rows = (k + 1) * l;
#pragma omp parallel
    {
        u64 *tmp = (u64 *)malloc(N * 8), *acc = (u64 *)malloc(N * 8);
#pragma omp for schedule(static)
        for (uint32_t ir = 0; ir < count * rows; ir++) {
            uint32_t i = ir / rows, r = ir % rows, comp = r / l, lev = r % l;
            u64 *row = out + (size_t)ir * (k + 1) * N;
            memset(acc, 0, N * 8);
            for (uint32_t j = 0; j < k; j++) {
                u64 *A = row + (size_t)j * N;
                for (uint32_t x = 0; x < N; x++) A[x] = rnd_modq(km, ((u64)ir * (k + 1) + j) * N + x);
                if (TORUS) { /* A * S with S binary: shifted adds, wrap-around (no transform exists mod 2^64) */
                    for (uint32_t sft = 0; sft < N; sft++) {
                        if (!sk_big[(size_t)j * N + sft]) continue;
                        for (uint32_t x = 0; x + sft < N; x++) acc[x + sft] += A[x];
                        for (uint32_t x = N - sft; x < N; x++) acc[x + sft - N] -= A[x];
                    }
                    continue;
                }
                memcpy(tmp, A, N * 8);
                ntt_fwd(t, tmp);
                for (uint32_t x = 0; x < N; x++) acc[x] = addq(acc[x], mulq(tmp[x], S[(size_t)j * N + x]));
            }
            if (!TORUS) ntt_inv(t, acc);
            u64 *B = row + (size_t)k * N;
            for (uint32_t x = 0; x < N; x++) B[x] = addq(acc[x], rnd_gauss(ke, (u64)ir * N + x, P->glwe_noise));
            if (msg[i]) {
                u64 g = (u64)1 << (QBITS - P->bs_base_log * (lev + 1));
                row[(size_t)comp * N] = addq(row[(size_t)comp * N], g);
            }
        }
        free(tmp); free(acc);
    }
    free(S); if (t) ntt_free(t);
}

void ora_keygen(const ora_params *P, u64 seed, u64 *sk_small, u64 *sk_big, u64 *bsk, u64 *ksk) {
    ora_set_field(P->q_bits);
    uint32_t n = P->n, N = 1u << P->log_N, k = P->k, lk = P->ks_levels;
    u64 k1 = stream_key(seed, ST_SK_SMALL), k2 = stream_key(seed, ST_SK_BIG);
    for (uint32_t i = 0; i < n; i++) sk_small[i] = rnd_u64(k1, i) & 1;
    for (uint32_t i = 0; i < k * N; i++) sk_big[i] = rnd_u64(k2, i) & 1;

    ggsw_rows(P, stream_key(seed, ST_BSK_MASK), stream_key(seed, ST_BSK_NOISE), sk_big, sk_small, n, bsk);

    u64 kkm = stream_key(seed, ST_KSK_MASK), kke = stream_key(seed, ST_KSK_NOISE);
#pragma omp parallel for schedule(static)
    for (uint32_t jr = 0; jr < k * N * lk; jr++) {
        uint32_t j = jr / lk, lev = jr % lk;
        u64 *row = ksk + (size_t)jr * (n + 1);
        u64 b = rnd_gauss(kke, jr, P->lwe_noise);
        for (uint32_t c = 0; c < n; c++) {
            row[c] = rnd_modq(kkm, (u64)jr * (n + 1) + c);
            if (sk_small[c]) b = addq(b, row[c]);
        }
        if (sk_big[j]) b = addq(b, (u64)1 << (QBITS - P->ks_base_log * (lev + 1)));
        row[n] = b;
    }
}

/* Unrolled bootstrap key (Zhou et al. 2018 / Bourse et al. 2018, unrolling factor 2):
 *   X^(a s + a' s') - 1 = s s' (X^(a+a') - 1) + s (1 - s') (X^a - 1) + (1 - s) s' (X^a' - 1)     for bits s, s',
 * so one step ACC <- ACC + sum_j (X^(c_j) - 1) * (K_j [.] ACC), c = (a + a', a, a'), absorbs two LWE coefficients with ONE
 * decomposition and one set of forward transforms.  Streams ST_BSK3_MASK / ST_BSK3_NOISE, rows numbered (3 i + j) rows + r. */
void ora_keygen_bsk_unrolled(const ora_params *P, u64 seed, const u64 *sk_small, const u64 *sk_big, u64 *bsk3) {
    ora_set_field(P->q_bits);
    uint32_t n = P->n, pairs = (n + 1) / 2;
    u64 *msg = (u64 *)malloc((size_t)pairs * 3 * 8);
    for (uint32_t i = 0; i < pairs; i++) {
        u64 s = sk_small[2 * i], s2 = 2 * i + 1 < n ? sk_small[2 * i + 1] : 0;
        msg[3 * i] = s & s2; msg[3 * i + 1] = s & (s2 ^ 1); msg[3 * i + 2] = (s ^ 1) & s2;
    }
    ggsw_rows(P, stream_key(seed, ST_BSK3_MASK), stream_key(seed, ST_BSK3_NOISE), sk_big, msg, pairs * 3, bsk3);
    free(msg);
}

/* 2^64 torus: a bootstrap key STORED AT `precision` BITS (64 = exact; the library's torus sets keep 48, or 42): every key
 * word, read as a signed integer, is rounded half up to a multiple of 2^(64 - precision) - the same rounding rule as the
 * decomposition's.  The rounded key is the key: it is what the library exports and what every consumer bootstraps with (the
 * rounding only adds a uniform error of variance 2^(2 (64 - precision)) / 12 to valid GLWE samples). */
void ora_round_key(u64 *key, size_t words, uint32_t precision) {
    if (precision >= 64) return;
    uint32_t drop = 64 - precision;
    for (size_t i = 0; i < words; i++) {
        i64 c = (i64)key[i];
        i64 r = (c >> drop) + ((c >> (drop - 1)) & 1); /* round half up, no 64-bit overflow */
        key[i] = (u64)r << drop;
    }
}

/* LWE encryption of torus values under a binary key of dimension dim; ciphertext i
 * uses mask indices [(first+i)*(dim+1), ...) of stream ST_ENC_MASK. */
void ora_lwe_encrypt(const u64 *key, uint32_t dim, double noise, u64 seed, u64 first, const u64 *torus, uint32_t count, u64 *out) {
    u64 km = stream_key(seed, ST_ENC_MASK), ke = stream_key(seed, ST_ENC_NOISE);
    for (uint32_t i = 0; i < count; i++) {
        u64 *ct = out + (size_t)i * (dim + 1);
        u64 b = addq(TORUS ? torus[i] : torus[i] % Q, rnd_gauss(ke, first + i, noise));
        for (uint32_t c = 0; c < dim; c++) {
            ct[c] = rnd_modq(km, (first + i) * (u64)(dim + 1) + c);
            if (key[c]) b = addq(b, ct[c]);
        }
        ct[dim] = b;
    }
}
void ora_lwe_phase(const u64 *key, uint32_t dim, const u64 *cts, uint32_t count, u64 *phase) {
    for (uint32_t i = 0; i < count; i++) {
        const u64 *ct = cts + (size_t)i * (dim + 1);
        u64 p = ct[dim];
        for (uint32_t c = 0; c < dim; c++) if (key[c]) p = subq(p, ct[c]);
        phase[i] = p;
    }
}
/* signed message = round(centred(phase) / 2^delta_log) */
void ora_decode(const u64 *phase, uint32_t count, uint32_t delta_log, i64 *msg) {
    for (uint32_t i = 0; i < count; i++) {
        i64 c = centered(phase[i]);
        msg[i] = (c + ((i64)1 << (delta_log - 1))) >> delta_log;
    }
}

/* ------------------------------------------------------------- test vector -- */
/* Signed message space [-2^(p-1), 2^(p-1)) spread over the whole negacyclic circle:
 * box width w = N / 2^p positions, boxes centred on m*w (half-box pre-rotation folded in).
 * table[m + 2^(p-1)] = f(m) as a signed integer; output torus value = f(m) * 2^out_delta_log. */
void ora_make_test_vector(uint32_t log_N, uint32_t p, const i64 *table, uint32_t out_delta_log, u64 *tv) {
    uint32_t N = 1u << log_N, w = N >> p, half = w >> 1, M = 1u << p, Mh = M >> 1;
    for (uint32_t j = 0; j < N; j++) {
        uint32_t idx = (j + half) / w; /* 0 .. M */
        i64 f; int neg;
        if (idx < Mh) { f = table[idx + Mh]; neg = 0; }
        else if (idx < M) { f = table[idx - Mh]; neg = 1; } /* m = idx - M in [-M/2, -1] */
        else { f = table[Mh]; neg = 1; }                     /* m = 0 approached from below */
        u64 v = from_i64(f);
        v = mulq(v, (u64)1 << out_delta_log);
        tv[j] = neg ? negq(v) : v;
    }
}

/* ------------------------------------------------------------ the PBS path -- */
typedef struct {
    ora_params P;
    ntt_tab *t;
    u64 *bsk_ntt; /* [n][(k+1)l][(k+1)][N] forward-NTT of every GGSW row polynomial */
    u64 *bsk3_ntt; /* unrolled key (ora_ctx_set_bsk_unrolled), same per-polynomial form, or NULL */
    const u64 *ksk;
} ora_ctx;

/* transform-domain copy of `polys` key polynomials (what the external product multiplies with) */
static u64 *key_to_ntt(const ora_ctx *c, const u64 *key, size_t polys) {
    uint32_t N = 1u << c->P.log_N;
    u64 *out;
    if (TORUS) {
        /* torus: Goldilocks transforms of the low and the high 32-bit half of every key polynomial: [poly][2][N] */
        out = (u64 *)malloc(polys * 2 * N * 8);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < polys; i++) {
            u64 *lo = out + i * 2 * N, *hi = lo + N;
            for (uint32_t x = 0; x < N; x++) { lo[x] = key[i * N + x] & 0xFFFFFFFFull; hi[x] = key[i * N + x] >> 32; }
            g_ntt_fwd(c->t, lo); g_ntt_fwd(c->t, hi);
        }
        return out;
    }
    out = (u64 *)malloc(polys * N * 8);
    memcpy(out, key, polys * N * 8);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < polys; i++) ntt_fwd(c->t, out + i * N);
    return out;
}
ora_ctx *ora_ctx_create(const ora_params *P, const u64 *bsk, const u64 *ksk) {
    ora_set_field(P->q_bits);
    ora_ctx *c = (ora_ctx *)malloc(sizeof *c);
    c->P = *P; c->ksk = ksk; c->bsk3_ntt = NULL;
    c->t = TORUS ? g_ntt_make(P->log_N) : ntt_make(P->log_N);
    c->bsk_ntt = key_to_ntt(c, bsk, (size_t)P->n * (P->k + 1) * P->bs_levels * (P->k + 1));
    return c;
}
/* attaches the unrolled bootstrap key bsk3[ceil(n/2)][3][(k+1)l][(k+1)][N] (ora_keygen_bsk_unrolled or an exported one) */
void ora_ctx_set_bsk_unrolled(ora_ctx *c, const u64 *bsk3) {
    free(c->bsk3_ntt);
    c->bsk3_ntt = key_to_ntt(c, bsk3, (size_t)((c->P.n + 1) / 2) * 3 * (c->P.k + 1) * c->P.bs_levels * (c->P.k + 1));
}
void ora_ctx_destroy(ora_ctx *c) { ntt_free(c->t); free(c->bsk_ntt); free(c->bsk3_ntt); free(c); }

/* big-key LWE (k*N+1 words) -> small-key LWE (n+1 words) */
void ora_keyswitch(const ora_ctx *c, const u64 *in, u64 *out) {
    const ora_params *P = &c->P;
    uint32_t n = P->n, kN = P->k << P->log_N, lk = P->ks_levels;
    i64 dig[64];
    /* 128-bit signed accumulators keep the sum exact; reduce once at the end */
    __int128 *acc = (__int128 *)calloc(n + 1, sizeof(__int128));
    for (uint32_t j = 0; j < kN; j++) {
        ora_decompose(in[j], lk, P->ks_base_log, dig);
        for (uint32_t lev = 0; lev < lk; lev++) {
            if (!dig[lev]) continue;
            const u64 *row = c->ksk + ((size_t)j * lk + lev) * (n + 1);
            for (uint32_t x = 0; x <= n; x++) acc[x] += (__int128)dig[lev] * (__int128)row[x];
        }
    }
    for (uint32_t x = 0; x <= n; x++) {
        __int128 a = -acc[x];
        if (TORUS) { out[x] = (u64)a; continue; }
        __int128 r = a % (__int128)Q; if (r < 0) r += Q;
        out[x] = (u64)r;
    }
    out[n] = addq(out[n], in[kN]);
    free(acc);
}

/* res[(k+1)][N] = sum over the (k+1) l rows of dec[row] * G[row][.]  (negacyclic, exact mod q): the external product of
 * one GGSW ciphertext (transform-domain polynomials at g) with the digit polynomials dec (field elements / two's
 * complement words; overwritten by their transforms). */
static void external_product(const ora_ctx *c, const u64 *g, u64 *dec, u64 *res) {
    const ora_params *P = &c->P;
    uint32_t N = 1u << P->log_N, k = P->k, rows = (k + 1) * P->bs_levels;
    if (TORUS) {
        /* exact product mod 2^64: digits (|d| <= 2^(Bg-1)) as Goldilocks elements against both key halves; every
         * half-sum is an integer below rows * N * 2^(Bg-1) * 2^32 < q/2, so its centred residue is the integer */
        u64 *res_hi = (u64 *)calloc((size_t)(k + 1) * N, 8);
        for (uint32_t r = 0; r < rows; r++) {
            u64 *d = dec + (size_t)r * N;
            for (uint32_t x = 0; x < N; x++) d[x] = g_from_i64((i64)d[x]);
            g_ntt_fwd(c->t, d);
        }
        memset(res, 0, (size_t)(k + 1) * N * 8);
        for (uint32_t r = 0; r < rows; r++)
            for (uint32_t oc = 0; oc <= k; oc++) {
                const u64 *blo = g + ((size_t)r * (k + 1) + oc) * 2 * N, *bhi = blo + N, *d = dec + (size_t)r * N;
                u64 *o = res + (size_t)oc * N, *oh = res_hi + (size_t)oc * N;
                for (uint32_t x = 0; x < N; x++) { o[x] = g_add(o[x], g_mul(d[x], blo[x])); oh[x] = g_add(oh[x], g_mul(d[x], bhi[x])); }
            }
        for (uint32_t oc = 0; oc <= k; oc++) {
            g_ntt_inv(c->t, res + (size_t)oc * N); g_ntt_inv(c->t, res_hi + (size_t)oc * N);
            u64 *o = res + (size_t)oc * N, *oh = res_hi + (size_t)oc * N;
            for (uint32_t x = 0; x < N; x++) o[x] = (u64)g_centered(o[x]) + ((u64)g_centered(oh[x]) << 32);
        }
        free(res_hi);
        return;
    }
    for (uint32_t r = 0; r < rows; r++) ntt_fwd(c->t, dec + (size_t)r * N);
    memset(res, 0, (size_t)(k + 1) * N * 8);
    for (uint32_t r = 0; r < rows; r++)
        for (uint32_t oc = 0; oc <= k; oc++) {
            const u64 *b = g + ((size_t)r * (k + 1) + oc) * N, *d = dec + (size_t)r * N;
            u64 *o = res + (size_t)oc * N;
            for (uint32_t x = 0; x < N; x++) o[x] = addq(o[x], mulq(d[x], b[x]));
        }
    for (uint32_t oc = 0; oc <= k; oc++) ntt_inv(c->t, res + (size_t)oc * N);
}

/* small-key LWE -> big-key LWE of tv[phase]: modulus switch, blind rotation, sample extraction.
 * tv has N coefficients (body polynomial; mask polynomials start at zero). */
void ora_blind_rotate_extract(const ora_ctx *c, const u64 *lwe, const u64 *tv, u64 *out) {
    const ora_params *P = &c->P;
    uint32_t n = P->n, N = 1u << P->log_N, k = P->k, l = P->bs_levels, rows = (k + 1) * l, log2N = P->log_N + 1;
    u64 *acc = (u64 *)calloc((size_t)(k + 1) * N, 8);
    u64 *diff = (u64 *)malloc((size_t)(k + 1) * N * 8);
    u64 *dec = (u64 *)malloc((size_t)rows * N * 8);
    u64 *res = (u64 *)malloc((size_t)(k + 1) * N * 8);
    i64 dig[64];
    uint32_t bt = ora_modswitch(lwe[n], log2N);
    poly_rot(N, tv, (2 * N - bt) & (2 * N - 1), acc + (size_t)k * N);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t at = ora_modswitch(lwe[i], log2N);
        if (at == 0) continue; /* X^0 * ACC - ACC = 0: the external product adds exactly zero */
        for (uint32_t comp = 0; comp <= k; comp++) {
            u64 *a = acc + (size_t)comp * N, *d = diff + (size_t)comp * N;
            poly_rot(N, a, at, d);
            for (uint32_t x = 0; x < N; x++) d[x] = subq(d[x], a[x]);
            for (uint32_t x = 0; x < N; x++) {
                ora_decompose(d[x], l, P->bs_base_log, dig);
                for (uint32_t lev = 0; lev < l; lev++) dec[((size_t)comp * l + lev) * N + x] = from_i64(dig[lev]);
            }
        }
        external_product(c, c->bsk_ntt + (size_t)i * rows * (k + 1) * N * (TORUS ? 2 : 1), dec, res);
        for (uint32_t x = 0; x < (k + 1) * N; x++) acc[x] = addq(acc[x], res[x]);
    }
    /* sample extraction of coefficient 0 */
    for (uint32_t j = 0; j < k; j++) {
        const u64 *A = acc + (size_t)j * N;
        out[(size_t)j * N] = A[0];
        for (uint32_t x = 1; x < N; x++) out[(size_t)j * N + x] = negq(A[N - x]);
    }
    out[(size_t)k * N] = acc[(size_t)k * N];
    free(acc); free(diff); free(dec); free(res);
}

/* The same with the UNROLLED key (ora_ctx_set_bsk_unrolled): step i absorbs the LWE coefficients (a, a') = (a_2i, a_2i+1)
 *   ACC <- ACC + sum_{j<3} (X^(c_j) - 1) * (K3[i][j] [.] ACC),    c = (a + a' mod 2N, a, a'),
 * with ONE decomposition of ACC itself per step (the rotation acts on the products, here in the coefficient domain).
 * Returns -1 when no unrolled key is attached. */
int ora_blind_rotate_extract_unrolled(const ora_ctx *c, const u64 *lwe, const u64 *tv, u64 *out) {
    const ora_params *P = &c->P;
    uint32_t n = P->n, N = 1u << P->log_N, k = P->k, l = P->bs_levels, rows = (k + 1) * l, log2N = P->log_N + 1;
    if (!c->bsk3_ntt) return -1;
    u64 *acc = (u64 *)calloc((size_t)(k + 1) * N, 8);
    u64 *dec0 = (u64 *)malloc((size_t)rows * N * 8), *dec = (u64 *)malloc((size_t)rows * N * 8);
    u64 *res = (u64 *)malloc((size_t)(k + 1) * N * 8), *rot = (u64 *)malloc(N * 8);
    i64 dig[64];
    uint32_t bt = ora_modswitch(lwe[n], log2N);
    poly_rot(N, tv, (2 * N - bt) & (2 * N - 1), acc + (size_t)k * N);
    for (uint32_t i = 0; i < (n + 1) / 2; i++) {
        uint32_t a1 = ora_modswitch(lwe[2 * i], log2N), a2 = 2 * i + 1 < n ? ora_modswitch(lwe[2 * i + 1], log2N) : 0;
        uint32_t cj[3] = {(a1 + a2) & (2 * N - 1), a1, a2};
        if ((a1 | a2) == 0) continue; /* every factor X^0 - 1 vanishes */
        for (uint32_t comp = 0; comp <= k; comp++)
            for (uint32_t x = 0; x < N; x++) {
                ora_decompose(acc[(size_t)comp * N + x], l, P->bs_base_log, dig);
                for (uint32_t lev = 0; lev < l; lev++) dec0[((size_t)comp * l + lev) * N + x] = from_i64(dig[lev]);
            }
        for (uint32_t j = 0; j < 3; j++) {
            if (cj[j] == 0) continue;
            memcpy(dec, dec0, (size_t)rows * N * 8);   /* external_product overwrites its digit operand */
            external_product(c, c->bsk3_ntt + ((size_t)i * 3 + j) * rows * (k + 1) * N * (TORUS ? 2 : 1), dec, res);
            for (uint32_t oc = 0; oc <= k; oc++) {
                u64 *a = acc + (size_t)oc * N, *r = res + (size_t)oc * N;
                poly_rot(N, r, cj[j], rot);
                for (uint32_t x = 0; x < N; x++) a[x] = addq(a[x], subq(rot[x], r[x]));
            }
        }
    }
    for (uint32_t j = 0; j < k; j++) {
        const u64 *A = acc + (size_t)j * N;
        out[(size_t)j * N] = A[0];
        for (uint32_t x = 1; x < N; x++) out[(size_t)j * N + x] = negq(A[N - x]);
    }
    out[(size_t)k * N] = acc[(size_t)k * N];
    free(acc); free(dec0); free(dec); free(res); free(rot);
    return 0;
}

/* Concrete-order PBS on a batch: keyswitch -> blind rotate -> extract.
 * tvs: [n_tv][N]; tv_ids[count] selects the test vector per ciphertext.
 * ks_out (optional, may be NULL): the intermediate small-key ciphertexts. */
void ora_pbs_batch(const ora_ctx *c, const u64 *in, const u64 *tvs, const uint32_t *tv_ids, uint32_t count, u64 *out, u64 *ks_out) {
    uint32_t n = c->P.n, N = 1u << c->P.log_N, big = c->P.k * N + 1;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t i = 0; i < count; i++) {
        u64 *small = (u64 *)malloc((n + 1) * 8);
        ora_keyswitch(c, in + (size_t)i * big, small);
        if (ks_out) memcpy(ks_out + (size_t)i * (n + 1), small, (n + 1) * 8);
        ora_blind_rotate_extract(c, small, tvs + (size_t)tv_ids[i] * N, out + (size_t)i * big);
        free(small);
    }
}
void ora_keyswitch_batch(const ora_ctx *c, const u64 *in, uint32_t count, u64 *out) {
    uint32_t n = c->P.n, big = (c->P.k << c->P.log_N) + 1;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t i = 0; i < count; i++) ora_keyswitch(c, in + (size_t)i * big, out + (size_t)i * (n + 1));
}
void ora_blind_rotate_batch(const ora_ctx *c, const u64 *in, const u64 *tvs, const uint32_t *tv_ids, uint32_t count, u64 *out) {
    uint32_t n = c->P.n, N = 1u << c->P.log_N, big = c->P.k * N + 1;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t i = 0; i < count; i++)
        ora_blind_rotate_extract(c, in + (size_t)i * (n + 1), tvs + (size_t)tv_ids[i] * N, out + (size_t)i * big);
}

/* unrolled-key twins of the two batch entry points above */
int ora_pbs_batch_unrolled(const ora_ctx *c, const u64 *in, const u64 *tvs, const uint32_t *tv_ids, uint32_t count, u64 *out, u64 *ks_out) {
    uint32_t n = c->P.n, N = 1u << c->P.log_N, big = c->P.k * N + 1;
    if (!c->bsk3_ntt) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t i = 0; i < count; i++) {
        u64 *small = (u64 *)malloc((n + 1) * 8);
        ora_keyswitch(c, in + (size_t)i * big, small);
        if (ks_out) memcpy(ks_out + (size_t)i * (n + 1), small, (n + 1) * 8);
        ora_blind_rotate_extract_unrolled(c, small, tvs + (size_t)tv_ids[i] * N, out + (size_t)i * big);
        free(small);
    }
    return 0;
}
int ora_blind_rotate_batch_unrolled(const ora_ctx *c, const u64 *in, const u64 *tvs, const uint32_t *tv_ids, uint32_t count, u64 *out) {
    uint32_t n = c->P.n, N = 1u << c->P.log_N, big = c->P.k * N + 1;
    if (!c->bsk3_ntt) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t i = 0; i < count; i++)
        ora_blind_rotate_extract_unrolled(c, in + (size_t)i * (n + 1), tvs + (size_t)tv_ids[i] * N, out + (size_t)i * big);
    return 0;
}

/* out[i] = const_i + sum_j coef[j] * in[idx[j]]  over CSR rows (leveled linear ops) */
void ora_lincomb(uint32_t width, const u64 *in, const uint32_t *row_ptr, const uint32_t *idx, const i64 *coef,
                 const u64 *const_body, uint32_t count, u64 *out) {
    for (uint32_t i = 0; i < count; i++) {
        u64 *o = out + (size_t)i * width;
        memset(o, 0, (size_t)width * 8);
        for (uint32_t e = row_ptr[i]; e < row_ptr[i + 1]; e++) {
            u64 cq = from_i64(coef[e]);
            const u64 *s = in + (size_t)idx[e] * width;
            for (uint32_t x = 0; x < width; x++) o[x] = addq(o[x], mulq(cq, s[x]));
        }
        o[width - 1] = addq(o[width - 1], TORUS ? const_body[i] : const_body[i] % Q);
    }
}

/* size of the OpenMP team of the batch entry points (bench.py: the host cores the process may actually use) */
void ora_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int ora_num_threads(void) {
    int n = 1;
#ifdef _OPENMP
#pragma omp parallel
    {
#pragma omp master
        n = omp_get_num_threads();
    }
#endif
    return n;
}

/* ====================================================================== fast path (49-bit field) ====
 * The CPU BASELINE of bench.py: the same PBS, same results bit for bit (ora_fast_pbs_batch == ora_pbs_batch, held by
 * tests/test_oracle_tfhe.py), written the way a competent CPU implementation is: no 128-bit divisions, no allocation per
 * call, vectorisable loops.  Arithmetic: exact integers mod p = 2^49 - 720895 carried in doubles (products through one
 * FMA pair, reductions x - p rint(x / p): the representation the GPU kernels use, here in plain radix-2 transforms
 * that gcc vectorises), keyswitch as exact f64 FMAs on the two 25-bit halves of the key words.
 * The generic path above stays the definition; this one is only ever compared with it. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define F_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))   /* picked at load time by the host CPU */
#else
#define F_CLONES
#endif
#define FP 562949952700417.0
#define FPINV (1.0 / 562949952700417.0)
static inline double f_red(double x) { return __builtin_fma(-__builtin_rint(x * FPINV), FP, x); }
static inline double f_mul(double a, double b) { /* |a| < 2^53, |b| <= p/2: exact product, result |r| <= 0.5p + |a|/8 */
    double h = a * b, l = __builtin_fma(a, b, -h);
    return __builtin_fma(-__builtin_rint(h * FPINV), FP, h) + l;
}
static inline double f_center(u64 v) { return v > (P49 >> 1) ? -(double)(P49 - v) : (double)v; }
static inline u64 f_canon(double x) { double r = f_red(x); if (r < 0) r += FP; return (u64)r; }

typedef struct {
    ora_params P;
    uint32_t N;
    double *psi_br, *ipsi_br; /* centred twiddles, bit-reversed order (same tables as ntt_tab) */
    double inv_N;
    double *bsk;              /* [n][rows][k+1][N] transform domain, centred */
    double *ksk_lo, *ksk_hi;  /* [kN*lk][n+1]: word = hi * 2^25 + lo, both exact in f64 */
    /* N = 1024 as 32 x 32 ("four-step"): every butterfly loop runs over 32 contiguous doubles.  t1: 32-point negacyclic
     * tree (root psi^32) over the row index, tw: psi^(c (2 br5(rho) + 1)), t2: 32-point cyclic tree (root psi^64) over
     * the column index after a transpose; *_i the inverses (1/N folded into twi).  NULL for other N (radix-2 path). */
    double *t1, *t1i, *t2, *t2i, *tw, *twi;
} ora_fctx;

/* butterflies between ROWS of an R x C matrix (Cooley-Tukey, natural in / tree order out), inner loop over the C columns */
F_CLONES static void f_rows_fwd(double *A, uint32_t R, uint32_t C, const double *tw) {
    for (uint32_t m = 1, len = R >> 1; m < R; m <<= 1, len >>= 1)
        for (uint32_t i = 0; i < m; i++) {
            const double w = tw[m + i];
            for (uint32_t j = 0; j < len; j++) {
                double *x = A + (size_t)(2 * i * len + j) * C, *y = x + (size_t)len * C;
                for (uint32_t cc = 0; cc < C; cc++) { double u = x[cc], v = f_mul(y[cc], w); x[cc] = u + v; y[cc] = u - v; }
            }
        }
}
F_CLONES static void f_rows_inv(double *A, uint32_t R, uint32_t C, const double *itw) {
    for (uint32_t m = R >> 1, len = 1; m >= 1; m >>= 1, len <<= 1)
        for (uint32_t i = 0; i < m; i++) {
            const double w = itw[m + i];
            for (uint32_t j = 0; j < len; j++) {
                double *x = A + (size_t)(2 * i * len + j) * C, *y = x + (size_t)len * C;
                for (uint32_t cc = 0; cc < C; cc++) { double u = x[cc], v = y[cc]; x[cc] = f_red(u + v); y[cc] = f_mul(u - v, w); }
            }
        }
}
F_CLONES static void f_transpose32(const double *a, double *b) {
    for (uint32_t r = 0; r < 32; r++)
        for (uint32_t cc = 0; cc < 32; cc++) b[cc * 32 + r] = a[r * 32 + cc];
}
/* N = 1024: a[32 r + c] -> transform in (column position, row position) order; |out| <= 8.4 p (lazy), callers reduce */
F_CLONES static void f_ntt1024_fwd(const ora_fctx *c, double *a, double *tmp) {
    f_rows_fwd(a, 32, 32, c->t1);
    for (uint32_t x = 0; x < 1024; x++) a[x] = f_mul(a[x], c->tw[x]);
    f_transpose32(a, tmp);
    f_rows_fwd(tmp, 32, 32, c->t2);
    memcpy(a, tmp, 1024 * 8);
}
F_CLONES static void f_ntt1024_inv(const ora_fctx *c, double *a, double *tmp) { /* input |.| <= 0.51 p */
    f_rows_inv(a, 32, 32, c->t2i);
    f_transpose32(a, tmp);
    for (uint32_t x = 0; x < 1024; x++) tmp[x] = f_mul(tmp[x], c->twi[x]);
    f_rows_inv(tmp, 32, 32, c->t1i);
    memcpy(a, tmp, 1024 * 8);
}

F_CLONES static void f_ntt_fwd(const ora_fctx *c, double *a) {
    if (c->t1) { double tmp[1024]; f_ntt1024_fwd(c, a, tmp); return; }
    uint32_t N = c->N, stage = 0;
    for (uint32_t m = 1, len = N >> 1; m < N; m <<= 1, len >>= 1, stage++) {
        for (uint32_t i = 0; i < m; i++) {
            double w = c->psi_br[m + i];
            double *x = a + 2 * i * len, *y = x + len;
            for (uint32_t j = 0; j < len; j++) { double u = x[j], v = f_mul(y[j], w); x[j] = u + v; y[j] = u - v; }
        }
        if (stage == 4) for (uint32_t j = 0; j < N; j++) a[j] = f_red(a[j]); /* lazy sums: 0.5p + 5 x 1.4p at most before */
    }
}
F_CLONES static void f_ntt_inv(const ora_fctx *c, double *a) { /* input |.| <= 0.51 p */
    if (c->t1) { double tmp[1024]; f_ntt1024_inv(c, a, tmp); return; }
    uint32_t N = c->N;
    for (uint32_t m = N >> 1, len = 1; m >= 1; m >>= 1, len <<= 1)
        for (uint32_t i = 0; i < m; i++) {
            double w = c->ipsi_br[m + i];
            double *x = a + 2 * i * len, *y = x + len;
            for (uint32_t j = 0; j < len; j++) { double u = x[j], v = y[j]; x[j] = f_red(u + v); y[j] = f_mul(u - v, w); }
        }
    for (uint32_t i = 0; i < N; i++) a[i] = f_mul(a[i], c->inv_N);
}

/* transform tables of the 49-bit field (no keys); the caller has selected that field */
static ora_fctx *f_tables_create(const ora_params *P) {
    ora_fctx *c = (ora_fctx *)calloc(1, sizeof *c);
    c->P = *P; c->N = 1u << P->log_N;
    uint32_t N = c->N;
    ntt_tab *t = ntt_make(P->log_N);
    c->psi_br = (double *)malloc(N * 8); c->ipsi_br = (double *)malloc(N * 8);
    for (uint32_t i = 0; i < N; i++) { c->psi_br[i] = f_center(t->psi_br[i]); c->ipsi_br[i] = f_center(t->ipsi_br[i]); }
    c->inv_N = f_center(t->inv_N);
    if (P->log_N == 10) {
        /* Cooley-Tukey tree over x^R - g^e: node idx (heap order) splits x^(2 len) - g^e into x^len -+ g^(e/2); its
         * twiddle is g^(e/2), its children carry e/2 and e/2 + ord/2.  Negacyclic 32 with g = psi^32 (order 64): e_root =
         * 32; cyclic 32 with g = psi^64 (order 32): e_root = 0. */
        const u64 psi = powq(GEN, (Q - 1) / 2048), ipsi = powq(psi, Q - 2);
        double **fw[2] = {&c->t1, &c->t2}, **iw[2] = {&c->t1i, &c->t2i};
        const u64 g[2] = {powq(psi, 32), powq(psi, 64)}, ord[2] = {64, 32}, root_e[2] = {32, 0};
        uint32_t leaf_e[32];
        for (int t = 0; t < 2; t++) {
            *fw[t] = (double *)calloc(32, 8); *iw[t] = (double *)calloc(32, 8);
            uint32_t e[64]; e[1] = (uint32_t)root_e[t];
            for (uint32_t idx = 1; idx < 32; idx++) {
                uint32_t h = (e[idx] / 2) % ord[t];
                (*fw[t])[idx] = f_center(powq(g[t], h));
                (*iw[t])[idx] = f_center(powq(powq(g[t], h), Q - 2));
                e[2 * idx] = h; e[2 * idx + 1] = (uint32_t)((h + ord[t] / 2) % ord[t]);
            }
            if (t == 0) for (uint32_t r = 0; r < 32; r++) leaf_e[r] = e[32 + r];   /* row position rho evaluates at (psi^32)^leaf_e */
        }
        c->tw = (double *)malloc(1024 * 8); c->twi = (double *)malloc(1024 * 8);
        for (uint32_t r = 0; r < 32; r++)
            for (uint32_t cc = 0; cc < 32; cc++) {
                u64 w = powq(psi, (u64)cc * leaf_e[r] % 2048);
                c->tw[r * 32 + cc] = f_center(w);
                c->twi[r * 32 + cc] = f_center(mulq(powq(ipsi, (u64)cc * leaf_e[r] % 2048), t->inv_N));
            }
    }
    ntt_free(t);
    return c;
}

ora_fctx *ora_fctx_create(const ora_params *P, const u64 *bsk, const u64 *ksk) {
    if (P->q_bits != 49) return NULL;
    ora_set_field(49);
    ora_fctx *c = f_tables_create(P);
    uint32_t N = c->N, n = P->n, k = P->k, lk = P->ks_levels;
    size_t polys = (size_t)n * (k + 1) * P->bs_levels * (k + 1);
    c->bsk = (double *)malloc(polys * N * 8);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < polys; i++) {
        double *d = c->bsk + i * N;
        for (uint32_t x = 0; x < N; x++) d[x] = f_center(bsk[i * N + x]);
        f_ntt_fwd(c, d);
        for (uint32_t x = 0; x < N; x++) d[x] = f_red(d[x]);
    }
    size_t kw = (size_t)k * N * lk * (n + 1);
    c->ksk_lo = (double *)malloc(kw * 8); c->ksk_hi = (double *)malloc(kw * 8);
    for (size_t i = 0; i < kw; i++) { c->ksk_lo[i] = (double)(ksk[i] & 0x1FFFFFFull); c->ksk_hi[i] = (double)(ksk[i] >> 25); }
    return c;
}
void ora_fctx_destroy(ora_fctx *c) { if (!c) return; free(c->t1); free(c->t1i); free(c->t2); free(c->t2i); free(c->tw); free(c->twi); free(c->psi_br); free(c->ipsi_br); free(c->bsk); free(c->ksk_lo); free(c->ksk_hi); free(c); }

/* per-thread scratch of the fast path */
typedef struct { double *acc, *dec, *res, *ks_lo, *ks_hi; u64 *small; } f_scratch;
static void f_scratch_make(const ora_fctx *c, f_scratch *s) {
    uint32_t N = c->N, k = c->P.k, rows = (k + 1) * c->P.bs_levels, n = c->P.n;
    s->acc = (double *)malloc((size_t)(k + 1) * N * 8); s->dec = (double *)malloc((size_t)rows * N * 8);
    s->res = (double *)malloc((size_t)(k + 1) * N * 8);
    s->ks_lo = (double *)malloc((n + 1) * 8); s->ks_hi = (double *)malloc((n + 1) * 8); s->small = (u64 *)malloc((n + 1) * 8);
}
static void f_scratch_free(f_scratch *s) { free(s->acc); free(s->dec); free(s->res); free(s->ks_lo); free(s->ks_hi); free(s->small); }

F_CLONES static void f_keyswitch(const ora_fctx *c, const u64 *in, f_scratch *s, u64 *out) {
    const ora_params *P = &c->P;
    uint32_t n = P->n, kN = P->k << P->log_N, lk = P->ks_levels;
    i64 dig[64];
    double *alo = s->ks_lo, *ahi = s->ks_hi;
    for (uint32_t x = 0; x <= n; x++) alo[x] = ahi[x] = 0.0;
    /* |digit| <= 2^(Bks-1) <= 64, halves < 2^25, kN*lk <= 2^15 rows: every partial sum stays below 2^46, exact in f64 */
    for (uint32_t j = 0; j < kN; j++) {
        ora_decompose(in[j], lk, P->ks_base_log, dig);
        for (uint32_t lev = 0; lev < lk; lev++) {
            if (!dig[lev]) continue;
            const double d = (double)dig[lev];
            const double *rlo = c->ksk_lo + ((size_t)j * lk + lev) * (n + 1), *rhi = c->ksk_hi + ((size_t)j * lk + lev) * (n + 1);
            for (uint32_t x = 0; x <= n; x++) { alo[x] = __builtin_fma(d, rlo[x], alo[x]); ahi[x] = __builtin_fma(d, rhi[x], ahi[x]); }
        }
    }
    for (uint32_t x = 0; x <= n; x++) {
        /* -(hi * 2^25 + lo) mod p: hi < 2^46 -> hi * 2^25 through the exact product */
        double v = f_mul(-ahi[x], 33554432.0) - alo[x];
        out[x] = f_canon(v);
    }
    out[n] = addq(out[n], in[kN]);
}

/* blind rotation of one ciphertext in three pieces (accumulator set-up, one CMUX, sample extraction) so that a thread
 * can walk a GROUP of ciphertexts through the key in step: every key row is then used F_GROUP times while it is in cache */
static void f_br_init(const ora_fctx *c, const u64 *lwe, const u64 *tv, f_scratch *s) {
    const ora_params *P = &c->P;
    uint32_t n = P->n, N = c->N, k = P->k, l = P->bs_levels, rows = (k + 1) * l, log2N = P->log_N + 1;
    const uint32_t shift = 49 - l * P->bs_base_log;
    const double sc = 1.0 / (double)((u64)1 << shift), B = (double)((u64)1 << P->bs_base_log), Binv = 1.0 / B;
    double *acc = s->acc, *dec = s->dec, *res = s->res;
    (void)n; (void)rows; (void)log2N; (void)sc; (void)B; (void)Binv; (void)dec; (void)res; (void)shift;
    uint32_t bt = ora_modswitch(lwe[n], log2N);
    for (uint32_t x = 0; x < (k + 1) * N; x++) acc[x] = 0.0;
    for (uint32_t j = 0; j < N; j++) { /* X^(-bt) * tv */
        uint32_t pos = (j + 2 * N - bt) & (2 * N - 1);
        double v = f_center(tv[j]);
        if (pos < N) acc[(size_t)k * N + pos] = v; else acc[(size_t)k * N + pos - N] = -v;
    }
}
F_CLONES static void f_br_step(const ora_fctx *c, const u64 *lwe, uint32_t i, f_scratch *s) {
    const ora_params *P = &c->P;
    uint32_t n = P->n, N = c->N, k = P->k, l = P->bs_levels, rows = (k + 1) * l, log2N = P->log_N + 1;
    const uint32_t shift = 49 - l * P->bs_base_log;
    const double sc = 1.0 / (double)((u64)1 << shift), B = (double)((u64)1 << P->bs_base_log), Binv = 1.0 / B;
    double *acc = s->acc, *dec = s->dec, *res = s->res;
    (void)n; (void)rows; (void)log2N; (void)sc; (void)B; (void)Binv; (void)dec; (void)res; (void)shift;
        uint32_t at = ora_modswitch(lwe[i], log2N);
        if (at == 0) return;
        for (uint32_t comp = 0; comp <= k; comp++) {
            const double *a = acc + (size_t)comp * N;
            double *d0 = dec + (size_t)comp * l * N;
            /* rot = X^at * a as two contiguous runs (no per-element branch), then (rot - a) centred, rounded (half
             * up) to its top l * Bg bits and peeled into signed digits: all three loops vectorise */
            double *rot = res;   /* scratch: res is rewritten below */
            {
                const uint32_t b = at & (N - 1);
                const double sg = at < N ? 1.0 : -1.0;
                for (uint32_t x = 0; x < b; x++) rot[x] = -sg * a[x + N - b];
                for (uint32_t x = b; x < N; x++) rot[x] = sg * a[x - b];
            }
            for (uint32_t x = 0; x < N; x++) d0[x] = __builtin_floor(__builtin_fma(f_red(rot[x] - a[x]), sc, 0.5));
            for (int lev = (int)l - 1; lev >= 1; lev--) {
                double *dl_ = d0 + (size_t)lev * N;
                for (uint32_t x = 0; x < N; x++) {
                    double r = d0[x], rn = __builtin_floor(__builtin_fma(r, Binv, 0.5));
                    dl_[x] = __builtin_fma(-B, rn, r);
                    d0[x] = rn;
                }
            }
        }
        for (uint32_t r = 0; r < rows; r++) {
            double *d = dec + (size_t)r * N;
            f_ntt_fwd(c, d);
            for (uint32_t x = 0; x < N; x++) d[x] = f_red(d[x]);
        }
        const double *g = c->bsk + (size_t)i * rows * (k + 1) * N;
        for (uint32_t oc = 0; oc <= k; oc++) {
            double *o = res + (size_t)oc * N;
            for (uint32_t x = 0; x < N; x++) o[x] = 0.0;
            for (uint32_t r = 0; r < rows; r++) { /* lazy: rows * 0.57p stays far below 2^53 for rows <= 12 */
                const double *b = g + ((size_t)r * (k + 1) + oc) * N, *d = dec + (size_t)r * N;
                for (uint32_t x = 0; x < N; x++) o[x] += f_mul(d[x], b[x]);
            }
            for (uint32_t x = 0; x < N; x++) o[x] = f_red(o[x]);
            f_ntt_inv(c, o);
            double *a = acc + (size_t)oc * N;
            for (uint32_t x = 0; x < N; x++) a[x] = f_red(a[x] + o[x]);
        }
}
static void f_br_extract(const ora_fctx *c, f_scratch *s, u64 *out) {
    const ora_params *P = &c->P;
    uint32_t n = P->n, N = c->N, k = P->k, l = P->bs_levels, rows = (k + 1) * l, log2N = P->log_N + 1;
    const uint32_t shift = 49 - l * P->bs_base_log;
    const double sc = 1.0 / (double)((u64)1 << shift), B = (double)((u64)1 << P->bs_base_log), Binv = 1.0 / B;
    double *acc = s->acc, *dec = s->dec, *res = s->res;
    (void)n; (void)rows; (void)log2N; (void)sc; (void)B; (void)Binv; (void)dec; (void)res; (void)shift;
    for (uint32_t j = 0; j < k; j++) {
        const double *A = acc + (size_t)j * N;
        out[(size_t)j * N] = f_canon(A[0]);
        for (uint32_t x = 1; x < N; x++) out[(size_t)j * N + x] = f_canon(-A[N - x]);
    }
    out[(size_t)k * N] = f_canon(acc[(size_t)k * N]);
}
/* same contract as ora_pbs_batch (ks_out optional) */
#define F_GROUP 8
void ora_fast_pbs_batch(const ora_fctx *c, const u64 *in, const u64 *tvs, const uint32_t *tv_ids, uint32_t count, u64 *out, u64 *ks_out) {
    uint32_t n = c->P.n, N = c->N, big = c->P.k * N + 1;
    uint32_t groups = (count + F_GROUP - 1) / F_GROUP;
#pragma omp parallel
    {
        f_scratch s[F_GROUP];
        for (int g = 0; g < F_GROUP; g++) f_scratch_make(c, &s[g]);
#pragma omp for schedule(dynamic, 1)
        for (uint32_t gi = 0; gi < groups; gi++) {
            const uint32_t first = gi * F_GROUP, m = count - first < F_GROUP ? count - first : F_GROUP;
            for (uint32_t g = 0; g < m; g++) {
                f_keyswitch(c, in + (size_t)(first + g) * big, &s[g], s[g].small);
                if (ks_out) memcpy(ks_out + (size_t)(first + g) * (n + 1), s[g].small, (n + 1) * 8);
                f_br_init(c, s[g].small, tvs + (size_t)tv_ids[first + g] * N, &s[g]);
            }
            for (uint32_t i = 0; i < n; i++)
                for (uint32_t g = 0; g < m; g++) f_br_step(c, s[g].small, i, &s[g]);
            for (uint32_t g = 0; g < m; g++) f_br_extract(c, &s[g], out + (size_t)(first + g) * big);
        }
        for (int g = 0; g < F_GROUP; g++) f_scratch_free(&s[g]);
    }
}

/* ====================================================================== fast path (2^64 torus, 48-bit key) ====
 * The CPU baseline of bench.py on the torus: the same PBS, same words as ora_pbs_batch on the same (rounded) key - held by
 * tests/test_oracle_tfhe.py - with the arithmetic of the fast path above: bootstrap key stored at 48 bits of precision as two
 * balanced 24-bit limbs, digits in base 2^10 (|d| <= 2^9), every limb sum an integer below p / 2 = 2^48 computed EXACTLY mod
 * p = 2^49 - 720895 in doubles (the route the GPU's exact-transform kernel k_blind_rotate_t64 takes; the generic path above
 * goes through Goldilocks transforms of the key's 32-bit halves), accumulator = word / 2^16 as an exact double centred mod
 * 2^48.  Keyswitch: wrap-around u64 arithmetic. */
typedef struct {
    ora_fctx *f;     /* transform tables of the 49-bit field */
    ora_params P;
    double *bsk;     /* [n][rows][k+1][2 limbs][N] transform domain, centred */
    u64 *ksk;        /* [kN*lk][n+1] words */
} ora_tctx;

static inline i64 t_limb(i64 kword, int j) { /* balanced 24-bit limbs of the 48-bit key word k / 2^16; the second takes the rest */
    kword >>= 16;
    const i64 d = ((kword + (1 << 23)) & ((1 << 24) - 1)) - (1 << 23);
    return j == 0 ? d : (kword - d) >> 24;
}
static inline double t_mod48(double t) { /* centred residue mod 2^48 of an exact integer |t| < 2^53, ties to the negative end */
    return __builtin_fma(-0x1p48, __builtin_floor(__builtin_fma(t, 0x1p-48, 0.5)), t);
}
static inline u64 t_word(double v) { return (u64)(i64)v << 16; } /* |v| < 2^48 exact integer -> torus word */

ora_tctx *ora_tctx_create(const ora_params *P, const u64 *bsk, const u64 *ksk) {
    if (P->q_bits != Q_TORUS64 || P->log_N != 10 || P->bs_base_log > 10 || P->bs_levels * P->bs_base_log >= 48) return NULL;
    size_t polys = (size_t)P->n * (P->k + 1) * P->bs_levels * (P->k + 1), N = (size_t)1 << P->log_N;
    for (size_t i = 0; i < polys * N; i++) if (bsk[i] & 0xFFFF) return NULL;   /* the key must be the 48-bit (rounded) one */
    ora_tctx *c = (ora_tctx *)calloc(1, sizeof *c);
    c->P = *P;
    ora_set_field(49);
    c->f = f_tables_create(P);
    c->bsk = (double *)malloc(polys * 2 * N * 8);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < polys * 2; i++) {
        double *d = c->bsk + i * N;
        const u64 *src = bsk + (i >> 1) * N;
        for (size_t x = 0; x < N; x++) d[x] = (double)t_limb((i64)src[x], (int)(i & 1));
        f_ntt_fwd(c->f, d);
        for (size_t x = 0; x < N; x++) d[x] = f_red(d[x]);
    }
    ora_set_field(Q_TORUS64);
    size_t kw = (size_t)P->k * N * P->ks_levels * (P->n + 1);
    c->ksk = (u64 *)malloc(kw * 8);
    memcpy(c->ksk, ksk, kw * 8);
    return c;
}
void ora_tctx_destroy(ora_tctx *c) { if (!c) return; ora_fctx_destroy(c->f); free(c->bsk); free(c->ksk); free(c); }

F_CLONES static void t_keyswitch(const ora_tctx *c, const u64 *in, u64 *out) {
    const ora_params *P = &c->P;
    uint32_t n = P->n, kN = P->k << P->log_N, lk = P->ks_levels;
    i64 dig[64];
    for (uint32_t x = 0; x <= n; x++) out[x] = 0;
    for (uint32_t j = 0; j < kN; j++) {
        ora_decompose(in[j], lk, P->ks_base_log, dig);
        for (uint32_t lev = 0; lev < lk; lev++) {
            if (!dig[lev]) continue;
            const u64 d = (u64)dig[lev];
            const u64 *row = c->ksk + ((size_t)j * lk + lev) * (n + 1);
            for (uint32_t x = 0; x <= n; x++) out[x] -= d * row[x];   /* mod 2^64 */
        }
    }
    out[n] += in[kN];
}
static void t_br_init(const ora_tctx *c, const u64 *lwe, const u64 *tv, f_scratch *s) {
    const ora_params *P = &c->P;
    uint32_t n = P->n, N = 1u << P->log_N, k = P->k;
    uint32_t bt = ora_modswitch(lwe[n], P->log_N + 1);
    for (uint32_t x = 0; x < (k + 1) * N; x++) s->acc[x] = 0.0;
    for (uint32_t j = 0; j < N; j++) { /* X^(-bt) * tv; test polynomials are multiples of 2^16 */
        uint32_t pos = (j + 2 * N - bt) & (2 * N - 1);
        double v = (double)((i64)tv[j] >> 16);
        if (pos < N) s->acc[(size_t)k * N + pos] = v; else s->acc[(size_t)k * N + pos - N] = t_mod48(-v);
    }
}
F_CLONES static void t_br_step(const ora_tctx *c, const u64 *lwe, uint32_t i, f_scratch *s) {
    const ora_params *P = &c->P;
    uint32_t N = 1u << P->log_N, k = P->k, l = P->bs_levels, rows = (k + 1) * l;
    const double sc = 1.0 / (double)((u64)1 << (48 - l * P->bs_base_log)), B = (double)((u64)1 << P->bs_base_log), Binv = 1.0 / B;
    double *acc = s->acc, *dec = s->dec, *res = s->res;
    uint32_t at = ora_modswitch(lwe[i], P->log_N + 1);
    if (at == 0) return;
    for (uint32_t comp = 0; comp <= k; comp++) {
        const double *a = acc + (size_t)comp * N;
        double *d0 = dec + (size_t)comp * l * N, *rot = res;
        {
            const uint32_t b = at & (N - 1);
            const double sg = at < N ? 1.0 : -1.0;
            for (uint32_t x = 0; x < b; x++) rot[x] = -sg * a[x + N - b];
            for (uint32_t x = b; x < N; x++) rot[x] = sg * a[x - b];
        }
        /* the oracle's rule on the 64-bit word (ora_decompose): centred, rounded half up to its top l * Bg bits - here on word / 2^16 */
        for (uint32_t x = 0; x < N; x++) d0[x] = __builtin_floor(__builtin_fma(t_mod48(rot[x] - a[x]), sc, 0.5));
        for (int lev = (int)l - 1; lev >= 1; lev--) {
            double *dl_ = d0 + (size_t)lev * N;
            for (uint32_t x = 0; x < N; x++) {
                double r = d0[x], rn = __builtin_floor(__builtin_fma(r, Binv, 0.5));
                dl_[x] = __builtin_fma(-B, rn, r);
                d0[x] = rn;
            }
        }
    }
    for (uint32_t r = 0; r < rows; r++) {
        double *d = dec + (size_t)r * N;
        f_ntt_fwd(c->f, d);
        for (uint32_t x = 0; x < N; x++) d[x] = f_red(d[x]);
    }
    const double *g = c->bsk + (size_t)i * rows * (k + 1) * 2 * N;
    for (uint32_t oc = 0; oc <= k; oc++)
        for (int j = 0; j < 2; j++) {
            double *o = res;
            for (uint32_t x = 0; x < N; x++) o[x] = 0.0;
            for (uint32_t r = 0; r < rows; r++) {
                const double *b = g + (((size_t)r * (k + 1) + oc) * 2 + j) * N, *d = dec + (size_t)r * N;
                for (uint32_t x = 0; x < N; x++) o[x] += f_mul(d[x], b[x]);
            }
            for (uint32_t x = 0; x < N; x++) o[x] = f_red(o[x]);
            f_ntt_inv(c->f, o);
            double *a = acc + (size_t)oc * N;
            if (j == 0) for (uint32_t x = 0; x < N; x++) a[x] = t_mod48(a[x] + f_red(o[x]));   /* the limb's exact integer, |.| < p / 2 */
            else for (uint32_t x = 0; x < N; x++) {
                double v = f_red(o[x]);
                v = __builtin_fma(-0x1p24, __builtin_rint(v * 0x1p-24), v);                    /* x 2^24 mod 2^48: the low 24 bits survive */
                a[x] = t_mod48(__builtin_fma(v, 0x1p24, a[x]));
            }
        }
}
static void t_br_extract(const ora_tctx *c, f_scratch *s, u64 *out) {
    uint32_t N = 1u << c->P.log_N, k = c->P.k;
    for (uint32_t j = 0; j < k; j++) {
        const double *A = s->acc + (size_t)j * N;
        out[(size_t)j * N] = t_word(A[0]);
        for (uint32_t x = 1; x < N; x++) out[(size_t)j * N + x] = (u64)0 - t_word(A[N - x]);
    }
    out[(size_t)k * N] = t_word(s->acc[(size_t)k * N]);
}
/* same contract as ora_pbs_batch (ks_out optional); selects the torus */
void ora_tfast_pbs_batch(const ora_tctx *c, const u64 *in, const u64 *tvs, const uint32_t *tv_ids, uint32_t count, u64 *out, u64 *ks_out) {
    ora_set_field(Q_TORUS64);
    uint32_t n = c->P.n, N = 1u << c->P.log_N, big = c->P.k * N + 1;
    uint32_t groups = (count + F_GROUP - 1) / F_GROUP;
#pragma omp parallel
    {
        f_scratch s[F_GROUP];
        for (int g = 0; g < F_GROUP; g++) f_scratch_make(c->f, &s[g]);
#pragma omp for schedule(dynamic, 1)
        for (uint32_t gi = 0; gi < groups; gi++) {
            const uint32_t first = gi * F_GROUP, m = count - first < F_GROUP ? count - first : F_GROUP;
            for (uint32_t g = 0; g < m; g++) {
                t_keyswitch(c, in + (size_t)(first + g) * big, s[g].small);
                if (ks_out) memcpy(ks_out + (size_t)(first + g) * (n + 1), s[g].small, (n + 1) * 8);
                t_br_init(c, s[g].small, tvs + (size_t)tv_ids[first + g] * N, &s[g]);
            }
            for (uint32_t i = 0; i < n; i++)
                for (uint32_t g = 0; g < m; g++) t_br_step(c, s[g].small, i, &s[g]);
            for (uint32_t g = 0; g < m; g++) t_br_extract(c, &s[g], out + (size_t)(first + g) * big);
        }
        for (int g = 0; g < F_GROUP; g++) f_scratch_free(&s[g]);
    }
}
