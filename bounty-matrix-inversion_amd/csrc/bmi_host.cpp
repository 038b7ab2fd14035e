// Host side of libbmi_tfhe.so: the C ABI of include/bmi_tfhe.h — context, deterministic key
// generation, encryption / decryption, LUT (test-polynomial) construction and the batch entry points
// that launch the HIP kernels of bmi_kernels.hip.  No CPU fallback exists for the PBS path: every
// batch call needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>

#include <errno.h>
#include <sys/random.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bmi_tfhe.h"
#include "bmi_internal.hpp"
#include "field49.hpp"
#include "ntt_wave.hpp"
#include "ntt_half_f64.hpp"
#include "ntt_wave_f64.hpp"
#include "fft_half_f64.hpp"
#include "fft_wave_f64.hpp"
#include "fft_eighth_f64.hpp"
#include "fft_quarter_f64.hpp"
#include "t64_common.hpp"

using gl::i64;
using gl::u64;

namespace {

// --------------------------------------------------------------------------- deterministic RNG
// Counter-based splitmix64 (same specification as the oracle so that key generation can be
// cross-checked bit for bit): value(stream, idx) = mix(stream_key + (idx + 1) * GOLDEN).
inline u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// Arithmetic mod the context's ciphertext modulus on the host (keygen, encryption, LUT construction): plain
// 128-bit remainders - none of this is on the hot path.
struct Fq {
    u64 q = gl::P;
    uint32_t bits = 64;
    bool torus = false;   // q = 2^64 exactly (q field unused): plain wrap-around arithmetic
    u64 add(u64 a, u64 b) const { return torus ? a + b : (u64)(((unsigned __int128)a + b) % q); }
    u64 sub(u64 a, u64 b) const { return torus ? a - b : (a >= b ? a - b : (u64)((unsigned __int128)a + q - b)); }
    u64 neg(u64 a) const { return torus ? (u64)0 - a : (a ? q - a : 0); }
    u64 mul(u64 a, u64 b) const { return torus ? a * b : (u64)(((unsigned __int128)a * b) % q); }
    u64 from_i64(long long v) const { return torus ? (u64)v : (v >= 0 ? (u64)v % q : q - ((u64)(-v) % q)); }
    long long centered(u64 a) const { return torus ? (long long)a : (a > (q >> 1) ? (long long)(a - q) : (long long)a); }
    bool canonical(u64 a) const { return torus || a < q; }
    u64 pow(u64 b, u64 e) const {
        u64 r = 1;
        while (e) {
            if (e & 1) r = mul(r, b);
            b = mul(b, b);
            e >>= 1;
        }
        return r;
    }
};

struct Stream {
    u64 key;
    Fq f;
    Stream(u64 seed, u64 id, const Fq &fq) : key(mix64(seed ^ (id * 0xD6E8FEB86659FD93ULL))), f(fq) {}
    u64 raw(u64 idx) const { return mix64(key + (idx + 1) * 0x9E3779B97F4A7C15ULL); }
    u64 bit(u64 idx) const { return raw(idx) & 1; }
    u64 uniform(u64 idx) const {
        u64 u = raw(idx);
        if (f.torus) return u;
        return f.bits == 64 ? (u >= f.q ? u - f.q : u) : u % f.q;
    }
    u64 gauss(u64 idx, double sigma) const {  // Box-Muller, rounded to an element of Z_q
        const double u1 = (double)((raw(2 * idx) >> 11) + 1) * (1.0 / 9007199254740992.0);
        const double u2 = (double)(raw(2 * idx + 1) >> 11) * (1.0 / 9007199254740992.0);
        const double g = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
        const long long e = std::llround(g * sigma * (f.bits == 64 ? 18446744073709551616.0 : (double)f.q));
        return f.from_i64(e);
    }
};
enum : u64 { S_SK_SMALL = 1, S_SK_BIG, S_BSK_MASK, S_BSK_NOISE, S_KSK_MASK, S_KSK_NOISE, S_ENC_MASK, S_ENC_NOISE, S_BSK3_MASK, S_BSK3_NOISE };

// --------------------------------------------------------------------------- CSPRNG (production keys and encryptions)
// ChaCha20 (D. J. Bernstein's original variant: 256-bit key, 64-bit nonce, 64-bit block counter) keyed from the
// operating system (getrandom).  Secret material (key bits, noise) and public material (masks) use two independent keys,
// so that the public evaluation-key words say nothing about the stream the secrets came from.  Every (domain, row) pair
// is its own nonce: rows are sampled in parallel and sequentially within a row.
struct ChaKey {
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool fill_from_os() {
        size_t got = 0;
        unsigned char *p = reinterpret_cast<unsigned char *>(k);
        while (got < sizeof(k)) {
            const ssize_t r = getrandom(p + got, sizeof(k) - got, 0);
            if (r < 0) {
                if (errno == EINTR) continue;
                return false;
            }
            got += (size_t)r;
        }
        return true;
    }
};
inline uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
inline void chacha_block(const ChaKey &key, u64 nonce, u64 counter, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                       key.k[4], key.k[5], key.k[6], key.k[7], (uint32_t)counter, (uint32_t)(counter >> 32),
                       (uint32_t)nonce, (uint32_t)(nonce >> 32)};
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = in[i];
#define BMI_QR(a, b, c, d)                  \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7)
    for (int r = 0; r < 10; r++) {
        BMI_QR(0, 4, 8, 12); BMI_QR(1, 5, 9, 13); BMI_QR(2, 6, 10, 14); BMI_QR(3, 7, 11, 15);
        BMI_QR(0, 5, 10, 15); BMI_QR(1, 6, 11, 12); BMI_QR(2, 7, 8, 13); BMI_QR(3, 4, 9, 14);
    }
#undef BMI_QR
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}
struct ChaStream {
    const ChaKey *key;
    u64 nonce, counter = 0;
    uint32_t buf[16];
    int pos = 16;
    ChaStream(const ChaKey &k, u64 domain, u64 row) : key(&k), nonce((domain << 56) | (row & ((1ULL << 56) - 1))) {}
    u64 next64() {
        if (pos > 14) {
            chacha_block(*key, nonce, counter++, buf);
            pos = 0;
        }
        const u64 v = (u64)buf[pos] | ((u64)buf[pos + 1] << 32);
        pos += 2;
        return v;
    }
    u64 bit() { return next64() & 1; }
    u64 uniform(const Fq &f) {   // rejection sampling: exactly uniform on [0, q)
        for (;;) {
            const u64 u = f.bits == 64 ? next64() : next64() >> (64 - f.bits);
            if (f.torus || u < f.q) return u;
        }
    }
    u64 gauss(const Fq &f, double sigma) {
        const double u1 = (double)((next64() >> 11) + 1) * (1.0 / 9007199254740992.0);
        const double u2 = (double)(next64() >> 11) * (1.0 / 9007199254740992.0);
        const double g = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
        return f.from_i64(std::llround(g * sigma * (f.bits == 64 ? 18446744073709551616.0 : (double)f.q)));
    }
};
// One row's source of masks / noise: the CSPRNG (sequential within the row) or, for the test-only deterministic key
// generation, the counter-indexed splitmix streams the oracle reproduces bit for bit.
struct RowRng {
    const Stream *det;
    ChaStream cha;
    bool secure;
    RowRng(const Stream *d, const ChaKey &k, u64 domain, u64 row, bool sec) : det(d), cha(k, domain, row), secure(sec) {}
    u64 uniform(u64 idx) { return secure ? cha.uniform(det->f) : det->uniform(idx); }
    u64 gauss(u64 idx, double sigma) { return secure ? cha.gauss(det->f, sigma) : det->gauss(idx, sigma); }
    u64 bit(u64 idx) { return secure ? cha.bit() : det->bit(idx); }
};

thread_local std::string g_create_error;

template <class F>
void parallel_for(size_t n, F f) {
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    if (n < 4 * nt) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([=]() {
            for (size_t i = t; i < n; i += nt) f(i);
        });
    for (auto &x : th) x.join();
}

}  // namespace

struct bmi_ctx {
    bmi_params P{};
    Fq f;
    int device = 0;
    uint32_t N = 0, big_n = 0, rows = 0, ks_stride = 0;
    hipStream_t stream = nullptr;  // the context's own stream (host-buffer entry points)
    bool have_keys = false;
    bool have_secret = false;  // false for a context that imported evaluation keys only (no encrypt / decrypt)
    u64 seed = 0, enc_counter = 0;
    int bsk_prec = 64;   // torus: bits of precision the bootstrap key is stored at (64 = exact, 48, 46, 42: t64_common.hpp; bmi_set_bsk_precision)
    bool bsk_prec_explicit = false;   // set by bmi_set_bsk_precision: bmi_set_bsk_unroll then leaves the precision alone
    int bsk_limbs() const { return t64::limbs_of(bsk_prec); }
    bool secure_rng = false;       // true: keys / encryptions drawn from the CSPRNG below; false: test-only seeded streams
    ChaKey rng_secret, rng_public; // independent ChaCha20 keys from getrandom(): secrets + noise / public masks
    std::vector<u64> sk_small, sk_big, bsk_std, ksk;
    void *d_bsk = nullptr, *d_tw = nullptr, *d_luts = nullptr;  // u64 words (Goldilocks) or f64 words (49-bit field)
    double *d_tw_half = nullptr, *d_bsk_lat = nullptr;          // 49-bit field: tables and key copy of the split-transform latency kernel
    double *d_tw_fft = nullptr, *d_bsk_fft = nullptr;           // 2^64 torus, key at 48 bits: tables and key copy of the floating-point-transform wave-pair kernel (bmi_kernels_t64f.hip)
    double *d_tw_fh = nullptr, *d_bsk_latf = nullptr;           // ... and of its latency form (half transforms, fft_half_f64.hpp; key in slot-pair order)
    double *d_zeta_pow = nullptr;                               // 2^64 torus at N = 1024: zeta^x, x in [0, 1024) as (re, im) - the factors X^c of the unrolled floating-point-transform kernel
    double *d_tw_fq = nullptr, *d_bsk_w = nullptr;              // 2^64 torus at N = 2048, key at 46 bits: tables and key copy of bmi_kernels_t64w.hip (quarter transforms, fft_quarter_f64.hpp);
                                                                // at N = 4096, key at 44 bits: those of bmi_kernels_t64q.hip (eighth transforms, fft_eighth_f64.hpp)
    double *d_bsk_w2 = nullptr;                                 // N = 2048: the key copy of the two-ciphertexts-per-workgroup kernel (bmi_kernels_t64w2.hip: [t 4][256 slots] per polynomial and limb)
    double *d_tw_wide = nullptr;                                // N = 2048: T / T^-1 of the even/odd combination (d_bsk_lat then holds the wide key copy)
    // bootstrap-key unrolling (49-bit field at N = 1024 / 2048, 2^64 torus; bmi_set_bsk_unroll): per pair of LWE coefficients the GGSW encryptions of
    // s s', s (1 - s'), (1 - s) s'; host copy in the standard domain, device copy in the slot order of the latency kernel
    uint32_t unroll = 1;
    std::vector<u64> bsk3_std;
    double *d_bsk3_lat = nullptr, *d_root_pow = nullptr;        // d_root_pow: psi^x, x in [0, 2N), centred doubles
    bool have_bsk3 = false;
    uint32_t pairs() const { return (P.n + 1) / 2; }
    size_t bsk3_words() const { return (size_t)pairs() * 3 * rows * (P.k + 1) * N; }
    bool wide() const { return N == 2048; }
    bool quad() const { return N == 4096; }   // d_tw_wide then holds T_1..T_3 and their inverses, d_bsk_lat the quad key copy
    u64 *d_ksk = nullptr, *d_ks_bias = nullptr;
    uint32_t n_luts = 0, lut_cap = 0;
    std::vector<std::vector<u64>> luts_host;
    // growable device scratch
    u64 *d_small = nullptr;
    size_t small_cap = 0;
    u64 *d_io_a = nullptr, *d_io_b = nullptr;
    uint32_t *d_io_ids = nullptr;
    size_t io_cap = 0;
    int variant = 0;
    uint32_t lat_threshold = 512;  // 2 rounds of 256 one-workgroup PBS (11.2 ms) tie with one round of the wave-pair kernel (11.3 ms)
    void *d_ks_partial = nullptr;
    size_t ks_partial_bytes = 0;
    // keyswitch on the matrix cores: limb-wise key (per keygen), digit matrix and int32 sums (growable scratch)
    int ks_variant = 0;  // 0 = auto (matrix cores when the shape allows), 1 = scalar kernel
    bool no_big_lds = false;  // set when a > 64 KB LDS kernel could not be configured on this device (auto mode only)
    bool ks_mfma_ok = false;
    signed char *d_ks_limbs = nullptr, *d_ks_digits = nullptr;
    int *d_ks_sums = nullptr;
    size_t ks_digits_bytes = 0, ks_sums_bytes = 0;
    uint32_t ks_limbs() const { return f64() ? bmi49::KS_LIMBS : (f.torus ? bmit::KS_LIMBS : bmi::KS_LIMBS); }
    mutable std::string err;
    bool f64() const { return f.bits == 49; }
    bool t64() const { return f.torus; }
};

namespace {

int fail(const bmi_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}
#define HIP_OK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(ctx, -2, std::string(#call) + ": " + hipGetErrorString(e__));                  \
    } while (0)

int ensure_small(bmi_ctx *c, size_t count) {
    if (count <= c->small_cap) return 0;
    if (c->d_small) {   // queued work may still read the old buffer
        HIP_OK(c, hipDeviceSynchronize());
        HIP_OK(c, hipFree(c->d_small));
        c->d_small = nullptr;
        c->small_cap = 0;
    }
    size_t cap = std::max<size_t>(count, 1024);
    HIP_OK(c, hipMalloc(&c->d_small, cap * (c->P.n + 1) * sizeof(u64)));
    c->small_cap = cap;
    return 0;
}
int ensure_io(bmi_ctx *c, size_t count) {
    if (count <= c->io_cap) return 0;
    if (c->d_io_a) HIP_OK(c, hipFree(c->d_io_a));
    if (c->d_io_b) HIP_OK(c, hipFree(c->d_io_b));
    if (c->d_io_ids) HIP_OK(c, hipFree(c->d_io_ids));
    size_t cap = std::max<size_t>(count, 256);
    HIP_OK(c, hipMalloc(&c->d_io_a, cap * (c->big_n + 1) * sizeof(u64)));
    HIP_OK(c, hipMalloc(&c->d_io_b, cap * (c->big_n + 1) * sizeof(u64)));
    HIP_OK(c, hipMalloc(&c->d_io_ids, cap * sizeof(uint32_t)));
    c->io_cap = cap;
    return 0;
}

// out += X^t * a  (negacyclic), coefficient-wise in Z_q
void add_shifted(const Fq &f, u64 *out, const u64 *a, uint32_t t, uint32_t N) {
    for (uint32_t j = 0; j + t < N; j++) out[j + t] = f.add(out[j + t], a[j]);
    for (uint32_t j = N - t; j < N; j++) out[j + t - N] = f.sub(out[j + t - N], a[j]);
}

// Twiddle tables of the wave NTT as canonical integers mod q: W1[k1][l] = psi^(l (2 k1 + 1)),
// W1I = psi^-(..) / N, W2[v][t] = (psi^32)^(t v), W2I its inverse (layout shared by both fields).
std::vector<u64> build_twiddles(const Fq &f, u64 psi, u64 psi_inv, u64 n_inv) {
    using namespace nttw;
    std::vector<u64> tw(TW_WORDS);
    for (int k1 = 0; k1 < 16; k1++)
        for (int l = 0; l < 64; l++) {
            const u64 e = (u64)l * (2 * k1 + 1);
            tw[TW_W1 + k1 * 64 + l] = f.pow(psi, e);
            tw[TW_W1I + k1 * 64 + l] = f.mul(f.pow(psi_inv, e), n_inv);
        }
    const u64 w64 = f.pow(psi, 32), w64i = f.pow(psi_inv, 32);
    for (int v = 0; v < 16; v++)
        for (int t = 0; t < 4; t++) {
            tw[TW_W2 + v * 4 + t] = f.pow(w64, (u64)t * v);
            tw[TW_W2I + v * 4 + t] = f.pow(w64i, (u64)t * v);
        }
    return tw;
}

std::vector<double> to_centred_doubles(const std::vector<u64> &v) {
    std::vector<double> d(v.size());
    for (size_t i = 0; i < v.size(); i++) d[i] = f49::to_f(v[i]);
    return d;
}

// Tables of the two-wave half transform (ntt_half_f64.hpp, tools/ntt_half_model.py), canonical integers mod q
std::vector<u64> build_twiddles_half(const Fq &f, u64 psi, u64 psi_inv) {
    using namespace ntth;
    std::vector<u64> tw(HT_WORDS);
    const u64 inv1024 = f.pow(1024 % f.q, f.q - 2);
    for (int k1 = 0; k1 < 8; k1++)
        for (int l = 0; l < 64; l++) {
            const u64 e = (u64)2 * (2 * k1 + 1) * l;
            tw[HT_W1 + k1 * 64 + l] = f.pow(psi, e);
            tw[HT_W1I + k1 * 64 + l] = f.mul(f.pow(psi_inv, e), inv1024);
        }
    for (int k2a = 0; k2a < 8; k2a++)
        for (int l0 = 0; l0 < 8; l0++) {
            tw[HT_W2 + k2a * 8 + l0] = f.pow(psi, (u64)32 * l0 * k2a);
            tw[HT_W2I + k2a * 8 + l0] = f.pow(psi_inv, (u64)32 * l0 * k2a);
        }
    for (int reg = 0; reg < 8; reg++)
        for (int lane = 0; lane < 64; lane++) {
            const u64 e = (u64)2 * kk_of(lane, reg) + 1;
            tw[HT_T + reg * 64 + lane] = f.pow(psi, e);
            tw[HT_TI + reg * 64 + lane] = f.pow(psi_inv, e);
        }
    return tw;
}

// N = 2048: T[reg * 64 + lane] = psi_4096^(2 kk + 1) at the evaluation slot (lane, reg) of the 1024-point wave transform
// (kk = k1 + 16 (4 vl + g) + 256 s for lane 4 k1 + g, register 4 vl + s), then the inverses.
std::vector<u64> build_twiddles_wide(const Fq &f) {
    const u64 psi4096 = f.pow(f49::GEN, (f.q - 1) / 4096), inv = f.pow(psi4096, f.q - 2);
    std::vector<u64> tw(2048);
    for (int reg = 0; reg < 16; reg++)
        for (int lane = 0; lane < 64; lane++) {
            const u64 kk = (u64)(lane >> 2) + 16 * (4 * (reg >> 2) + (lane & 3)) + 256 * (reg & 3);
            tw[reg * 64 + lane] = f.pow(psi4096, 2 * kk + 1);
            tw[1024 + reg * 64 + lane] = f.pow(inv, 2 * kk + 1);
        }
    return tw;
}

// N = 4096: T_j[reg * 64 + lane] = psi_8192^((2 kk + 1) j) for j = 1, 2, 3, then the three inverse tables
std::vector<u64> build_twiddles_quad(const Fq &f) {
    const u64 psi = f.pow(f49::GEN, (f.q - 1) / 8192), inv = f.pow(psi, f.q - 2);
    std::vector<u64> tw(6 * 1024);
    for (int j = 1; j <= 3; j++)
        for (int reg = 0; reg < 16; reg++)
            for (int lane = 0; lane < 64; lane++) {
                const u64 kk = (u64)(lane >> 2) + 16 * (4 * (reg >> 2) + (lane & 3)) + 256 * (reg & 3);
                tw[(j - 1) * 1024 + reg * 64 + lane] = f.pow(psi, (2 * kk + 1) * j);
                tw[(3 + j - 1) * 1024 + reg * 64 + lane] = f.pow(inv, (2 * kk + 1) * j);
            }
    return tw;
}

// precision a torus context stores its bootstrap key at unless bmi_set_bsk_precision says otherwise: 48 bits (two 24-bit limbs)
// where the decomposition base leaves room for it (Bg <= 2^10: the default torus set), else the exact key (three 22-bit limbs)
// (N = 2048: 46 bits, two 23-bit limbs; N = 4096: 44 bits, two 22-bit limbs - the lengths at which the floating-point transform's error
// bound still certifies the rounding)
int default_bsk_precision(const bmi_params &P) { return P.log_N == 12 ? 44 : P.log_N == 11 ? 46 : (P.bs_base_log <= 10 ? 48 : 64); }

bool params_supported(const bmi_params &P, std::string &why) {
    if (P.log_N < 10 || P.log_N > 12 || (P.log_N != 10 && P.q_bits != 49 && P.q_bits != BMI_Q_TORUS64)) {
        why = "log_N must be 10 (N = 1024), 11 (N = 2048: 49-bit field or 2^64 torus) or 12 (N = 4096: 49-bit field or 2^64 torus)";
        return false;
    }
    if (P.k != 1) { why = "only k = 1 has a HIP kernel in this build"; return false; }
    const bool lb_default = P.bs_levels == 3 && P.bs_base_log == 15;
    const bool lb_f64 = lb_default || (P.bs_levels == 2 && P.bs_base_log == 15) || (P.bs_levels == 1 && P.bs_base_log == 23);
    // (2, 2^15) and (1, 2^23): the templated 49-bit kernels (N = 1024 wave-pair / latency kernels, N = 2048), and (2, 2^15) on the torus
    const bool ok_lb = (lb_default && P.q_bits != BMI_Q_TORUS64) || (P.q_bits == 49 && P.log_N <= 11 && lb_f64) ||
                       (P.q_bits == BMI_Q_TORUS64 && P.log_N == 10 && bmit::shape_supported(default_bsk_precision(P), P.bs_levels, P.bs_base_log)) ||
                       (P.q_bits == BMI_Q_TORUS64 && P.log_N == 11 && bmit::shape_supported_wide(default_bsk_precision(P), P.bs_levels, P.bs_base_log)) ||
                       (P.q_bits == BMI_Q_TORUS64 && P.log_N == 12 && bmit::shape_supported_quad(default_bsk_precision(P), P.bs_levels, P.bs_base_log));
    if (!ok_lb) {
        why = "(l, Bg) must be (3, 2^15); the 49-bit field at N <= 2048 also takes (2, 2^15) and (1, 2^23), the 2^64 torus (3 or 2, 2^10) "
              "and, at N = 1024, (3 or 2, 2^15)";
        return false;
    }
    if (P.n == 0 || P.n > BMI_MAX_LWE_N) { why = "n must be in [1, 1024]"; return false; }
    if (P.q_bits != 0 && P.q_bits != 64 && P.q_bits != 49 && P.q_bits != BMI_Q_TORUS64) {
        why = "q_bits must be 64 (2^64-2^32+1), 49 (2^49-720895) or BMI_Q_TORUS64 (2^64)";
        return false;
    }
    const uint32_t qb = P.q_bits == 49 ? 49 : 64;
    if (P.ks_levels * P.ks_base_log >= qb || P.ks_base_log > 7 || P.ks_levels == 0) { why = "unsupported keyswitch decomposition"; return false; }
    return true;
}

}  // namespace

extern "C" {

int bmi_default_params_for(uint32_t q_bits, bmi_params *out) {
    if (!out || (q_bits != 64 && q_bits != 49 && q_bits != BMI_Q_TORUS64)) return -1;
    // same shape for every modulus; the 49-bit modulus keeps the absolute bootstrap-key noise above the integer grid
    // the torus set decomposes in base 2^10: the limb sums of its exact products then leave room for a two-limb key (48 bits of
    // precision) in the plain AND the unrolled blind rotation, and the output noise is lower than at 2^15 (t64_common.hpp)
    *out = bmi_params{630, 10, 1, 3, q_bits == BMI_Q_TORUS64 ? 10u : 15u, 8, 4, q_bits, std::ldexp(1.0, -25),
                      std::ldexp(1.0, q_bits == 49 ? -40 : -44)};
    return 0;
}

int bmi_default_params(bmi_params *out) { return bmi_default_params_for(BMI_DEFAULT_Q_BITS, out); }

int bmi_preset_params(const char *name, bmi_params *out) {
    if (!name || !out) return -1;
    const std::string s(name);
    if (s == "north_star") return bmi_default_params_for(49, out);
    if (s == "north_star_torus64") return bmi_default_params_for(BMI_Q_TORUS64, out);
    if (s == "north_star_goldilocks") return bmi_default_params_for(64, out);
    if (s == "secure128") {
        // n = 742 with LWE noise 7.07e-6 (2^-17.11) and a GLWE of size k N = 2048: the two security-relevant pairs of
        // TFHE-rs' published 128-bit set PARAM_MESSAGE_2_CARRY_2_KS_PBS (see bmi_tfhe.h); the GLWE noise is kept at
        // 2^-44 >= that set's 2.94e-16, so the GLWE side is at least as hard.  Decompositions are this build's: bootstrap 2 x 15
        // bits (output noise 2^-19.5: the circuits' linear combinations amplify it up to 75x), keyswitch 8 x 2 bits (its noise,
        // set by the LWE noise that security dictates, is what bounds the look-up margin: finer digits = less of it).
        *out = bmi_params{742, 11, 1, 2, 15, 8, 2, 49, 7.069849454709433e-6, std::ldexp(1.0, -44)};
        return 0;
    }
    if (s == "secure128_torus") {
        // the same two security-relevant pairs (n = 742 at LWE noise 2^-17.11; GLWE of size 2048 at noise 2^-44 >= 2.94e-16) on
        // Concrete's own modulus q = 2^64.  Decompositions: bootstrap 3 x 10 bits against a key stored at 46 bits of precision
        // (two 23-bit limbs: exact limb sums through the floating-point transform, fft_quarter_f64.hpp) - output noise 2^-22.9,
        // below the 49-bit preset's 2^-19.5; keyswitch 8 x 2 bits as there.
        *out = bmi_params{742, 11, 1, 3, 10, 8, 2, BMI_Q_TORUS64, 7.069849454709433e-6, std::ldexp(1.0, -44)};
        return 0;
    }
    if (s == "secure128_torus_wide") {
        // the same LWE pair under a GLWE of size 4096 (harder than the 2048 the noise 2^-44 is rated for) on q = 2^64, for the 5-bit
        // look-ups of the reference's unmodified circuits and of bases other than 2: key at 44 bits of precision (two 22-bit limbs,
        // fft_eighth_f64.hpp), bootstrap 3 x 10 bits.  The keyswitch noise (4,096 x levels rows at the LWE noise security dictates)
        // is what bounds the margin: 16 levels of 1 bit (mean-square digit 1/2) put a 5-bit look-up at 5.4 sigma, against 4.6 with
        // 8 x 2 bits and 4.4 at N = 2048 - whether that carries a circuit is error_budget's call (p_error), look-up count by count.
        *out = bmi_params{742, 12, 1, 3, 10, 16, 1, BMI_Q_TORUS64, 7.069849454709433e-6, std::ldexp(1.0, -44)};
        return 0;
    }
    return -1;
}

const char *bmi_last_error(const bmi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int bmi_ctx_create(const bmi_params *params, int device, bmi_ctx **out) {
    if (!params || !out) return fail(nullptr, -1, "null argument");
    std::string why;
    if (!params_supported(*params, why)) return fail(nullptr, -1, why);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, -3, "no HIP device: the PBS path has no CPU fallback (hipGetDeviceCount: " +
                                     std::string(hipGetErrorString(e)) + ")");
    if (device < 0 || device >= ndev) return fail(nullptr, -1, "bad device index");
    bmi_ctx *c = new bmi_ctx();
    c->P = *params;
    if (c->P.q_bits == 0) c->P.q_bits = 64;
    if (c->P.q_bits == 49) c->f = Fq{f49::Q, 49};
    if (c->P.q_bits == BMI_Q_TORUS64) {
        c->f = Fq{0, 64, true};
        c->bsk_prec = default_bsk_precision(c->P);
    }
    c->device = device;
    c->N = 1u << params->log_N;
    c->big_n = params->k * c->N;
    c->rows = (params->k + 1) * params->bs_levels;
    c->ks_stride = (params->n + 1 + 7) & ~7u;
    auto bail = [&](const std::string &m) {
        g_create_error = m;
        delete c;
        return -2;
    };
    if (hipSetDevice(device) != hipSuccess) return bail("hipSetDevice failed");
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate failed");
    {
        const bool f64 = c->f64() || c->t64();   // the torus kernels transform mod the 49-bit prime as well
        const Fq f49q{f49::Q, 49};
        const std::vector<u64> tw = f64 ? build_twiddles(f49q, nttf::PSI_U, nttf::PSI_INV_U, nttf::N_INV_U)
                                        : build_twiddles(c->f, nttw::PSI, nttw::PSI_INV, nttw::N_INV);
        const std::vector<double> twd = f64 ? to_centred_doubles(tw) : std::vector<double>();
        const void *src = f64 ? (const void *)twd.data() : (const void *)tw.data();
        if (hipMalloc(&c->d_tw, tw.size() * 8) != hipSuccess) return bail("hipMalloc(twiddles) failed");
        if (hipMemcpy(c->d_tw, src, tw.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail("hipMemcpy(twiddles) failed");
    }
    if (c->f64() || c->t64()) {
        const std::vector<double> th = to_centred_doubles(build_twiddles_half(Fq{f49::Q, 49}, nttf::PSI_U, nttf::PSI_INV_U));
        if (hipMalloc(&c->d_tw_half, th.size() * 8) != hipSuccess) return bail("hipMalloc(half-transform twiddles) failed");
        if (hipMemcpy(c->d_tw_half, th.data(), th.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail("hipMemcpy(half-transform twiddles) failed");
    }
    if (c->t64() && c->N == 1024) {   // tables of the folded 512-point complex transform (fft_wave_f64.hpp): powers of zeta = exp(i pi / 1024)
        std::vector<double> tf(fftw::TW_WORDS);
        auto zeta_pow = [](uint32_t e, double *dst) {
            const long double ang = 3.14159265358979323846264338327950288L * (long double)(e % 2048) / 1024.0L;
            dst[0] = (double)cosl(ang);
            dst[1] = (double)sinl(ang);
        };
        for (uint32_t k2 = 0; k2 < 8; k2++)
            for (uint32_t lane = 0; lane < 64; lane++) zeta_pow(lane * (4 * k2 + 1), &tf[fftw::TW_T1 + (k2 * 64 + lane) * 2]);
        for (uint32_t d = 0; d < 8; d++)
            for (uint32_t a = 0; a < 8; a++) zeta_pow(32 * a * d, &tf[fftw::TW_T2 + (d * 8 + a) * 2]);
        if (hipMalloc(&c->d_tw_fft, tf.size() * 8) != hipSuccess) return bail("hipMalloc(fft twiddles) failed");
        if (hipMemcpy(c->d_tw_fft, tf.data(), tf.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail("hipMemcpy(fft twiddles) failed");
        std::vector<double> th(ffth::HT_WORDS);
        ffth::build_tables(th.data());
        if (hipMalloc(&c->d_tw_fh, th.size() * 8) != hipSuccess) return bail("hipMalloc(half-fft twiddles) failed");
        if (hipMemcpy(c->d_tw_fh, th.data(), th.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail("hipMemcpy(half-fft twiddles) failed");
        std::vector<double> zp(2048);
        for (uint32_t x = 0; x < 1024; x++) zeta_pow(x, &zp[2 * x]);
        if (hipMalloc(&c->d_zeta_pow, zp.size() * 8) != hipSuccess) return bail("hipMalloc(zeta powers) failed");
        if (hipMemcpy(c->d_zeta_pow, zp.data(), zp.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail("hipMemcpy(zeta powers) failed");
    }
    if ((c->f64() && !c->quad()) || c->t64()) {   // psi^x for the unrolled blind rotation (X^c at the root psi^e is psi^(e c))
        // N = 1024: psi = psi_2048, x in [0, 2048).  N = 2048: psi = psi_4096, x in [0, 2048) (the upper half is the negative)
        std::vector<u64> rp(2048);
        const Fq fp{f49::Q, 49};   // the torus kernels transform mod the 49-bit prime as well
        const u64 psi = c->wide() ? fp.pow(f49::GEN, (fp.q - 1) / 4096) : (u64)nttf::PSI_U;
        rp[0] = 1;
        for (uint32_t x = 1; x < 2048; x++) rp[x] = fp.mul(rp[x - 1], psi);
        const std::vector<double> rpd = to_centred_doubles(rp);
        if (hipMalloc(&c->d_root_pow, rpd.size() * 8) != hipSuccess) return bail("hipMalloc(root powers) failed");
        if (hipMemcpy(c->d_root_pow, rpd.data(), rpd.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail("hipMemcpy(root powers) failed");
    }
    if (c->t64() && (c->wide() || c->quad())) {   // tables of the quarter / eighth transforms: powers of zeta = exp(i pi / N)
        static_assert(ffte::ET_WORDS == fftq::QT_WORDS, "the two table sets share a size");
        std::vector<double> tq(fftq::QT_WORDS);
        if (c->quad()) ffte::build_tables(tq.data());
        else fftq::build_tables(tq.data());
        if (hipMalloc(&c->d_tw_fq, tq.size() * 8) != hipSuccess) return bail("hipMalloc(quarter-fft twiddles) failed");
        if (hipMemcpy(c->d_tw_fq, tq.data(), tq.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail("hipMemcpy(quarter-fft twiddles) failed");
    }
    if ((c->wide() || c->quad()) && !c->t64()) {
        const std::vector<double> tw = to_centred_doubles(c->quad() ? build_twiddles_quad(c->f) : build_twiddles_wide(c->f));
        if (hipMalloc(&c->d_tw_wide, tw.size() * 8) != hipSuccess) return bail("hipMalloc(wide twiddles) failed");
        if (hipMemcpy(c->d_tw_wide, tw.data(), tw.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail("hipMemcpy(wide twiddles) failed");
    }
    c->lut_cap = BMI_LUT_CAP;
    if (hipMalloc(&c->d_luts, (size_t)c->lut_cap * c->N * 8) != hipSuccess) return bail("hipMalloc(luts) failed");
    *out = c;
    return 0;
}

void bmi_ctx_destroy(bmi_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (void *p : {c->d_bsk, (void *)c->d_ksk, (void *)c->d_ks_bias, c->d_tw, c->d_luts, (void *)c->d_small,
                    (void *)c->d_io_a, (void *)c->d_io_b, (void *)c->d_io_ids, c->d_ks_partial,
                    (void *)c->d_ks_limbs, (void *)c->d_ks_digits, (void *)c->d_ks_sums, (void *)c->d_tw_half,
                    (void *)c->d_bsk_lat, (void *)c->d_tw_wide, (void *)c->d_bsk3_lat, (void *)c->d_root_pow,
                    (void *)c->d_tw_fft, (void *)c->d_bsk_fft, (void *)c->d_tw_fh, (void *)c->d_bsk_latf, (void *)c->d_tw_fq,
                    (void *)c->d_bsk_w, (void *)c->d_bsk_w2, (void *)c->d_zeta_pow})
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bmi_get_params(const bmi_ctx *c, bmi_params *out) {
    if (!c || !out) return -1;
    *out = c->P;
    return 0;
}

namespace {
int upload_eval_keys(bmi_ctx *c);
int upload_bsk3(bmi_ctx *c);
int build_exact_torus_copies(bmi_ctx *c, const u64 *d_std);
}

namespace {
int gen_eval_keys(bmi_ctx *c, uint64_t seed);
// fresh CSPRNG keys for this context (secret-class and public-class streams, never shared between them)
int rekey_csprng(bmi_ctx *c) {
    if (!c->rng_secret.fill_from_os() || !c->rng_public.fill_from_os())
        return fail(c, -2, "getrandom() failed: no entropy source for key generation");
    c->secure_rng = true;
    c->enc_counter = 0;   // a new key: the (key, nonce) pairs of the new streams have never been used
    return 0;
}
int gen_secret_keys(bmi_ctx *c, uint64_t seed) {
    const uint32_t n = c->P.n, kN = c->P.k * c->N;
    c->sk_small.assign(n, 0);
    c->sk_big.assign(kN, 0);
    Stream s1(seed, S_SK_SMALL, c->f), s2(seed, S_SK_BIG, c->f);
    RowRng r1(&s1, c->rng_secret, S_SK_SMALL, 0, c->secure_rng), r2(&s2, c->rng_secret, S_SK_BIG, 0, c->secure_rng);
    for (uint32_t i = 0; i < n; i++) c->sk_small[i] = r1.bit(i);
    for (uint32_t i = 0; i < kN; i++) c->sk_big[i] = r2.bit(i);
    return 0;
}
}

int bmi_keygen(bmi_ctx *c) {
    if (!c) return -1;
    if (int rc = rekey_csprng(c)) return rc;
    gen_secret_keys(c, 0);
    return gen_eval_keys(c, 0);
}

int bmi_keygen_insecure_deterministic(bmi_ctx *c, uint64_t seed) {
    if (!c) return -1;
    c->secure_rng = false;
    gen_secret_keys(c, seed);
    return gen_eval_keys(c, seed);
}

int bmi_keygen_from_secret(bmi_ctx *c, const uint64_t *sk_small, const uint64_t *sk_big, uint64_t seed) {
    if (!c || !sk_small || !sk_big) return -1;
    const uint32_t n = c->P.n, kN = c->P.k * c->N;
    for (uint32_t i = 0; i < n; i++)
        if (sk_small[i] > 1) return fail(c, -1, "secret keys are binary");
    for (uint32_t i = 0; i < kN; i++)
        if (sk_big[i] > 1) return fail(c, -1, "secret keys are binary");
    c->sk_small.assign(sk_small, sk_small + n);
    c->sk_big.assign(sk_big, sk_big + kN);
    if (seed == 0) {            // production: masks and noise from the CSPRNG
        if (int rc = rekey_csprng(c)) return rc;
    } else {
        c->secure_rng = false;  // test vectors: deterministic in seed
    }
    return gen_eval_keys(c, seed);
}

// ct words on the 2^64 torus <-> words mod q: the modulus switch round(x * q / 2^64) and back.  Context-free (host).
int bmi_torus64_to_field(uint32_t q_bits, const uint64_t *in, uint64_t words, uint64_t *out) {
    if (!in || !out || (q_bits != 64 && q_bits != 49)) return -1;
    const u64 q = q_bits == 49 ? f49::Q : gl::P;
    for (u64 i = 0; i < words; i++) {
        const unsigned __int128 t = (unsigned __int128)in[i] * q + ((unsigned __int128)1 << 63);
        const u64 r = (u64)(t >> 64);
        out[i] = r >= q ? r - q : r;
    }
    return 0;
}

int bmi_field_to_torus64(uint32_t q_bits, const uint64_t *in, uint64_t words, uint64_t *out) {
    if (!in || !out || (q_bits != 64 && q_bits != 49)) return -1;
    const u64 q = q_bits == 49 ? f49::Q : gl::P;
    for (u64 i = 0; i < words; i++) {
        if (in[i] >= q) return -1;
        const unsigned __int128 t = ((unsigned __int128)in[i] << 64) + (q >> 1);   // round(in * 2^64 / q) mod 2^64
        out[i] = (u64)(t / q);
    }
    return 0;
}

namespace {
// GGSW encryptions (standard domain) of the bits msg[0..) under the GLWE key, rows numbered g * rows + comp * l + lev:
// B = sum_j A_j * S_j + E by shifted adds (S binary); mask / noise streams as in oracle/tfhe_oracle.c ggsw_rows.
void gen_ggsw_rows(bmi_ctx *c, uint64_t seed, u64 mask_stream, u64 noise_stream, const std::vector<u64> &msg, u64 *out) {
    const bmi_params &P = c->P;
    const uint32_t N = c->N, k = P.k, l = P.bs_levels, rows = c->rows;
    const Stream sm_det(seed, mask_stream, c->f), se_det(seed, noise_stream, c->f);
    const Stream *smp = &sm_det, *sep = &se_det;
    const ChaKey *ksec = &c->rng_secret, *kpub = &c->rng_public;
    const bool secure = c->secure_rng;
    const Fq f = c->f;
    const u64 *skb = c->sk_big.data();
    const u64 *bits = msg.data();
    const double sigma = P.glwe_noise;
    const uint32_t bl = P.bs_base_log;
    parallel_for(msg.size() * rows, [=](size_t ir) {
        RowRng sm(smp, *kpub, mask_stream, ir, secure), se(sep, *ksec, noise_stream, ir, secure);
        const uint32_t i = (uint32_t)(ir / rows), r = (uint32_t)(ir % rows), comp = r / l, lev = r % l;
        u64 *row = out + ir * (k + 1) * N;
        u64 *B = row + (size_t)k * N;
        for (uint32_t x = 0; x < N; x++) B[x] = se.gauss((u64)ir * N + x, sigma);
        for (uint32_t j = 0; j < k; j++) {
            u64 *A = row + (size_t)j * N;
            for (uint32_t x = 0; x < N; x++) A[x] = sm.uniform(((u64)ir * (k + 1) + j) * N + x);
            for (uint32_t t = 0; t < N; t++)
                if (skb[(size_t)j * N + t]) add_shifted(f, B, A, t, N);
        }
        if (bits[i]) row[(size_t)comp * N] = f.add(row[(size_t)comp * N], (u64)1 << (f.bits - bl * (lev + 1)));
    });
}

// the unrolled bootstrap key of the secret keys held: per pair (s, s') = (s_2i, s_2i+1) the GGSW encryptions of s s', s (1 - s'),
// (1 - s) s'; an odd n is completed by s_n = 0 (host copy only; upload_bsk3 sends it to the device)
void gen_bsk3(bmi_ctx *c, uint64_t seed) {
    const uint32_t n = c->P.n;
    std::vector<u64> msg((size_t)c->pairs() * 3);
    for (uint32_t i = 0; i < c->pairs(); i++) {
        const u64 s1 = c->sk_small[2 * i], s2 = 2 * i + 1 < n ? c->sk_small[2 * i + 1] : 0;
        msg[3 * i] = s1 & s2;
        msg[3 * i + 1] = s1 & (s2 ^ 1);
        msg[3 * i + 2] = (s1 ^ 1) & s2;
    }
    c->bsk3_std.assign(c->bsk3_words(), 0);
    gen_ggsw_rows(c, seed, S_BSK3_MASK, S_BSK3_NOISE, msg, c->bsk3_std.data());
}

// evaluation keys for the secret keys held in c->sk_small / c->sk_big, deterministic in `seed`
int gen_eval_keys(bmi_ctx *c, uint64_t seed) {
    HIP_OK(c, hipSetDevice(c->device));
    const bmi_params &P = c->P;
    const uint32_t n = P.n, N = c->N, k = P.k, lk = P.ks_levels, rows = c->rows;
    c->seed = seed;
    if (!c->secure_rng) c->enc_counter = 0;   // deterministic test keys: encryption i of a key set is reproducible
    // --- bootstrap key: GGSW(s_i) rows, standard domain
    c->bsk_std.assign((size_t)n * rows * (k + 1) * N, 0);
    // (A torus key at 42 bits of precision is ROUNDED from this full-precision key at upload.  Drawing the masks on the
    // 2^22 grid instead would keep the rounding error away from the secret key - measured: output noise 2^-20 - but the
    // body's noise, std 2^20, is then rounded to the grid too and vanishes in 95 % of the words: not an LWE sample any more.)
    gen_ggsw_rows(c, seed, S_BSK_MASK, S_BSK_NOISE, c->sk_small, c->bsk_std.data());
    c->have_bsk3 = false;
    if (c->unroll == 2) gen_bsk3(c, seed);
    // --- keyswitch key
    c->ksk.assign((size_t)k * N * lk * (n + 1), 0);
    {
        const Stream sm_det(seed, S_KSK_MASK, c->f), se_det(seed, S_KSK_NOISE, c->f);
        const Stream *smp = &sm_det, *sep = &se_det;
        const ChaKey *ksec = &c->rng_secret, *kpub = &c->rng_public;
        const bool secure = c->secure_rng;
        const Fq f = c->f;
        const u64 *skb = c->sk_big.data();
        const u64 *sks = c->sk_small.data();
        u64 *ksk = c->ksk.data();
        const double sigma = P.lwe_noise;
        const uint32_t bl = P.ks_base_log;
        parallel_for((size_t)k * N * lk, [=](size_t jr) {
            RowRng sm(smp, *kpub, S_KSK_MASK, jr, secure), se(sep, *ksec, S_KSK_NOISE, jr, secure);
            const uint32_t j = (uint32_t)(jr / lk), lev = (uint32_t)(jr % lk);
            u64 *row = ksk + jr * (n + 1);
            u64 b = se.gauss(jr, sigma);
            for (uint32_t x = 0; x < n; x++) {
                row[x] = sm.uniform((u64)jr * (n + 1) + x);
                if (sks[x]) b = f.add(b, row[x]);
            }
            if (skb[j]) b = f.add(b, (u64)1 << (f.bits - bl * (lev + 1)));
            row[n] = b;
        });
    }
    c->have_secret = true;
    if (int rc = upload_eval_keys(c)) return rc;
    return c->unroll == 2 ? upload_bsk3(c) : 0;
}

// unrolled bootstrap key (host copy in c->bsk3_std) -> device, in the slot order of the latency kernel
int upload_bsk3(bmi_ctx *c) {
    HIP_OK(c, hipSetDevice(c->device));
    const size_t words = c->bsk3_words();
    if (c->bsk3_std.size() != words) return fail(c, -1, "no unrolled bootstrap key to upload");
    // torus: the key stored at bsk_prec bits IS the key from here on (what bmi_export_bsk_unrolled returns)
    if (c->t64() && c->bsk_prec != 64)
        for (u64 &w : c->bsk3_std) w = t64::round_key_word(w, c->bsk_prec);
    u64 *d_tmp = nullptr;
    HIP_OK(c, hipMalloc(&d_tmp, words * 8));
    const size_t lat_bytes = words * 8 * (c->t64() ? c->bsk_limbs() : 1);
    if (!c->d_bsk3_lat && hipMalloc(&c->d_bsk3_lat, lat_bytes) != hipSuccess) {
        (void)hipFree(d_tmp);
        return fail(c, -2, "hipMalloc(unrolled key) failed");
    }
    HIP_OK(c, hipMemcpy(d_tmp, c->bsk3_std.data(), words * 8, hipMemcpyHostToDevice));
    const bool ufft = c->t64() && bmit::shape_supported_unrolled_fft(c->bsk_prec, c->P.bs_levels, c->P.bs_base_log);
    const int rc = ufft ? bmit::launch_bsk_to_latf(d_tmp, c->d_bsk3_lat, c->d_tw_fh, (uint32_t)(words / c->N), c->bsk_prec, c->stream)   // slot-pair order of the half FFT
                   : c->t64() ? bmit::launch_bsk_to_lat(d_tmp, c->d_bsk3_lat, c->d_tw_half, (uint32_t)(words / c->N), c->bsk_prec, c->stream)
                   : c->wide() ? bmi49::launch_bsk_to_wide(d_tmp, c->d_bsk3_lat, (const double *)c->d_tw, c->d_tw_wide, (uint32_t)(words / c->N), c->stream)
                               : bmi49::launch_bsk_to_lat(d_tmp, c->d_bsk3_lat, c->d_tw_half, (uint32_t)(words / c->N), true, c->stream);
    if (rc) { (void)hipFree(d_tmp); return fail(c, -2, "bsk_to_lat (unrolled key) launch failed"); }
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipFree(d_tmp));
    c->have_bsk3 = true;
    return 0;
}

// 2^64 torus at N = 1024: the key copies of the exact-transform kernels (wave pairs: d_bsk; latency form: d_bsk_lat).  d_std = the
// standard-domain key on the device, or null to upload it from the host copy.  Contexts whose key has floating-point-transform
// copies build these on demand only (kernel variants 1 / 3 / 4).
int build_exact_torus_copies(bmi_ctx *c, const u64 *d_std) {
    if (c->d_bsk && c->d_bsk_lat) return 0;
    const size_t bsk_words = c->bsk_std.size();
    u64 *d_own = nullptr;
    if (!d_std) {
        HIP_OK(c, hipMalloc(&d_own, bsk_words * 8));
        HIP_OK(c, hipMemcpy(d_own, c->bsk_std.data(), bsk_words * 8, hipMemcpyHostToDevice));
        d_std = d_own;
    }
    auto bail = [&](const char *m) {
        if (d_own) (void)hipFree(d_own);
        return fail(c, -2, m);
    };
    if (!c->d_bsk) {
        if (hipMalloc(&c->d_bsk, bsk_words * 8 * c->bsk_limbs()) != hipSuccess) return bail("hipMalloc(torus limb key) failed");
        if (bmit::launch_bsk_to_limbs(d_std, (double *)c->d_bsk, (const double *)c->d_tw, (uint32_t)(bsk_words / c->N), c->bsk_prec, c->stream))
            return bail("bsk_to_limbs launch failed");
    }
    if (!c->d_bsk_lat) {
        if (hipMalloc(&c->d_bsk_lat, bsk_words * 8 * c->bsk_limbs()) != hipSuccess) return bail("hipMalloc(torus latency-kernel key) failed");
        if (bmit::launch_bsk_to_lat(d_std, c->d_bsk_lat, c->d_tw_half, (uint32_t)(bsk_words / c->N), c->bsk_prec, c->stream))
            return bail("bsk_to_lat (torus) launch failed");
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (d_own) (void)hipFree(d_own);
    return e == hipSuccess ? 0 : fail(c, -2, "hipStreamSynchronize failed");
}

// Evaluation keys (host copies in c->bsk_std / c->ksk) -> device: bootstrap key to the NTT domain (both layouts for
// the 49-bit field), keyswitch key in word form (+ bias vector) and in limb form.
int upload_eval_keys(bmi_ctx *c) {
    const bmi_params &P = c->P;
    const uint32_t n = P.n, N = c->N, k = P.k, lk = P.ks_levels;
    // --- upload: bootstrap key -> NTT domain on the GPU; keyswitch key with padded rows
    if (c->t64() && c->bsk_prec != 64) {
        // key stored at 48 / 42 bits of precision: every word rounded (half up, as a signed integer) to a multiple of 2^16 / 2^22.
        // The rounded key IS the key from here on (bmi_export_keys returns it), so every consumer agrees on it.
        for (u64 &w : c->bsk_std) w = t64::round_key_word(w, c->bsk_prec);
    }
    const size_t bsk_words = c->bsk_std.size();
    if (!c->d_bsk && !c->wide() && !c->quad() && !c->t64()) HIP_OK(c, hipMalloc(&c->d_bsk, bsk_words * 8));
    u64 *d_tmp = nullptr;
    HIP_OK(c, hipMalloc(&d_tmp, bsk_words * sizeof(u64)));
    HIP_OK(c, hipMemcpy(d_tmp, c->bsk_std.data(), bsk_words * sizeof(u64), hipMemcpyHostToDevice));
    int rc = 0;
    if (c->t64() && (c->wide() || c->quad())) {  // 2^64 torus at N = 2048 / 4096: ONE key copy, two limb polynomials per key polynomial in the order of bmi_kernels_t64w.hip / t64q.hip
        if (c->d_bsk_w) { (void)hipFree(c->d_bsk_w); c->d_bsk_w = nullptr; }
        if (hipMalloc(&c->d_bsk_w, bsk_words * 8 * c->bsk_limbs()) != hipSuccess) {
            (void)hipFree(d_tmp);
            return fail(c, -2, "hipMalloc(torus N = 2048 / 4096 key) failed");
        }
        rc = (c->quad() ? bmit::launch_bsk_to_quad : bmit::launch_bsk_to_wide)(d_tmp, c->d_bsk_w, c->d_tw_fq, (uint32_t)(bsk_words / N), c->bsk_prec, c->stream);
        if (!rc && c->wide()) {   // ... and at N = 2048 the copy of the throughput form (two ciphertexts per workgroup: batches beyond 256)
            if (c->d_bsk_w2) { (void)hipFree(c->d_bsk_w2); c->d_bsk_w2 = nullptr; }
            if (hipMalloc(&c->d_bsk_w2, bsk_words * 8 * c->bsk_limbs()) != hipSuccess) {
                (void)hipFree(d_tmp);
                return fail(c, -2, "hipMalloc(torus N = 2048 throughput-form key) failed");
            }
            rc = bmit::launch_bsk_to_wide2(d_tmp, c->d_bsk_w2, c->d_tw_fq, (uint32_t)(bsk_words / N), c->bsk_prec, c->stream);
        }
        if (rc) { (void)hipFree(d_tmp); return fail(c, -2, "bsk_to_wide (torus) launch failed"); }
    } else if (c->t64()) {  // 2^64 torus at N = 1024: bsk_limbs transform-domain limb polynomials per key polynomial
        for (void **p : {&c->d_bsk, (void **)&c->d_bsk_lat, (void **)&c->d_bsk_fft, (void **)&c->d_bsk_latf})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        const bool fft = c->d_tw_fft && bmit::shape_supported_fft(c->bsk_prec, P.bs_levels, P.bs_base_log);
        if (fft) {
            // the kernels auto dispatch runs on this key (48 bits, base 2^10): the wave-pair and the latency form of
            // bmi_kernels_t64f.hip, exact limb products through the floating-point transform.  The two copies of the
            // exact-transform kernels (variants 1 / 3 / 4: A/B only) are built when such a variant is first pinned.
            if (hipMalloc(&c->d_bsk_fft, bsk_words * 8 * c->bsk_limbs()) != hipSuccess) {
                (void)hipFree(d_tmp);
                return fail(c, -2, "hipMalloc(torus fft key) failed");
            }
            rc = bmit::launch_bsk_to_fft(d_tmp, c->d_bsk_fft, c->d_tw_fft, (uint32_t)(bsk_words / N), c->bsk_prec, c->stream);
            if (rc) { (void)hipFree(d_tmp); return fail(c, -2, "bsk_to_fft (torus) launch failed"); }
            if (hipMalloc(&c->d_bsk_latf, bsk_words * 8 * c->bsk_limbs()) != hipSuccess) {
                (void)hipFree(d_tmp);
                return fail(c, -2, "hipMalloc(torus fft latency-kernel key) failed");
            }
            rc = bmit::launch_bsk_to_latf(d_tmp, c->d_bsk_latf, c->d_tw_fh, (uint32_t)(bsk_words / N), c->bsk_prec, c->stream);
            if (rc) { (void)hipFree(d_tmp); return fail(c, -2, "bsk_to_latf (torus) launch failed"); }
        } else if (bmit::shape_supported(c->bsk_prec, P.bs_levels, P.bs_base_log)) {
            HIP_OK(c, hipStreamSynchronize(c->stream));
            if (int rc2 = build_exact_torus_copies(c, d_tmp)) { (void)hipFree(d_tmp); return rc2; }
        }   // (else: a precision that only the unrolled kernel takes - 42 bits at base 2^10: no plain key copy)
    } else if (c->wide() || c->quad()) {  // N = 2048 / 4096: one key copy, in the slot order of k_blind_rotate_wide49 / quad49
        if (!c->d_bsk_lat && hipMalloc(&c->d_bsk_lat, bsk_words * 8) != hipSuccess) {
            (void)hipFree(d_tmp);
            return fail(c, -2, "hipMalloc(wide key) failed");
        }
        rc = (c->quad() ? bmi49::launch_bsk_to_quad : bmi49::launch_bsk_to_wide)(d_tmp, c->d_bsk_lat, (const double *)c->d_tw, c->d_tw_wide, (uint32_t)(bsk_words / N), c->stream);
        if (rc) { (void)hipFree(d_tmp); return fail(c, -2, "bsk_to_wide launch failed"); }
    } else {
        rc = c->f64() ? bmi49::launch_bsk_to_ntt(d_tmp, (double *)c->d_bsk, (const double *)c->d_tw, (uint32_t)(bsk_words / N), c->stream)
                      : bmi::launch_bsk_to_ntt(d_tmp, (u64 *)c->d_bsk, (const u64 *)c->d_tw, (uint32_t)(bsk_words / N), c->stream);
        if (rc) { (void)hipFree(d_tmp); return fail(c, -2, "bsk_to_ntt launch failed"); }
        if (c->f64()) {  // second copy of the key, in the slot order of the split-transform latency kernel
            if (!c->d_bsk_lat && hipMalloc(&c->d_bsk_lat, bsk_words * 8) != hipSuccess) {
                (void)hipFree(d_tmp);
                return fail(c, -2, "hipMalloc(latency-kernel key) failed");
            }
            rc = bmi49::launch_bsk_to_lat(d_tmp, c->d_bsk_lat, c->d_tw_half, (uint32_t)(bsk_words / N), false, c->stream);
            if (rc) { (void)hipFree(d_tmp); return fail(c, -2, "bsk_to_lat launch failed"); }
        }
    }
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipFree(d_tmp));
    const size_t ksk_rows = (size_t)k * N * lk;
    if (!c->d_ksk) HIP_OK(c, hipMalloc(&c->d_ksk, ksk_rows * c->ks_stride * sizeof(u64)));
    HIP_OK(c, hipMemset(c->d_ksk, 0, ksk_rows * c->ks_stride * sizeof(u64)));
    HIP_OK(c, hipMemcpy2D(c->d_ksk, c->ks_stride * sizeof(u64), c->ksk.data(), (n + 1) * sizeof(u64),
                          (n + 1) * sizeof(u64), ksk_rows, hipMemcpyHostToDevice));
    // bias vector of the unsigned-digit keyswitch: (B/2) * sum_rows ksk[row][col]
    {
        std::vector<u64> bias(c->ks_stride, 0);
        const u64 half = (u64)1 << (P.ks_base_log - 1);
        for (size_t r = 0; r < ksk_rows; r++) {
            const u64 *row = c->ksk.data() + r * (n + 1);
            for (uint32_t x = 0; x <= n; x++) bias[x] = c->f.add(bias[x], row[x]);
        }
        for (uint32_t x = 0; x <= n; x++) bias[x] = c->f.mul(bias[x], half);
        if (!c->d_ks_bias) HIP_OK(c, hipMalloc(&c->d_ks_bias, c->ks_stride * sizeof(u64)));
        HIP_OK(c, hipMemcpy(c->d_ks_bias, bias.data(), c->ks_stride * sizeof(u64), hipMemcpyHostToDevice));
    }
    // limb-wise copy of the keyswitch key for the matrix-core keyswitch (int8 operands, int32 sums: the digits must
    // fit int8 and a column sum of rows * (B/2) * 128 must stay below 2^31)
    c->ks_mfma_ok = ksk_rows % 32 == 0 && P.ks_levels <= 16 && P.ks_base_log <= 7 &&
                    (ksk_rows << (P.ks_base_log - 1)) < ((size_t)1 << 24);
    if (c->ks_mfma_ok) {
        const uint32_t cbs = (n + 1 + 31) / 32;
        const size_t bytes = (size_t)cbs * (ksk_rows / 32) * c->ks_limbs() * 1024;
        if (!c->d_ks_limbs) HIP_OK(c, hipMalloc(&c->d_ks_limbs, bytes));
        rc = (c->f64() ? bmi49::launch_ksk_to_limbs : (c->t64() ? bmit::launch_ksk_to_limbs : bmi::launch_ksk_to_limbs))(c->d_ksk, c->d_ks_limbs, (uint32_t)ksk_rows, n,
                                                                                c->ks_stride, c->stream);
        if (rc) return fail(c, -2, "ksk_to_limbs launch failed");
        HIP_OK(c, hipStreamSynchronize(c->stream));
    }
    c->have_keys = true;
    return 0;
}
}  // namespace

int bmi_import_keys(bmi_ctx *c, const uint64_t *sk_small, const uint64_t *sk_big, const uint64_t *bsk, const uint64_t *ksk) {
    if (!c || !bsk || !ksk) return -1;
    if ((sk_small == nullptr) != (sk_big == nullptr)) return fail(c, -1, "pass both secret keys or neither");
    HIP_OK(c, hipSetDevice(c->device));
    const bmi_params &P = c->P;
    const size_t bsk_words = (size_t)P.n * c->rows * (P.k + 1) * c->N, ksk_words = (size_t)c->big_n * P.ks_levels * (P.n + 1);
    for (size_t i = 0; i < bsk_words; i++)
        if (!c->f.canonical(bsk[i])) return fail(c, -1, "bootstrap key word not reduced mod q");
    for (size_t i = 0; i < ksk_words; i++)
        if (!c->f.canonical(ksk[i])) return fail(c, -1, "keyswitch key word not reduced mod q");
    c->have_keys = false;
    c->have_bsk3 = false;   // an unrolled key belongs to the key set it was generated with: import it again (bmi_import_bsk_unrolled)
    c->bsk_std.assign(bsk, bsk + bsk_words);
    c->ksk.assign(ksk, ksk + ksk_words);
    c->have_secret = sk_small != nullptr;
    if (c->have_secret) {
        for (uint32_t i = 0; i < P.n; i++)
            if (sk_small[i] > 1) return fail(c, -1, "secret keys are binary");
        for (uint32_t i = 0; i < c->big_n; i++)
            if (sk_big[i] > 1) return fail(c, -1, "secret keys are binary");
        c->sk_small.assign(sk_small, sk_small + P.n);
        c->sk_big.assign(sk_big, sk_big + c->big_n);
    } else {
        c->sk_small.clear();
        c->sk_big.clear();
    }
    if (int rc = rekey_csprng(c)) return rc;   // encryptions under an imported key set draw fresh CSPRNG randomness
    return upload_eval_keys(c);
}

int bmi_export_keys(const bmi_ctx *c, uint64_t *sk_small, uint64_t *sk_big, uint64_t *bsk, uint64_t *ksk) {
    if (!c) return -1;
    if (!c->have_keys) return fail(c, -1, "no keys: call bmi_keygen first");
    if ((sk_small || sk_big) && !c->have_secret) return fail(c, -1, "evaluation-only context: it holds no secret key");
    if (sk_small) std::memcpy(sk_small, c->sk_small.data(), c->sk_small.size() * 8);
    if (sk_big) std::memcpy(sk_big, c->sk_big.data(), c->sk_big.size() * 8);
    if (bsk) std::memcpy(bsk, c->bsk_std.data(), c->bsk_std.size() * 8);
    if (ksk) std::memcpy(ksk, c->ksk.data(), c->ksk.size() * 8);
    return 0;
}

int bmi_key_bytes(const bmi_ctx *c, uint64_t *bsk_bytes, uint64_t *ksk_bytes) {
    if (!c) return -1;
    if (bsk_bytes) {   // every resident device copy (one per kernel family that is selectable on this context), plus the unrolled key
        const u64 one = (u64)c->P.n * c->rows * (c->P.k + 1) * c->N * 8 * (c->t64() ? c->bsk_limbs() : 1);
        u64 copies = 0;
        for (const void *p : {(const void *)c->d_bsk, (const void *)c->d_bsk_lat, (const void *)c->d_bsk_fft, (const void *)c->d_bsk_latf,
                              (const void *)c->d_bsk_w, (const void *)c->d_bsk_w2})
            copies += p != nullptr;
        *bsk_bytes = one * copies + (c->have_bsk3 ? (u64)c->bsk3_words() * 8 * (c->t64() ? c->bsk_limbs() : 1) : 0);
    }
    if (ksk_bytes) *ksk_bytes = (u64)c->big_n * c->P.ks_levels * (c->P.n + 1) * 8;
    return 0;
}

int bmi_encrypt(bmi_ctx *c, const int64_t *msgs, uint32_t count, uint32_t delta_log, uint64_t *ct_out) {
    if (!c || !msgs || !ct_out) return -1;
    if (!c->have_keys) return fail(c, -1, "no keys: call bmi_keygen first");
    if (!c->have_secret) return fail(c, -1, "evaluation-only context: it holds no secret key");
    if (delta_log >= c->f.bits - 1) return fail(c, -1, "delta_log out of range");
    const Stream sm_det(c->seed, S_ENC_MASK, c->f), se_det(c->seed, S_ENC_NOISE, c->f);
    const Stream *smp = &sm_det, *sep = &se_det;
    const ChaKey *ksec = &c->rng_secret, *kpub = &c->rng_public;
    const bool secure = c->secure_rng;
    const Fq f = c->f;
    const uint32_t dim = c->big_n;
    const u64 first = c->enc_counter;   // secure contexts: never reset while the CSPRNG keys live (one nonce per ciphertext)
    const u64 *key = c->sk_big.data();
    const double sigma = c->P.glwe_noise;
    parallel_for(count, [=](size_t i) {
        RowRng sm(smp, *kpub, S_ENC_MASK, first + i, secure), se(sep, *ksec, S_ENC_NOISE, first + i, secure);
        u64 *ct = ct_out + i * (dim + 1);
        const u64 torus = f.mul(f.from_i64(msgs[i]), (u64)1 << delta_log);
        u64 b = f.add(torus, se.gauss(first + i, sigma));
        for (uint32_t x = 0; x < dim; x++) {
            ct[x] = sm.uniform((first + i) * (u64)(dim + 1) + x);
            if (key[x]) b = f.add(b, ct[x]);
        }
        ct[dim] = b;
    });
    c->enc_counter += count;
    return 0;
}

int bmi_phase(const bmi_ctx *c, const uint64_t *ct_in, uint32_t count, uint64_t *phase) {
    if (!c || !ct_in || !phase) return -1;
    if (!c->have_keys) return fail(c, -1, "no keys: call bmi_keygen first");
    if (!c->have_secret) return fail(c, -1, "evaluation-only context: it holds no secret key");
    const uint32_t dim = c->big_n;
    const u64 *key = c->sk_big.data();
    for (uint32_t i = 0; i < count; i++) {
        const u64 *ct = ct_in + (size_t)i * (dim + 1);
        u64 p = ct[dim];
        for (uint32_t x = 0; x < dim; x++)
            if (key[x]) p = c->f.sub(p, ct[x]);
        phase[i] = p;
    }
    return 0;
}

int bmi_decrypt(const bmi_ctx *c, const uint64_t *ct_in, uint32_t count, uint32_t delta_log, int64_t *msgs) {
    if (!c || !ct_in || !msgs) return -1;
    if (delta_log == 0 || delta_log >= c->f.bits - 1) return fail(c, -1, "delta_log out of range");
    std::vector<u64> ph(count);
    int rc = bmi_phase(c, ct_in, count, ph.data());
    if (rc) return rc;
    for (uint32_t i = 0; i < count; i++) {
        const i64 v = c->f.centered(ph[i]);
        msgs[i] = (v >> delta_log) + ((v >> (delta_log - 1)) & 1);
    }
    return 0;
}

int bmi_lut_register(bmi_ctx *c, const int64_t *table, uint32_t msg_bits, uint32_t out_delta_log, uint32_t *lut_id) {
    if (!c || !table || !lut_id) return -1;
    const uint32_t N = c->N;
    if (msg_bits == 0 || (1u << msg_bits) * 2 > N) return fail(c, -1, "msg_bits out of range for N");
    if (out_delta_log >= c->f.bits - 1) return fail(c, -1, "out_delta_log out of range");
    if (c->t64() && c->bsk_prec != 64 && out_delta_log < (uint32_t)(64 - c->bsk_prec))
        // the torus kernels on a rounded key (48 / 46 / 42 bits) keep the accumulator as a multiple of 2^16 / 2^18 / 2^22: a table
        // encoded below that scale would lose its low bits - and sits ~2^40 below the noise anyway.  (The exact key has no such grid.)
        return fail(c, -1, "out_delta_log below 64 - (bootstrap-key precision) on the 2^64 torus: test polynomials must be multiples of the key's grid");
    if (c->n_luts == c->lut_cap) return fail(c, -1, "LUT table full");
    // Signed messages on the whole negacyclic circle: box width w = N / 2^p, boxes centred on m*w.
    const uint32_t M = 1u << msg_bits, Mh = M >> 1, w = N >> msg_bits, half = w >> 1;
    std::vector<u64> tv(N);
    for (uint32_t j = 0; j < N; j++) {
        const uint32_t box = (j + half) / w;  // 0 .. M
        i64 f;
        bool negate;
        if (box < Mh) { f = table[box + Mh]; negate = false; }        // m = box >= 0
        else if (box < M) { f = table[box - Mh]; negate = true; }      // m = box - M < 0, reached as -X^(j-N)
        else { f = table[Mh]; negate = true; }                         // m = 0 from below
        const u64 v = c->f.mul(c->f.from_i64(f), (u64)1 << out_delta_log);
        tv[j] = negate ? c->f.neg(v) : v;
    }
    HIP_OK(c, hipSetDevice(c->device));
    {
        const std::vector<double> tvd = c->f64() ? to_centred_doubles(tv) : std::vector<double>();
        const void *src = c->f64() ? (const void *)tvd.data() : (const void *)tv.data();
        HIP_OK(c, hipMemcpy((char *)c->d_luts + (size_t)c->n_luts * N * 8, src, N * 8, hipMemcpyHostToDevice));
    }
    c->luts_host.push_back(std::move(tv));
    *lut_id = c->n_luts++;
    return 0;
}

int bmi_lut_get(const bmi_ctx *c, uint32_t lut_id, uint64_t *test_vector) {
    if (!c || !test_vector) return -1;
    if (lut_id >= c->n_luts) return fail(c, -1, "unknown LUT id");
    std::memcpy(test_vector, c->luts_host[lut_id].data(), c->N * 8);
    return 0;
}

int bmi_set_bsk_precision(bmi_ctx *c, uint32_t bits) {
    if (!c) return -1;
    if (!c->t64()) return fail(c, -1, "the bootstrap-key precision option exists on the 2^64 torus only");
    if (!t64::precision_ok((int)bits)) return fail(c, -1, "bootstrap-key precision must be 64 (exact), 48, 46, 44 or 42 bits");
    if (c->have_keys) return fail(c, -1, "set the bootstrap-key precision before generating or importing keys");
    if (c->wide() || c->quad()) {
        if (!(c->quad() ? bmit::shape_supported_quad : bmit::shape_supported_wide)((int)bits, c->P.bs_levels, c->P.bs_base_log))
            return fail(c, -1, "at N = 2048 the torus kernel takes the key at 46 bits of precision (two 23-bit limbs) only, at N = 4096 at 44 bits (two "
                               "22-bit limbs): the lengths at which the floating-point transform's error bound certifies the rounding "
                               "(fft_quarter_f64.hpp, fft_eighth_f64.hpp)");
        c->bsk_prec = (int)bits;
        c->bsk_prec_explicit = true;
        return 0;
    }
    // 42 bits at base 2^10: in unrolled mode only (bmi_set_bsk_unroll first - which selects it by itself when no precision was set)
    const bool unrolled_fft = c->unroll == 2 && bmit::shape_supported_unrolled_fft((int)bits, c->P.bs_levels, c->P.bs_base_log);
    if (!bmit::shape_supported((int)bits, c->P.bs_levels, c->P.bs_base_log) && !unrolled_fft)
        return fail(c, -1, "no kernel for this precision at this decomposition: base 2^10 takes 48 (default) or 64 bits - and 42 bits for the "
                           "unrolled floating-point-transform kernel (bmi_set_bsk_unroll) -, base 2^15 takes 64 (default) or 42 bits (a limb sum "
                           "must stay below p/2: t64_common.hpp)");
    if (c->unroll == 2 && !unrolled_fft && !bmit::shape_supported_unrolled((int)bits, c->P.bs_levels, c->P.bs_base_log))
        return fail(c, -1, "the unrolled torus kernels take the key at 48 bits (exact transform) or 42 bits (floating-point transform) at base 2^10");
    c->bsk_prec = (int)bits;
    c->bsk_prec_explicit = true;
    return 0;
}

int bmi_get_bsk_precision(const bmi_ctx *c, uint32_t *bits) {
    if (!c || !bits) return -1;
    *bits = c->t64() ? (uint32_t)c->bsk_prec : 64u;
    return 0;
}

int bmi_set_bsk_unroll(bmi_ctx *c, uint32_t factor) {
    if (!c) return -1;
    if (factor != 1 && factor != 2) return fail(c, -1, "the unrolling factor is 1 or 2");
    if (factor == 2 && !((c->f64() && !c->quad()) || c->t64()))
        return fail(c, -1, "bootstrap-key unrolling has HIP kernels for the 49-bit field at N = 1024 and N = 2048 and for the 2^64 torus only");
    if (factor == 2 && c->t64() && (c->wide() || c->quad()))
        return fail(c, -1, "bootstrap-key unrolling has no HIP kernel on the 2^64 torus at N = 2048 / 4096");
    if (c->t64() && !c->bsk_prec_explicit && !c->have_keys) {
        // the precision nobody chose follows the mode: the unrolled step runs through the floating-point transform on a key stored
        // at 42 bits (k_blind_rotate_lat2u_t64f: 2.9 ms per bootstrap against 3.3 of the exact-transform kernel on the 48-bit key,
        // which bmi_set_bsk_precision(ctx, 48) still selects); back to the set's default when unrolling is switched off
        if (factor == 2 && bmit::shape_supported_unrolled_fft(42, c->P.bs_levels, c->P.bs_base_log)) c->bsk_prec = 42;
        if (factor == 1) c->bsk_prec = default_bsk_precision(c->P);
    }
    if (factor == 2 && c->t64() && !bmit::shape_supported_unrolled(c->bsk_prec, c->P.bs_levels, c->P.bs_base_log) &&
        !bmit::shape_supported_unrolled_fft(c->bsk_prec, c->P.bs_levels, c->P.bs_base_log))
        return fail(c, -1, "on the 2^64 torus the unrolled kernels take the key at 48 or 42 bits at base 2^10 (the default torus set): the limb sums of "
                           "its three scaled products must stay below p/2");
    if (factor == 2 && c->wide() && c->P.bs_levels > 2)
        return fail(c, -1, "at N = 2048 the unrolled kernel exists for l <= 2 (at l = 3 it would not fit the registers: the plain kernel is faster)");
    c->unroll = factor;
    if (factor == 2 && c->have_keys && !c->have_bsk3) {
        if (!c->have_secret) return 0;   // evaluation-only context: the key arrives through bmi_import_bsk_unrolled
        // the unrolled key of the secret keys already held: fresh masks and noise (CSPRNG, or the seeded test streams)
        gen_bsk3(c, c->seed);
        return upload_bsk3(c);
    }
    return 0;
}

int bmi_import_bsk_unrolled(bmi_ctx *c, const uint64_t *bsk3) {
    if (!c || !bsk3) return -1;
    if (!((c->f64() && !c->quad()) || (c->t64() && !c->wide() && !c->quad())))
        return fail(c, -1, "bootstrap-key unrolling has HIP kernels for the 49-bit field at N = 1024 and N = 2048 and for the 2^64 torus at N = 1024 only");
    if (!c->have_keys) return fail(c, -1, "no keys: import or generate the key set first");
    const size_t words = c->bsk3_words();
    for (size_t i = 0; i < words; i++)
        if (!c->f.canonical(bsk3[i])) return fail(c, -1, "unrolled bootstrap key word not reduced mod q");
    c->bsk3_std.assign(bsk3, bsk3 + words);
    return upload_bsk3(c);
}

int bmi_export_bsk_unrolled(const bmi_ctx *c, uint64_t *bsk3) {
    if (!c || !bsk3) return -1;
    if (!c->have_bsk3) return fail(c, -1, "no unrolled bootstrap key: bmi_set_bsk_unroll(ctx, 2) before keygen, or import one");
    std::memcpy(bsk3, c->bsk3_std.data(), c->bsk3_std.size() * 8);
    return 0;
}

int bmi_set_keyswitch_variant(bmi_ctx *c, int variant) {
    if (!c) return -1;
    if (variant < 0 || variant > 1) return fail(c, -1, "keyswitch variant must be 0 or 1");
    c->ks_variant = variant;
    return 0;
}

int bmi_set_kernel_variant(bmi_ctx *c, int variant) {
    if (!c) return -1;
    if (variant < 0 || variant > 6) return fail(c, -1, "variant must be 0..6");
    if (variant >= 5 && !c->t64()) return fail(c, -1, "kernel variants 5 and 6 (floating-point transform) exist on the 2^64 torus only");
#ifndef BMI_AB_KERNELS
    if (c->f64() && !c->wide() && !c->quad() && (variant == 1 || variant == 4))
        return fail(c, -1, "kernel variants 1 and 4 of the 49-bit field (the predecessors of the wave-pair and latency kernels) are not "
                           "in the product build: make -C csrc ab builds libbmi_tfhe_ab.so with them");
#endif
    c->variant = variant;
    return 0;
}

// ------------------------------------------------------------------------------- the hot path
namespace {
// grows a device scratch buffer (never shrinks); the old one may still be in use by queued work -> synchronise first
int ensure_bytes(bmi_ctx *c, void **p, size_t *cap, size_t need, size_t floor_bytes) {
    if (need <= *cap) return 0;
    if (*p) {
        HIP_OK(c, hipDeviceSynchronize());
        HIP_OK(c, hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    const size_t bytes = std::max(need, floor_bytes);
    HIP_OK(c, hipMalloc(p, bytes));
    *cap = bytes;
    return 0;
}

// K-slices of the matrix-core keyswitch: enough wavefronts (tiles x column blocks x slices) to fill 1024 SIMDs twice
uint32_t ks_mfma_slices(const bmi_ctx *c, uint32_t count) {
    const uint32_t tiles = (count + 31) / 32, cbs = (c->P.n + 1 + 31) / 32;
    const uint32_t ksteps = c->big_n * c->P.ks_levels / 32;
    uint32_t slices = 1;
    while (slices < 64 && slices * 2 <= ksteps && ksteps % (slices * 2) == 0 && (size_t)tiles * cbs * slices < 2048) slices *= 2;
    return slices;
}

int ensure_ks_mfma(bmi_ctx *c, uint32_t count) {
    const uint32_t cbs = (c->P.n + 1 + 31) / 32;
    const size_t dig = (size_t)count * c->big_n * c->P.ks_levels;
    const size_t sums = (size_t)ks_mfma_slices(c, count) * count * c->ks_limbs() * cbs * 32 * sizeof(int);
    int rc = ensure_bytes(c, (void **)&c->d_ks_digits, &c->ks_digits_bytes, dig, (size_t)8 << 20);
    if (rc) return rc;
    return ensure_bytes(c, (void **)&c->d_ks_sums, &c->ks_sums_bytes, sums, (size_t)32 << 20);
}

int keyswitch_mfma(bmi_ctx *c, const uint64_t *d_in, uint32_t count, uint64_t *d_small, hipStream_t stream) {
    if (count == 0) return 0;
    int rc = ensure_ks_mfma(c, count);
    if (rc) return rc;
    rc = (c->f64() ? bmi49::launch_keyswitch_mfma : (c->t64() ? bmit::launch_keyswitch_mfma : bmi::launch_keyswitch_mfma))(
        d_in, c->d_ks_limbs, c->d_ks_digits, c->d_ks_sums, d_small, ks_mfma_slices(c, count), count, c->P.n, c->big_n,
        c->P.ks_levels, c->P.ks_base_log, stream);
    return rc ? fail(c, -2, std::string("keyswitch (matrix cores) launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
}
}  // namespace

int bmi_keyswitch_batch(bmi_ctx *c, const uint64_t *d_in, uint32_t count, uint64_t *d_small, void *stream) {
    if (!c || (count && (!d_in || !d_small))) return -1;
    if (!c->have_keys) return fail(c, -1, "no keys: call bmi_keygen first");
    if (count == 0) return 0;
    HIP_OK(c, hipSetDevice(c->device));
    if (c->ks_variant == 0 && c->ks_mfma_ok && count >= BMI_KS_MFMA_MIN) return keyswitch_mfma(c, d_in, count, d_small, (hipStream_t)stream);
    if (c->P.n + 1 > 3 * 256)   // ks_lincomb.hpp: KS_COLS x KS_THREADS output columns per workgroup
        return fail(c, -1, "the scalar keyswitch kernel takes n <= 767; this parameter set needs the matrix-core form");
    // Scalar form.  The row walk (k*N*levels rows) of one workgroup is the latency of a small batch, so it is split over
    // `slices` workgroups per tile of 8 ciphertexts (partial 128-bit sums + a reduce kernel) until the launch
    // has ~1024 workgroups; large batches fill the chip with one slice.
    const uint32_t tiles = (count + 7) / 8;
    uint32_t slices = 1;
    if (c->variant == 0 || c->variant == 2)
        while (slices < 64 && tiles * slices * 2 <= 1024) slices *= 2;
    const size_t need = (size_t)slices * count * c->ks_stride * 16;
    if (slices > 1 && need > c->ks_partial_bytes) {
        if (c->d_ks_partial) {
            HIP_OK(c, hipDeviceSynchronize());
            HIP_OK(c, hipFree(c->d_ks_partial));
            c->d_ks_partial = nullptr;
        }
        const size_t cap = std::max(need, (size_t)96 << 20);
        HIP_OK(c, hipMalloc(&c->d_ks_partial, cap));
        c->ks_partial_bytes = cap;
    }
    int rc = (c->f64() ? bmi49::launch_keyswitch : (c->t64() ? bmit::launch_keyswitch : bmi::launch_keyswitch))(
        d_in, c->d_ksk, c->d_ks_bias, d_small, slices > 1 ? c->d_ks_partial : nullptr, slices, count, c->P.n, c->big_n,
        c->P.ks_levels, c->P.ks_base_log, c->ks_stride, (hipStream_t)stream);
    return rc ? fail(c, -2, std::string("keyswitch launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
}

int bmi_blind_rotate_batch(bmi_ctx *c, const uint64_t *d_small, const uint32_t *d_lut_ids, uint32_t count,
                           uint64_t *d_out, void *stream) {
    if (!c || (count && (!d_small || !d_lut_ids || !d_out))) return -1;
    if (!c->have_keys) return fail(c, -1, "no keys: call bmi_keygen first");
    HIP_OK(c, hipSetDevice(c->device));
    // variant 0 = auto: the latency kernel (one workgroup per ciphertext) while the batch cannot fill the chip
    // with wave-pair work, the throughput kernel beyond that (49-bit field: the exchange-once form).
    const bool latency = c->variant == 2 || (c->variant == 0 && count <= c->lat_threshold);
    int rc;
    if (c->t64() && c->wide() && c->d_bsk_w2 && (c->variant == 1 || c->variant == 3 || (c->variant == 0 && count > 256))) {
        // N = 2048, batches beyond one round of 256: two ciphertexts per workgroup sharing the key words (variants 1 / 3 pin it, 2 pins the
        // one-ciphertext form below); the same words
        rc = bmit::launch_blind_rotate_wide2(d_small, d_lut_ids, (const u64 *)c->d_luts, c->d_bsk_w2, c->d_tw_fq, d_out, count, c->P.n, c->bsk_prec,
                                             c->P.bs_levels, c->P.bs_base_log, (hipStream_t)stream);
        return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
    }
    if (c->t64() && (c->wide() || c->quad())) {   // 2^64 torus at N = 2048 / 4096: one workgroup per ciphertext (N = 2048: up to 256 ciphertexts, see above)
        rc = (c->quad() ? bmit::launch_blind_rotate_quad : bmit::launch_blind_rotate_wide)(d_small, d_lut_ids, (const u64 *)c->d_luts, c->d_bsk_w, c->d_tw_fq, d_out, count, c->P.n,
                                            c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, nullptr, (hipStream_t)stream);
        return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
    }
    if (c->t64()) {   // 2^64 torus: latency kernel (one workgroup per ciphertext) for small batches, wave pairs beyond
        if (c->unroll == 2) {   // unrolled key: one kernel (one workgroup per ciphertext) for every batch size
            if (!c->have_bsk3) return fail(c, -1, "unrolling selected but the context holds no unrolled key: generate keys after bmi_set_bsk_unroll, or bmi_import_bsk_unrolled");
            // the floating-point-transform route (42-bit key): one ciphertext per workgroup up to a full round of 256, two per workgroup
            // (key words shared in registers: half the key bytes per bootstrap) beyond - the same words either way; variant 2 pins
            // the former, variants 1 / 3 the latter
            const bool ufft = bmit::shape_supported_unrolled_fft(c->bsk_prec, c->P.bs_levels, c->P.bs_base_log);
            const bool two = ufft && (c->variant == 1 || c->variant == 3 || (c->variant == 0 && count > 256));
            rc = two ? bmit::launch_blind_rotate_tp2u_fft(d_small, d_lut_ids, (const u64 *)c->d_luts, c->d_bsk3_lat, c->d_tw_fh, c->d_zeta_pow, d_out,
                                                          count, c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, (hipStream_t)stream)
                 : ufft
                     ? bmit::launch_blind_rotate_lat2u_fft(d_small, d_lut_ids, (const u64 *)c->d_luts, c->d_bsk3_lat, c->d_tw_fh, c->d_zeta_pow, d_out,
                                                           count, c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, nullptr, (hipStream_t)stream)
                     : bmit::launch_blind_rotate_lat2u(d_small, d_lut_ids, (const u64 *)c->d_luts, c->d_bsk3_lat, c->d_tw_half, c->d_root_pow, d_out,
                                                       count, c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, (hipStream_t)stream);
            return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
        }
        if (!c->d_bsk_fft && !bmit::shape_supported(c->bsk_prec, c->P.bs_levels, c->P.bs_base_log))
            return fail(c, -1, "this bootstrap-key precision exists for the unrolled kernel only: bmi_set_bsk_unroll(ctx, 2) before keygen");
        // (floating-point-transform kernels: two rounds of 256 one-workgroup bootstraps, 7.7 ms, still beat the wave-pair kernel's
        // 8.4 ms up to 1,024 ciphertexts; three rounds do not)
        const bool lat_t = c->variant == 2 || c->variant == 4 || c->variant == 6 || (c->variant == 0 && count <= c->lat_threshold);
        if (c->variant == 6 && !c->d_bsk_fft)
            return fail(c, -1, "kernel variant 6 needs the bootstrap key at 48 bits of precision in base 2^10 (the torus default)");
        if (lat_t && c->d_bsk_fft && c->variant != 4) {   // latency form through the floating-point transform (variant 4 pins the exact one)
            rc = bmit::launch_blind_rotate_lat_fft(d_small, d_lut_ids, (const u64 *)c->d_luts, c->d_bsk_latf, c->d_tw_fh, d_out, count,
                                                   c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, nullptr, (hipStream_t)stream);
            return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
        }
        if (lat_t) {
            if (int rcb = build_exact_torus_copies(c, nullptr)) return rcb;   // (variant 4, or a key without FFT copies: a no-op once built)
            rc = bmit::launch_blind_rotate_lat(d_small, d_lut_ids, (const u64 *)c->d_luts, c->d_bsk_lat, c->d_tw_half, d_out,
                                               count, c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, (hipStream_t)stream);
            return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
        }
        // wave pairs: through the floating-point transform where its key copy exists (48-bit key, base 2^10; variant 5 pins it,
        // variants 1 / 3 pin the exact transform mod 2^49 - 720895)
        if (c->variant == 5 && !c->d_bsk_fft)
            return fail(c, -1, "kernel variant 5 needs the bootstrap key at 48 bits of precision in base 2^10 (the torus default)");
        if (c->d_bsk_fft && (c->variant == 0 || c->variant == 5)) {
            rc = bmit::launch_blind_rotate_fft(d_small, d_lut_ids, (const u64 *)c->d_luts, c->d_bsk_fft, c->d_tw_fft, d_out, count, c->P.n,
                                               c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, nullptr, (hipStream_t)stream);
            return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
        }
        if (int rcb = build_exact_torus_copies(c, nullptr)) return rcb;       // (variants 1 / 3, or a key without FFT copies)
        rc = bmit::launch_blind_rotate(d_small, d_lut_ids, (const u64 *)c->d_luts, (const double *)c->d_bsk,
                                       (const double *)c->d_tw, d_out, count, c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log,
                                       (hipStream_t)stream);
        return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
    }
    if (c->f64()) {
        const double *luts = (const double *)c->d_luts, *bsk = (const double *)c->d_bsk, *tw = (const double *)c->d_tw;
        hipStream_t st = (hipStream_t)stream;
        if (c->wide() && c->unroll == 2) {   // N = 2048 with the unrolled key
            if (!c->have_bsk3) return fail(c, -1, "unrolling selected but the context holds no unrolled key: generate keys after bmi_set_bsk_unroll, or bmi_import_bsk_unrolled");
            rc = bmi49::launch_blind_rotate_wide_u(d_small, d_lut_ids, luts, c->d_bsk3_lat, tw, c->d_tw_wide, c->d_root_pow, d_out, count,
                                                   c->P.n, c->P.bs_levels, c->P.bs_base_log, st);
            return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
        }
        if (c->wide() || c->quad()) {   // N = 2048 / 4096: one kernel for every batch size
            rc = c->quad() ? bmi49::launch_blind_rotate_quad(d_small, d_lut_ids, luts, c->d_bsk_lat, tw, c->d_tw_wide, d_out, count, c->P.n, st)
                           : bmi49::launch_blind_rotate_wide(d_small, d_lut_ids, luts, c->d_bsk_lat, tw, c->d_tw_wide, d_out, count, c->P.n,
                                                             c->P.bs_levels, c->P.bs_base_log, st);
            return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
        }
        // auto mode falls back from the kernels that need > 64 KB of LDS per workgroup to their predecessors (still on
        // the GPU) if the device refuses the configuration; a pinned variant reports the error instead
        const bool lb3 = c->P.bs_levels == 3 && c->P.bs_base_log == 15;   // variants 1 and 4 exist for (3, 2^15) only
        if (!lb3 && (c->variant == 1 || c->variant == 4))
            return fail(c, -1, "kernel variants 1 and 4 exist for (l, Bg) = (3, 2^15) only");
        const uint32_t lv = c->P.bs_levels, bl = c->P.bs_base_log;
        if (c->unroll == 2) {   // unrolled key: one kernel (one workgroup per ciphertext) for every batch size, so that a
                                // ciphertext's bits never depend on the batch it travelled in
            if (!c->have_bsk3) return fail(c, -1, "unrolling selected but the context holds no unrolled key: generate keys after bmi_set_bsk_unroll, or bmi_import_bsk_unrolled");
            rc = bmi49::launch_blind_rotate_lat2u(d_small, d_lut_ids, luts, c->d_bsk3_lat, c->d_tw_half, c->d_root_pow, d_out, count,
                                                  c->P.n, lv, bl, st);
            return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
        }
        if (c->variant == 4 || (latency && c->no_big_lds && lb3)) {
            rc = bmi49::launch_blind_rotate_lat(d_small, d_lut_ids, luts, bsk, tw, d_out, count, c->P.n, st);
        } else if (latency) {
            rc = bmi49::launch_blind_rotate_lat2(d_small, d_lut_ids, luts, c->d_bsk_lat, c->d_tw_half, d_out, count, c->P.n, lv, bl, st);
            if (rc && c->variant == 0 && lb3) {
                (void)hipGetLastError();
                c->no_big_lds = true;
                rc = bmi49::launch_blind_rotate_lat(d_small, d_lut_ids, luts, bsk, tw, d_out, count, c->P.n, st);
            }
        } else if (c->variant == 1 || (c->no_big_lds && lb3)) {
            rc = bmi49::launch_blind_rotate_tp(d_small, d_lut_ids, luts, bsk, tw, d_out, count, c->P.n, st);
        } else {
            rc = bmi49::launch_blind_rotate_tpx(d_small, d_lut_ids, luts, bsk, tw, d_out, count, c->P.n, lv, bl, st);
            if (rc && c->variant == 0 && lb3) {
                (void)hipGetLastError();
                c->no_big_lds = true;
                rc = bmi49::launch_blind_rotate_tp(d_small, d_lut_ids, luts, bsk, tw, d_out, count, c->P.n, st);
            }
        }
    } else {
        const u64 *luts = (const u64 *)c->d_luts, *bsk = (const u64 *)c->d_bsk, *tw = (const u64 *)c->d_tw;
        rc = latency ? bmi::launch_blind_rotate_lat(d_small, d_lut_ids, luts, bsk, tw, d_out, count, c->P.n, (hipStream_t)stream)
                     : bmi::launch_blind_rotate_tp(d_small, d_lut_ids, luts, bsk, tw, d_out, count, c->P.n, (hipStream_t)stream);
    }
    return rc ? fail(c, -2, std::string("blind_rotate launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
}

int bmi_pbs_batch(bmi_ctx *c, const uint64_t *d_in, const uint32_t *d_lut_ids, uint32_t count, uint64_t *d_out,
                  void *stream) {
    if (!c) return -1;
    HIP_OK(c, hipSetDevice(c->device));
    int rc = ensure_small(c, count);
    if (rc) return rc;
    rc = bmi_keyswitch_batch(c, d_in, count, c->d_small, stream);
    if (rc) return rc;
    return bmi_blind_rotate_batch(c, c->d_small, d_lut_ids, count, d_out, stream);
}

int bmi_lincomb_batch(bmi_ctx *c, const uint64_t *d_store, const uint32_t *d_row_ptr, const uint32_t *d_idx,
                      const int64_t *d_coef, const uint64_t *d_const_body, uint32_t count, uint64_t *d_out,
                      void *stream) {
    if (!c || (count && (!d_store || !d_row_ptr || !d_const_body || !d_out))) return -1;
    HIP_OK(c, hipSetDevice(c->device));
    int rc = (c->f64() ? bmi49::launch_lincomb : (c->t64() ? bmit::launch_lincomb : bmi::launch_lincomb))(d_store, d_row_ptr, d_idx, (const i64 *)d_coef,
                                                                      d_const_body, d_out, count, c->big_n + 1,
                                                                      (hipStream_t)stream);
    return rc ? fail(c, -2, std::string("lincomb launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
}

int bmi_scatter_rows(bmi_ctx *c, const uint64_t *d_src, uint32_t count, uint64_t *d_store, const uint32_t *d_rows,
                     void *stream) {
    if (!c || (count && (!d_src || !d_store || !d_rows))) return -1;
    HIP_OK(c, hipSetDevice(c->device));
    int rc = bmi::launch_scatter_rows(d_src, d_store, d_rows, count, c->big_n + 1, (hipStream_t)stream);
    return rc ? fail(c, -2, std::string("scatter_rows launch: ") + hipGetErrorString((hipError_t)rc)) : 0;
}

int bmi_reserve(bmi_ctx *c, uint32_t max_count) {
    if (!c) return -1;
    HIP_OK(c, hipSetDevice(c->device));
    HIP_OK(c, hipDeviceSynchronize());
    if (!c->d_ks_partial) {
        c->ks_partial_bytes = (size_t)96 << 20;  // covers every split configuration (<= 1024 workgroups x 8 ciphertexts)
        HIP_OK(c, hipMalloc(&c->d_ks_partial, c->ks_partial_bytes));
    }
    if (c->ks_mfma_ok) {
        // the sums buffer peaks where the K-split is still active (small batches), not at max_count, and the slice
        // count is not monotonic in the batch size: take the largest need over every tile count up to max_count
        uint32_t worst = max_count;
        size_t worst_sums = 0;
        for (uint32_t tiles = 1; tiles <= (max_count + 31) / 32; tiles++) {
            const uint32_t cnt = std::min(tiles * 32, max_count);
            const size_t sums = (size_t)ks_mfma_slices(c, cnt) * cnt;
            if (sums > worst_sums) { worst_sums = sums; worst = cnt; }
        }
        int rc = ensure_ks_mfma(c, worst);
        if (rc) return rc;
        rc = ensure_ks_mfma(c, max_count);
        if (rc) return rc;
    }
    return ensure_small(c, max_count);
}

int bmi_sync(bmi_ctx *c, void *stream) {
    if (!c) return -1;
    HIP_OK(c, hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------- host-buffer convenience forms
namespace {
// a look-up id beyond the registered tables would make the kernels read past the table buffer: refused on the host
// (the device-pointer entry points cannot look at their ids without a synchronisation; their contract is ids < count)
int check_lut_ids(bmi_ctx *c, const uint32_t *lut_ids, uint32_t count) {
    for (uint32_t i = 0; i < count; i++)
        if (lut_ids[i] >= c->n_luts) return fail(c, -1, "look-up id " + std::to_string(lut_ids[i]) + " is not registered");
    return 0;
}
}  // namespace

int bmi_pbs_batch_host(bmi_ctx *c, const uint64_t *in, const uint32_t *lut_ids, uint32_t count, uint64_t *out) {
    if (!c || !in || !lut_ids || !out) return -1;
    if (int bad = check_lut_ids(c, lut_ids, count)) return bad;
    HIP_OK(c, hipSetDevice(c->device));
    int rc = ensure_io(c, count);
    if (rc) return rc;
    const size_t w = (size_t)(c->big_n + 1) * 8;
    HIP_OK(c, hipMemcpyAsync(c->d_io_a, in, count * w, hipMemcpyHostToDevice, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->d_io_ids, lut_ids, count * 4, hipMemcpyHostToDevice, c->stream));
    rc = bmi_pbs_batch(c, c->d_io_a, c->d_io_ids, count, c->d_io_b, c->stream);
    if (rc) return rc;
    HIP_OK(c, hipMemcpyAsync(out, c->d_io_b, count * w, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int bmi_keyswitch_batch_host(bmi_ctx *c, const uint64_t *in, uint32_t count, uint64_t *small_out) {
    if (!c || !in || !small_out) return -1;
    HIP_OK(c, hipSetDevice(c->device));
    int rc = ensure_io(c, count);
    if (rc) return rc;
    rc = ensure_small(c, count);
    if (rc) return rc;
    HIP_OK(c, hipMemcpyAsync(c->d_io_a, in, (size_t)count * (c->big_n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    rc = bmi_keyswitch_batch(c, c->d_io_a, count, c->d_small, c->stream);
    if (rc) return rc;
    HIP_OK(c, hipMemcpyAsync(small_out, c->d_small, (size_t)count * (c->P.n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int bmi_blind_rotate_batch_host(bmi_ctx *c, const uint64_t *small_in, const uint32_t *lut_ids, uint32_t count,
                                uint64_t *out) {
    if (!c || !small_in || !lut_ids || !out) return -1;
    if (int bad = check_lut_ids(c, lut_ids, count)) return bad;
    HIP_OK(c, hipSetDevice(c->device));
    int rc = ensure_io(c, count);
    if (rc) return rc;
    rc = ensure_small(c, count);
    if (rc) return rc;
    HIP_OK(c, hipMemcpyAsync(c->d_small, small_in, (size_t)count * (c->P.n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->d_io_ids, lut_ids, count * 4, hipMemcpyHostToDevice, c->stream));
    rc = bmi_blind_rotate_batch(c, c->d_small, c->d_io_ids, count, c->d_io_b, c->stream);
    if (rc) return rc;
    HIP_OK(c, hipMemcpyAsync(out, c->d_io_b, (size_t)count * (c->big_n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int bmi_fft_margin_host(bmi_ctx *c, const uint64_t *small_in, const uint32_t *lut_ids, uint32_t count, uint64_t *out,
                        double *max_distance) {
    if (!c || !small_in || !lut_ids || !out || !max_distance) return -1;
    if (!c->have_keys) return fail(c, -1, "no keys: call bmi_keygen first");
    const bool ufft = c->t64() && c->unroll == 2 && c->have_bsk3 && bmit::shape_supported_unrolled_fft(c->bsk_prec, c->P.bs_levels, c->P.bs_base_log);
    if (!c->d_bsk_fft && !c->d_bsk_w && !ufft)
        return fail(c, -1, "the floating-point-transform kernels exist on the 2^64 torus with the bootstrap key at 48 bits (N = 1024), 46 bits "
                           "(N = 2048) or 44 bits (N = 4096) in base 2^10");
    if (int bad = check_lut_ids(c, lut_ids, count)) return bad;
    HIP_OK(c, hipSetDevice(c->device));
    int rc = ensure_io(c, count);
    if (rc) return rc;
    rc = ensure_small(c, count);
    if (rc) return rc;
    unsigned long long *d_stat = nullptr;
    HIP_OK(c, hipMalloc(&d_stat, 8));
    hipError_t e = hipMemsetAsync(d_stat, 0, 8, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_small, small_in, (size_t)count * (c->P.n + 1) * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_io_ids, lut_ids, count * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = ufft ? (hipError_t)bmit::launch_blind_rotate_lat2u_fft(c->d_small, c->d_io_ids, (const u64 *)c->d_luts, c->d_bsk3_lat, c->d_tw_fh, c->d_zeta_pow,
                                                                   c->d_io_b, count, c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, d_stat, c->stream)
            : (!c->d_bsk_w && (c->variant == 6 || c->variant == 2))   // the latency form of the N = 1024 transform (kernel variant 6 / 2 selected)
                ? (hipError_t)bmit::launch_blind_rotate_lat_fft(c->d_small, c->d_io_ids, (const u64 *)c->d_luts, c->d_bsk_latf, c->d_tw_fh, c->d_io_b,
                                                                count, c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, d_stat, c->stream)
            : c->d_bsk_w ? (hipError_t)(c->quad() ? bmit::launch_blind_rotate_quad : bmit::launch_blind_rotate_wide)(c->d_small, c->d_io_ids, (const u64 *)c->d_luts, c->d_bsk_w, c->d_tw_fq, c->d_io_b,
                                                                    count, c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, d_stat, c->stream)
                       : (hipError_t)bmit::launch_blind_rotate_fft(c->d_small, c->d_io_ids, (const u64 *)c->d_luts, c->d_bsk_fft, c->d_tw_fft, c->d_io_b,
                                                                   count, c->P.n, c->bsk_prec, c->P.bs_levels, c->P.bs_base_log, d_stat, c->stream);
    unsigned long long bits = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(out, c->d_io_b, (size_t)count * (c->big_n + 1) * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&bits, d_stat, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_stat);
    if (e != hipSuccess) return fail(c, -2, std::string("bmi_fft_margin_host: ") + hipGetErrorString(e));
    std::memcpy(max_distance, &bits, 8);
    return 0;
}

int bmi_negacyclic_mul_host(bmi_ctx *c, const uint64_t *a, const uint64_t *b, uint32_t count, uint64_t *out) {
    if (!c || !a || !b || !out) return -1;
    if (c->wide() || c->quad()) return fail(c, -1, "the transform test hook exists for N = 1024 only");
    if (c->t64()) return fail(c, -1, "no transform exists mod 2^64: the test hook covers the two prime fields");
    HIP_OK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)count * c->N * 8;
    u64 *da = nullptr, *db = nullptr, *dc = nullptr;
    HIP_OK(c, hipMalloc(&da, bytes));
    HIP_OK(c, hipMalloc(&db, bytes));
    HIP_OK(c, hipMalloc(&dc, bytes));
    HIP_OK(c, hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    HIP_OK(c, hipMemcpy(db, b, bytes, hipMemcpyHostToDevice));
    int rc = c->f64() ? bmi49::launch_negacyclic_mul(da, db, dc, (const double *)c->d_tw, count, c->stream)
                      : bmi::launch_negacyclic_mul(da, db, dc, (const u64 *)c->d_tw, count, c->stream);
    if (rc) return fail(c, -2, "negacyclic_mul launch failed");
    HIP_OK(c, hipStreamSynchronize(c->stream));
    HIP_OK(c, hipMemcpy(out, dc, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc);
    return 0;
}

}  // extern "C"
