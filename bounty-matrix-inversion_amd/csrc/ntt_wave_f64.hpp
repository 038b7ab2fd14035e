// Wave-level 1024-point negacyclic NTT over Z_q, q = 2^49 - 720895, in exact f64 arithmetic (field49.hpp).
//
// Same data flow as ntt_wave.hpp (one wavefront per polynomial, 16 coefficients per lane, 1024 = 16 x 16 x 4:
// register DFT16 -> table twiddle -> LDS transpose -> register DFT16 -> table twiddle -> LDS transpose inside
// quads -> register DFT4), but the roots are generic (no power-of-two shifts in this field) and the
// butterflies are LAZY: additions are plain v_add_f64 on centred values, reductions happen inside the
// multiplications (mul(a, b) is exact for any |a| < 2^53 and returns |r| <= (0.5 + |a|/8p) p) and at two running
// sums per DFT16 where a bound would otherwise approach the 53-bit mantissa.  tools/f64_bounds.py is the bound
// model: forward() takes |x| <= 0.8 p and returns |.| <= 6.9 p; inverse() takes |x| <= 0.51 p and returns
// |.| <= 10.6 p; callers reduce.  Evaluation layout identical to ntt_wave.hpp (eval_offset).
#pragma once
#include <type_traits>

#include "field49.hpp"

namespace nttf {

using f49::u64;

constexpr int LOG_N = 10;
constexpr int N = 1 << LOG_N;
constexpr int ROW = 68;
constexpr int SCRATCH_WORDS = 16 * ROW;

constexpr u64 PSI_U = f49::powmod_c(f49::GEN, (f49::Q - 1) / (2 * N));  // primitive 2N-th root of unity
constexpr u64 PSI_INV_U = f49::powmod_c(PSI_U, f49::Q - 2);
constexpr u64 N_INV_U = f49::powmod_c(N, f49::Q - 2);
static_assert(f49::powmod_c(PSI_U, N) == f49::Q - 1, "psi^N = -1");

// twiddle tables (doubles, centred), built on the host once per context, staged into LDS by every workgroup
constexpr int TW_W1 = 0;            // [k1][lane]  psi^(lane (2 k1 + 1))
constexpr int TW_W1I = 1024;        // [k1][lane]  psi^-(lane (2 k1 + 1)) / N
constexpr int TW_W2 = 2048;         // [v][t]      (psi^32)^(t v)
constexpr int TW_W2I = 2048 + 64;   // [v][t]      (psi^32)^-(t v)
constexpr int TW_WORDS = 2048 + 128;

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int br4(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// psi^(E) (INV: psi^(-E)) as a centred double, E taken mod 2N
template <bool INV, int E>
constexpr double psi_pow() {
    return f49::centred_c(f49::powmod_c(INV ? PSI_INV_U : PSI_U, (u64)(E % (2 * N))));
}

// 16-point DFT over the register array, root psi^128 (INV: its inverse), natural order in and out, lazy.
// RED = true reduces the two running sums x[0], x[1] after the second stage; nothing else is reduced explicitly
// (every twiddled difference is reduced by its multiplication).  tools/f64_bounds.py tracks the magnitudes through
// the whole pipeline: with RED only in the second DFT16 of forward() and in both of inverse() no value exceeds
// 10.6 p (the exact-integer limit of a double is 2^53 = 16 p).
template <bool INV, bool RED>
__device__ __forceinline__ void dft16(double (&x)[16]) {
    static_for<0, 8>([&](auto I) {  // half = 8, twiddle w16^i = psi^(128 i)
        constexpr int i = I;
        const double u = x[i] + x[i + 8], d = x[i] - x[i + 8];
        x[i] = u;
        if constexpr (i == 0) x[i + 8] = d;
        else x[i + 8] = f49::mul(d, psi_pow<INV, 128 * i>());
    });
    static_for<0, 2>([&](auto B) {  // half = 4, twiddle w8^i = psi^(256 i)
        static_for<0, 4>([&](auto I) {
            constexpr int b = B * 8, i = I;
            const double u = x[b + i] + x[b + i + 4], d = x[b + i] - x[b + i + 4];
            x[b + i] = (RED && b == 0 && i < 2) ? f49::red(u) : u;
            if constexpr (i == 0) x[b + i + 4] = d;
            else x[b + i + 4] = f49::mul(d, psi_pow<INV, 256 * i>());
        });
    });
    static_for<0, 4>([&](auto B) {  // half = 2, twiddle w4^i = psi^(512 i)
        static_for<0, 2>([&](auto I) {
            constexpr int b = B * 4, i = I;
            const double u = x[b + i] + x[b + i + 2], d = x[b + i] - x[b + i + 2];
            x[b + i] = u;
            if constexpr (i == 0) x[b + i + 2] = d;
            else x[b + i + 2] = f49::mul(d, psi_pow<INV, 512>());
        });
    });
    static_for<0, 8>([&](auto B) {
        constexpr int b = B * 2;
        const double u = x[b] + x[b + 1], d = x[b] - x[b + 1];
        x[b] = u;
        x[b + 1] = d;
    });
    double y[16];
    static_for<0, 16>([&](auto R) { y[br4(R)] = x[R]; });
    static_for<0, 16>([&](auto R) { x[R] = y[R]; });
}

// 4-point DFT over x[B..B+3] (index = t), root psi^512, lazy
template <bool INV, int B>
__device__ __forceinline__ void dft4(double (&x)[16]) {
    const double e0 = x[B] + x[B + 2], o0 = x[B] - x[B + 2];
    const double e1 = x[B + 1] + x[B + 3];
    const double o1 = f49::mul(x[B + 1] - x[B + 3], psi_pow<INV, 512>());
    x[B] = e0 + e1;
    x[B + 2] = e0 - e1;
    x[B + 1] = o0 + o1;
    x[B + 3] = o0 - o1;
}

// The scheduler may not move anything across this point.  Used to keep batches of LDS twiddle reads ahead of the
// arithmetic that hides their latency: left alone, the compiler issues each read right before its use and waits
// for it (lgkmcnt(0) after every ds_read: ~30 exposed LDS latencies per transform, seen in the ISA).
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// Forward transform.  x[j] = a[lane + 64 j] (|.| <= 0.8p) on entry; evaluation layout (|.| <= 6.9p) on exit.
// mid() runs before the second DFT16, late() before the quad transpose: callers issue global loads there whose
// latency the rest of the transform hides.
template <class Mid = NoHook, class Late = NoHook>
__device__ __forceinline__ void forward(double (&x)[16], int lane, const double *tw, double *scratch, Mid mid = Mid(),
                                        Late late = Late()) {
    double wa[8], wb[8];
    static_for<0, 8>([&](auto K) { wa[K] = tw[TW_W1 + K * 64 + lane]; });
    sched_fence();
    static_for<1, 16>([&](auto J) { x[J] = f49::mul(x[J], psi_pow<false, 64 * J>()); });  // psi^(64 j)
    dft16<false, false>(x);
    static_for<0, 8>([&](auto K) { wb[K] = tw[TW_W1 + (K + 8) * 64 + lane]; });
    sched_fence();
    static_for<0, 8>([&](auto K) { x[K] = f49::mul(x[K], wa[K]); });
    static_for<0, 8>([&](auto K) { x[K + 8] = f49::mul(x[K + 8], wb[K]); });
    wave_sync();
    static_for<0, 16>([&](auto K) { scratch[K * ROW + lane] = x[K]; });
    wave_sync();
    const int k1 = lane >> 2, t = lane & 3;
    double *row = scratch + k1 * ROW;
    static_for<0, 16>([&](auto U) { x[U] = row[t + 4 * U]; });
    static_for<1, 8>([&](auto V) { wa[V] = tw[TW_W2 + V * 4 + t]; });
    mid();
    sched_fence();
    dft16<false, true>(x);
    static_for<0, 8>([&](auto V) { wb[V] = tw[TW_W2 + (V + 8) * 4 + t]; });
    sched_fence();
    x[0] = f49::red(x[0]);
    static_for<1, 8>([&](auto V) { x[V] = f49::mul(x[V], wa[V]); });
    static_for<0, 8>([&](auto V) { x[V + 8] = f49::mul(x[V + 8], wb[V]); });
    late();
    const int sw0 = t & 3, sw1 = (t + 1) & 3, sw2 = (t + 2) & 3, sw3 = (t + 3) & 3;
    wave_sync();
    static_for<0, 16>([&](auto V) {
        constexpr int v = V;
        const int sw = (v & 3) == 0 ? sw0 : ((v & 3) == 1 ? sw1 : ((v & 3) == 2 ? sw2 : sw3));
        row[16 * t + (v & 12) + sw] = x[V];
    });
    wave_sync();
    static_for<0, 16>([&](auto R) {
        constexpr int vl = R / 4, tt = R % 4;
        const int sw = tt == 0 ? sw0 : (tt == 1 ? sw1 : (tt == 2 ? sw2 : sw3));
        x[R] = row[16 * tt + 4 * vl + sw];
    });
    dft4<false, 0>(x);
    dft4<false, 4>(x);
    dft4<false, 8>(x);
    dft4<false, 12>(x);
}

// Inverse transform (includes 1/N): evaluation layout in (|.| <= 0.51p), x[j] = a[lane + 64 j] out (|.| <= 10.6p).
__device__ __forceinline__ void inverse(double (&x)[16], int lane, const double *tw, double *scratch) {
    const int k1 = lane >> 2, t = lane & 3;
    double *row = scratch + k1 * ROW;
    const int sw0 = t & 3, sw1 = (t + 1) & 3, sw2 = (t + 2) & 3, sw3 = (t + 3) & 3;
    double wa[8], wb[8];
    static_for<1, 8>([&](auto V) { wa[V] = tw[TW_W2I + V * 4 + t]; });
    static_for<0, 8>([&](auto V) { wb[V] = tw[TW_W2I + (V + 8) * 4 + t]; });
    sched_fence();
    dft4<true, 0>(x);
    dft4<true, 4>(x);
    dft4<true, 8>(x);
    dft4<true, 12>(x);
    wave_sync();
    static_for<0, 16>([&](auto R) {
        constexpr int vl = R / 4, tt = R % 4;
        const int sw = tt == 0 ? sw0 : (tt == 1 ? sw1 : (tt == 2 ? sw2 : sw3));
        row[16 * tt + 4 * vl + sw] = x[R];
    });
    wave_sync();
    static_for<0, 16>([&](auto V) {
        constexpr int v = V;
        const int sw = (v & 3) == 0 ? sw0 : ((v & 3) == 1 ? sw1 : ((v & 3) == 2 ? sw2 : sw3));
        x[V] = row[16 * t + (v & 12) + sw];
    });
    x[0] = f49::red(x[0]);
    static_for<1, 8>([&](auto V) { x[V] = f49::mul(x[V], wa[V]); });
    static_for<0, 8>([&](auto V) { x[V + 8] = f49::mul(x[V + 8], wb[V]); });
    dft16<true, true>(x);
    wave_sync();
    static_for<0, 16>([&](auto U) { row[t + 4 * U] = x[U]; });
    wave_sync();
    double y[16];
    static_for<0, 16>([&](auto K) { y[K] = scratch[K * ROW + lane]; });
    static_for<0, 8>([&](auto K) { wa[K] = tw[TW_W1I + K * 64 + lane]; });
    static_for<0, 8>([&](auto K) { wb[K] = tw[TW_W1I + (K + 8) * 64 + lane]; });
    sched_fence();
    static_for<0, 8>([&](auto K) { x[K] = f49::mul(y[K], wa[K]); });
    static_for<0, 8>([&](auto K) { x[K + 8] = f49::mul(y[K + 8], wb[K]); });
    dft16<true, true>(x);
    static_for<1, 16>([&](auto J) { x[J] = f49::mul(x[J], psi_pow<true, 64 * J>()); });
}

__host__ __device__ __forceinline__ int eval_offset(int th, int v) { return ((v >> 1) * 64 + th) * 2 + (v & 1); }

}  // namespace nttf
