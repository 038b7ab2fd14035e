// Wave-level 1024-point negacyclic NTT over Z_q (q = 2^64 - 2^32 + 1) for gfx950.
//
// One 64-lane wavefront owns one polynomial: lane l holds 16 coefficients a[l + 64 j] (j = register
// index), so every global/LDS access of a register is a contiguous 512-byte row.  The transform is
// 1024 = 16 x 16 x 4:
//   P1  16-point DFT over the registers (stride-64 elements), all twiddles powers of two (shifts);
//       the negacyclic twist along this axis is psi^(64 j) = 2^(6 j), also a shift;
//   W1  one general multiplication per element by psi^(l (2 k1 + 1)) (LDS-staged table) - this also
//       carries the rest of the negacyclic twist;
//   T   transpose through a wave-private LDS tile (16 rows of 64 words, padded to 68: conflict-free);
//   P2  16-point DFT over the registers again (root 2^12), then twiddle 8^(t v) (LDS table);
//   T2  a second transpose inside each group of 4 lanes (same tile, XOR-free swizzle (v + t) & 3 on the low
//       index bits: conflict-free for both the writes and the reads);
//   P3  four 4-point DFTs in registers (root 2^48).  (A first version did P3 across the lanes of a quad with
//       DPP moves; every lane then computes both butterfly branches and a mostly discarded shift: 923 VALU
//       instructions per transform against ~290 for the in-register form, measured from the ISA.)
// No workgroup barrier is needed anywhere: all exchanges stay inside the wavefront.
// Output slot (thread 4*k1 + g, register 4*vl + s) holds the evaluation at psi^(2k+1), k = k1 + 16 (4 vl + g) + 256 s;
// the inverse transform consumes exactly that layout and returns coefficients in the input layout,
// already scaled by 1/N.  tools/ntt_model.py is the index/twiddle model this file follows.
#pragma once
#include <type_traits>
#include <utility>

#include "goldilocks.hpp"

namespace nttw {

using gl::u64;

constexpr int LOG_N = 10;
constexpr int N = 1 << LOG_N;
constexpr int ROW = 68;                     // padded LDS row (64-bit words)
constexpr int SCRATCH_WORDS = 16 * ROW;     // wave-private transpose tile (>= N words)
// psi: primitive 2N-th root of unity with psi^32 = 8 (so psi^64 = 2^6), see tools/ntt_model.py
constexpr u64 PSI = 0x7a591595e67c27e8ULL;
constexpr u64 PSI_INV = 0x88faac55bfee9b74ULL;
constexpr u64 N_INV = 0xffbfffff00400001ULL;

// twiddle tables (64-bit words), built on the host once per context and staged into LDS by every workgroup
constexpr int TW_W1 = 0;             // [k1][lane]  psi^(lane (2 k1 + 1))
constexpr int TW_W1I = 1024;         // [k1][lane]  psi^-(lane (2 k1 + 1)) / N
constexpr int TW_W2 = 2048;          // [v][t]      8^(t v)
constexpr int TW_W2I = 2048 + 64;    // [v][t]      8^-(t v)
constexpr int TW_WORDS = 2048 + 128;

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wavefront execute in program order; only the compiler must not reorder.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int br4(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// x * 2^(S) for the forward root, x * 2^(-S) for the inverse root (2 has order 192)
template <bool INV, int S>
__device__ __forceinline__ u64 tw_shift(u64 x) {
    return gl::mul_pow2<INV ? (192 - (S % 192)) % 192 : (S % 192)>(x);
}

// 16-point DFT over the register array, root 2^12 (INV: 2^-12), natural order in and out, unnormalised.
template <bool INV>
__device__ __forceinline__ void dft16(u64 (&x)[16]) {
    // decimation in frequency: half = 8, 4, 2, 1 ; twiddle w_{2 half}^i = 2^(12 * (8/half) * i)
    static_for<0, 8>([&](auto I) {
        constexpr int i = I;
        u64 u = x[i], v = x[i + 8];
        x[i] = gl::add(u, v);
        x[i + 8] = tw_shift<INV, 12 * i>(gl::sub(u, v));
    });
    static_for<0, 2>([&](auto B) {
        static_for<0, 4>([&](auto I) {
            constexpr int b = B * 8, i = I;
            u64 u = x[b + i], v = x[b + i + 4];
            x[b + i] = gl::add(u, v);
            x[b + i + 4] = tw_shift<INV, 24 * i>(gl::sub(u, v));
        });
    });
    static_for<0, 4>([&](auto B) {
        static_for<0, 2>([&](auto I) {
            constexpr int b = B * 4, i = I;
            u64 u = x[b + i], v = x[b + i + 2];
            x[b + i] = gl::add(u, v);
            x[b + i + 2] = tw_shift<INV, 48 * i>(gl::sub(u, v));
        });
    });
    static_for<0, 8>([&](auto B) {
        constexpr int b = B * 2;
        u64 u = x[b], v = x[b + 1];
        x[b] = gl::add(u, v);
        x[b + 1] = gl::sub(u, v);
    });
    // bit-reversed -> natural (register renaming only)
    u64 y[16];
    static_for<0, 16>([&](auto R) { y[br4(R)] = x[R]; });
    static_for<0, 16>([&](auto R) { x[R] = y[R]; });
}

// 4-point DFT over x[B], x[B+1], x[B+2], x[B+3] (index = t), root i = 2^48 (INV: 2^-48 = 2^144), unnormalised
template <bool INV, int B>
__device__ __forceinline__ void dft4(u64 (&x)[16]) {
    const u64 e0 = gl::add(x[B], x[B + 2]), o0 = gl::sub(x[B], x[B + 2]);
    const u64 e1 = gl::add(x[B + 1], x[B + 3]);
    const u64 o1 = gl::mul_pow2<INV ? 144 : 48>(gl::sub(x[B + 1], x[B + 3]));
    x[B] = gl::add(e0, e1);
    x[B + 2] = gl::sub(e0, e1);
    x[B + 1] = gl::add(o0, o1);
    x[B + 3] = gl::sub(o0, o1);
}

// Forward transform.  x[j] = a[lane + 64 j] on entry; evaluation layout on exit.
// tw: LDS twiddle tables (TW_* offsets); scratch: wave-private LDS tile of SCRATCH_WORDS words.
__device__ __forceinline__ void forward(u64 (&x)[16], int lane, const u64 *tw, u64 *scratch) {
    static_for<1, 16>([&](auto J) { x[J] = gl::mul_pow2<6 * J>(x[J]); });  // psi^(64 j)
    dft16<false>(x);
    static_for<0, 16>([&](auto K) { x[K] = gl::mul(x[K], tw[TW_W1 + K * 64 + lane]); });
    wave_sync();
    static_for<0, 16>([&](auto K) { scratch[K * ROW + lane] = x[K]; });
    wave_sync();
    const int k1 = lane >> 2, t = lane & 3;
    u64 *row = scratch + k1 * ROW;
    static_for<0, 16>([&](auto U) { x[U] = row[t + 4 * U]; });
    dft16<false>(x);
    static_for<1, 16>([&](auto V) { x[V] = gl::mul(x[V], tw[TW_W2 + V * 4 + t]); });
    // T2: lane (k1, t), register v  ->  lane (k1, g), register 4*vl + tt  with v = 4*vl + g
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
        const int sw = tt == 0 ? sw0 : (tt == 1 ? sw1 : (tt == 2 ? sw2 : sw3));  // (g + tt) & 3 with g = lane & 3
        x[R] = row[16 * tt + 4 * vl + sw];
    });
    dft4<false, 0>(x);
    dft4<false, 4>(x);
    dft4<false, 8>(x);
    dft4<false, 12>(x);
}

// Inverse transform (includes the 1/N factor): evaluation layout in, x[j] = a[lane + 64 j] out.
__device__ __forceinline__ void inverse(u64 (&x)[16], int lane, const u64 *tw, u64 *scratch) {
    const int k1 = lane >> 2, t = lane & 3;
    u64 *row = scratch + k1 * ROW;
    const int sw0 = t & 3, sw1 = (t + 1) & 3, sw2 = (t + 2) & 3, sw3 = (t + 3) & 3;
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
    static_for<1, 16>([&](auto V) { x[V] = gl::mul(x[V], tw[TW_W2I + V * 4 + t]); });
    dft16<true>(x);
    wave_sync();
    static_for<0, 16>([&](auto U) { row[t + 4 * U] = x[U]; });
    wave_sync();
    static_for<0, 16>([&](auto K) { x[K] = gl::mul(scratch[K * ROW + lane], tw[TW_W1I + K * 64 + lane]); });
    dft16<true>(x);
    static_for<1, 16>([&](auto J) { x[J] = gl::mul_pow2<(192 - 6 * J) % 192>(x[J]); });
}

// Offset (in words) of evaluation slot (thread th, register v) inside a stored NTT-domain polynomial:
// two registers per lane are adjacent so that a lane loads 16 bytes and a wave 1 KiB per instruction.
__host__ __device__ __forceinline__ int eval_offset(int th, int v) { return ((v >> 1) * 64 + th) * 2 + (v & 1); }

}  // namespace nttw
