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
//   P3  4-point DFT across the 4 lanes of a quad with DPP quad_perm moves (root 2^48), no LDS.
// No workgroup barrier is needed anywhere: all exchanges stay inside the wavefront.
// Output slot (thread 4*k1 + t, register v) holds the evaluation at psi^(2k+1), k = k1 + 16 v + 256 br2(t);
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

template <int CTRL>
__device__ __forceinline__ u64 dpp64(u64 x) {
    int lo = (int)(unsigned)x, hi = (int)(unsigned)(x >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return ((u64)(unsigned)hi << 32) | (u64)(unsigned)lo;
}
constexpr int QUAD_XOR1 = 0xB1;  // quad_perm [1,0,3,2]
constexpr int QUAD_XOR2 = 0x4E;  // quad_perm [2,3,0,1]

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
    static_for<0, 16>([&](auto U) { x[U] = scratch[k1 * ROW + t + 4 * U]; });
    dft16<false>(x);
    static_for<1, 16>([&](auto V) { x[V] = gl::mul(x[V], tw[TW_W2 + V * 4 + t]); });
    const bool hi2 = (t & 2) != 0, odd = (t & 1) != 0, l3 = (t == 3);
    static_for<0, 16>([&](auto V) {
        u64 a = x[V];
        u64 p = dpp64<QUAD_XOR2>(a);
        u64 y = gl::add(p, hi2 ? gl::neg(a) : a);  // t<2: x_t + x_{t+2} ; t>=2: x_{t-2} - x_t
        u64 y3 = gl::mul_pow2<48>(y);
        y = l3 ? y3 : y;
        p = dpp64<QUAD_XOR1>(y);
        x[V] = gl::add(p, odd ? gl::neg(y) : y);
    });
}

// Inverse transform (includes the 1/N factor): evaluation layout in, x[j] = a[lane + 64 j] out.
__device__ __forceinline__ void inverse(u64 (&x)[16], int lane, const u64 *tw, u64 *scratch) {
    const int k1 = lane >> 2, t = lane & 3;
    const bool hi2 = (t & 2) != 0, odd = (t & 1) != 0, l3 = (t == 3);
    static_for<0, 16>([&](auto V) {
        u64 z = x[V];
        u64 p = dpp64<QUAD_XOR1>(z);
        u64 y = gl::add(p, odd ? gl::neg(z) : z);
        u64 y3 = gl::mul_pow2<144>(y);  // / 2^48
        y = l3 ? y3 : y;
        p = dpp64<QUAD_XOR2>(y);
        x[V] = gl::add(p, hi2 ? gl::neg(y) : y);
    });
    static_for<1, 16>([&](auto V) { x[V] = gl::mul(x[V], tw[TW_W2I + V * 4 + t]); });
    dft16<true>(x);
    wave_sync();
    static_for<0, 16>([&](auto U) { scratch[k1 * ROW + t + 4 * U] = x[U]; });
    wave_sync();
    static_for<0, 16>([&](auto K) { x[K] = gl::mul(scratch[K * ROW + lane], tw[TW_W1I + K * 64 + lane]); });
    dft16<true>(x);
    static_for<1, 16>([&](auto J) { x[J] = gl::mul_pow2<(192 - 6 * J) % 192>(x[J]); });
}

// Offset (in words) of evaluation slot (thread th, register v) inside a stored NTT-domain polynomial:
// two registers per lane are adjacent so that a lane loads 16 bytes and a wave 1 KiB per instruction.
__host__ __device__ __forceinline__ int eval_offset(int th, int v) { return ((v >> 1) * 64 + th) * 2 + (v & 1); }

}  // namespace nttw
