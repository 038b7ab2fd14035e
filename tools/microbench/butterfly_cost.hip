// Cost of one radix-2 NTT butterfly (t = w*b; a' = a + t; b' = a - t) per wave, chip-wide wall clock, for
//   G  : the Goldilocks field used by the kernels (csrc/goldilocks.hpp)
//   R2 : two 31-bit primes (RNS), Shoup multiplication with precomputed quotient, v_min_u32 corrections
// Each lane runs UNROLL independent butterflies per iteration; 2 waves per SIMD (512-thread blocks, 1 per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -I../../bounty-matrix-inversion_amd/csrc butterfly_cost.hip -o butterfly_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "goldilocks.hpp"

constexpr uint32_t P1 = 2147473409u;  // 2^31 - 10239, = 1 mod 2048
constexpr uint32_t P2 = 2147389441u;  // = 1 mod 2048
#define ITER 4000
#define UNROLL 8

__device__ __forceinline__ uint32_t shoup_mul(uint32_t b, uint32_t w, uint32_t wq, uint32_t p) {
    // wq = floor(w * 2^32 / p); result in [0, p)
    const uint32_t q = __umulhi(b, wq);
    uint32_t r = b * w - q * p;      // in [0, 2p)
    return min(r, r - p);            // unsigned: r - p wraps to a huge value when r < p
}
__device__ __forceinline__ uint32_t addp(uint32_t a, uint32_t b, uint32_t p) { uint32_t s = a + b; return min(s, s - p); }
__device__ __forceinline__ uint32_t subp(uint32_t a, uint32_t b, uint32_t p) { uint32_t d = a - b; return min(d, d + p); }

template <int T>
__global__ void __launch_bounds__(512) k(uint64_t *out, uint64_t seed) {
    if constexpr (T == 0) {
        gl::u64 a[UNROLL], b[UNROLL];
        gl::u64 w = (seed * 0x9E3779B97F4A7C15ull + threadIdx.x) % gl::P;
        for (int i = 0; i < UNROLL; i++) { a[i] = (seed + 11 * i + threadIdx.x) % gl::P; b[i] = (seed * 3 + 7 * i + threadIdx.x) % gl::P; }
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) {
                const gl::u64 t = gl::mul(b[i], w);
                const gl::u64 x = gl::add(a[i], t), y = gl::sub(a[i], t);
                a[i] = x; b[i] = y;
            }
        }
        gl::u64 acc = 0;
        for (int i = 0; i < UNROLL; i++) acc ^= a[i] ^ b[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    } else {
        uint32_t a1[UNROLL], b1[UNROLL], a2[UNROLL], b2[UNROLL];
        const uint32_t w1 = (uint32_t)(seed * 2654435761u + threadIdx.x) % P1, w2 = (uint32_t)(seed * 40503u + threadIdx.x) % P2;
        const uint32_t wq1 = (uint32_t)(((uint64_t)w1 << 32) / P1), wq2 = (uint32_t)(((uint64_t)w2 << 32) / P2);
        for (int i = 0; i < UNROLL; i++) { a1[i] = (seed + 11 * i + threadIdx.x) % P1; b1[i] = (seed * 3 + i) % P1; a2[i] = (seed + 5 * i + threadIdx.x) % P2; b2[i] = (seed * 7 + i) % P2; }
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) {
                const uint32_t t1 = shoup_mul(b1[i], w1, wq1, P1), t2 = shoup_mul(b2[i], w2, wq2, P2);
                const uint32_t x1 = addp(a1[i], t1, P1), y1 = subp(a1[i], t1, P1);
                const uint32_t x2 = addp(a2[i], t2, P2), y2 = subp(a2[i], t2, P2);
                a1[i] = x1; b1[i] = y1; a2[i] = x2; b2[i] = y2;
            }
        }
        uint32_t acc = 0;
        for (int i = 0; i < UNROLL; i++) acc ^= a1[i] ^ b1[i] ^ a2[i] ^ b2[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    }
}

// F : a 50-bit prime in f64, centred representatives in (-p/2, p/2], every reduction is x - p * rint(x / p)
constexpr double PF = 1125899906826241.0;      // a 50-bit odd modulus for the timing (value irrelevant to the cost)
constexpr double PFINV = 1.0 / 1125899906826241.0;
__device__ __forceinline__ double redf(double x) { return __builtin_fma(-__builtin_rint(x * PFINV), PF, x); }
__device__ __forceinline__ double mulf(double a, double b) {
    const double h = a * b, l = __builtin_fma(a, b, -h);
    return __builtin_fma(-__builtin_rint(h * PFINV), PF, h) + l;
}
__global__ void __launch_bounds__(512) kf(uint64_t *out, uint64_t seed) {
    double a[UNROLL], b[UNROLL];
    const double w = (double)((seed * 0x9E3779B97F4A7C15ull + threadIdx.x) % 1125899906826241ull) - 5e14;
    for (int i = 0; i < UNROLL; i++) { a[i] = (double)((seed + 11 * i + threadIdx.x) % 1000003); b[i] = (double)((seed * 3 + 7 * i + threadIdx.x) % 1000033); }
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            const double t = mulf(b[i], w);
            const double x = redf(a[i] + t), y = redf(a[i] - t);
            a[i] = x; b[i] = y;
        }
    }
    double acc = 0;
    for (int i = 0; i < UNROLL; i++) acc += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)(long long)acc;
}

template <int T>
void run(const char *name) {
    uint64_t *d;
    const int blocks = 256, threads = 512;
    hipMalloc(&d, blocks * threads * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&](uint64_t sd) {
        if constexpr (T == 2) hipLaunchKernelGGL(kf, dim3(blocks), dim3(threads), 0, 0, d, sd);
        else hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, sd);
    };
    launch(12345ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) launch(12345ull + r);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bf = (double)blocks * threads * ITER * UNROLL * 5;  // lane-butterflies
    printf("%-44s %.3f ms  %.2f T lane-butterflies/s  -> %.1f cycles per wave-butterfly per SIMD @2.1GHz\n", name, ms / 5, bf / (ms * 1e-3) / 1e12,
           1024.0 * 2.1e9 / (bf / 64 / (ms * 1e-3)));
    hipFree(d);
}

int main() {
    run<0>("Goldilocks butterfly (mul + add + sub)");
    run<1>("2 x 31-bit RNS butterfly (Shoup + min)");
    run<2>("50-bit prime in f64, centred (fma + rint)");
    return 0;
}
