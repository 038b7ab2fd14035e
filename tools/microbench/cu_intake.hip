// How fast can ONE compute unit take in key words?  One workgroup of 1,024 threads (the latency kernels' shape) streams a
// 93 MB buffer the way k_blind_rotate_lat2u_49 reads its key: per step every thread requests W words (8 B each, or pairs as
// one 16 B request) from rows 8 KB apart, waits for them, adds them up.  Reported: bytes per cycle per compute unit at the
// measured shader clock estimate (wall-clock x 2.4 GHz), for 1 workgroup and for 256 workgroups reading the SAME addresses
// (the blind rotation's case: every ciphertext walks the same key) or DISTINCT ones.
// Build: hipcc --offload-arch=gfx950 -O3 cu_intake.hip -o cu_intake
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int THREADS = 1024;

template <int W, bool WIDE>
__global__ void __launch_bounds__(THREADS) k(const double *__restrict__ buf, double *out, size_t words_per_wg, int steps, int distinct) {
    const double *base = buf + (distinct ? (size_t)blockIdx.x * words_per_wg : 0);
    double s = 0.0;
    const int t = threadIdx.x;
    for (int it = 0; it < steps; it++) {
        const double *p = base + (size_t)it * W * THREADS;
        if constexpr (WIDE) {
            double2 v[W / 2];
#pragma unroll
            for (int r = 0; r < W / 2; r++) v[r] = reinterpret_cast<const double2 *>(p + (size_t)r * 2 * THREADS)[t];
#pragma unroll
            for (int r = 0; r < W / 2; r++) s += v[r].x + v[r].y;
        } else {
            double v[W];
#pragma unroll
            for (int r = 0; r < W; r++) v[r] = p[(size_t)r * THREADS + t];
#pragma unroll
            for (int r = 0; r < W; r++) s += v[r];
        }
    }
    out[(size_t)blockIdx.x * THREADS + t] = s;
}

template <int W, bool WIDE>
void run(const char *name, const double *d_buf, double *d_out, size_t total_words, int grid, int distinct) {
    const size_t words_per_wg = distinct ? total_words / grid : total_words;
    const int steps = (int)(words_per_wg / ((size_t)W * THREADS));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<W, WIDE>), dim3(grid), dim3(THREADS), 0, 0, d_buf, d_out, words_per_wg, steps, distinct);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<W, WIDE>), dim3(grid), dim3(THREADS), 0, 0, d_buf, d_out, words_per_wg, steps, distinct);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes_per_wg = (double)steps * W * THREADS * 8;
    printf("%-34s grid %3d %s: %7.3f ms, %6.1f GB/s per CU, %5.1f B/cycle per CU at 2.4 GHz\n", name, grid,
           distinct ? "distinct" : "same    ", ms, bytes_per_wg / (ms * 1e-3) / 1e9, bytes_per_wg / (ms * 1e-3) / 2.4e9);
}

int main() {
    const size_t total_words = (size_t)93 * 1024 * 1024 / 8;
    double *d_buf, *d_out;
    hipMalloc(&d_buf, total_words * 8);
    hipMalloc(&d_out, (size_t)256 * THREADS * 8);
    hipMemset(d_buf, 0, total_words * 8);
    for (int grid : {1, 256}) {
        for (int distinct : {0, 1}) {
            if (grid == 1 && distinct) continue;
            run<12, false>("12 x 8 B per thread per step", d_buf, d_out, total_words, grid, distinct);
            run<24, false>("24 x 8 B per thread per step", d_buf, d_out, total_words, grid, distinct);
            run<36, false>("36 x 8 B per thread per step", d_buf, d_out, total_words, grid, distinct);
            run<12, true>("6 x 16 B per thread per step", d_buf, d_out, total_words, grid, distinct);
            run<24, true>("12 x 16 B per thread per step", d_buf, d_out, total_words, grid, distinct);
        }
    }
    return 0;
}
