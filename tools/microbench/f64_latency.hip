// Dependent-issue latency of f64 VALU instructions on gfx950: C independent chains per wave, W waves per SIMD.
// ns per wave-instruction per SIMD; with enough independent work the cost is the issue rate (~2.0 ns at ~2.2 GHz).
// Build: hipcc --offload-arch=gfx950 -O3 f64_latency.hip -o f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2000
template <int C, int OP>
__global__ void k(double *out, double seed) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 64 / C; r++) {
#pragma unroll
            for (int i = 0; i < C; i++) {
                if constexpr (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else if constexpr (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int C, int OP>
void run(const char *name, int threads) {
    double *d;
    hipMalloc(&d, 256 * 1024 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<C, OP><<<256, threads>>>(d, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<C, OP><<<256, threads>>>(d, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int waves_per_simd = threads / 256;
    printf("%-10s chains %d  waves/SIMD %d : %6.2f ns per wave-instruction per SIMD (%6.2f ns between dependent instructions)\n", name, C,
           waves_per_simd, ms * 1e6 / ((double)waves_per_simd * ITER * 64), ms * 1e6 / (ITER * 64.0) * C);
    hipFree(d);
}
int main() {
    run<1, 0>("fma", 256); run<2, 0>("fma", 256); run<4, 0>("fma", 256); run<8, 0>("fma", 256);
    run<1, 0>("fma", 512); run<2, 0>("fma", 512); run<4, 0>("fma", 512); run<8, 0>("fma", 512);
    run<1, 1>("add", 256); run<2, 1>("add", 256); run<4, 1>("add", 256); run<2, 1>("add", 512); run<4, 1>("add", 512);
    run<1, 2>("rndne", 256); run<2, 2>("rndne", 256); run<4, 2>("rndne", 512);
    return 0;
}
