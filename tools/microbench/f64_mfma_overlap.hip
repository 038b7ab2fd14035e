// Do f64 matrix-core instructions (v_mfma_f64_16x16x4_f64) and f64 vector instructions of DIFFERENT wavefronts on
// one SIMD overlap on gfx950?  Three runs of a workgroup of 512 threads per CU (2 wavefronts per SIMD):
//   valu  : both wavefronts of a SIMD run independent v_fma_f64 chains
//   mfma  : both run independent v_mfma_f64_16x16x4_f64 chains
//   mixed : one runs the v_fma_f64 loop, the other the mfma loop (same instruction counts per wave as above)
// If the two pipes overlap, `mixed` takes max(valu, mfma) / 1 (each loop at its own wave's share); if they share the
// FP64 datapath it takes (valu + mfma) / 2 ... the printout gives ns per wave-instruction for each class.
// Build: hipcc --offload-arch=gfx950 -O3 f64_mfma_overlap.hip -o f64_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4000
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void valu_loop(double (&a)[8]) {
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
}
__device__ __forceinline__ void mfma_loop(double4_t (&acc)[4], double x, double y) {
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
}
// mode 0: all valu, 1: all mfma, 2: even waves valu / odd waves... no: waves w and w + 4 share a SIMD -> split by w < 4
__global__ void __launch_bounds__(512) k(double *out, double seed, int mode) {
    const int wave = threadIdx.x >> 6;
    double a[8];
    double4_t acc[4];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    for (int i = 0; i < 4; i++) acc[i] = double4_t{seed, seed, seed, seed};
    const bool do_valu = mode == 0 || (mode == 2 && wave < 4);
    if (do_valu) valu_loop(a);
    else mfma_loop(acc, seed * 1e-3, seed * 1e-4);
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
float run(int mode) {
    double *d;
    hipMalloc(&d, 256 * 512 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<<<256, 512>>>(d, 1.0, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, 512>>>(d, 1.0, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms;
}
int main() {
    const float v = run(0), m = run(1), x = run(2);
    printf("valu  (2 waves/SIMD x %d v_fma_f64)              : %8.3f ms  %6.2f ns per wave-instruction per SIMD\n", ITER * 32, v, v * 1e6 / (2.0 * ITER * 32));
    printf("mfma  (2 waves/SIMD x %d v_mfma_f64_16x16x4)      : %8.3f ms  %6.2f ns per wave-instruction per SIMD\n", ITER * 8, m, m * 1e6 / (2.0 * ITER * 8));
    printf("mixed (1 valu wave + 1 mfma wave per SIMD)          : %8.3f ms\n", x);
    printf("  if the pipes overlap fully  : max(valu, mfma) / 2 = %8.3f ms\n", (v > m ? v : m) / 2);
    printf("  if they share one datapath  : (valu + mfma) / 2   = %8.3f ms\n", (v + m) / 2);
    return 0;
}
