// Does the VGPR bank of the source operands change the issue cost of f64 VALU instructions on gfx950?
// Explicit registers: 64-bit operand v[2k:2k+1] occupies banks (2k % 4, 2k % 4 + 1).  Wall clock, 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 f64_bank.hip -o f64_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4000
#define R8(s) s s s s s s s s
template <int T>
__global__ void __launch_bounds__(512) k(double *out, double seed) {
    asm volatile("v_cvt_f64_i32 v[20:21], %0\n v_cvt_f64_i32 v[22:23], %0\n v_cvt_f64_i32 v[24:25], %0\n v_cvt_f64_i32 v[26:27], %0\n"
                 "v_cvt_f64_i32 v[28:29], %0\n v_cvt_f64_i32 v[30:31], %0\n v_cvt_f64_i32 v[32:33], %0\n v_cvt_f64_i32 v[34:35], %0\n"
                 :: "v"((int)threadIdx.x) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
    for (int it = 0; it < ITER; it++) {
        if constexpr (T == 0)       // add, sources in different bank pairs: (2,3) + (0,1)
            asm volatile(R8("v_add_f64 v[40:41], v[22:23], v[24:25]\n v_add_f64 v[42:43], v[26:27], v[28:29]\n v_add_f64 v[44:45], v[30:31], v[32:33]\n v_add_f64 v[46:47], v[34:35], v[20:21]\n v_add_f64 v[48:49], v[22:23], v[28:29]\n v_add_f64 v[50:51], v[26:27], v[32:33]\n v_add_f64 v[52:53], v[30:31], v[20:21]\n v_add_f64 v[54:55], v[34:35], v[24:25]\n")
                         ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        else if constexpr (T == 1)  // add, both sources in bank pair (0,1)
            asm volatile(R8("v_add_f64 v[40:41], v[20:21], v[24:25]\n v_add_f64 v[42:43], v[24:25], v[28:29]\n v_add_f64 v[44:45], v[28:29], v[32:33]\n v_add_f64 v[46:47], v[32:33], v[20:21]\n v_add_f64 v[48:49], v[20:21], v[28:29]\n v_add_f64 v[50:51], v[24:25], v[32:33]\n v_add_f64 v[52:53], v[28:29], v[20:21]\n v_add_f64 v[54:55], v[32:33], v[24:25]\n")
                         ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        else if constexpr (T == 2)  // fma, three VGPR sources: (0,1) (2,3) (0,1)
            asm volatile(R8("v_fma_f64 v[40:41], v[20:21], v[22:23], v[24:25]\n v_fma_f64 v[42:43], v[24:25], v[26:27], v[28:29]\n v_fma_f64 v[44:45], v[28:29], v[30:31], v[32:33]\n v_fma_f64 v[46:47], v[32:33], v[34:35], v[20:21]\n v_fma_f64 v[48:49], v[20:21], v[26:27], v[28:29]\n v_fma_f64 v[50:51], v[24:25], v[30:31], v[32:33]\n v_fma_f64 v[52:53], v[28:29], v[34:35], v[20:21]\n v_fma_f64 v[54:55], v[32:33], v[22:23], v[24:25]\n")
                         ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        else if constexpr (T == 3)  // fma, two VGPR sources in different bank pairs + an SGPR pair
            asm volatile(R8("v_fma_f64 v[40:41], v[20:21], v[22:23], s[20:21]\n v_fma_f64 v[42:43], v[24:25], v[26:27], s[20:21]\n v_fma_f64 v[44:45], v[28:29], v[30:31], s[20:21]\n v_fma_f64 v[46:47], v[32:33], v[34:35], s[20:21]\n v_fma_f64 v[48:49], v[20:21], v[26:27], s[20:21]\n v_fma_f64 v[50:51], v[24:25], v[30:31], s[20:21]\n v_fma_f64 v[52:53], v[28:29], v[34:35], s[20:21]\n v_fma_f64 v[54:55], v[32:33], v[22:23], s[20:21]\n")
                         ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s20","s21");
        else if constexpr (T == 4)  // fma, all three sources in bank pair (0,1)
            asm volatile(R8("v_fma_f64 v[40:41], v[20:21], v[24:25], v[28:29]\n v_fma_f64 v[42:43], v[24:25], v[28:29], v[32:33]\n v_fma_f64 v[44:45], v[28:29], v[32:33], v[20:21]\n v_fma_f64 v[46:47], v[32:33], v[20:21], v[24:25]\n v_fma_f64 v[48:49], v[20:21], v[28:29], v[32:33]\n v_fma_f64 v[50:51], v[24:25], v[32:33], v[20:21]\n v_fma_f64 v[52:53], v[28:29], v[20:21], v[24:25]\n v_fma_f64 v[54:55], v[32:33], v[24:25], v[28:29]\n")
                         ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        else if constexpr (T == 5)  // mul, one VGPR source + SGPR
            asm volatile(R8("v_mul_f64 v[40:41], v[20:21], s[20:21]\n v_mul_f64 v[42:43], v[22:23], s[20:21]\n v_mul_f64 v[44:45], v[24:25], s[20:21]\n v_mul_f64 v[46:47], v[26:27], s[20:21]\n v_mul_f64 v[48:49], v[28:29], s[20:21]\n v_mul_f64 v[50:51], v[30:31], s[20:21]\n v_mul_f64 v[52:53], v[32:33], s[20:21]\n v_mul_f64 v[54:55], v[34:35], s[20:21]\n")
                         ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s20","s21");
    }
    double r;
    asm volatile("v_add_f64 %0, v[40:41], v[54:55]" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + seed;
}
template <int T>
void run(const char *name) {
    double *d;
    hipMalloc(&d, 256 * 512 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<T><<<256, 512>>>(d, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T><<<256, 512>>>(d, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %7.3f ms  %5.2f ns per wave-instruction\n", name, ms, ms * 1e6 / (2.0 * ITER * 64));
    hipFree(d);
}
int main() {
    run<0>("v_add_f64, sources in different bank pairs");
    run<1>("v_add_f64, both sources in the same bank pair");
    run<2>("v_fma_f64, three VGPR sources (two share a pair)");
    run<3>("v_fma_f64, two VGPR sources + SGPR pair");
    run<4>("v_fma_f64, three sources in one bank pair");
    run<5>("v_mul_f64, one VGPR source + SGPR pair");
    return 0;
}
