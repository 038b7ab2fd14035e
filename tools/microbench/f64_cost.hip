// Issue cost of the f64 VALU instructions the 49-bit field arithmetic is made of (gfx950), relative to v_add_u32.
// Wall-clock (hipEvent) over a long unrolled sequence, 8 independent chains per lane, 2 waves per SIMD on every CU.
// Build: hipcc --offload-arch=gfx950 -O3 f64_cost.hip -o f64_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4000

#define OP8(fmt)                                                                                                   \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)                                            \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c))

#define FMA(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define MUL(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define ADD(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define RND(i) "v_rndne_f64 %" #i ", %" #i "\n"
#define LDX(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define CVT(i) "v_cvt_f64_i32 %" #i ", %" #i "\n"
#define FLR(i) "v_floor_f64 %" #i ", %" #i "\n"

template <int T>
__global__ void __launch_bounds__(512) k(double *out, double seed) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 0.001 + i;
    double b = 1.0000001, c = 1e-9;
    uint32_t u[8];
    for (int i = 0; i < 8; i++) u[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if constexpr (T == 0) OP8(FMA);
            else if constexpr (T == 1) OP8(MUL);
            else if constexpr (T == 2) OP8(ADD);
            else if constexpr (T == 3) OP8(RND);
            else if constexpr (T == 4) OP8(LDX);
            else if constexpr (T == 5) OP8(FLR);
            else if constexpr (T == 6)
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                             : "v"(u[0] | 1u));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int T>
double run(const char *name, double base) {
    double *d;
    const int blocks = 256, threads = 512;
    hipMalloc(&d, blocks * threads * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<T><<<blocks, threads>>>(d, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T><<<blocks, threads>>>(d, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 2 waves per SIMD, each issues ITER*64 instructions
    const double ns_per_instr = ms * 1e6 / (2.0 * ITER * 64);
    printf("%-14s %8.3f ms  %6.2f ns per wave-instruction  (x%.2f of v_add_u32)\n", name, ms, ns_per_instr,
           base > 0 ? ns_per_instr / base : 1.0);
    hipFree(d);
    return ns_per_instr;
}

int main() {
    const double base = run<6>("v_add_u32", 0);
    run<0>("v_fma_f64", base);
    run<1>("v_mul_f64", base);
    run<2>("v_add_f64", base);
    run<3>("v_rndne_f64", base);
    run<4>("v_ldexp_f64", base);
    run<5>("v_floor_f64", base);
    return 0;
}
