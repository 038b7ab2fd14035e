// Issue cost (cycles per wave-instruction) of the VALU instructions the Goldilocks arithmetic is made of, on gfx950.
// Each test runs a long unrolled sequence in every wave of a 256-thread block (1 or 2 waves per SIMD) and
// reports cycles / instruction from s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 64
#define ITER 2000

template <int T>
__global__ void __launch_bounds__(1024) k(uint64_t *out, uint64_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint64_t b = seed * 31 + 7;
    uint32_t c0 = (uint32_t)a0, c1 = (uint32_t)a1, c2 = c0 * 3, c3 = c1 * 5;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if constexpr (T == 0) {  // v_lshl_add_u64 independent x8
                asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                             "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (T == 1) {  // v_mad_u64_u32 independent x8
                asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n"
                             "v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n v_mad_u64_u32 %7, s[20:21], %8, %9, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c0), "v"(c1) : "s20", "s21");
            } else if constexpr (T == 2) {  // v_add_u32 independent x8 (baseline 32-bit op)
                asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                             "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"((uint32_t)b));
            } else if constexpr (T == 3) {  // v_cmp_lt_u64 + 2x v_cndmask (the conditional-fix pattern), 4 independent
                asm volatile("v_cmp_lt_u64 vcc, %4, %6\n v_cmp_lt_u64 s[20:21], %5, %6\n v_cmp_lt_u64 s[22:23], %6, %4\n v_cmp_lt_u64 s[24:25], %6, %5\n"
                             "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[22:23]\n v_cndmask_b32 %3, %3, %0, s[24:25]\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(b) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25");
            } else if constexpr (T == 4) {  // carry chain: v_add_co_u32 + v_addc_co_u32, 4 independent chains
                asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, s[20:21], %1, %4\n v_add_co_u32 %2, s[22:23], %2, %4\n v_add_co_u32 %3, s[24:25], %3, %4\n"
                             "v_addc_co_u32 %0, vcc, %0, %5, vcc\n v_addc_co_u32 %1, s[20:21], %1, %5, s[20:21]\n v_addc_co_u32 %2, s[22:23], %2, %5, s[22:23]\n v_addc_co_u32 %3, s[24:25], %3, %5, s[24:25]\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"((uint32_t)b), "v"((uint32_t)(b >> 32)) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25");
            } else if constexpr (T == 5) {  // v_lshlrev_b64 x8
                asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3\n"
                             "v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (T == 6) {  // v_mul_lo_u32 + v_mul_hi_u32 x4
                asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                             "v_mul_lo_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"((uint32_t)b));
            } else if constexpr (T == 7) {  // dependent chain of v_add_u32 (latency)
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                             "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                             : "+v"(c0) : "v"((uint32_t)b));
            } else if constexpr (T == 8) {  // dependent chain of v_lshl_add_u64 (latency)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %0, %0, 0, %1\n"
                             "v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %0, %0, 0, %1\n"
                             : "+v"(a0) : "v"(b));
            } else if constexpr (T == 9) {  // v_cmp_lt_u64 x8 to distinct SGPR pairs (no consumers)
                asm volatile("v_cmp_lt_u64 s[20:21], %0, %1\n v_cmp_lt_u64 s[22:23], %1, %0\n v_cmp_lt_u64 s[24:25], %0, %1\n v_cmp_lt_u64 s[26:27], %1, %0\n"
                             "v_cmp_lt_u64 s[20:21], %0, %1\n v_cmp_lt_u64 s[22:23], %1, %0\n v_cmp_lt_u64 s[24:25], %0, %1\n v_cmp_lt_u64 s[26:27], %1, %0\n"
                             : : "v"(a0), "v"(a1) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            } else if constexpr (T == 10) {  // v_mov_b32 dpp quad_perm x8
                asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
            }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ c0 ^ c1 ^ c2 ^ c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = t1 - t0;
}

template <int T>
void run(const char *name, int threads) {
    uint64_t *d;
    const int blocks = 256;
    hipMalloc(&d, (blocks * threads + 1) * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, 12345ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, 12345ull);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t cyc;
    hipMemcpy(&cyc, d + blocks * threads, 8, hipMemcpyDeviceToHost);
    double winstr = (double)blocks * (threads / 64) * ITER * REP * 10;
    printf("%-40s waves/SIMD=%d  %.2f ticks/instr/wave   wall %.3f ms   %.1f G wave-instr/s  (%.2f per SIMD-cycle @2.4GHz)\n", name,
           threads / 256, (double)cyc / (ITER * REP), ms / 10, winstr / (ms * 1e-3) / 1e9, winstr / (ms * 1e-3) / (1024 * 2.4e9));
    hipFree(d);
}

int main() {
    for (int th : {256, 512, 768, 1024}) {
        run<2>("v_add_u32 (independent)", th);
        run<7>("v_add_u32 (dependent chain)", th);
        run<0>("v_lshl_add_u64 (independent)", th);
        run<8>("v_lshl_add_u64 (dependent chain)", th);
        run<1>("v_mad_u64_u32 (independent)", th);
        run<6>("v_mul_lo/hi_u32 (independent)", th);
        run<5>("v_lshlrev_b64 (independent)", th);
        run<9>("v_cmp_lt_u64 -> sgpr (independent)", th);
        run<3>("4x v_cmp_lt_u64 then 4x v_cndmask", th);
        run<4>("4x v_add_co then 4x v_addc_co", th);
        run<10>("v_mov_b32_dpp quad_perm (independent)", th);
    }
    return 0;
}
