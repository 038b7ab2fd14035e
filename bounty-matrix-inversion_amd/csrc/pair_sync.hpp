// Synchronisation of the two wavefronts that share a ciphertext in the torus wave-pair kernels (bmi_kernels_t64.hip,
// bmi_kernels_t64f.hip): LDS counters posted with release semantics and polled by the partner.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void pair_post(uint32_t *flag, uint32_t v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// one opaque asm block (as C++ control flow the poll loop makes the register allocator spill, see bmi_kernels_f64.hip)
__device__ __forceinline__ void pair_wait(uint32_t *flag, uint32_t v) {
    const uint32_t addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)flag;
    uint32_t tmp;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %0, %1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32 vcc, %2, %0\n\t"
        "s_cbranch_vccnz 2f\n\t"
        "s_sleep 1\n\t"
        "s_branch 1b\n"
        "2:"
        : "=&v"(tmp)
        : "v"(addr), "s"(v)
        : "vcc", "memory");
}
// nothing moves across this point (neither the compiler's memory operations nor the instruction scheduler's)
__device__ __forceinline__ void pin() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
