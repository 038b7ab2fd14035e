// 2 x 2 transposes between a pair of registers and one lane bit of a wavefront, without LDS: v_permlane32_swap / v_permlane16_swap
// for lane bits 5 and 4 (one instruction per 32-bit register pair), DPP moves (row_ror:8, row_half_mirror + quad_perm, quad_perm)
// and two selects for bits 3..0.  Used by the half transforms of the torus latency kernel (fft_half_f64.hpp) and, as an option,
// by the exchanges of the whole-wave transform (fft_wave_f64.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef BMI_LANETR_ASM
#define BMI_LANETR_ASM 1   // lane bits 1 and 0: 1 = v_cndmask_b32_dpp by inline assembly (2 instructions per register pair), 0 = compiler (4)
#endif

namespace lanetr {

// (a: the register whose index bit is clear, b: the one whose bit is set)
__device__ __forceinline__ void swap_dw32(uint32_t &a, uint32_t &b) {   // lane bit 5
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swap_dw16(uint32_t &a, uint32_t &b) {   // lane bit 4
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// lane ^ MASK for MASK in {8, 4, 2, 1}
template <int MASK>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (MASK == 8) return dpp<0x128>(v);                       // row_ror:8
    else if constexpr (MASK == 4) return dpp<0x1B>(dpp<0x141>(v));       // row_half_mirror, then quad_perm [3,2,1,0]
    else if constexpr (MASK == 2) return dpp<0x4E>(v);                   // quad_perm [2,3,0,1]
    else return dpp<0xB1>(v);                                            // quad_perm [1,0,3,2]
}
// DPP move that writes only the lanes of the banks (groups of four lanes within a row of sixteen) selected by BANKS; the other
// lanes keep `old`
template <int CTRL, int BANKS>
__device__ __forceinline__ uint32_t dpp_banks(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, BANKS, false);
}
template <int MASK>
__device__ __forceinline__ void swap_dw_dpp(uint32_t &a, uint32_t &b, bool bit) {
    if constexpr (MASK == 8) {
        // lanes with bit 3 set are banks 2, 3 of a row: they take b from the lane 8 below (row_shr:8); the others take a from
        // the lane 8 above (row_shl:8) - one instruction per direction, no select
        const uint32_t na = dpp_banks<0x118, 0xC>(a, b), nb = dpp_banks<0x108, 0x3>(b, a);
        a = na;
        b = nb;
    } else if constexpr (MASK == 4) {
        const uint32_t na = dpp_banks<0x114, 0xA>(a, b), nb = dpp_banks<0x104, 0x5>(b, a);   // row_shr:4 into banks 1, 3; row_shl:4 into banks 0, 2
        a = na;
        b = nb;
    } else {
        const uint32_t ta = lane_xor<MASK>(a), tb = lane_xor<MASK>(b);
        a = bit ? tb : a;
        b = bit ? b : ta;
    }
}
template <int LANE_BIT>
__device__ __forceinline__ void tr_double(double &a, double &b, int lane) {
    uint32_t al = (uint32_t)__double2loint(a), ah = (uint32_t)__double2hiint(a);
    uint32_t bl = (uint32_t)__double2loint(b), bh = (uint32_t)__double2hiint(b);
    if constexpr (LANE_BIT == 5) {
        swap_dw32(al, bl);
        swap_dw32(ah, bh);
    } else if constexpr (LANE_BIT == 4) {
        swap_dw16(al, bl);
        swap_dw16(ah, bh);
    } else if constexpr (LANE_BIT >= 2 || !BMI_LANETR_ASM) {
        const bool bit = (lane >> LANE_BIT) & 1;
        swap_dw_dpp<(1 << LANE_BIT)>(al, bl, bit);
        swap_dw_dpp<(1 << LANE_BIT)>(ah, bh, bit);
    } else {
        // lane bits 1 and 0 (inside a quad, below the granularity of DPP's bank mask): the select and the quad permutation in ONE
        // instruction each way, v_cndmask_b32_dpp with the lane mask in VCC (the compiler leaves them apart: four instructions)
        constexpr unsigned long long SET = LANE_BIT == 1 ? 0xCCCCCCCCCCCCCCCCull : 0xAAAAAAAAAAAAAAAAull;
        uint32_t nal, nah, nbl, nbh;
        if constexpr (LANE_BIT == 1) {
            asm volatile(
                "s_mov_b64 vcc, %8\n\t"
                "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 vcc, %9\n\t"
                "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                : "=&v"(nal), "=&v"(nah), "=&v"(nbl), "=&v"(nbh)
                : "v"(al), "v"(ah), "v"(bl), "v"(bh), "s"(~SET), "s"(SET)
                : "vcc");
        } else {
            asm volatile(
                "s_mov_b64 vcc, %8\n\t"
                "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 vcc, %9\n\t"
                "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                : "=&v"(nal), "=&v"(nah), "=&v"(nbl), "=&v"(nbh)
                : "v"(al), "v"(ah), "v"(bl), "v"(bh), "s"(~SET), "s"(SET)
                : "vcc");
        }
        al = nal; ah = nah; bl = nbl; bh = nbh;
    }
    a = __hiloint2double((int)ah, (int)al);
    b = __hiloint2double((int)bh, (int)bl);
}
}  // namespace lanetr
