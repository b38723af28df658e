// Device helpers shared by the LDS-window plane sweeps (sweep_ring.hip, sweep_zp.hip); gfx950 only.
#pragma once
#include "bmv_common.hpp"

namespace bmv {
namespace sweep_util {

using i32x4 = __attribute__((ext_vector_type(4))) int;

// min and max over every 16-lane row (4 DPP steps each; the two chains fill each other's DPP wait states)
__device__ __forceinline__ void row_min_max16(float& lo, float& hi) {
  asm volatile(
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(lo), "+v"(hi));
}
// min and max over every group of 8 lanes (3 DPP steps), two pairs at once
__device__ __forceinline__ void oct_min_max2x(float& lo0, float& hi0, float& lo1, float& hi1) {
  asm volatile(
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1));
}

// mask ? b : a per lane, mask = a 64-bit lane mask in scalar registers (written as asm so that the compiler cannot
// merge two select stages into one 16-way select on a computed index: it did, 45 instructions per output)
__device__ __forceinline__ float lane_select(float a, float b, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
  return r;
}

__device__ __forceinline__ int floor_to_int(float x) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

__device__ __forceinline__ float4 lds4(const char* base, unsigned byte) {
  return *reinterpret_cast<const float4*>(base + byte);
}

// workgroup barrier that waits for this wave's LDS traffic only (a __syncthreads() fence may wait for vmcnt(0))
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float rl(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

}  // namespace sweep_util
}  // namespace bmv
