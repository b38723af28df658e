// What of the three-piece operand split hides under the matrix instructions of the SAME wave on gfx950?  One wave per SIMD
// (as the MVSNeRF MLP runs: 460 registers), groups of six dependent v_mfma_f32_32x32x16_bf16 (192 matrix cycles) with P
// split pairs interleaved between them (one pair's instructions behind each of the first P MFMAs):
//   form B  round-to-nearest pieces: v_cvt_pk_bf16_f32, 2 expands (shift / and), v_pk_add_f32, ...   9 instr., 5 of them fp
//   form T  truncated pieces: 2 x v_and, v_perm_b32, v_pk_add_f32, 2 x v_and, v_perm_b32, v_pk_add_f32, v_perm_b32
//                                                                                            9 instr., 2 of them fp
// (T is an error-free split as well: 8 + 8 + 8 significand bits by truncation.)
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/sum scripts/ubench/split_under_mfma.hip && /tmp/sum
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int FORM>
__device__ __forceinline__ unsigned split_pair(f2& v) {
  unsigned ph, pm, pl;
  if constexpr (FORM == 0) {
    ph = cvt_pk(v[0], v[1]);
    const f2 r1 = v - f2{__uint_as_float(ph << 16), __uint_as_float(ph & 0xffff0000u)};
    pm = cvt_pk(r1[0], r1[1]);
    const f2 r2 = r1 - f2{__uint_as_float(pm << 16), __uint_as_float(pm & 0xffff0000u)};
    pl = cvt_pk(r2[0], r2[1]);
  } else {
    const f2 h = {__uint_as_float(__float_as_uint(v[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(v[1]) & 0xffff0000u)};
    ph = __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u);
    const f2 r1 = v - h;
    const f2 m = {__uint_as_float(__float_as_uint(r1[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(r1[1]) & 0xffff0000u)};
    pm = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
    const f2 r2 = r1 - m;
    pl = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
  }
  v[0] = __uint_as_float(__float_as_uint(v[0]) ^ (pl & 0xff));      // keeps the values changing
  return ph ^ pm ^ pl;
}
template <int FORM, int P, bool MFMA>
__global__ void __launch_bounds__(256, 1) k(const float* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc0 = {}, acc1 = {};
  u32x4 a = {(unsigned)lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, (unsigned)lane, 0x3f803f80u, 0x3f803f80u};
  f2 v[6];
  for (int j = 0; j < 6; ++j) v[j] = f2{in[(threadIdx.x * 12 + 2 * j) & 4095], in[(threadIdx.x * 12 + 2 * j + 1) & 4095]};
  unsigned x = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x16& acc = g ? acc1 : acc0;
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        if constexpr (MFMA)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (m < P) x ^= split_pair<FORM>(v[m]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)x;
}
template <int FORM, int P, bool MFMA>
float run(const float* in, float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FORM, P, MFMA>), dim3(256), dim3(256), 0, 0, in, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms * 1e-3f * 2.4e9f / (2.f * iters);   // cycles at 2.4 GHz per group of six MFMAs
}
int main() {
  float *in, *out;
  (void)hipMalloc(&in, 4096 * 4), (void)hipMalloc(&out, 256 * 256 * 4);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 0.37f + 1e-3f * i;
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int it = 4000;
  printf("cycles (wall time x 2.4 GHz) per group of six dependent v_mfma_f32_32x32x16_bf16, one wave per SIMD\n");
  printf("pairs per group         0      1      2      3      4      6\n");
  printf("form B + MFMAs    %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f\n", run<0, 0, true>(in, out, it), run<0, 1, true>(in, out, it), run<0, 2, true>(in, out, it), run<0, 3, true>(in, out, it), run<0, 4, true>(in, out, it), run<0, 6, true>(in, out, it));
  printf("form T + MFMAs    %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f\n", run<1, 0, true>(in, out, it), run<1, 1, true>(in, out, it), run<1, 2, true>(in, out, it), run<1, 3, true>(in, out, it), run<1, 4, true>(in, out, it), run<1, 6, true>(in, out, it));
  printf("form B alone      %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f\n", run<0, 0, false>(in, out, it), run<0, 1, false>(in, out, it), run<0, 2, false>(in, out, it), run<0, 3, false>(in, out, it), run<0, 4, false>(in, out, it), run<0, 6, false>(in, out, it));
  printf("form T alone      %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f\n", run<1, 0, false>(in, out, it), run<1, 1, false>(in, out, it), run<1, 2, false>(in, out, it), run<1, 3, false>(in, out, it), run<1, 4, false>(in, out, it), run<1, 6, false>(in, out, it));
  return 0;
}
