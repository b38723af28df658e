// VALU issue-rate microbenchmark (gfx950): v_fma_f32 vs v_pk_fma_f32 vs v_pk_mul_f32, 1..8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float m = 1.0001f, c = 0.5f;
  const float2_ m2 = {m, m}, c2 = {c, c};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
      }
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y +
            p6.x + p6.y + p7.x + p7.y;
  if (s == 12345.f) out[0] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode)
    for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
      int blocks = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double winstr = (double)iters * 64;                  // wave-instructions per wave
      double cyc = ms * 1e-3 * 2.4e9 / (winstr * wps);     // cycles per wave-instruction per SIMD (at 2.4 GHz)
      double tflops = (double)blocks * 256 * iters * 64 * (mode ? 4 : 2) / (ms * 1e-3) / 1e12;
      printf("%s waves/SIMD %d: %.3f ms  %.2f cyc/instr/SIMD (if 2.4 GHz)  %.1f TFLOP/s\n", mode ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms,
             cyc, tflops);
    }
  return 0;
}
