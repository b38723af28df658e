// Is a packed fp32 instruction (v_pk_fma_f32: two FMAs per lane) worth two plain ones on gfx950?  8 independent chains of
// each per wave, 1 / 2 / 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_pk_rate valu_pk_rate.hip && ./valu_pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(1024) rate(float* out, int iters, float seed) {
  const float a = seed + threadIdx.x, b = seed * 0.5f + 1e-3f * threadIdx.x;
  if (PK) {
    f2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f2{a + i, a - i};
    const f2 bb = {b, b * 0.5f}, aa = {a, -a};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(bb), "v"(aa));
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    if (s == 12345.f) out[0] = s;
  } else {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[0] = s;
  }
}

int main() {
  float* out;
  (void)hipMalloc(&out, 64);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int pk = 0; pk < 2; ++pk)
    for (int wps = 1; wps <= 4; wps *= 2) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.f);
        else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
      }
      const double instr = (double)iters * 8 * wps;   // per SIMD
      printf("%s  waves/SIMD %d: %.3f ms  %.2f cycles per instruction per SIMD at 2.4 GHz  (%.1f TFLOP/s)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps,
             ms, ms * 1e-3 * 2.4e9 / instr, instr * 1024 * 64 * 2 * (pk ? 2 : 1) / (ms * 1e-3) / 1e12);
    }
  return 0;
}
