// fp32 MFMA shapes on gfx950: lane layout of v_mfma_f32_4x4x1_16b_f32 (probed, not assumed) and the issue rate of
// 4x4x1 (16 blocks) against 16x16x4 and 32x32x2, 1..4 waves per SIMD, independent accumulators.
// The convolution engine's 8-output-channel layers fill only part of a 16-row tile (row pairing: 75 %); the 4-row
// blocks of 4x4x1 would fill 100 % -- IF the small shape issues at the same FLOP rate.
// hipcc --offload-arch=gfx950 -O3 -o mfma_f32_shapes mfma_f32_shapes.hip && ./mfma_f32_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

__global__ void layout_probe(float* out) {
  // A[lane] = 2 lane + 1 (odd), B[lane] = 2^lane: every product (odd x power of two) names its two lanes uniquely
  const int l = threadIdx.x;
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(2 * l + 1), ldexpf(1.f, l), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <int SHAPE>
__global__ void __launch_bounds__(256) rate(float* out, int iters, float seed) {
  const float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  if (SHAPE == 0) {
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 12345.f) out[0] = s;
  } else if (SHAPE == 1) {
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 12345.f) out[0] = s;
  } else {
    f16 c[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += c[i][r];
    if (s == 12345.f) out[0] = s;
  }
}

int main() {
  float* out;
  hipMalloc(&out, 64 * 4 * 4);
  hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, out);
  float h[256];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("v_mfma_f32_4x4x1_16b_f32 with A[lane] = 2 lane + 1, B[lane] = 2^lane: D[reg] at lane l = A[la] * B[lb]\n");
  for (int l = 0; l < 64; l += (l < 8 ? 1 : 13)) {
    printf("  lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      int e = 0;
      const float m = frexpf(h[l * 4 + r], &e);          // value = m 2^e, m in [0.5, 1): odd part = m 2^k
      float odd = m;
      int lb = e;
      while (odd != floorf(odd)) odd *= 2.f, --lb;
      printf("  reg %d = A[%2d] B[%2d]", r, ((int)odd - 1) / 2, lb);
    }
    printf("\n");
  }
  printf("  (expected: reg r at lane l = A[4 (l / 4) + r] * B[l]: block l / 4, row r, column l %% 4)\n");
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 4000;
  const char* names[3] = {"4x4x1 x16 blocks", "16x16x4         ", "32x32x2         "};
  const double flops[3] = {512, 2048, 4096};
  const int per_iter[3] = {8, 8, 4};
  for (int shape = 0; shape < 3; ++shape)
    for (int wps = 1; wps <= 4; wps *= 2) {
      const int blocks = 256 * wps;
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
        if (shape == 1) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
        if (shape == 2) hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double n = (double)iters * per_iter[shape];          // MFMAs per wave
      const double tf = (double)blocks * 4 * n * flops[shape] / (ms * 1e-3) / 1e12;
      printf("%s  waves/SIMD %d: %.3f ms  %.1f TFLOP/s  %.2f cycles per MFMA per SIMD at 2.4 GHz\n", names[shape], wps, ms, tf,
             ms * 1e-3 * 2.4e9 / (n * wps));
    }
  return 0;
}
