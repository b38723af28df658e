// Do fp32 MFMAs of one wave and vector-ALU work of ANOTHER wave of the same SIMD overlap on gfx950?
// The fused renderer (csrc/render.hip, render_pc_kernel) puts one MLP wave (206 v_mfma_f32_32x32x2_f32 + ~920 vector
// instructions per 32-sample tile) and two gather waves (~1900 vector instructions per tile) on every SIMD; its frame time
// is close to the SUM of the matrix time and the vector time.  This probe runs, per SIMD, one wave that only issues
// MFMAs (independent accumulators) and one or two waves that only issue v_fma_f32 chains (independent registers), alone and
// together.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coissue mfma_valu_coissue.hip && ./mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// role by wave: waves [0, 4) = one per SIMD: MFMA; waves [4, 4 + 4 NV) = VALU.  mode bit 0: MFMA waves work; bit 1: VALU
// waves work.  SHAPE 0 = 32x32x2, 1 = 16x16x4, 2 = 4x4x1
template <int SHAPE>
__global__ void __launch_bounds__(768) mix(float* out, int iters_m, int iters_v, int mode, float seed) {
  const int wave = threadIdx.x >> 6;
  const float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  if (wave < 4) {
    if (!(mode & 1)) return;
    if (SHAPE == 0) {
      f16 c[4];
      for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
      for (int it = 0; it < iters_m; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
      float s = 0.f;
      for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += c[i][r];
      if (s == 12345.f) out[0] = s;
    } else {
      f4 c[8];
      for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters_m; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          c[i] = SHAPE == 1 ? __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
      float s = 0.f;
      for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
      if (s == 12345.f) out[0] = s;
    }
  } else {
    if (!(mode & 2)) return;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters_v; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], b, a);      // 8 independent chains: one v_fma_f32 per 4 cycles
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[1] = s;
  }
}

template <int SHAPE>
static float run(float* out, int threads, int im, int iv, int mode) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mix<SHAPE>, dim3(256), dim3(threads), 0, 0, out, im, iv, mode, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

template <int SHAPE>
static void study(float* out, const char* name, int mfma_per_iter, double cyc) {
  // matrix work and vector work sized to take about the same time alone
  const int im = 4000;
  const double t_m_cycles = (double)im * mfma_per_iter * cyc;
  for (int nv = 1; nv <= 2; ++nv) {
    const int threads = 256 + 256 * nv;
    const int iv = (int)(t_m_cycles / (8 * 4 * nv));            // each VALU wave: 8 FMAs x 4 cycles per iteration
    const float tm = run<SHAPE>(out, threads, im, iv, 1), tv = run<SHAPE>(out, threads, im, iv, 2), tb = run<SHAPE>(out, threads, im, iv, 3);
    printf("%s + %d VALU wave(s) per SIMD: MFMA alone %.3f ms, VALU alone %.3f ms, together %.3f ms  (sum %.3f, max %.3f: overlap %.0f %%)\n",
           name, nv, tm, tv, tb, tm + tv, tm > tv ? tm : tv, 100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
  }
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  study<0>(out, "v_mfma_f32_32x32x2_f32", 4, 65);
  study<1>(out, "v_mfma_f32_16x16x4_f32", 8, 36);
  study<2>(out, "v_mfma_f32_4x4x1_16b  ", 8, 10.3);
  return 0;
}
