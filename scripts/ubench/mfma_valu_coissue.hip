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
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef int i4 __attribute__((ext_vector_type(4)));

// role by wave: waves [0, 4) = one per SIMD: MFMA; waves [4, 4 + 4 NV) = VALU.  mode bit 0: MFMA waves work; bit 1: VALU
// waves work.  SHAPE 0 = 32x32x2, 1 = 16x16x4, 2 = 4x4x1 (fp32); 3 = v_mfma_f32_32x32x16_bf16, 4 = v_mfma_f32_16x16x32_bf16
template <int SHAPE>
__global__ void __launch_bounds__(768) mix(float* out, int iters_m, int iters_v, int mode, float seed) {
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  // mode bit 2: roles swapped -- the MFMA waves are the workgroup's LAST four (issue arbitration favours older waves)
  if ((mode & 4) ? wave >= nwaves - 4 : wave < 4) {
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
    } else if (SHAPE == 3) {
      const i4 ai = {(int)threadIdx.x, 2, 3, 4}, bi = {5, (int)threadIdx.x, 7, 8};
      const bf8 ab = __builtin_bit_cast(bf8, ai), bb = __builtin_bit_cast(bf8, bi);
      f16 c[4];
      for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
      for (int it = 0; it < iters_m; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c[i], 0, 0, 0);
      float s = 0.f;
      for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += c[i][r];
      if (s == 12345.f) out[0] = s;
    } else if (SHAPE == 4) {
      const i4 ai = {(int)threadIdx.x, 2, 3, 4}, bi = {5, (int)threadIdx.x, 7, 8};
      const bf8 ab = __builtin_bit_cast(bf8, ai), bb = __builtin_bit_cast(bf8, bi);
      f4 c[8];
      for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters_m; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c[i], 0, 0, 0);
      float s = 0.f;
      for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
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
    if (mode & 8) {           // integer vector work instead of fp32 FMAs
      unsigned u[8];
      const unsigned ua = threadIdx.x * 2654435761u;
      for (int i = 0; i < 8; ++i) u[i] = ua + i;
      for (int it = 0; it < iters_v; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ua), "v"(it));
      unsigned s = 0;
      for (int i = 0; i < 8; ++i) s += u[i];
      if (s == 12345u) out[1] = (float)s;
      return;
    }
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

// the same question INSIDE one wave: every MFMA (32x32x2 fp32, independent accumulators) is followed by K independent
// v_fma_f32 of the same wave, order pinned (volatile asm + scheduling barriers).  One wave per SIMD.
template <int K>
__global__ void __launch_bounds__(256) intra(float* out, int iters, float seed) {
  const float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  f16 c[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += c[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.f) out[0] = s;
}

// ... and with the accumulators in the ACC register file (AGPRs: their own read / write ports)
template <int K>
__global__ void __launch_bounds__(256) intra_agpr(float* out, int iters, float seed) {
  const float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  f16 c[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += c[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.f) out[0] = s;
}

template <int K>
static void intra_run(float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float ms = 0;
  const int iters = 4000;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(intra<K>, dim3(256), dim3(256), 0, 0, out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  float ms2 = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(intra_agpr<K>, dim3(256), dim3(256), 0, 0, out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms2, e0, e1);
  }
  printf("one wave per SIMD, each 32x32x2 fp32 MFMA followed by %2d independent v_fma_f32 of the same wave: %.3f ms = %.1f cycles per MFMA group at 2.4 GHz;  accumulators in AGPRs: %.3f ms = %.1f cycles\n",
         K, ms, ms * 1e-3 * 2.4e9 / (iters * 4.0), ms2, ms2 * 1e-3 * 2.4e9 / (iters * 4.0));
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
    const float ts = run<SHAPE>(out, threads, im, iv, 7);
    const float ti = run<SHAPE>(out, threads, im, iv, 8 | 2), tbi = run<SHAPE>(out, threads, im, iv, 8 | 3);
    printf("   integer VALU (v_xad_u32) instead: alone %.3f ms, together with the MFMA waves %.3f ms (sum %.3f)\n", ti, tbi, tm + ti);
    printf("%s + %d VALU wave(s) per SIMD: MFMA alone %.3f ms, VALU alone %.3f ms, together %.3f ms (MFMA waves last: %.3f)  (sum %.3f, max %.3f: overlap %.0f %%)\n",
           name, nv, tm, tv, tb, ts, tm + tv, tm > tv ? tm : tv, 100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
  }
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  study<0>(out, "v_mfma_f32_32x32x2_f32", 4, 65);
  study<1>(out, "v_mfma_f32_16x16x4_f32", 8, 36);
  study<2>(out, "v_mfma_f32_4x4x1_16b  ", 8, 10.3);
  study<3>(out, "v_mfma_f32_32x32x16_bf16", 4, 32);
  study<4>(out, "v_mfma_f32_16x16x32_bf16", 8, 16);
  intra_run<0>(out), intra_run<4>(out), intra_run<8>(out), intra_run<12>(out), intra_run<16>(out), intra_run<24>(out), intra_run<32>(out);
  return 0;
}
