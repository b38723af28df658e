// Three ways to split a pair of fp32 values into three bf16 pieces each (the B operands of the bf16 x 3 matrix kernels):
//   A  cvt_pk, 2 x (expand, expand, sub, sub), ...            11 vector instructions per pair (mlp.hpp, round 5)
//   B  the two residuals as v_pk_add_f32                        9
//   C  the residuals as v_dot2_f32_bf16 (v - 1 * hi + 0 * hi')  7      D  ... and the compiler's own fp32 -> bf16 conversion
// Checks that B, C and D produce A's pieces bit for bit on random and special values, and times each form at 1, 2 and 4
// waves per SIMD.  RESULT (MI355X, ROCm 7.2, profiles/r6/split3_forms_ubench.txt): B is bit-identical and ~5 % faster --
// shipped in csrc/mvs.hip; C / D do NOT reproduce A (hand-written or through the builtin, the dot form's first residual
// comes back as the value itself and the mid / low pieces differ from A's on every pair -- whatever v_dot2_f32_bf16 does
// with these operands, it is not v - hi) and, behind the s_nop 2 the compiler puts between the dot and its reader, they
// are no faster than B: not pursued.     hipcc --offload-arch=gfx950 -O3 -o /tmp/split3 scripts/ubench/split3_dot2.hip && /tmp/split3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (the residual through the compiler's builtin: hand-written v_dot2_f32_bf16 directly behind a hand-written v_cvt_pk_bf16_f32
// read a STALE register -- the hazard is the compiler's to cover, it puts s_nop 2 between the dot and its reader)
__device__ __forceinline__ f2 resid(f2 v, unsigned p) {
  const b2 h = __builtin_bit_cast(b2, p);
  const b2 c0 = {(__bf16)-1.0f, (__bf16)0.0f}, c1 = {(__bf16)0.0f, (__bf16)-1.0f};
  return f2{__builtin_amdgcn_fdot2_f32_bf16(h, c0, v[0], false), __builtin_amdgcn_fdot2_f32_bf16(h, c1, v[1], false)};
}
template <int MODE>
__device__ __forceinline__ void split(float v0, float v1, unsigned& ph, unsigned& pm, unsigned& pl) {
  if constexpr (MODE >= 2) {
    auto cv = [](f2 x) { return MODE == 3 ? __builtin_bit_cast(unsigned, __builtin_convertvector(x, b2)) : cvt_pk(x[0], x[1]); };
    const f2 v = {v0, v1};
    ph = cv(v);
    const f2 r1 = resid(v, ph);
    pm = cv(r1);
    const f2 r2 = resid(r1, pm);
    pl = cv(r2);
    return;
  }
  ph = cvt_pk(v0, v1);
  float r10, r11;
  if constexpr (MODE == 0) {
    r10 = v0 - __uint_as_float(ph << 16), r11 = v1 - __uint_as_float(ph & 0xffff0000u);
  } else if constexpr (MODE == 1) {
    f2 r = f2{v0, v1} - f2{__uint_as_float(ph << 16), __uint_as_float(ph & 0xffff0000u)};
    r10 = r[0], r11 = r[1];
  } else {
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r10) : "v"(ph), "v"(0x0000BF80u), "v"(v0));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r11) : "v"(ph), "v"(0xBF800000u), "v"(v1));
  }
  pm = cvt_pk(r10, r11);
  float r20, r21;
  if constexpr (MODE == 0) {
    r20 = r10 - __uint_as_float(pm << 16), r21 = r11 - __uint_as_float(pm & 0xffff0000u);
  } else if constexpr (MODE == 1) {
    f2 r = f2{r10, r11} - f2{__uint_as_float(pm << 16), __uint_as_float(pm & 0xffff0000u)};
    r20 = r[0], r21 = r[1];
  } else {
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r20) : "v"(pm), "v"(0x0000BF80u), "v"(r10));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r21) : "v"(pm), "v"(0xBF800000u), "v"(r11));
  }
  pl = cvt_pk(r20, r21);
}
template <int MODE>
__global__ void check_kernel(const float* in, unsigned* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned a, b, c;
  split<MODE>(in[2 * i], in[2 * i + 1], a, b, c);
  out[3 * i] = a, out[3 * i + 1] = b, out[3 * i + 2] = c;
}
template <int MODE>
__global__ void __launch_bounds__(256) time_kernel(const float* in, unsigned* out, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = in[(i * 8 + j) & 4095];
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {          // 4 independent pairs per iteration
      unsigned a, b, c;
      split<MODE>(v[j], v[j + 1], a, b, c);
      acc ^= a ^ b ^ c;
      v[j] = __uint_as_float(__float_as_uint(v[j]) ^ (acc & 0x3ff));   // keeps the values changing, in range
    }
  }
  out[i] = acc;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(2 * n);
  srand(1);
  for (int i = 0; i < 2 * n; ++i) {
    unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
    unsigned ex = (u >> 23) & 0xff;
    if (ex == 255) u &= ~(1u << 30);                 // no inf / nan in the random part
    if (i % 3 == 0) u = (u & 0x807fffffu) | ((100u + (unsigned)(rand() % 56)) << 23);   // ordinary magnitudes
    memcpy(&h[i], &u, 4);
  }
  const float special[] = {0.f, -0.f, 1.f, -1.f, 1.17549435e-38f, 1e-39f, 3.4e38f, 1.0039062f, 0.99609375f, 65504.f, 1e-30f, -1e-30f};
  for (int i = 0; i < 12; ++i) h[i] = special[i], h[24 + 2 * i] = special[i], h[24 + 2 * i + 1] = special[11 - i];
  float* d_in; unsigned* d_out[4];
  hipMalloc(&d_in, 8 * n);
  hipMemcpy(d_in, h.data(), 8 * n, hipMemcpyHostToDevice);
  std::vector<unsigned> o[4];
  for (int m = 0; m < 4; ++m) {
    hipMalloc(&d_out[m], 12 * (size_t)n);
    if (m == 0) hipLaunchKernelGGL(check_kernel<0>, dim3(n / 256), dim3(256), 0, 0, d_in, d_out[m], n);
    if (m == 1) hipLaunchKernelGGL(check_kernel<1>, dim3(n / 256), dim3(256), 0, 0, d_in, d_out[m], n);
    if (m == 2) hipLaunchKernelGGL(check_kernel<2>, dim3(n / 256), dim3(256), 0, 0, d_in, d_out[m], n);
    if (m == 3) hipLaunchKernelGGL(check_kernel<3>, dim3(n / 256), dim3(256), 0, 0, d_in, d_out[m], n);
    o[m].resize(3 * (size_t)n);
    hipMemcpy(o[m].data(), d_out[m], 12 * (size_t)n, hipMemcpyDeviceToHost);
  }
  for (int m = 1; m < 4; ++m) {
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < 3 * (size_t)n; ++i)
      if (o[m][i] != o[0][i]) { if (!bad) first = i; ++bad; }
    printf("form %c vs A on %d pairs: %zu of %zu piece words differ", "ABCD"[m], n, bad, 3 * (size_t)n);
    if (bad) printf(" (first: pair %zu values %g %g: A %08x, %c %08x)", first / 3, h[2 * (first / 3)], h[2 * (first / 3) + 1], o[0][first], "ABCD"[m], o[m][first]);
    printf("\n");
  }
  if (getenv("SPLIT3_DUMP"))
    for (int i = 0; i < 14; ++i)
      printf("pair %2d (%13g, %13g): A %08x %08x %08x | B %08x %08x %08x | C %08x %08x %08x\n", i, h[2 * i], h[2 * i + 1], o[0][3 * i],
             o[0][3 * i + 1], o[0][3 * i + 2], o[1][3 * i], o[1][3 * i + 1], o[1][3 * i + 2], o[2][3 * i], o[2][3 * i + 1], o[2][3 * i + 2]);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 2000;
  for (int wps = 1; wps <= 4; wps *= 2)
    for (int m = 0; m < 4; ++m) {
      dim3 grid(256 * wps), block(256);               // 256 CUs x 4 SIMDs x wps waves
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(time_kernel<0>, grid, block, 0, 0, d_in, d_out[0], iters);
        if (m == 1) hipLaunchKernelGGL(time_kernel<1>, grid, block, 0, 0, d_in, d_out[0], iters);
        if (m == 2) hipLaunchKernelGGL(time_kernel<2>, grid, block, 0, 0, d_in, d_out[0], iters);
        if (m == 3) hipLaunchKernelGGL(time_kernel<3>, grid, block, 0, 0, d_in, d_out[0], iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      double pairs_per_simd = (double)iters * 4 * wps;
      printf("form %c, %d wave(s) per SIMD: %.3f ms, %.1f cycles per pair and SIMD at 2.4 GHz\n", "ABCD"[m], wps, ms,
             ms * 1e-3 * 2.4e9 / pairs_per_simd);
    }
  return 0;
}
