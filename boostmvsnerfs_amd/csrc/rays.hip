// Target-ray generation on the device (the data format in front of the path).
// Reference: lib/datasets/enerf_utils.py:25-31, 62-71 (`build_rays`, full-image branch): numpy on the host, float64,
// then a 10.5 MB host -> device copy per level and frame.  Same arithmetic here (float64 for the two small inverses and
// the per-pixel product, rounded to float32 once at the end), written straight into batch['rays_i'] layout:
//   rays[b, y*w + x] = [ c2w[:3,3] | [x, y, 1] @ (inv(K_s)^T @ c2w[:3,:3]^T) | x, y ],  K_s = K with rows 0-1 * scale.
#include "bmv_common.hpp"

namespace bmv {

// Gauss-Jordan with partial pivoting on an n x n system (n <= 4), float64; returns false for a singular matrix
template <int N>
__device__ bool invert(const double* a_in, double* inv) {
  double a[N][2 * N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) a[i][j] = a_in[i * N + j], a[i][N + j] = (i == j) ? 1.0 : 0.0;
  for (int c = 0; c < N; ++c) {
    int piv = c;
    for (int r = c + 1; r < N; ++r)
      if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    if (a[piv][c] == 0.0) return false;
    if (piv != c)
      for (int j = 0; j < 2 * N; ++j) {
        double t = a[c][j];
        a[c][j] = a[piv][j], a[piv][j] = t;
      }
    const double d = 1.0 / a[c][c];
    for (int j = 0; j < 2 * N; ++j) a[c][j] *= d;
    for (int r = 0; r < N; ++r)
      if (r != c) {
        const double f = a[r][c];
        for (int j = 0; j < 2 * N; ++j) a[r][j] -= f * a[c][j];
      }
  }
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) inv[i * N + j] = a[i][N + j];
  return true;
}

__global__ void __launch_bounds__(256) make_rays_kernel(const float* __restrict__ tar_ext, const float* __restrict__ tar_ixt,
                                                         double scale, int h, int w, float* __restrict__ rays) {
  __shared__ double M[9], O[3];   // M = inv(K_s)^T @ c2w[:3,:3]^T  (row vector [x y 1] @ M), O = c2w[:3,3]
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {
    double E[16], K[9], c2w[16], Ki[9];
    for (int i = 0; i < 16; ++i) E[i] = (double)tar_ext[b * 16 + i];
    for (int i = 0; i < 9; ++i) K[i] = (double)tar_ixt[b * 9 + i];
    if (scale != 1.0)
      for (int i = 0; i < 6; ++i) K[i] *= scale;
    const bool ok4 = invert<4>(E, c2w), ok3 = invert<3>(K, Ki);
    const bool ok = ok4 && ok3;
    for (int r = 0; r < 3; ++r) {
      O[r] = ok ? c2w[r * 4 + 3] : __builtin_nan("");
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;   // (inv(K)^T @ R^T)[r][c] = sum_k inv(K)[k][r] * c2w[c][k]
        for (int k = 0; k < 3; ++k) s += Ki[k * 3 + r] * c2w[c * 4 + k];
        M[r * 3 + c] = ok ? s : __builtin_nan("");
      }
    }
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h * w) return;
  const int y = i / w, x = i - y * w;
  const double fx = (double)x, fy = (double)y;
  float* o = rays + ((size_t)b * h * w + i) * 8;
  float4 lo, hi;
  lo.x = (float)O[0], lo.y = (float)O[1], lo.z = (float)O[2];
  // numpy evaluates the row-vector product left to right: (x*M0c + y*M1c) + 1*M2c
  lo.w = (float)((fx * M[0] + fy * M[3]) + M[6]);
  hi.x = (float)((fx * M[1] + fy * M[4]) + M[7]);
  hi.y = (float)((fx * M[2] + fy * M[5]) + M[8]);
  hi.z = (float)x, hi.w = (float)y;
  reinterpret_cast<float4*>(o)[0] = lo;
  reinterpret_cast<float4*>(o)[1] = hi;
}

}  // namespace bmv

using namespace bmv;

extern "C" int bmv_make_rays(const float* tar_ext, const float* tar_ixt, int B, int h, int w, double scale, float* rays,
                             bmv_stream_t stream) {
  BMV_REQUIRE(tar_ext && tar_ixt && rays, "bmv_make_rays: null pointer");
  BMV_REQUIRE(B > 0 && h > 0 && w > 0 && scale > 0.0, "bmv_make_rays: bad shape (h=%d, w=%d, scale=%g)", h, w, scale);
  hipLaunchKernelGGL(make_rays_kernel, dim3(cdiv((long)h * w, 256), B), dim3(256), 0, as_stream(stream), tar_ext, tar_ixt,
                     scale, h, w, rays);
  BMV_LAUNCH_END("bmv_make_rays");
}
