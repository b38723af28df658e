// Training-mode batch normalisation (+ ReLU) of the convolution stacks (ConvBnReLU / ConvBnReLU3D,
// lib/networks/enerf/utils.py:10-33; FeatureNet feature_net.py:4-36, cost regularisers cost_reg_net.py:4-86) for the
// fine-tune leg.  MIOpen's spatial batch norm costs ~73 us per call on gfx950 whatever the tensor size (23 layers x
// forward + backward = 3.4 ms of a 22 ms step); these kernels are plain HBM-bound passes: one statistics pass + one
// apply pass forward, one reduction pass + one apply pass backward.
//
// x (N, C, S) planar (S = D*H*W or H*W).  Statistics are merged from per-block (count, mean, M2) triples (Chan et
// al.), each block accumulating around a pivot (its first element), so nothing cancels against the channel mean.
#include "bmv_common.hpp"

namespace bmv {

constexpr int kBnThreads = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// grid (N * cpp, C): block (n, k) reduces elements [k * per_chunk, ...) of plane (n, c);
// partial[(c * chunks + block) * 3 + {0,1,2}] = count, mean, M2 of the chunk
__global__ void __launch_bounds__(kBnThreads) bn_stats_kernel(const float* __restrict__ x, int C, long S, int cpp,
                                                              long per_chunk, float* __restrict__ partial) {
  __shared__ float sh[4];
  const int c = blockIdx.y, chunks = gridDim.x, n = blockIdx.x / cpp, k = blockIdx.x - n * cpp;
  const long begin = (long)k * per_chunk, end = min(begin + per_chunk, S);
  const float* __restrict__ xp = x + ((long)n * C + c) * S;
  float cnt = 0.f, mean = 0.f, m2 = 0.f;
  if (begin < end) {
    const float pivot = xp[begin];
    float sd = 0.f, sd2 = 0.f;
    for (long i = begin + threadIdx.x; i < end; i += kBnThreads) {
      const float d = xp[i] - pivot;
      sd += d, sd2 += d * d;
    }
    sd = block_sum(sd, sh), sd2 = block_sum(sd2, sh);
    cnt = (float)(end - begin);
    mean = pivot + sd / cnt;
    m2 = fmaxf(sd2 - sd * sd / cnt, 0.f);
  }
  if (threadIdx.x == 0) {
    float* p = partial + ((long)c * chunks + blockIdx.x) * 3;
    p[0] = cnt, p[1] = mean, p[2] = m2;
  }
}

// one thread per channel: merge the chunks (fixed order), write mean / invstd, update the running statistics
__global__ void bn_stats_finish_kernel(const float* __restrict__ partial, int C, int chunks, float eps, float momentum,
                                       float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                       float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int k = 0; k < chunks; ++k) {
    const float* p = partial + ((long)c * chunks + k) * 3;
    const float nb = p[0];
    if (nb == 0.f) continue;
    const float delta = p[1] - mean, nn = n + nb;
    mean += delta * nb / nn;
    m2 += p[2] + delta * delta * n * nb / nn;
    n = nn;
  }
  const float var = m2 / n;                                  // biased: what normalises the batch
  mean_out[c] = mean;
  invstd_out[c] = rsqrtf(var + eps);
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);   // unbiased
  }
}

// y = act((x - mean) * invstd * w + b), act(v) = v > 0 ? v : slope * v (1 = none, 0 = ReLU, 0.01 = InPlaceABN's leaky
// ReLU); 4 elements per thread where the plane allows it
__global__ void __launch_bounds__(kBnThreads) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              int C, long S, float slope, float* __restrict__ y) {
  const long plane = blockIdx.y;                         // n * C + c
  const int c = (int)(plane % C);
  const float sc = invstd[c] * (w ? w[c] : 1.f), sh = (b ? b[c] : 0.f) - mean[c] * sc;
  const float* xp = x + plane * S;
  float* yp = y + plane * S;
  if ((S & 3) == 0) {
    for (long i = (long)blockIdx.x * kBnThreads + threadIdx.x; i < S / 4; i += (long)gridDim.x * kBnThreads) {
      float4 v = reinterpret_cast<const float4*>(xp)[i];
      v.x = v.x * sc + sh, v.y = v.y * sc + sh, v.z = v.z * sc + sh, v.w = v.w * sc + sh;
      v.x = v.x > 0.f ? v.x : slope * v.x, v.y = v.y > 0.f ? v.y : slope * v.y;
      v.z = v.z > 0.f ? v.z : slope * v.z, v.w = v.w > 0.f ? v.w : slope * v.w;
      reinterpret_cast<float4*>(yp)[i] = v;
    }
  } else {
    for (long i = (long)blockIdx.x * kBnThreads + threadIdx.x; i < S; i += (long)gridDim.x * kBnThreads) {
      float v = xp[i] * sc + sh;
      yp[i] = v > 0.f ? v : slope * v;
    }
  }
}

// backward reductions per channel: sum(g), sum(g * xhat) with g = dy * act'(y) (y > 0 ? 1 : slope); grid as bn_stats_kernel
__global__ void __launch_bounds__(kBnThreads) bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ dy,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, int C, long S, int cpp,
                                                                   long per_chunk, float slope, float* __restrict__ partial) {
  __shared__ float sh[4];
  const int c = blockIdx.y, chunks = gridDim.x, n = blockIdx.x / cpp, k = blockIdx.x - n * cpp;
  const long begin = (long)k * per_chunk, end = min(begin + per_chunk, S), base = ((long)n * C + c) * S;
  const float mu = mean[c], is = invstd[c];
  float sg = 0.f, sgx = 0.f;
  for (long i = begin + threadIdx.x; i < end; i += kBnThreads) {
    float g = dy[base + i];
    if (slope != 1.f && !(y[base + i] > 0.f)) g *= slope;
    sg += g, sgx += g * (x[base + i] - mu) * is;
  }
  sg = block_sum(sg, sh), sgx = block_sum(sgx, sh);
  if (threadIdx.x == 0) {
    float* p = partial + ((long)c * chunks + blockIdx.x) * 2;
    p[0] = sg, p[1] = sgx;
  }
}

__global__ void bn_bwd_finish_kernel(const float* __restrict__ partial, int C, int chunks, float* __restrict__ sums,
                                     float* __restrict__ dw, float* __restrict__ db) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sg = 0.f, sgx = 0.f;
  for (int k = 0; k < chunks; ++k) sg += partial[((long)c * chunks + k) * 2], sgx += partial[((long)c * chunks + k) * 2 + 1];
  sums[2 * c] = sg, sums[2 * c + 1] = sgx;
  if (dw) dw[c] = sgx;
  if (db) db[c] = sg;
}

// dx = w * invstd * (g - sum(g)/n - xhat * sum(g xhat)/n)
__global__ void __launch_bounds__(kBnThreads) bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  const float* __restrict__ dy,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ sums, int C, long S,
                                                                  float inv_n, float slope, float* __restrict__ dx) {
  const long plane = blockIdx.y;
  const int c = (int)(plane % C);
  const float mu = mean[c], is = invstd[c], k = is * (w ? w[c] : 1.f);
  const float a = sums[2 * c] * inv_n, bq = sums[2 * c + 1] * inv_n;
  const long base = plane * S;
  for (long i = (long)blockIdx.x * kBnThreads + threadIdx.x; i < S; i += (long)gridDim.x * kBnThreads) {
    float g = dy[base + i];
    if (slope != 1.f && !(y[base + i] > 0.f)) g *= slope;
    const float xh = (x[base + i] - mu) * is;
    dx[base + i] = k * (g - a - xh * bq);
  }
}

// chunks per (n, c) plane: >= 16 k elements per block, at most ~256 blocks per channel
static int bn_cpp(int N, long S) {
  long c = (S + 16383) / 16384;
  const long cap = 256 / N > 1 ? 256 / N : 1;
  return (int)(c < 1 ? 1 : (c > cap ? cap : c));
}

}  // namespace bmv

using namespace bmv;

extern "C" {

int bmv_bn_chunks(int N, long S) { return N * bn_cpp(N, S); }

int bmv_bn_train_fwd(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                     int N, int C, long S, float eps, float momentum, float act_slope, float* workspace, float* save_mean,
                     float* save_invstd, float* y, bmv_stream_t stream) {
  BMV_REQUIRE(x && workspace && save_mean && save_invstd && y, "bmv_bn_train_fwd: null pointer");
  BMV_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bmv_bn_train_fwd: running stats come in pairs");
  BMV_REQUIRE(N > 0 && C > 0 && S > 0 && (long)N * S > 1, "bmv_bn_train_fwd: bad shape (needs more than one value per channel)");
  BMV_REQUIRE((long)N * C <= 65535 && C <= 65535, "bmv_bn_train_fwd: N * C = %ld planes exceed the launch grid", (long)N * C);
  const int cpp = bn_cpp(N, S), chunks = N * cpp;
  const long per_chunk = (S + cpp - 1) / cpp;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(chunks, C), dim3(kBnThreads), 0, st, x, C, S, cpp, per_chunk, workspace);
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, workspace, C, chunks, eps, momentum,
                     save_mean, save_invstd, running_mean, running_var);
  const long per_plane = (S & 3) == 0 ? S / 4 : S;
  unsigned gx = (unsigned)((per_plane + kBnThreads - 1) / kBnThreads);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(gx, N * C), dim3(kBnThreads), 0, st, x, save_mean, save_invstd, weight, bias, C,
                     S, act_slope, y);
  BMV_LAUNCH_END("bmv_bn_train_fwd");
}

int bmv_bn_train_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* save_mean,
                     const float* save_invstd, int N, int C, long S, float act_slope, float* workspace, float* dx, float* dweight,
                     float* dbias, bmv_stream_t stream) {
  BMV_REQUIRE(x && dy && save_mean && save_invstd && workspace && dx, "bmv_bn_train_bwd: null pointer");
  BMV_REQUIRE(act_slope == 1.f || y, "bmv_bn_train_bwd: the fused activation needs the forward output");
  BMV_REQUIRE(act_slope >= 0.f, "bmv_bn_train_bwd: the activation's mask is read off the sign of y: slope must be >= 0");
  BMV_REQUIRE(N > 0 && C > 0 && S > 0, "bmv_bn_train_bwd: bad shape");
  BMV_REQUIRE((long)N * C <= 65535 && C <= 65535, "bmv_bn_train_bwd: N * C = %ld planes exceed the launch grid", (long)N * C);
  const long total = (long)N * S;
  const int cpp = bn_cpp(N, S), chunks = N * cpp;
  const long per_chunk = (S + cpp - 1) / cpp;
  hipStream_t st = as_stream(stream);
  float* sums = workspace + (long)C * chunks * 2;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(chunks, C), dim3(kBnThreads), 0, st, x, y, dy, save_mean, save_invstd, C,
                     S, cpp, per_chunk, act_slope, workspace);
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, workspace, C, chunks, sums, dweight, dbias);
  unsigned gx = (unsigned)((S + kBnThreads - 1) / kBnThreads);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gx, N * C), dim3(kBnThreads), 0, st, x, y, dy, save_mean, save_invstd,
                     weight, sums, C, S, 1.f / (float)total, act_slope, dx);
  BMV_LAUNCH_END("bmv_bn_train_bwd");
}

}  // extern "C"
