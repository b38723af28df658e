// Plane-sweep cost volume (a1-a5): projection matrices, depth hypotheses,
// homography warp, fused variance sweep, depth regression.
// Reference: lib/networks/enerf/utils.py:35-153, 324-351, 722-731.
#include <stdarg.h>

#include <stdlib.h>

#include "bmv_common.hpp"

namespace bmv {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------
// a1: one thread per (b, s).  The 4x4 inverse is done in fp64 (Gauss-Jordan with
// partial pivoting); the result is rounded to fp32 once.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void proj_mats_one(int idx, const float* __restrict__ src_exts,
                                              const float* __restrict__ src_ixts, const float* __restrict__ tar_ext,
                                              const float* __restrict__ tar_ixt, float src_scale, float tar_scale,
                                              int S, float* __restrict__ proj) {
  int b = idx / S;
  const float* Es = src_exts + (size_t)idx * 16;
  const float* Ks = src_ixts + (size_t)idx * 9;
  const float* Et = tar_ext + (size_t)b * 16;
  const float* Kt = tar_ixt + (size_t)b * 9;
  double sp[3][4], tp[4][8];
  for (int r = 0; r < 3; ++r) {
    // the reference scales rows 0-1 of K in fp32 and multiplies in fp32
    float ks[3], kt[3];
    for (int c = 0; c < 3; ++c) {
      ks[c] = r < 2 ? Ks[r * 3 + c] * src_scale : Ks[r * 3 + c];
      kt[c] = r < 2 ? Kt[r * 3 + c] * tar_scale : Kt[r * 3 + c];
    }
    for (int c = 0; c < 4; ++c) {
      float a = 0.f, t = 0.f;
      for (int k = 0; k < 3; ++k) {
        a += ks[k] * Es[k * 4 + c];
        t += kt[k] * Et[k * 4 + c];
      }
      sp[r][c] = (double)a;
      tp[r][c] = (double)t;
    }
  }
  for (int c = 0; c < 4; ++c) tp[3][c] = c == 3 ? 1.0 : 0.0;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) tp[r][4 + c] = r == c ? 1.0 : 0.0;
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    double best = fabs(tp[col][col]);
    for (int r = col + 1; r < 4; ++r)
      if (fabs(tp[r][col]) > best) {
        best = fabs(tp[r][col]);
        piv = r;
      }
    if (piv != col)
      for (int c = 0; c < 8; ++c) {
        double t = tp[col][c];
        tp[col][c] = tp[piv][c];
        tp[piv][c] = t;
      }
    double inv = 1.0 / tp[col][col];
    for (int c = 0; c < 8; ++c) tp[col][c] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == col) continue;
      double f = tp[r][col];
      for (int c = 0; c < 8; ++c) tp[r][c] -= f * tp[col][c];
    }
  }
  float* P = proj + (size_t)idx * 12;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      double a = 0.0;
      for (int k = 0; k < 4; ++k) a += sp[r][k] * (double)(float)tp[k][4 + c];
      P[r * 4 + c] = (float)a;
    }
}

__global__ void proj_mats_kernel(const float* __restrict__ src_exts, const float* __restrict__ src_ixts,
                                 const float* __restrict__ tar_ext, const float* __restrict__ tar_ixt,
                                 float src_scale, float tar_scale, int B, int S, float* __restrict__ proj) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * S) return;
  proj_mats_one(idx, src_exts, src_ixts, tar_ext, tar_ixt, src_scale, tar_scale, S, proj);
}

// ---------------------------------------------------------------------------
// a2
// ---------------------------------------------------------------------------
// grid.z = groups of 8 planes (a 64-plane level on a 64 x 80 map is otherwise 20 workgroups of 64 divisions per thread)
__device__ __forceinline__ void depth_values_uniform_group(int zgroup, const float* __restrict__ near_far, int D,
                                                           int hw, int depth_inv, float* __restrict__ dv,
                                                           float* __restrict__ nf_out) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  float n = near_far[b * 2], f = near_far[b * 2 + 1];
  float step = D > 1 ? 1.f / (float)(D - 1) : 0.f;
  auto value = [&](int d) {
    // torch.linspace(0, 1, D): start + d*step for the first half, end - (D-1-d)*step after
    float t = d < D / 2 ? (float)d * step : 1.f - (float)(D - 1 - d) * step;
    return depth_inv ? 1.f / (1.f / n + t * (1.f / f - 1.f / n)) : n + (f - n) * t;
  };
  const int d0 = zgroup * 8, d1 = min(D, d0 + 8);
  for (int d = d0; d < d1; ++d) dv[((size_t)b * D + d) * hw + i] = value(d);
  if (zgroup == 0) {
    float first = value(0), last = value(D - 1);
    if (depth_inv) {
      first = 1.f / fmaxf(first, 1e-6f);
      last = 1.f / fmaxf(last, 1e-6f);
    }
    nf_out[((size_t)b * 2 + 0) * hw + i] = first;
    nf_out[((size_t)b * 2 + 1) * hw + i] = last;
  }
}

__global__ void depth_values_uniform_kernel(const float* __restrict__ near_far, int D, int hw, int depth_inv,
                                            float* __restrict__ dv, float* __restrict__ nf_out) {
  depth_values_uniform_group(blockIdx.z, near_far, D, hw, depth_inv, dv, nf_out);
}

// Everything a frame needs from the cameras alone, one launch (inference: three tiny launches -- the projection
// matrices of each cascade level and level 0's hypotheses -- otherwise sit in front of the sweeps, ~5 us each of launch
// and single-thread fp64 latency): grid.z = plane groups of the hypotheses + one slice for the projection matrices.
struct FrameSetupArgs {
  const float* src_exts;
  const float* src_ixts;
  const float* tar_ext;
  const float* tar_ixt;
  const float* near_far;
  float* proj;        // (L, B, S, 3, 4)
  float* dv;          // (B, D, h, w)
  float* nf_out;      // (B, 2, h, w)
  float src_scale[4], tar_scale[4];
  int L, B, S, D, hw, depth_inv;
};
__global__ void frame_setup_kernel(FrameSetupArgs a) {
  const int zgroups = (a.D + 7) / 8;
  if ((int)blockIdx.z < zgroups) {
    depth_values_uniform_group(blockIdx.z, a.near_far, a.D, a.hw, a.depth_inv, a.dv, a.nf_out);
    return;
  }
  if (blockIdx.y != 0) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, per = a.B * a.S;
  if (idx >= a.L * per) return;
  const int l = idx / per;
  proj_mats_one(idx - l * per, a.src_exts, a.src_ixts, a.tar_ext, a.tar_ixt, a.src_scale[l], a.tar_scale[l], a.S,
                a.proj + (size_t)l * per * 12);
}

__global__ void depth_values_cascade_kernel(const float* __restrict__ depth, const float* __restrict__ std_,
                                            const float* __restrict__ near_far, int h0, int w0, int h, int w,
                                            int D, float* __restrict__ dv, float* __restrict__ nf_out) {
  int b = blockIdx.y;
  // XCD x writes a contiguous eighth of the pixels = the rows band the plane sweep's workgroups of that XCD read
  int i = xcd_contiguous(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  int hw = h * w;
  if (i >= hw) return;
  int y = i / w, x = i - y * w;
  Lerp1 ly = upsample_axis(y, h0, h), lx = upsample_axis(x, w0, w);
  size_t o = (size_t)b * h0 * w0;
  float dep = upsample_fetch(depth + o, w0, ly, lx);
  float sd = upsample_fetch(std_ + o, w0, ly, lx);
  float nf0 = upsample_fetch(near_far + o * 2, w0, ly, lx);
  float nf1 = upsample_fetch(near_far + o * 2 + (size_t)h0 * w0, w0, ly, lx);
  float hi = dep + sd, lo = dep - sd;
  if (hi > nf0) hi = nf0;  // masked assignment of utils.py:124-127
  if (lo < nf1) lo = nf1;
  float nearv = 1.f / hi, farv = 1.f / lo;
  float step = D > 1 ? 1.f / (float)(D - 1) : 0.f;
  float first = 0.f, last = 0.f;
  for (int d = 0; d < D; ++d) {
    float t = d < D / 2 ? (float)d * step : 1.f - (float)(D - 1 - d) * step;
    float v = nearv + t * (farv - nearv);
    dv[((size_t)b * D + d) * hw + i] = v;
    if (d == 0) first = v;
    if (d == D - 1) last = v;
  }
  nf_out[((size_t)b * 2 + 0) * hw + i] = first;
  nf_out[((size_t)b * 2 + 1) * hw + i] = last;
}

// ---------------------------------------------------------------------------
// a3 / a4 geometry shared by every sweep variant.
// ---------------------------------------------------------------------------
struct Proj {
  float r[12];
};

__device__ __forceinline__ void warp_coord(const float* __restrict__ P, float x, float y, float depth, int Ws, int Hs,
                                           float& gx, float& gy) {
  // R @ [x,y,1] + T / depth  (utils.py:72), sum order of a 3-term matmul row
  float inv = 1.f / depth;  // torch computes T / depth element-wise
  float px = P[0] * x + P[1] * y + P[2] + P[3] / depth;
  float py = P[4] * x + P[5] * y + P[6] + P[7] / depth;
  float pz = P[8] * x + P[9] * y + P[10] + P[11] / depth;
  (void)inv;
  float z = fmaxf(pz, 1e-6f);
  gx = (px / z) / ((float)(Ws - 1) * 0.5f) - 1.f;
  gy = (py / z) / ((float)(Hs - 1) * 0.5f) - 1.f;
}

__global__ void homo_warp_kernel(const float* __restrict__ src, const float* __restrict__ proj,
                                 const float* __restrict__ dv, int C, int Hs, int Ws, int D, int h, int w,
                                 float* __restrict__ warped, float* __restrict__ grid) {
  int b = blockIdx.y;
  size_t nvox = (size_t)D * h * w;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox) return;
  int x = (int)(i % w), y = (int)((i / w) % h);
  float depth = dv[(size_t)b * nvox + i];
  float gx, gy;
  warp_coord(proj + (size_t)b * 12, (float)x, (float)y, depth, Ws, Hs, gx, gy);
  if (grid) {
    grid[((size_t)b * nvox + i) * 2] = gx;
    grid[((size_t)b * nvox + i) * 2 + 1] = gy;
  }
  Taps2 t = taps_zeros(unnorm(gx, Ws), unnorm(gy, Hs), Ws, Hs);
  const float* f = src + (size_t)b * C * Hs * Ws;
  for (int c = 0; c < C; ++c) warped[((size_t)b * C + c) * nvox + i] = tap_fetch(f + (size_t)c * Hs * Ws, t);
}

// Direct-gather sweep: one thread per voxel, CB channels of sum / sum-of-squares
// in registers, the S views innermost so no warped volume ever exists in memory.
template <int CB>
__global__ void __launch_bounds__(256) sweep_direct_kernel(const float* __restrict__ feats,
                                                            const float* __restrict__ proj,
                                                            const float* __restrict__ dv, int S, int C, int Hs,
                                                            int Ws, int D, int h, int w, float* __restrict__ out) {
  int b = blockIdx.z;
  int c0 = blockIdx.y * CB;
  size_t nvox = (size_t)D * h * w;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox) return;
  int x = (int)(i % w), y = (int)((i / w) % h);
  float depth = dv[(size_t)b * nvox + i];
  float acc[CB], acc2[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) acc[c] = acc2[c] = 0.f;
  size_t plane = (size_t)Hs * Ws;
  for (int s = 0; s < S; ++s) {
    float gx, gy;
    warp_coord(proj + ((size_t)b * S + s) * 12, (float)x, (float)y, depth, Ws, Hs, gx, gy);
    Taps2 t = taps_zeros(unnorm(gx, Ws), unnorm(gy, Hs), Ws, Hs);
    const float* f = feats + (((size_t)b * S + s) * C + c0) * plane;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      float v = tap_fetch(f + (size_t)c * plane, t);
      acc[c] += v;
      acc2[c] += v * v;
    }
  }
  float fs = (float)S;
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    float m = acc[c] / fs;
    out[((size_t)b * C + c0 + c) * nvox + i] = acc2[c] / fs - m * m;
  }
}

// ---------------------------------------------------------------------------
// a5: one thread per pixel.  DT > 0: the D logits and hypotheses are loaded once
// (all loads independent and in flight together) and the softmax / mean / variance
// passes run out of registers; DT == 0: generic three-pass fallback for any D.
// ---------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(64) depth_regress_kernel(const float* __restrict__ prob,
                                                            const float* __restrict__ dv, int D, int hw,
                                                            int depth_inv, float* __restrict__ depth,
                                                            float* __restrict__ std_, const void* const* table,
                                                            int s_depth, int s_std) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  const float* p = prob + (size_t)b * D * hw + i;
  const float* v = dv + (size_t)b * D * hw + i;
  float mean = 0.f, var = 0.f;
  if constexpr (DT > 0) {
    float e[DT], val[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      e[d] = p[(size_t)d * hw];
      val[d] = v[(size_t)d * hw];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int d = 0; d < DT; ++d) mx = fmaxf(mx, e[d]);
    float den = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      e[d] = expf(e[d] - mx);
      den += e[d];
      if (depth_inv) val[d] = 1.f / fmaxf(val[d], 1e-6f);
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      e[d] = e[d] / den;
      mean += e[d] * val[d];
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      float df = val[d] - mean;
      var += e[d] * (df * df);
    }
  } else {
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) mx = fmaxf(mx, p[(size_t)d * hw]);
    float den = 0.f;
    for (int d = 0; d < D; ++d) den += expf(p[(size_t)d * hw] - mx);
    for (int d = 0; d < D; ++d) {
      float val = v[(size_t)d * hw];
      if (depth_inv) val = 1.f / fmaxf(val, 1e-6f);
      mean += (expf(p[(size_t)d * hw] - mx) / den) * val;
    }
    for (int d = 0; d < D; ++d) {
      float val = v[(size_t)d * hw];
      if (depth_inv) val = 1.f / fmaxf(val, 1e-6f);
      float df = val - mean;
      var += (expf(p[(size_t)d * hw] - mx) / den) * (df * df);
    }
  }
  const float sd = sqrtf(fmaxf(var, 1e-10f));
  depth[(size_t)b * hw + i] = mean;
  std_[(size_t)b * hw + i] = sd;
  // deferred outputs (bmv_defer_pointer): the maps are ALSO written to the tensors the table points at -- the later
  // kernels of a captured frame read the captured buffers above, the caller gets these (no copy node at the frame's end)
  if (table) {
    float* d2 = deferred_load(table, s_depth, (float*)nullptr);
    float* s2 = deferred_load(table, s_std, (float*)nullptr);
    if (d2 && d2 != depth) d2[(size_t)b * hw + i] = mean;
    if (s2 && s2 != std_) s2[(size_t)b * hw + i] = sd;
  }
}

// The coarse level (64 planes on a few thousand pixels) is 80 one-wave workgroups for the kernel above, each issuing
// ~2500 dependent instructions: latency, not throughput.  Here a pixel is FOUR lanes (16 planes each; lane = pixel %
// 16 + 16 * quarter, so a plane's 16 pixels are one 64-byte run), the four partial results meet by two shuffles.
template <int DT>
__global__ void __launch_bounds__(64) depth_regress_quad_kernel(const float* __restrict__ prob,
                                                                 const float* __restrict__ dv, int hw, int depth_inv,
                                                                 float* __restrict__ depth, float* __restrict__ std_) {
  static_assert(DT % 4 == 0, "planes per quarter");
  constexpr int Q = DT / 4;
  const int b = blockIdx.y;
  const int lane = threadIdx.x, q = lane >> 4;
  const int i = blockIdx.x * 16 + (lane & 15);
  const int ic = i < hw ? i : hw - 1;
  const float* p = prob + ((size_t)b * DT + q * Q) * hw + ic;
  const float* v = dv + ((size_t)b * DT + q * Q) * hw + ic;
  float e[Q], val[Q];
#pragma unroll
  for (int d = 0; d < Q; ++d) {
    e[d] = p[(size_t)d * hw];
    val[d] = v[(size_t)d * hw];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int d = 0; d < Q; ++d) mx = fmaxf(mx, e[d]);
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float den = 0.f;
#pragma unroll
  for (int d = 0; d < Q; ++d) {
    e[d] = expf(e[d] - mx);
    den += e[d];
    if (depth_inv) val[d] = 1.f / fmaxf(val[d], 1e-6f);
  }
  den += __shfl_xor(den, 16);
  den += __shfl_xor(den, 32);
  float mean = 0.f, var = 0.f;
#pragma unroll
  for (int d = 0; d < Q; ++d) {
    e[d] = e[d] / den;
    mean += e[d] * val[d];
  }
  mean += __shfl_xor(mean, 16);
  mean += __shfl_xor(mean, 32);
#pragma unroll
  for (int d = 0; d < Q; ++d) {
    float df = val[d] - mean;
    var += e[d] * (df * df);
  }
  var += __shfl_xor(var, 16);
  var += __shfl_xor(var, 32);
  if (q == 0 && i < hw) {
    depth[(size_t)b * hw + i] = mean;
    std_[(size_t)b * hw + i] = sqrtf(fmaxf(var, 1e-10f));
  }
}


}  // namespace bmv

using namespace bmv;

extern "C" {

int bmv_version(void) { return 1; }
const char* bmv_last_error(void) { return bmv::g_err; }

int bmv_proj_mats(const float* src_exts, const float* src_ixts, const float* tar_ext, const float* tar_ixt,
                  float src_scale, float tar_scale, int B, int S, float* proj, bmv_stream_t stream) {
  BMV_REQUIRE(src_exts && src_ixts && tar_ext && tar_ixt && proj, "bmv_proj_mats: null pointer");
  BMV_REQUIRE(B > 0 && S > 0, "bmv_proj_mats: B=%d S=%d", B, S);
  hipLaunchKernelGGL(proj_mats_kernel, dim3(cdiv(B * S, 64)), dim3(64), 0, as_stream(stream), src_exts, src_ixts,
                     tar_ext, tar_ixt, src_scale, tar_scale, B, S, proj);
  BMV_LAUNCH_END("bmv_proj_mats");
}

int bmv_depth_values_uniform(const float* near_far, int B, int D, int h, int w, int depth_inv, float* depth_values,
                             float* near_far_out, bmv_stream_t stream) {
  BMV_REQUIRE(near_far && depth_values && near_far_out, "bmv_depth_values_uniform: null pointer");
  BMV_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "bmv_depth_values_uniform: bad shape");
  hipLaunchKernelGGL(depth_values_uniform_kernel, dim3(cdiv(h * w, 256), B, cdiv(D, 8)), dim3(256), 0, as_stream(stream),
                     near_far, D, h * w, depth_inv, depth_values, near_far_out);
  BMV_LAUNCH_END("bmv_depth_values_uniform");
}

int bmv_frame_setup(const float* src_exts, const float* src_ixts, const float* tar_ext, const float* tar_ixt,
                    const float* src_scales, const float* tar_scales, int L, int B, int S, float* proj,
                    const float* near_far, int D, int h, int w, int depth_inv, float* depth_values,
                    float* near_far_out, bmv_stream_t stream) {
  BMV_REQUIRE(src_exts && src_ixts && tar_ext && tar_ixt && src_scales && tar_scales && proj && near_far &&
                  depth_values && near_far_out,
              "bmv_frame_setup: null pointer");
  BMV_REQUIRE(L > 0 && L <= 4 && B > 0 && S > 0 && D > 0 && h > 0 && w > 0, "bmv_frame_setup: bad shape (L=%d)", L);
  BMV_REQUIRE(L * B * S <= 256 * (int)cdiv(h * w, 256), "bmv_frame_setup: %d projection matrices do not fit one grid slice",
              L * B * S);
  FrameSetupArgs a;
  a.src_exts = src_exts, a.src_ixts = src_ixts, a.tar_ext = tar_ext, a.tar_ixt = tar_ixt, a.near_far = near_far;
  a.proj = proj, a.dv = depth_values, a.nf_out = near_far_out;
  for (int l = 0; l < L; ++l) a.src_scale[l] = src_scales[l], a.tar_scale[l] = tar_scales[l];
  a.L = L, a.B = B, a.S = S, a.D = D, a.hw = h * w, a.depth_inv = depth_inv;
  hipLaunchKernelGGL(frame_setup_kernel, dim3(cdiv(h * w, 256), B, cdiv(D, 8) + 1), dim3(256), 0, as_stream(stream), a);
  BMV_LAUNCH_END("bmv_frame_setup");
}

int bmv_depth_values_cascade(const float* depth, const float* std_, const float* near_far, int B, int h0, int w0,
                             int h, int w, int D, float* depth_values, float* near_far_out, bmv_stream_t stream) {
  BMV_REQUIRE(depth && std_ && near_far && depth_values && near_far_out, "bmv_depth_values_cascade: null pointer");
  BMV_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0 && h0 > 0 && w0 > 0, "bmv_depth_values_cascade: bad shape");
  hipLaunchKernelGGL(depth_values_cascade_kernel, dim3(cdiv(h * w, 256), B), dim3(256), 0, as_stream(stream), depth,
                     std_, near_far, h0, w0, h, w, D, depth_values, near_far_out);
  BMV_LAUNCH_END("bmv_depth_values_cascade");
}

int bmv_homo_warp_fwd(const float* src_feat, const float* proj, const float* depth_values, int B, int C, int Hs,
                      int Ws, int D, int h, int w, float* warped, float* grid, bmv_stream_t stream) {
  BMV_REQUIRE(src_feat && proj && depth_values && warped, "bmv_homo_warp_fwd: null pointer");
  BMV_REQUIRE(B > 0 && C > 0 && Hs > 1 && Ws > 1 && D > 0 && h > 0 && w > 0, "bmv_homo_warp_fwd: bad shape");
  size_t nvox = (size_t)D * h * w;
  hipLaunchKernelGGL(homo_warp_kernel, dim3(cdiv(nvox, 256), B), dim3(256), 0, as_stream(stream), src_feat, proj,
                     depth_values, C, Hs, Ws, D, h, w, warped, grid);
  BMV_LAUNCH_END("bmv_homo_warp_fwd");
}

int bmv_sweep_win_launch(const float* feats, const float* proj, const float* dv, int B, int S, int C, int Hs, int Ws,
                         int D, int h, int w, float* out, const int* view_ids, int n_all, int variant,
                         hipStream_t stream);

int bmv_sweep_variance_views_fwd(const float* feats_all, const int* view_ids, int n_all, const float* proj,
                                 const float* depth_values, int B, int S, int C, int Hs, int Ws, int D, int h, int w,
                                 float* variance, bmv_stream_t stream) {
  BMV_REQUIRE(feats_all && view_ids && proj && depth_values && variance, "bmv_sweep_variance_views_fwd: null pointer");
  BMV_REQUIRE(B > 0 && S > 0 && n_all >= S && C > 0 && Hs > 1 && Ws > 1 && D > 0 && h > 0 && w > 0,
              "bmv_sweep_variance_views_fwd: bad shape");
  const int rc = bmv_sweep_win_launch(feats_all, proj, depth_values, B, S, C, Hs, Ws, D, h, w, variance, view_ids, n_all, -1,
                                      as_stream(stream));
  if (rc == BMV_ERR_UNSUPPORTED)
    set_error("bmv_sweep_variance_views_fwd: needs channel-last features with C in {16, 32} and 2..4 views (C=%d, S=%d)", C,
              S);
  return rc;
}

int bmv_sweep_variance_fwd(const float* feats, const float* proj, const float* depth_values, int B, int S, int C,
                           int Hs, int Ws, int D, int h, int w, float* variance, int feat_layout, int algo,
                           bmv_stream_t stream) {
  BMV_REQUIRE(feats && proj && depth_values && variance, "bmv_sweep_variance_fwd: null pointer");
  BMV_REQUIRE(B > 0 && S > 0 && C > 0 && Hs > 1 && Ws > 1 && D > 0 && h > 0 && w > 0,
              "bmv_sweep_variance_fwd: bad shape");
  BMV_REQUIRE(feat_layout == 0 || feat_layout == 1, "bmv_sweep_variance_fwd: feat_layout=%d", feat_layout);
  BMV_REQUIRE(algo == 0 || (algo == 1 && feat_layout == 0) || ((algo == 4 || (algo >= 40 && algo < 100)) && feat_layout == 1),
              "bmv_sweep_variance_fwd: algo=%d with feat_layout=%d", algo, feat_layout);
  size_t nvox = (size_t)D * h * w;
  if (feat_layout == 1) {
    // channel-last: LDS-staged exact windows (sweep_win.hip); 40 + i = its tuning variant i
    const int rc = bmv_sweep_win_launch(feats, proj, depth_values, B, S, C, Hs, Ws, D, h, w, variance, nullptr, 0,
                                        algo >= 40 ? algo - 40 : -1, as_stream(stream));
    if (rc == BMV_ERR_UNSUPPORTED)
      set_error("bmv_sweep_variance_fwd: the channel-last sweep needs C in {16, 32}, 2..4 views and a known variant "
                "(C=%d, S=%d, algo=%d); use feat_layout 0 or the quad-planar entry point", C, S, algo);
    return rc;
  }
  dim3 block(256);
#define LAUNCH(CB)                                                                                              \
  hipLaunchKernelGGL(sweep_direct_kernel<CB>, dim3(cdiv(nvox, 256), C / CB, B), block, 0, as_stream(stream), feats, \
                     proj, depth_values, S, C, Hs, Ws, D, h, w, variance)
  if (C % 16 == 0)
    LAUNCH(16);
  else if (C % 8 == 0)
    LAUNCH(8);
  else if (C % 4 == 0)
    LAUNCH(4);
  else
    LAUNCH(1);
#undef LAUNCH
  BMV_LAUNCH_END("bmv_sweep_variance_fwd");
}

int bmv_depth_regress_fwd(const float* depth_prob, const float* depth_values, int B, int D, int h, int w,
                          int depth_inv, float* depth, float* std_, bmv_stream_t stream) {
  BMV_REQUIRE(depth_prob && depth_values && depth && std_, "bmv_depth_regress_fwd: null pointer");
  BMV_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "bmv_depth_regress_fwd: bad shape");
  const DeferredPtr dd = deferred_for(depth), ds = deferred_for(std_);
  BMV_REQUIRE(!dd.table || !ds.table || dd.table == ds.table, "bmv_depth_regress_fwd: the deferred outputs must share one table");
  const void* const* table = dd.table ? dd.table : ds.table;
#define DR(DT)                                                                                                 \
  hipLaunchKernelGGL(depth_regress_kernel<DT>, dim3(cdiv(h * w, 64), B), dim3(64), 0, as_stream(stream), depth_prob, \
                     depth_values, D, h * w, depth_inv, depth, std_, table, dd.table ? dd.slot : -1, ds.table ? ds.slot : -1)
  if (D == 64 && h * w <= 65536 && !table) {   // coarse level: four lanes per pixel
    hipLaunchKernelGGL(depth_regress_quad_kernel<64>, dim3(cdiv(h * w, 16), B), dim3(64), 0, as_stream(stream),
                       depth_prob, depth_values, h * w, depth_inv, depth, std_);
    BMV_LAUNCH_END("bmv_depth_regress_fwd");
  }
  switch (D) {
    case 8: DR(8); break;
    case 16: DR(16); break;
    case 32: DR(32); break;
    case 64: DR(64); break;
    default: DR(0); break;
  }
#undef DR
  BMV_LAUNCH_END("bmv_depth_regress_fwd");
}

}  // extern "C"
