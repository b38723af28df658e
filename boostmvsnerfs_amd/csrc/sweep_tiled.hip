// Plane sweep on channel-last source features (a3+a4), the fast variants.
//
// Layout: feats (B,S,Hs,Ws,C), channels innermost.  A bilinear tap is then ONE
// contiguous C*4-byte record, fetched as 16-byte pieces by a small group of
// adjacent lanes, instead of C scattered dwords.
//
// Lane mapping: lane = (voxel x, channel slice q).  CQ = 4 lanes share one voxel,
// each owns 4*QPL consecutive channels (QPL 16-byte pieces per tap), so a wave
// covers 16 consecutive x of one (d, y) row.  For every view the 4 lanes of a
// voxel read 4 taps x (C*4) contiguous bytes; neighbouring voxels hit
// neighbouring records, so a wave-wide load touches a handful of cache lines.
// The S=3 views are innermost, sum and sum-of-squares live in registers and
// the variance is written once: no warped volume ever reaches memory.
//
// XCD-aware dispatch: blockIdx % 8 (blocks that share an XCD, hence an L2) owns
// one horizontal band of the target volume for ALL depth planes, so every XCD's
// 4 MB L2 only has to hold ~1/8 of the source features (+ the epipolar halo).
#include "bmv_common.hpp"

namespace bmv {

__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ src, int C, int HW,
                                                            float* __restrict__ dst) {
  // one thread per pixel: C coalesced plane reads, then C contiguous floats out (C % 4 == 0)
  int n = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW) return;
  const float* s = src + (size_t)n * C * HW + i;
  float4* d = reinterpret_cast<float4*>(dst + ((size_t)n * HW + i) * C);
  for (int c = 0; c < C; c += 4) {
    float4 v = {s[(size_t)c * HW], s[(size_t)(c + 1) * HW], s[(size_t)(c + 2) * HW], s[(size_t)(c + 3) * HW]};
    d[c >> 2] = v;
  }
}

__device__ __forceinline__ float4 fma4(float w, float4 a, float4 acc) {
  acc.x += w * a.x, acc.y += w * a.y, acc.z += w * a.z, acc.w += w * a.w;
  return acc;
}

// broadcast lane S of every quad (4 consecutive lanes) with a DPP move: no LDS, no latency to hide
template <int SRC>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), SRC * 0x55, 0xF, 0xF, true));
}

// Source pixel of target voxel (x, y, depth) in one view (utils.py:72-84).  Divisions are
// v_rcp_f32 + multiply (1 ulp; the coordinate moves by < 1e-4 px).
__device__ __forceinline__ void project_pixel(const float* __restrict__ P, float x, float y, float inv_depth,
                                              float inv_half_w, float inv_half_h, float wm1, float hm1, float& ix,
                                              float& iy) {
  float px = P[0] * x + P[1] * y + P[2] + P[3] * inv_depth;
  float py = P[4] * x + P[5] * y + P[6] + P[7] * inv_depth;
  float pz = P[8] * x + P[9] * y + P[10] + P[11] * inv_depth;
  float iz = __builtin_amdgcn_rcpf(fmaxf(pz, 1e-6f));
  float gx = (px * iz) * inv_half_w - 1.f;   // grid in [-1, 1] ...
  float gy = (py * iz) * inv_half_h - 1.f;
  ix = ((gx + 1.f) * 0.5f) * wm1;            // ... and back to pixels, as grid_sample does
  iy = ((gy + 1.f) * 0.5f) * hm1;
}

template <int QPL, int S>
__global__ void __launch_bounds__(256) sweep_nhwc_kernel(const float* __restrict__ feats,
                                                          const float* __restrict__ proj,
                                                          const float* __restrict__ dv, int Hs, int Ws, int D, int h,
                                                          int w, float* __restrict__ out, int rows_per_band,
                                                          int groups_per_row, int blocks_per_plane,
                                                          const int* __restrict__ view_ids, int n_all) {
  static_assert(S >= 1 && S <= 4, "one quad lane projects one view");
  constexpr int CQ = 4, VPW = 64 / CQ, C = 4 * CQ * QPL;
  const int b = blockIdx.y;
  const int band = blockIdx.x & 7;
  const int k = blockIdx.x >> 3;
  const int d = k / blocks_per_plane;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a workgroup is a 4-row x 16-column patch (one row per wave): the source rows two
  // neighbouring target rows tap overlap, and the waves of one CU share them through its L1
  const int t = k - d * blocks_per_plane;
  const int yb = (t / groups_per_row) * 4 + wave;
  const int y = band * rows_per_band + yb;
  if (yb >= rows_per_band || y >= h) return;  // wave-uniform
  const int x = (t % groups_per_row) * VPW + lane / CQ;
  const int q = lane % CQ;
  const bool valid = x < w;
  const int xc = valid ? x : w - 1;
  const size_t hw = (size_t)h * w;
  const float inv_depth = __builtin_amdgcn_rcpf(dv[((size_t)b * D + d) * hw + (size_t)y * w + xc]);
  const float wm1 = (float)(Ws - 1), hm1 = (float)(Hs - 1);
  const float inv_half_w = 2.f / wm1, inv_half_h = 2.f / hm1;

  // Lane q of each quad does ALL the per-view geometry of view q (projection, floor, bounds,
  // weights, byte offsets of the 4 taps); the quad then shares the 8 results by DPP, so the
  // geometry costs each lane one view instead of S.
  constexpr unsigned REC = C * 4;  // bytes per source pixel record
  unsigned my_o[4] = {0, 0, 0, 0};
  float my_w[4] = {0.f, 0.f, 0.f, 0.f};
  if (q < S) {
    float ix, iy;
    project_pixel(proj + ((size_t)b * S + q) * 12, (float)xc, (float)y, inv_depth, inv_half_w, inv_half_h, wm1, hm1,
                  ix, iy);
    Taps2 t = taps_zeros(ix, iy, Ws, Hs);
    // view q of this cost volume inside the feature tensor: q, or view_ids[b*S + q] of an (B, n_all, ...) tensor
    const unsigned vbase = (unsigned)(view_ids ? view_ids[b * S + q] : q) * (unsigned)(Hs * Ws) * REC;
    my_o[0] = vbase + (unsigned)t.o00 * REC, my_o[1] = vbase + (unsigned)t.o01 * REC;
    my_o[2] = vbase + (unsigned)t.o10 * REC, my_o[3] = vbase + (unsigned)t.o11 * REC;
    my_w[0] = t.w00, my_w[1] = t.w01, my_w[2] = t.w10, my_w[3] = t.w11;
  }

  float4 acc[QPL], acc2[QPL];
#pragma unroll
  for (int p = 0; p < QPL; ++p) acc[p] = acc2[p] = make_float4(0.f, 0.f, 0.f, 0.f);

  // buffer loads: one scalar resource for this batch item + 32-bit byte offsets per lane
  const int item_views = view_ids ? n_all : S;
  const size_t item_bytes = (size_t)item_views * Hs * Ws * REC;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(feats + (size_t)b * item_views * Hs * Ws * C), 0, (int)item_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)q * (16u * QPL);
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  auto one_view = [&](unsigned o0, unsigned o1, unsigned o2, unsigned o3, float w0, float w1, float w2, float w3) {
#pragma unroll
    for (int p = 0; p < QPL; ++p) {
      i32x4 ra = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o0 + lane_off + 16u * p, 0, 0);
      i32x4 rb = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o1 + lane_off + 16u * p, 0, 0);
      i32x4 rc = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o2 + lane_off + 16u * p, 0, 0);
      i32x4 rd = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o3 + lane_off + 16u * p, 0, 0);
      float4 a = *reinterpret_cast<float4*>(&ra), bq = *reinterpret_cast<float4*>(&rb);
      float4 c = *reinterpret_cast<float4*>(&rc), e = *reinterpret_cast<float4*>(&rd);
      float4 v = make_float4(a.x * w0, a.y * w0, a.z * w0, a.w * w0);
      v = fma4(w1, bq, v);
      v = fma4(w2, c, v);
      v = fma4(w3, e, v);
      acc[p].x += v.x, acc[p].y += v.y, acc[p].z += v.z, acc[p].w += v.w;
      acc2[p].x += v.x * v.x, acc2[p].y += v.y * v.y, acc2[p].z += v.z * v.z, acc2[p].w += v.w * v.w;
    }
  };
#define BMV_QB(SRC, v) quad_bcast<SRC>(v)
#define BMV_QBU(SRC, v) (unsigned)__builtin_amdgcn_mov_dpp((int)(v), SRC * 0x55, 0xF, 0xF, true)
#define BMV_VIEW(SRC)                                                                                       \
  one_view(BMV_QBU(SRC, my_o[0]), BMV_QBU(SRC, my_o[1]), BMV_QBU(SRC, my_o[2]), BMV_QBU(SRC, my_o[3]),       \
           BMV_QB(SRC, my_w[0]), BMV_QB(SRC, my_w[1]), BMV_QB(SRC, my_w[2]), BMV_QB(SRC, my_w[3]))
  BMV_VIEW(0);
  if constexpr (S > 1) BMV_VIEW(1);
  if constexpr (S > 2) BMV_VIEW(2);
  if constexpr (S > 3) BMV_VIEW(3);
#undef BMV_VIEW
#undef BMV_QB
#undef BMV_QBU

  if (!valid) return;
  const float inv_s = 1.f / (float)S;
  float* o = out + (((size_t)b * C + q * (4 * QPL)) * D + d) * hw + (size_t)y * w + x;
  const size_t cstride = (size_t)D * hw;
#pragma unroll
  for (int p = 0; p < QPL; ++p) {
    float m;
    m = acc[p].x * inv_s, o[(p * 4 + 0) * cstride] = acc2[p].x * inv_s - m * m;
    m = acc[p].y * inv_s, o[(p * 4 + 1) * cstride] = acc2[p].y * inv_s - m * m;
    m = acc[p].z * inv_s, o[(p * 4 + 2) * cstride] = acc2[p].z * inv_s - m * m;
    m = acc[p].w * inv_s, o[(p * 4 + 3) * cstride] = acc2[p].w * inv_s - m * m;
  }
}

}  // namespace bmv

using namespace bmv;

extern "C" {

int bmv_nchw_to_nhwc(const float* src, int n, int C, int H, int W, float* dst, bmv_stream_t stream) {
  BMV_REQUIRE(src && dst, "bmv_nchw_to_nhwc: null pointer");
  BMV_REQUIRE(n > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, "bmv_nchw_to_nhwc: bad shape (C must be a multiple of 4)");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv((long)H * W, 256), n), dim3(256), 0, as_stream(stream), src, C,
                     H * W, dst);
  BMV_LAUNCH_END("bmv_nchw_to_nhwc");
}

// Channel-last sweep.  Returns BMV_ERR_UNSUPPORTED for channel counts it has no kernel for.
int bmv_sweep_nhwc_launch(const float* feats, const float* proj, const float* dv, int B, int S, int C, int Hs, int Ws,
                          int D, int h, int w, float* out, const int* view_ids, int n_all, hipStream_t stream) {
  if ((C != 16 && C != 32) || S < 2 || S > 4) return BMV_ERR_UNSUPPORTED;
  if ((size_t)(view_ids ? n_all : S) * Hs * Ws * C * 4 >= ((size_t)1 << 31)) return BMV_ERR_UNSUPPORTED;  // 32-bit offsets
  int rows_per_band = (h + 7) / 8;
  int groups_per_row = (w + 15) / 16;
  int blocks_per_plane = ((rows_per_band + 3) / 4) * groups_per_row;
  dim3 grid(8u * (unsigned)(D * blocks_per_plane), B), block(256);
#define SW(QPL, SV)                                                                                            \
  hipLaunchKernelGGL((sweep_nhwc_kernel<QPL, SV>), grid, block, 0, stream, feats, proj, dv, Hs, Ws, D, h, w, out, \
                     rows_per_band, groups_per_row, blocks_per_plane, view_ids, n_all)
  if (C == 16) {
    if (S == 2) SW(1, 2); else if (S == 3) SW(1, 3); else SW(1, 4);
  } else {
    if (S == 2) SW(2, 2); else if (S == 3) SW(2, 3); else SW(2, 4);
  }
#undef SW
  BMV_LAUNCH_END("bmv_sweep_variance_fwd(nhwc)");
}

}  // extern "C"
