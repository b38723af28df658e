// Plane sweep, split-geometry variant (a3+a4, channel-last features, C in {16, 32}).
//
// sweep_tiled.hip computes the per-view geometry of a voxel (projection, floor, bounds, bilinear weights, tap
// offsets) in the 4-lanes-per-voxel layout its gathers want: every geometry instruction runs for a quarter of the
// voxels a wave could cover, and its ablation shows that skeleton alone is half the kernel.  Here a wave owns a
// 4-row x 16-column patch of one plane and works in two phases:
//   1. lane = voxel (64 voxels): project into the S views, reduce every view to (first tap offset | validity bits,
//      fractional x, fractional y) and park the three numbers per view in LDS;
//   2. four passes, one patch row each, in the gather layout (lane = voxel x 16-byte channel slice): a lane reads
//      its voxel's three numbers (the quad reads one address: broadcast), rebuilds 4 weights / 4 offsets with a
//      dozen instructions and does the same buffer_load_dwordx4 gathers, blend and variance as sweep_tiled.hip.
// Same arithmetic as sweep_tiled.hip for every tap that carries weight; a tap outside the image has weight 0 and
// is parked on its in-image neighbour (any in-bounds address does).
#include "bmv_common.hpp"

namespace bmv {

namespace {

__device__ __forceinline__ float4 fma4s(float w, float4 a, float4 acc) {
  acc.x += w * a.x, acc.y += w * a.y, acc.z += w * a.z, acc.w += w * a.w;
  return acc;
}

}  // namespace

template <int QPL, int S>
__global__ void __launch_bounds__(256) sweep_split_kernel(const float* __restrict__ feats,
                                                           const float* __restrict__ proj,
                                                           const float* __restrict__ dv, int Hs, int Ws, int D, int h,
                                                           int w, float* __restrict__ out, int rows_per_band,
                                                           int groups_per_row, int blocks_per_pg, int tall,
                                                           const int* __restrict__ view_ids, int n_all) {
  constexpr int C = 16 * QPL;
  constexpr unsigned REC = C * 4;                 // bytes per source pixel record
  __shared__ float geo[4][S][3][64];               // [wave][view][o|ax|ay][voxel]
  const int b = blockIdx.y;
  const int band = blockIdx.x & 7;
  const int kk = blockIdx.x >> 3;
  const int pg = kk / blocks_per_pg;
  const int t = kk - pg * blocks_per_pg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // tall bands (cascade level 1): the 4 waves are 4 vertically adjacent patches of ONE plane (16 x 16 voxels), whose
  // source footprints overlap (HBM fetch 36 -> 30 MB at the same speed);  short bands (level 0: 8 rows): the waves
  // take 4 planes of one patch (measured 26.8 us vs 37.5 us for the one-plane tiling there)
  const int d = tall ? pg : pg * 4 + wave;
  const int y0 = band * rows_per_band + (t / groups_per_row) * (tall ? 16 : 4) + (tall ? wave * 4 : 0);
  const int x0 = (t % groups_per_row) * 16;
  const int rows_left = min(band * rows_per_band + rows_per_band, h) - y0;   // rows of this patch inside band / volume
  if (d >= D || rows_left <= 0) return;            // wave-uniform; no block barrier is used below
  const size_t hw = (size_t)h * w;
  const float wm1 = (float)(Ws - 1), hm1 = (float)(Hs - 1);
  const float inv_half_w = 2.f / wm1, inv_half_h = 2.f / hm1;

  {  // ---- phase 1: lane = voxel (row = lane >> 4, column = lane & 15)
    const int vx = min(x0 + (lane & 15), w - 1), vy = min(y0 + (lane >> 4), h - 1);
    const float inv_depth = __builtin_amdgcn_rcpf(dv[((size_t)b * D + d) * hw + (size_t)vy * w + vx]);
    const float fx = (float)vx, fy = (float)vy;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float* P = proj + ((size_t)b * S + s) * 12;
      const float px = P[0] * fx + P[1] * fy + P[2] + P[3] * inv_depth;
      const float py = P[4] * fx + P[5] * fy + P[6] + P[7] * inv_depth;
      const float pz = P[8] * fx + P[9] * fy + P[10] + P[11] * inv_depth;
      const float iz = __builtin_amdgcn_rcpf(fmaxf(pz, 1e-6f));
      const float gx = (px * iz) * inv_half_w - 1.f, gy = (py * iz) * inv_half_h - 1.f;
      const float ix = ((gx + 1.f) * 0.5f) * wm1, iy = ((gy + 1.f) * 0.5f) * hm1;
      // taps_zeros (bmv_common.hpp) reduced to 3 numbers
      const float flx = floorf(ix), fly = floorf(iy);
      const int tx0 = (int)fminf(fmaxf(flx, -2.f), (float)Ws), ty0 = (int)fminf(fmaxf(fly, -2.f), (float)Hs);
      const bool vx0 = (tx0 >= 0) & (tx0 <= Ws - 1), vx1 = (tx0 + 1 >= 0) & (tx0 + 1 <= Ws - 1);
      const bool vy0 = (ty0 >= 0) & (ty0 <= Hs - 1), vy1 = (ty0 + 1 >= 0) & (ty0 + 1 <= Hs - 1);
      const int cx = vx0 ? tx0 : (vx1 ? tx0 + 1 : 0), cy = vy0 ? ty0 : (vy1 ? ty0 + 1 : 0);
      const unsigned o = (unsigned)(cy * Ws + cx) | (vx0 ? 1u << 28 : 0u) | (vx1 ? 1u << 29 : 0u) |
                         (vy0 ? 1u << 30 : 0u) | (vy1 ? 1u << 31 : 0u);
      geo[wave][s][0][lane] = __uint_as_float(o);
      geo[wave][s][1][lane] = ix - flx;
      geo[wave][s][2][lane] = iy - fly;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- phase 2: lane = (voxel column, 16-byte channel slice)
  const int c = lane >> 2, q = lane & 3;
  const int x = x0 + c;
  const bool xvalid = x < w;
  const int item_views = view_ids ? n_all : S;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(feats + (size_t)b * item_views * Hs * Ws * C), 0, (int)((size_t)item_views * Hs * Ws * REC),
      0x00020000);
  const unsigned lane_off = (unsigned)q * (16u * QPL);
  unsigned vbase[S];
#pragma unroll
  for (int s = 0; s < S; ++s) vbase[s] = (unsigned)(view_ids ? view_ids[b * S + s] : s) * (unsigned)(Hs * Ws) * REC;
  const float inv_s = 1.f / (float)S;
  const size_t cstride = (size_t)D * hw;
  using i32x4 = __attribute__((ext_vector_type(4))) int;

#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r >= rows_left) continue;  // wave-uniform
    float4 acc[QPL], acc2[QPL];
#pragma unroll
    for (int p = 0; p < QPL; ++p) acc[p] = acc2[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int v = r * 16 + c;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const unsigned o = __float_as_uint(geo[wave][s][0][v]);
      const float ax = geo[wave][s][1][v], ay = geo[wave][s][2][v];
      const bool vx0 = o & (1u << 28), vx1 = o & (1u << 29), vy0 = o & (1u << 30), vy1 = o & (1u << 31);
      const float ex = 1.f - ax, ey = 1.f - ay;
      const float w00 = (vx0 & vy0) ? ex * ey : 0.f, w01 = (vx1 & vy0) ? ax * ey : 0.f;
      const float w10 = (vx0 & vy1) ? ex * ay : 0.f, w11 = (vx1 & vy1) ? ax * ay : 0.f;
      const unsigned o00 = vbase[s] + (o & 0x0fffffffu) * REC + lane_off;
      const unsigned dxo = (vx0 & vx1) ? REC : 0u, dyo = (vy0 & vy1) ? (unsigned)Ws * REC : 0u;
#pragma unroll
      for (int p = 0; p < QPL; ++p) {
        i32x4 ra = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o00 + 16u * p, 0, 0);
        i32x4 rb = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o00 + dxo + 16u * p, 0, 0);
        i32x4 rc = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o00 + dyo + 16u * p, 0, 0);
        i32x4 rd = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o00 + dxo + dyo + 16u * p, 0, 0);
        float4 a = *reinterpret_cast<float4*>(&ra), bq = *reinterpret_cast<float4*>(&rb);
        float4 cc = *reinterpret_cast<float4*>(&rc), e = *reinterpret_cast<float4*>(&rd);
        float4 val = make_float4(a.x * w00, a.y * w00, a.z * w00, a.w * w00);
        val = fma4s(w01, bq, val);
        val = fma4s(w10, cc, val);
        val = fma4s(w11, e, val);
        acc[p].x += val.x, acc[p].y += val.y, acc[p].z += val.z, acc[p].w += val.w;
        acc2[p].x += val.x * val.x, acc2[p].y += val.y * val.y, acc2[p].z += val.z * val.z, acc2[p].w += val.w * val.w;
      }
    }
    if (xvalid) {
      float* op = out + (((size_t)b * C + q * (4 * QPL)) * D + d) * hw + (size_t)(y0 + r) * w + x;
#pragma unroll
      for (int p = 0; p < QPL; ++p) {
        float m;
        m = acc[p].x * inv_s, op[(p * 4 + 0) * cstride] = acc2[p].x * inv_s - m * m;
        m = acc[p].y * inv_s, op[(p * 4 + 1) * cstride] = acc2[p].y * inv_s - m * m;
        m = acc[p].z * inv_s, op[(p * 4 + 2) * cstride] = acc2[p].z * inv_s - m * m;
        m = acc[p].w * inv_s, op[(p * 4 + 3) * cstride] = acc2[p].w * inv_s - m * m;
      }
    }
  }
}

}  // namespace bmv

using namespace bmv;

extern "C" int bmv_sweep_split_launch(const float* feats, const float* proj, const float* dv, int B, int S, int C, int Hs,
                                      int Ws, int D, int h, int w, float* out, const int* view_ids, int n_all,
                                      hipStream_t stream) {
  if ((C != 16 && C != 32) || S < 2 || S > 4) return BMV_ERR_UNSUPPORTED;
  if ((size_t)(view_ids ? n_all : S) * Hs * Ws * C * 4 >= ((size_t)1 << 31)) return BMV_ERR_UNSUPPORTED;
  if ((size_t)Hs * Ws >= ((size_t)1 << 28)) return BMV_ERR_UNSUPPORTED;  // 28-bit tap offsets
  int rows_per_band = (h + 7) / 8;
  int groups_per_row = (w + 15) / 16;
  const int tall = rows_per_band >= 16;
  int blocks_per_pg = ((rows_per_band + (tall ? 15 : 3)) / (tall ? 16 : 4)) * groups_per_row;
  dim3 grid(8u * (unsigned)((tall ? D : (D + 3) / 4) * blocks_per_pg), B), block(256);
#define SW(QPL, SV)                                                                                               \
  hipLaunchKernelGGL((sweep_split_kernel<QPL, SV>), grid, block, 0, stream, feats, proj, dv, Hs, Ws, D, h, w, out, \
                     rows_per_band, groups_per_row, blocks_per_pg, tall, view_ids, n_all)
  if (C == 16) {
    if (S == 2) SW(1, 2); else if (S == 3) SW(1, 3); else SW(1, 4);
  } else {
    if (S == 2) SW(2, 2); else if (S == 3) SW(2, 3); else SW(2, 4);
  }
#undef SW
  BMV_LAUNCH_END("bmv_sweep_variance_fwd(split)");
}
