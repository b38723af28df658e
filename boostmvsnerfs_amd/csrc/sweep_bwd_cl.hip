// a3+a4 plane-sweep backward on CHANNEL-LAST tensors (round 3): d_variance (B,C,D,h,w) -> d_feats (B,S,Hs,Ws,C) and,
// when asked, d_depth_values (B,D,h,w).      lib/networks/enerf/utils.py:57-95, 324-351 under torch.autograd
//
// Why: fp32 atomic adds are the floor of this backward (3 views x 4 taps x C channels per voxel: 135 M per level of a
// 512x640 frame) and their rate depends on the access pattern only (scripts/ubench/atomic_rate.hip, MI355X):
//     every lane its own texel of a planar gradient (64 lines per wave instruction)      116 G lane-atomics / s
//     LDS ds_add_f32 into a window + flush (what backward.hip's sweep_bwd_kernel does)   203 G / s
//     >= 16 consecutive channels of one tap per lane group (1-4 lines per instruction)   330 G / s
// so the gradient is accumulated in the channel-last layout with the CHANNEL on the lane.
//
// A wave owns a run of 64 pixels of one plane.  Phase 1, lane = voxel: tap corner, the 4 bilinear weights (0 outside the map,
// exactly taps_zeros of bmv_common.hpp) and, for the depth gradient, the tap fractions, per view; the wave's tile of
// d_variance is staged [channel][voxel] in LDS (coalesced reads of the planar tensor, read back transposed).  Phase 2,
// lane = (tap slot, channel): for every voxel the tap data is broadcast with v_readlane, a gather of a tap is one
// contiguous record per lane group, the per-view warped value is a cross-group DPP / shuffle sum, and the gradient goes
// out as 64 / C taps per atomic instruction.  Depth gradient: per voxel a wave sum of f_t * (d w_t / d ix, iy) over taps
// and channels (row DPP sums + 4 readlanes), handed back to the voxel's own lane; one plain store per voxel.
#include "sweep_util.hpp"

namespace bmv {

using sweep_util::rl;

__device__ __forceinline__ int rli(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// sums of a and b over the whole wave, in every lane (4 row-DPP steps + 4 readlanes each)
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
  asm volatile(
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(a), "+v"(b));
  a = (rl(a, 0) + rl(a, 16)) + (rl(a, 32) + rl(a, 48));
  b = (rl(b, 0) + rl(b, 16)) + (rl(b, 32) + rl(b, 48));
}

template <int C>
__global__ void __launch_bounds__(256) sweep_bwd_cl_kernel(const float* __restrict__ feats, const float* __restrict__ proj,
                                                            const float* __restrict__ dv, const float* __restrict__ g_var,
                                                            int Hs, int Ws, int D, int h, int w,
                                                            float* __restrict__ d_feats, float* __restrict__ d_dv) {
  constexpr int S = 3, TPI = 64 / C, ROUNDS = 4 / TPI;     // taps per instruction: 2 (C = 32) or 4 (C = 16)
  static_assert(C == 16 || C == 32, "channel-on-lane layout");
  __shared__ float gt[4][C][65];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const long nvox = (long)D * h * w;
  // a wave owns a run of 64 pixels of one plane
  const int hw = h * w, runs = (hw + 63) / 64;
  const long wid = (long)blockIdx.x * 4 + wave;
  const int run = (int)(wid % runs), d = (int)(wid / runs);
  if (d >= D) return;                                       // (no workgroup barrier below: waves are independent)
  const long base = (long)d * hw + (long)run * 64;
  const int nv = min(64, hw - run * 64);
  // ---------------------------------------------------------------- phase 1: lane = voxel
  const long i = base + lane;
  const bool valid = lane < nv;
  const long ic = valid ? i : base;
  const int x = (int)(ic % w), y = (int)((ic / w) % h);
  const float depth = dv[(long)b * nvox + ic];
  int x0[S], y0[S];
  float wgt[S][4], ex[S], ey[S], ax[S], ay[S], pxs[S], pys[S], pzs[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float* P = proj + ((long)b * S + s) * 12;
    const float px = P[0] * x + P[1] * y + P[2] + P[3] / depth;
    const float py = P[4] * x + P[5] * y + P[6] + P[7] / depth;
    const float pz = P[8] * x + P[9] * y + P[10] + P[11] / depth;
    pxs[s] = px, pys[s] = py, pzs[s] = pz;
    const float z = fmaxf(pz, 1e-6f);
    const float ix = unnorm((px / z) / ((float)(Ws - 1) * 0.5f) - 1.f, Ws), iy = unnorm((py / z) / ((float)(Hs - 1) * 0.5f) - 1.f, Hs);
    const Taps2 t = taps_zeros(ix, iy, Ws, Hs);
    const float fx = floorf(ix), fy = floorf(iy);
    ex[s] = (fx + 1.f) - ix, ey[s] = (fy + 1.f) - iy, ax[s] = ix - fx, ay[s] = iy - fy;
    x0[s] = (int)fminf(fmaxf(fx, -2.f), (float)Ws), y0[s] = (int)fminf(fmaxf(fy, -2.f), (float)Hs);
    wgt[s][0] = valid ? t.w00 : 0.f, wgt[s][1] = valid ? t.w01 : 0.f, wgt[s][2] = valid ? t.w10 : 0.f, wgt[s][3] = valid ? t.w11 : 0.f;
  }
#pragma unroll 8
  for (int c = 0; c < C; ++c) gt[wave][c][lane] = valid ? g_var[((long)b * C + c) * nvox + i] : 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---------------------------------------------------------------- phase 2: lane = (tap slot, channel)
  const int ch = lane % C, tp = lane / C;
  float gix[S], giy[S];
#pragma unroll
  for (int s = 0; s < S; ++s) gix[s] = giy[s] = 0.f;
  const bool want_dv = d_dv != nullptr;
  // The gathers of voxel k + 1 are ISSUED BEFORE the atomics of voxel k: loads and return-less atomics share the wave's
  // in-order vmcnt queue on gfx9, so a load issued behind an atomic cannot be waited for without waiting for the
  // atomic's acknowledgement too.
  struct Taps {
    float f[S][ROUNDS], wt[S][ROUNDS];
    int off[S][ROUNDS];
  };
  auto fetch = [&](int k, Taps& T) {
    const int v = k;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int bx0 = rli(x0[s], v), by0 = rli(y0[s], v);
      const float bw0 = rl(wgt[s][0], v), bw1 = rl(wgt[s][1], v), bw2 = rl(wgt[s][2], v), bw3 = rl(wgt[s][3], v);
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int t = r * TPI + tp;                               // tap: bit 0 = +x, bit 1 = +y
        const int tx = bx0 + (t & 1), ty = by0 + (t >> 1);
        const float wsel = t == 0 ? bw0 : t == 1 ? bw1 : t == 2 ? bw2 : bw3;
        const bool inb = tx >= 0 && tx < Ws && ty >= 0 && ty < Hs;
        const int o = ((((b * S + s) * Hs + (inb ? ty : 0)) * Ws) + (inb ? tx : 0)) * C + ch;
        T.f[s][r] = inb ? feats[o] : 0.f;
        T.wt[s][r] = wsel, T.off[s][r] = o;
      }
    }
  };
  Taps nxt;
  fetch(0, nxt);
  for (int k = 0; k < nv; ++k) {
    const int v = k;
    const Taps cur = nxt;
    if (k + 1 < nv) fetch(k + 1, nxt);
    const float gc = gt[wave][ch][v];
    float wv[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) part += cur.wt[s][r] * cur.f[s][r];
      if constexpr (C == 32) {
        part += __shfl_xor(part, 32, 64);
      } else {
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
      }
      wv[s] = part;
    }
    const float mean = (wv[0] + wv[1] + wv[2]) / 3.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float gws = (2.f / 3.f) * gc * (wv[s] - mean);       // d var / d warped_s for this channel
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r)
        if (cur.wt[s][r] != 0.f) atomicAdd(d_feats + cur.off[s][r], cur.wt[s][r] * gws);
      if (want_dv) {
        // d warped / d (ix, iy) = sum over IN-BOUNDS taps of f_t * d w_t: w00 = ex ey, w01 = ax ey, w10 = ex ay, w11 = ax ay
        const float bex = rl(ex[s], v), bey = rl(ey[s], v), bax = rl(ax[s], v), bay = rl(ay[s], v);
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
          const int t = r * TPI + tp;
          const float cx = ((t & 1) ? 1.f : -1.f) * ((t >> 1) ? bay : bey);
          const float cy = ((t >> 1) ? 1.f : -1.f) * ((t & 1) ? bax : bex);
          sx += cur.f[s][r] * cx, sy += cur.f[s][r] * cy;
        }
        sx *= gws, sy *= gws;
        wave_sum2(sx, sy);
        if (lane == v) gix[s] = sx, giy[s] = sy;
      }
    }
  }
  // ---------------------------------------------------------------- phase 3: lane = voxel, depth gradient
  if (want_dv && valid) {
    float gdepth = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float* P = proj + ((long)b * S + s) * 12;
      const float z = fmaxf(pzs[s], 1e-6f);
      // ix = px / z (normalise / unnormalise cancel); p = A + T / depth
      const float gpx = gix[s] / z, gpy = giy[s] / z;
      const float gpz = pzs[s] > 1e-6f ? -(gix[s] * pxs[s] + giy[s] * pys[s]) / (z * z) : 0.f;
      gdepth += -(gpx * P[3] + gpy * P[7] + gpz * P[11]) / (depth * depth);
    }
    d_dv[(long)b * nvox + i] = gdepth;
  }
}

}  // namespace bmv

using namespace bmv;

extern "C" int bmv_sweep_variance_bwd_cl(const float* feats_cl, const float* proj, const float* depth_values,
                                         const float* d_variance, int B, int S, int C, int Hs, int Ws, int D, int h, int w,
                                         float* d_feats_cl, float* d_depth_values, bmv_stream_t stream) {
  BMV_REQUIRE(feats_cl && proj && depth_values && d_variance && d_feats_cl, "bmv_sweep_variance_bwd_cl: null pointer");
  BMV_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && D > 0 && h > 0 && w > 0, "bmv_sweep_variance_bwd_cl: bad shape");
  if (S != 3 || (C != 16 && C != 32)) {
    set_error("bmv_sweep_variance_bwd_cl: built for S=3 views and C in {16, 32} (got S=%d C=%d)", S, C);
    return BMV_ERR_UNSUPPORTED;
  }
  BMV_REQUIRE((long)B * S * Hs * Ws * C < (1L << 31), "bmv_sweep_variance_bwd_cl: source maps exceed 2^31 elements");
  const int runs = (h * w + 63) / 64;
  const dim3 grid(cdiv((long)runs * D, 4), B);
  hipStream_t st = as_stream(stream);
  if (C == 32)
    hipLaunchKernelGGL((sweep_bwd_cl_kernel<32>), grid, dim3(256), 0, st, feats_cl, proj, depth_values, d_variance, Hs, Ws, D,
                       h, w, d_feats_cl, d_depth_values);
  else
    hipLaunchKernelGGL((sweep_bwd_cl_kernel<16>), grid, dim3(256), 0, st, feats_cl, proj, depth_values, d_variance, Hs, Ws, D,
                       h, w, d_feats_cl, d_depth_values);
  BMV_LAUNCH_END("bmv_sweep_variance_bwd_cl");
}
