// LDS-staged plane sweep on channel-last features (a3+a4): the source window a
// tile of target voxels needs is copied into LDS once by LDS-DMA
// (global_load_lds_dwordx4, no VGPR round trip) and every bilinear tap becomes a
// ds_read_b128 instead of a trip through the vector L1.  Both the gather kernels
// and this one are bound by what a CU can pull from L2; staging lets every depth
// plane of a tile reuse ONE window per view, so the bytes pulled per CU drop by
// up to the number of planes sharing it.
//
// Work decomposition.  A workgroup (256 threads) owns a tile of 2 x 32 target
// pixels, 4 depth planes at a time (one per wave), PGI such plane groups in turn,
// and 16 channels.  One lane = one voxel with all 16 channels (sum and
// sum-of-squares in 32 VGPRs per plane group); a wave is two rows of 32 pixels,
// so each 16-lane ds_read_b128 group reads a run of neighbouring records.
// Per view:
//   window  <- bounding box of the tile's corner voxels over all 4*PGI planes
//              (wave-level min/max), width rounded to 16 records, clipped to
//              48 x 12 records (36.8 KB);
//   stage   <- one 1 KB DMA instruction per 16-record row segment; row / segment
//              arithmetic is wave-uniform (SALU), lanes only add a fixed offset;
//   taps    <- 4 x 4 ds_read_b128 per lane and plane group.
// Correctness never depends on the window: a lane whose taps fall outside the
// staged box (curved depth, long epipolar slide) reads them from global memory
// (wave-uniform branch on __any).
//
// Bank-conflict-free layout.  Records are 64 B, so 16 lanes reading the same
// 16-byte channel slice of 16 neighbouring records would hit only 4 of the 16
// four-bank groups.  Slice q of record r is therefore stored at slot
// q ^ ((r >> 2) & 3); LDS-DMA writes linearly, so the permutation is applied to
// the per-lane SOURCE address (linear destination + inverse-swizzled source +
// swizzled read).
#include "bmv_common.hpp"

namespace bmv {

constexpr int WWIN = 48, HWIN = 12, NREC = WWIN * HWIN;

__device__ __forceinline__ void project_pixel_l(const float* __restrict__ P, float x, float y, float inv_depth,
                                                float inv_half_w, float inv_half_h, float wm1, float hm1, float& ix,
                                                float& iy) {
  float px = P[0] * x + P[1] * y + P[2] + P[3] * inv_depth;
  float py = P[4] * x + P[5] * y + P[6] + P[7] * inv_depth;
  float pz = P[8] * x + P[9] * y + P[10] + P[11] * inv_depth;
  float iz = __builtin_amdgcn_rcpf(fmaxf(pz, 1e-6f));
  float gx = (px * iz) * inv_half_w - 1.f;
  float gy = (py * iz) * inv_half_h - 1.f;
  ix = ((gx + 1.f) * 0.5f) * wm1;
  iy = ((gy + 1.f) * 0.5f) * hm1;
}

__device__ __forceinline__ void accum4(float4 v, float4& a, float4& a2) {
  a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
  a2.x += v.x * v.x, a2.y += v.y * v.y, a2.z += v.z * v.z, a2.w += v.w * v.w;
}

__device__ __forceinline__ float4 blend4(float4 a, float4 b, float4 c, float4 e, float w00, float w01, float w10,
                                         float w11) {
  return make_float4(a.x * w00 + b.x * w01 + c.x * w10 + e.x * w11, a.y * w00 + b.y * w01 + c.y * w10 + e.y * w11,
                     a.z * w00 + b.z * w01 + c.z * w10 + e.z * w11, a.w * w00 + b.w * w01 + c.w * w10 + e.w * w11);
}

template <int PGI, int S>
__global__ void __launch_bounds__(256) sweep_lds_kernel(const float* __restrict__ feats,
                                                            const float* __restrict__ proj,
                                                            const float* __restrict__ dv, int C, int Hs, int Ws, int D,
                                                            int h, int w, float* __restrict__ out, int tiles_x,
                                                            int tiles_y, int rows_per_band, int plane_groups) {
  constexpr int TY = 2, TX = 32, NPW = 4;
  constexpr int NCORNER = 4 * NPW * PGI;  // lanes that evaluate a corner voxel
  static_assert(NCORNER <= 64 && (NCORNER & (NCORNER - 1)) == 0, "corner lanes");
  __shared__ __attribute__((aligned(16))) float lds[NREC * 16];

  // ---- block -> (band, channel group, plane super-group, tile); blockIdx % 8 shares an XCD / L2
  const int b = blockIdx.y;
  const int band = blockIdx.x & 7;
  int k = blockIdx.x >> 3;
  const int tx = k % tiles_x;
  k /= tiles_x;
  const int tyb = k % rows_per_band;
  k /= rows_per_band;
  const int pg = k % plane_groups;
  const int cg = k / plane_groups;  // 16-channel group
  const int ty = band * rows_per_band + tyb;
  if (ty >= tiles_y) return;  // whole workgroup, before any barrier
  const int c0 = cg * 16;
  const int d_base = pg * NPW * PGI;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = tx * TX + (lane & 31), y = ty * TY + (lane >> 5);
  const int xc = min(x, w - 1), yc = min(y, h - 1);
  const size_t hw = (size_t)h * w;
  const float* dvb = dv + (size_t)b * D * hw;
  float inv_depth[PGI];
#pragma unroll
  for (int g = 0; g < PGI; ++g)
    inv_depth[g] = __builtin_amdgcn_rcpf(dvb[(size_t)min(d_base + g * NPW + wave, D - 1) * hw + (size_t)yc * w + xc]);
  const float wm1 = (float)(Ws - 1), hm1 = (float)(Hs - 1);
  const float inv_half_w = 2.f / wm1, inv_half_h = 2.f / hm1;

  // corner voxels of the tile on every plane this workgroup touches (same values in every wave)
  const int cl = lane & (NCORNER - 1);
  const int ccx = min(tx * TX + ((cl & 1) ? TX - 1 : 0), w - 1);
  const int ccy = min(ty * TY + ((cl & 2) ? TY - 1 : 0), h - 1);
  const int ccd = min(d_base + (cl >> 2), D - 1);
  const float c_inv_depth = __builtin_amdgcn_rcpf(dvb[(size_t)ccd * hw + (size_t)ccy * w + ccx]);

  float4 acc[PGI][4], acc2[PGI][4];
#pragma unroll
  for (int g = 0; g < PGI; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[g][q] = acc2[g][q] = make_float4(0.f, 0.f, 0.f, 0.f);

  const float4* L4 = reinterpret_cast<const float4*>(lds);
  const int lrec = lane >> 2, lqs = lane & 3;  // this lane's record / slot inside a 16-record DMA piece
#pragma unroll 1
  for (int s = 0; s < S; ++s) {
    const float* P = proj + ((size_t)b * S + s) * 12;
    const float* fv = feats + ((size_t)b * S + s) * Hs * Ws * C + c0;
    // ---- window = bounding box of the corner voxels (+1 px margin), clipped to the LDS buffer
    int wx0, wy0, wcols, wrows;
    {
      float cix, ciy;
      project_pixel_l(P, (float)ccx, (float)ccy, c_inv_depth, inv_half_w, inv_half_h, wm1, hm1, cix, ciy);
      float lox = floorf(fminf(fmaxf(cix, -1.f), (float)Ws)), loy = floorf(fminf(fmaxf(ciy, -1.f), (float)Hs));
      float hix = lox, hiy = loy;
      if (lane >= NCORNER) lox = loy = 1e9f, hix = hiy = -1e9f;
#pragma unroll
      for (int m = 1; m < NCORNER; m <<= 1) {
        lox = fminf(lox, __shfl_xor(lox, m, 64)), loy = fminf(loy, __shfl_xor(loy, m, 64));
        hix = fmaxf(hix, __shfl_xor(hix, m, 64)), hiy = fmaxf(hiy, __shfl_xor(hiy, m, 64));
      }
      int ilox = __builtin_amdgcn_readfirstlane((int)lox) - 1, iloy = __builtin_amdgcn_readfirstlane((int)loy) - 1;
      int ihix = __builtin_amdgcn_readfirstlane((int)hix) + 2, ihiy = __builtin_amdgcn_readfirstlane((int)hiy) + 2;
      wx0 = max(0, min(ilox, Ws - 1));
      wy0 = max(0, min(iloy, Hs - 1));
      wcols = max(16, min(((ihix - wx0 + 1) + 15) & ~15, WWIN));  // whole 16-record DMA pieces
      wrows = max(1, min(min(ihiy - wy0 + 1, HWIN), Hs - wy0));
    }
    const int segs = wcols >> 4;
    const int npieces = wrows * segs;
    __syncthreads();  // every wave is done reading the previous view's window
    // ---- stage the window: piece p = (row, 16-record segment); everything but the lane offset is scalar
    for (int piece = wave; piece < npieces; piece += 4) {
      int row = piece / segs, seg = piece - row * segs;
      int rec = row * wcols + seg * 16 + lrec;              // record index inside the window
      int sx = min(wx0 + seg * 16 + lrec, Ws - 1);
      int q = lqs ^ ((rec >> 2) & 3);
      const float* src = fv + ((size_t)(wy0 + row) * Ws + sx) * C + q * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + (row * wcols + seg * 16) * 16), 16,
                                       0, 0);
    }
#pragma unroll
    for (int g = 0; g < PGI; ++g) {
      // ---- this lane's taps for plane group g (the first group overlaps the DMA latency)
      float ix, iy;
      project_pixel_l(P, (float)xc, (float)yc, inv_depth[g], inv_half_w, inv_half_h, wm1, hm1, ix, iy);
      float fx = floorf(ix), fy = floorf(iy);
      int x0 = (int)fminf(fmaxf(fx, -2.f), (float)Ws), y0 = (int)fminf(fmaxf(fy, -2.f), (float)Hs);
      int x1 = x0 + 1, y1 = y0 + 1;
      float ex = (fx + 1.f) - ix, ey = (fy + 1.f) - iy, ax = ix - fx, ay = iy - fy;
      bool vx0 = (x0 >= 0) & (x0 <= Ws - 1), vx1 = (x1 >= 0) & (x1 <= Ws - 1);
      bool vy0 = (y0 >= 0) & (y0 <= Hs - 1), vy1 = (y1 >= 0) & (y1 <= Hs - 1);
      float w00 = (vx0 & vy0) ? ex * ey : 0.f, w01 = (vx1 & vy0) ? ax * ey : 0.f;
      float w10 = (vx0 & vy1) ? ex * ay : 0.f, w11 = (vx1 & vy1) ? ax * ay : 0.f;
      // window-relative coordinates; axes that carry no valid tap are parked on the origin
      int rx0 = vx0 ? x0 - wx0 : 0, rx1 = vx1 ? x1 - wx0 : 0;
      int ry0 = vy0 ? y0 - wy0 : 0, ry1 = vy1 ? y1 - wy0 : 0;
      bool inwin = ((unsigned)rx0 < (unsigned)wcols) & ((unsigned)rx1 < (unsigned)wcols) &
                   ((unsigned)ry0 < (unsigned)wrows) & ((unsigned)ry1 < (unsigned)wrows);
      bool slow = (vx0 | vx1) & (vy0 | vy1) & !inwin;
      if (g == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // window landed
      }
      if (__any(slow)) {
        // rare: some lane's taps lie outside the staged box -> straight from global memory
        int gx0 = vx0 ? x0 : 0, gx1 = vx1 ? x1 : 0, gy0 = vy0 ? y0 : 0, gy1 = vy1 ? y1 : 0;
        const float4* g00 = reinterpret_cast<const float4*>(fv + ((size_t)gy0 * Ws + gx0) * C);
        const float4* g01 = reinterpret_cast<const float4*>(fv + ((size_t)gy0 * Ws + gx1) * C);
        const float4* g10 = reinterpret_cast<const float4*>(fv + ((size_t)gy1 * Ws + gx0) * C);
        const float4* g11 = reinterpret_cast<const float4*>(fv + ((size_t)gy1 * Ws + gx1) * C);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          accum4(blend4(g00[q], g01[q], g10[q], g11[q], w00, w01, w10, w11), acc[g][q], acc2[g][q]);
      } else {
        int r00 = ry0 * wcols + rx0, r01 = ry0 * wcols + rx1, r10 = ry1 * wcols + rx0, r11 = ry1 * wcols + rx1;
        int s00 = (r00 >> 2) & 3, s01 = (r01 >> 2) & 3, s10 = (r10 >> 2) & 3, s11 = (r11 >> 2) & 3;
        r00 = r00 * 4 + s00, r01 = r01 * 4 + s01, r10 = r10 * 4 + s10, r11 = r11 * 4 + s11;  // slot of slice 0
#pragma unroll
        for (int q = 0; q < 4; ++q)  // (r*4 + (q ^ s)) == ((r*4 + s) ^ q): the low two bits hold s
          accum4(blend4(L4[r00 ^ q], L4[r01 ^ q], L4[r10 ^ q], L4[r11 ^ q], w00, w01, w10, w11), acc[g][q], acc2[g][q]);
      }
    }
  }
  const float inv_s = 1.f / (float)S;
  const size_t cstride = (size_t)D * hw;
#pragma unroll
  for (int g = 0; g < PGI; ++g) {
    const int d = d_base + g * NPW + wave;
    if (!(x < w && y < h && d < D)) continue;
    float* o = out + (((size_t)b * C + c0) * D + d) * hw + (size_t)y * w + x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float m;
      m = acc[g][q].x * inv_s, o[(q * 4 + 0) * cstride] = acc2[g][q].x * inv_s - m * m;
      m = acc[g][q].y * inv_s, o[(q * 4 + 1) * cstride] = acc2[g][q].y * inv_s - m * m;
      m = acc[g][q].z * inv_s, o[(q * 4 + 2) * cstride] = acc2[g][q].z * inv_s - m * m;
      m = acc[g][q].w * inv_s, o[(q * 4 + 3) * cstride] = acc2[g][q].w * inv_s - m * m;
    }
  }
}

}  // namespace bmv

using namespace bmv;

// shape 0: source ~ volume resolution (cascade level 1): 8 planes (2 groups of 4) share one window per view.
// Other scales (level 0: source at 2x the volume resolution, 32-pixel rows need 66 source columns) stay on
// the direct-gather kernel: UNSUPPORTED here.
extern "C" int bmv_sweep_lds_launch(const float* feats, const float* proj, const float* dv, int B, int S, int C,
                                    int Hs, int Ws, int D, int h, int w, float* out, int shape, hipStream_t stream) {
  if (shape != 0 || S != 3 || C % 16 != 0 || Hs < 2 || Ws < 16) return BMV_ERR_UNSUPPORTED;
  const int TY = 2, TX = 32, NPW = 4, PGI = shape == 0 ? 2 : 1;
  int tiles_x = (w + TX - 1) / TX, tiles_y = (h + TY - 1) / TY;
  int rows_per_band = (tiles_y + 7) / 8;
  int plane_groups = (D + NPW * PGI - 1) / (NPW * PGI);
  dim3 grid(8u * (unsigned)(tiles_x * rows_per_band * plane_groups * (C / 16)), B), block(256);
  if (shape == 0)
    hipLaunchKernelGGL((sweep_lds_kernel<2, 3>), grid, block, 0, stream, feats, proj, dv, C, Hs, Ws, D, h, w, out,
                       tiles_x, tiles_y, rows_per_band, plane_groups);
  else
    hipLaunchKernelGGL((sweep_lds_kernel<1, 3>), grid, block, 0, stream, feats, proj, dv, C, Hs, Ws, D, h, w, out,
                       tiles_x, tiles_y, rows_per_band, plane_groups);
  BMV_LAUNCH_END("bmv_sweep_variance_fwd(lds)");
}
