// Plane sweep with LDS-staged ZERO-PADDED source windows (a3+a4, channel-last features, C in {16, 32}).
// Reference: lib/networks/enerf/utils.py:57-95 (homo_warp), :324-351 (build_feature_volume).
//
// Round 3.  Same decomposition and occupancy as sweep_win.hip (round 2: a workgroup = TXW x TYH pixels x DP planes x
// 16 channels, lane = voxel, one window per view, many small independent workgroups per CU) -- the round-3 attempts at
// a persistent producer / consumer pipeline (sweep_ring.hip) lost to it: an LDS-DMA piece occupies its issuing wave
// for ~90-130 cycles, so the fill has to be spread over many waves, and many independent waves per SIMD hide it
// better than a rigid pipeline.  What changes is the instruction stream, 1160 vector instructions per 64 voxels there:
//   * windows are NOT clipped to the image: the fill requests texels outside out of range (the DMA writes zeros), so
//     the blend has no validity logic, no parked taps, no per-voxel window test -- grid_sample's zero padding is in the
//     data.  A view whose box cannot be bounded (a corner behind the camera, non-finite hypotheses) or does not fit the
//     LDS budget is gathered from global memory with the full tap logic by the whole workgroup.
//   * the 16-byte slices of a record are read in a per-lane order (slice q ^ key, key = (lane >> 2) & 3): conflict-free
//     ds_read_b128 groups on a plain linear copy, tap addresses = ONE shift-add per slice (the right neighbour and
//     the row below are instruction offsets / one add), and the 16 variances go back into channel order with 2 x 16
//     v_cndmask before the stores (sweep_win: 6 instructions per tap for the record-keyed XOR).
//   * v_cvt_flr_i32_f32 / v_fract_f32 for the tap split, 24-bit multiplies, projection rows in scalar registers with
//     the multiply-add chains ordered so that no instruction needs two scalar sources.
//   * plane-uniform hypotheses (cascade level 0: dv is (B,D)): the depth load, the DPP reduction and the LDS exchange
//     disappear; the range of a workgroup's planes comes from scalar loads.
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "bmv_common.hpp"
#include "sweep_util.hpp"

namespace bmv {

using namespace sweep_util;

struct ZpArgs {
  const float* feats;
  const float* proj;
  const float* dv;
  float* out;
  const int* view_ids;
  int n_all, C, Hs, Ws, D, h, w;
  int tiles_x, tyb, pgroups, chalves, cap;
  unsigned tiles_x_magic;   // floor(2^32 / tiles_x) + 1: n / tiles_x = umulhi(n, magic) for n < 2^16
  int dv_ps, dv_rs, dv_cs;  // strides of the hypotheses in elements: plane, row, column ((B,D) planes: rs = cs = 0)
  long long dv_bs;          // batch stride
  int flags;                // tuning: 1 no fill, 2 no blend, 4 no store
};

// the scalar part of a fill piece's offset is added on the vector side: with a non-zero soffset the LDS-DMA form of
// buffer_load did not deliver the right texels on gfx950 (results wrong with it, right without; the same offsets)
#define BMV_ZP_SOFF(so) 0

#ifndef BMV_ZP_TAPBUF
#define BMV_ZP_TAPBUF 2   // slices of 4 taps in flight in the blend
#endif
#ifndef BMV_ZP_PHOIST
#define BMV_ZP_PHOIST 0   // 1: all projection rows loaded into scalar registers at kernel entry
#endif

// PU: plane-uniform hypotheses (one depth per plane)
template <int TXW, int TYH, int DP, int S, int WPE, bool PU>
__global__ void __launch_bounds__(TXW* TYH* DP) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
sweep_zp_kernel(const ZpArgs a) {
  constexpr int NT = TXW * TYH * DP, NW = NT / 64;
  constexpr int kStoreAux = TXW >= 32 ? 2 : 0;   // non-temporal where a wave row is a whole 128-byte line (sweep_win.hip)
  static_assert(NT % 64 == 0 && NT <= 1024, "workgroup size");
  static_assert((TXW * TYH) % 32 == 0, "a half wave covers 32 voxels of ONE plane");
  static_assert(8 * S <= 64, "corner lanes");
  // one window of cap records (64 B each), then NW (min, max) slots of the depth-range exchange.  No static LDS: the
  // window sits at LDS address 0 and the tap reads need no base add
  extern __shared__ __attribute__((aligned(64))) char win[];
  float2* slots = reinterpret_cast<float2*>(win + (size_t)a.cap * 64);

  // grid = (8 bands x channel halves x plane groups, tile columns x tile rows of a band, batch); blockIdx.x % 8 = the
  // band = the XCD whose L2 holds that band's source rows
  const int b = blockIdx.z;
  const int band = blockIdx.x & 7;
  const int kx = blockIdx.x >> 3;
  const int chh = kx & (a.chalves - 1);   // chalves is 1 or 2
  const int pg = kx >> (a.chalves - 1);
  const int j = a.tiles_x_magic ? (int)__umulhi((unsigned)blockIdx.y, (unsigned)a.tiles_x_magic) : (int)blockIdx.y;
  const int tx = blockIdx.y - j * a.tiles_x;
  const int ty = band * a.tyb + j;
  if (ty * TYH >= a.h) return;  // whole workgroup, before any barrier
  const int C = a.C, Hs = a.Hs, Ws = a.Ws, D = a.D, h = a.h, w = a.w;
  const unsigned REC = (unsigned)C * 4u;
  const unsigned hoff = (unsigned)chh * 64u;   // byte offset of the channel half inside a source record
  const size_t hw = (size_t)h * w;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = tid % TXW, ly = (tid / TXW) % TYH, ld = tid / (TXW * TYH);
  const int x = tx * TXW + lx, y = ty * TYH + ly, d = pg * DP + ld;
  const bool inb = (x < w) & (y < h) & (d < D);
  const int xc = min(x, w - 1), yc = min(y, h - 1), dc = min(d, D - 1);
  const float fx = (float)xc, fy = (float)yc;

  // corner lanes (lane < 8 S): corner (lane & 7) of the tile box in (x, y, 1/depth), view lane >> 3; their projection
  // rows are fetched now, under the latency of the depth load
  float cP[12];
  {
    const float* Pc = a.proj + ((size_t)b * S + min(lane >> 3, S - 1)) * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) cP[k] = Pc[k];
  }
  const float* dvb = a.dv + (size_t)b * a.dv_bs;
  const float inv_depth =
      __builtin_amdgcn_rcpf(dvb[(unsigned)(__mul24(dc, a.dv_ps) + __mul24(yc, a.dv_rs) + __mul24(xc, a.dv_cs))]);

  // ---- 1. range of 1/depth over the workgroup
  float ilo, ihi;
  if (PU) {
    // the planes of this group, from scalar loads (hypotheses are monotonic in the plane index or not: min / max)
    ilo = INFINITY, ihi = -INFINITY;
    const int d0 = pg * DP;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const float v = __builtin_amdgcn_rcpf(dvb[min(d0 + k, D - 1) * a.dv_ps]);
      ilo = fminf(ilo, v), ihi = fmaxf(ihi, v);
    }
  } else {
    ilo = inv_depth, ihi = inv_depth;
    row_min_max16(ilo, ihi);
    const float l0 = rl(ilo, 0), l1 = rl(ilo, 16), l2 = rl(ilo, 32), l3 = rl(ilo, 48);
    const float h0 = rl(ihi, 0), h1 = rl(ihi, 16), h2 = rl(ihi, 32), h3 = rl(ihi, 48);
    ilo = fminf(fminf(l0, l1), fminf(l2, l3)), ihi = fmaxf(fmaxf(h0, h1), fmaxf(h2, h3));
    if (NW > 1) {
      if (lane == 0) slots[wave] = make_float2(ilo, ihi);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const float2 v = slots[k];
        ilo = fminf(ilo, v.x), ihi = fmaxf(ihi, v.y);
      }
    }
  }

  // ---- 2. tap window per view, NOT clipped to the image: a projected coordinate is a ratio of affine functions of
  // (x, y, 1/depth), so over the tile's box its extremes sit on the 8 corners (while the box is in front of the camera)
  // mode: 0 = no tap can carry weight (box outside the image), 1 = box inside the image, 2 = box crosses the border
  // (the fill asks for zeros outside), 3 = no usable bound / does not fit the budget: gather from global memory
  int wx[S], wy[S], wc[S], wr[S], wmode[S], wdrow[S], wdcol[S];
  {
    const int bx0 = tx * TXW, bx1 = min(bx0 + TXW, w) - 1, by0 = ty * TYH, by1 = min(by0 + TYH, h) - 1;
    const float X = (float)((lane & 1) ? bx1 : bx0), Y = (float)((lane & 2) ? by1 : by0), I = (lane & 4) ? ihi : ilo;
    const float px = cP[0] * X + cP[1] * Y + cP[2] + cP[3] * I;
    const float py = cP[4] * X + cP[5] * Y + cP[6] + cP[7] * I;
    const float pz = cP[8] * X + cP[9] * Y + cP[10] + cP[11] * I;
    const bool bad = !(pz > 1e-6f);
    const float iz = __builtin_amdgcn_rcpf(fmaxf(pz, 1e-6f));
    const float uu = px * iz, vv = py * iz;
    float ulo = bad ? -INFINITY : uu, uhi = bad ? INFINITY : uu, vlo = bad ? -INFINITY : vv, vhi = bad ? INFINITY : vv;
    oct_min_max2x(ulo, uhi, vlo, vhi);
    // texel range [floor(lo), floor(hi) + 1] with a rounding margin
    const float fx_lo = floorf(ulo - 0.01f), fx_hi = floorf(uhi + 0.01f) + 1.f;
    const float fy_lo = floorf(vlo - 0.01f), fy_hi = floorf(vhi + 0.01f) + 1.f;
    const float fwc = fx_hi - fx_lo + 1.f, fwr = fy_hi - fy_lo + 1.f;
    // (comparisons that fail on NaN / infinity)
    const bool bounded = (fwc >= 2.f) && (fwc <= 4096.f) && (fwr >= 2.f) && (fwr <= 4096.f) && (fabsf(fx_lo) < 1e6f) && (fabsf(fy_lo) < 1e6f);
    const bool outside = (fx_hi < 0.f) | (fx_lo > (float)(Ws - 1)) | (fy_hi < 0.f) | (fy_lo > (float)(Hs - 1));
    const bool inside = (fx_lo >= 0.f) & (fx_hi <= (float)(Ws - 1)) & (fy_lo >= 0.f) & (fy_hi <= (float)(Hs - 1));
    const int x_lo = bounded ? (int)fx_lo : 0, y_lo = bounded ? (int)fy_lo : 0;
    const int wc_l = bounded ? (int)fwc : 1, wr_l = bounded ? (int)fwr : 1;
    const bool fits = bounded & ((wc_l * wr_l) <= a.cap);
    const int mode_l = !bounded ? 3 : outside ? 0 : !fits ? 3 : inside ? 1 : 2;
    // per round of pieces a lane's record advances by 16 NW: (rows, columns) of that step
    int drow_l = (int)((float)(16 * NW) * __builtin_amdgcn_rcpf((float)wc_l));
    drow_l += ((drow_l + 1) * wc_l <= 16 * NW) ? 1 : 0;
    drow_l -= (drow_l * wc_l > 16 * NW) ? 1 : 0;
    const int dcol_l = 16 * NW - drow_l * wc_l;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      wx[s] = __builtin_amdgcn_readlane(x_lo, 8 * s), wy[s] = __builtin_amdgcn_readlane(y_lo, 8 * s);
      wc[s] = __builtin_amdgcn_readlane(wc_l, 8 * s), wr[s] = __builtin_amdgcn_readlane(wr_l, 8 * s);
      wmode[s] = __builtin_amdgcn_readlane(mode_l, 8 * s);
      wdrow[s] = __builtin_amdgcn_readlane(drow_l, 8 * s), wdcol[s] = __builtin_amdgcn_readlane(dcol_l, 8 * s);
    }
  }

  const int item_views = a.view_ids ? a.n_all : S;
  const char* fbytes = reinterpret_cast<const char*>(a.feats + (size_t)b * item_views * Hs * Ws * C);
  unsigned vbase[S];
#pragma unroll
  for (int s = 0; s < S; ++s)
    vbase[s] = (unsigned)(a.view_ids ? a.view_ids[b * S + s] : s) * (unsigned)(Hs * Ws) * REC;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(fbytes), 0, (int)((size_t)item_views * Hs * Ws * REC), 0x00020000);

  // ---- 3. fill: piece p = records [16 p, 16 p + 16) of the row-major box; lane = (record lane >> 2, slice lane & 3),
  // a plain copy.  Reads past the box land in LDS nobody reads.
  const int lrec = lane >> 2;
  const unsigned lslice = (unsigned)(lane & 3) * 16u;
  auto issue_fill = [&](int s) {
    const int mode = wmode[s];
    if (mode == 0 || mode == 3 || (a.flags & 1)) return;
    const int wcs = wc[s], ntex = wcs * wr[s];
    const int npieces = (ntex + 15) >> 4;
    char* dst = win + wave * 1024;
    // record L = 16 p + lrec of the row-major box, p = wave, wave + NW, ...: L = L0 + 16 NW i.  With (Q, Rm) =
    // divmod(16 NW i, wc) kept in SCALAR registers, the record sits at (row0 + Q, col0 + Rm), one row further when the
    // column wraps: the lane's byte offset is one of two per-lane constants (selected by one compare) plus a scalar.
    const int L0 = wave * 16 + lrec;
    int row0 = (int)((float)L0 * __builtin_amdgcn_rcpf((float)wcs));
    int col0 = L0 - row0 * wcs;
    if (col0 >= wcs) col0 -= wcs, ++row0;
    if (col0 < 0) col0 += wcs, --row0;
    const int gy0 = wy[s] + row0, gx0 = wx[s] + col0;
    const unsigned off0 = vbase[s] + hoff + (unsigned)(__mul24(gy0, Ws) + gx0) * REC + lslice;
    const unsigned off1 = off0 + (unsigned)(Ws - wcs) * REC;   // ... after a column wrap
    int Q = 0, Rm = 0;                                         // scalar: divmod(16 NW i, wc)
    const int dq = wdrow[s], dr = wdcol[s];
    if (mode == 1) {
      for (int p = wave; p < npieces; p += NW) {
        const bool wrapped = col0 >= wcs - Rm;
        const unsigned so = (unsigned)(Q * Ws + Rm) * REC;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16,
                                                 (int)((wrapped ? off1 : off0) + so), BMV_ZP_SOFF(so), 0, 0);
        dst += NW * 1024;
        Q += dq, Rm += dr;
        if (Rm >= wcs) Rm -= wcs, ++Q;
      }
    } else {   // the box crosses the image border: texels outside are requested out of range (the DMA writes zeros)
      for (int p = wave; p < npieces; p += NW) {
        const bool wrapped = col0 >= wcs - Rm;
        const int gy = gy0 + Q + (wrapped ? 1 : 0), gx = gx0 + Rm - (wrapped ? wcs : 0);
        const bool ok = ((unsigned)gy < (unsigned)Hs) & ((unsigned)gx < (unsigned)Ws);
        const unsigned so = (unsigned)(Q * Ws + Rm) * REC;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16,
                                                 (int)(ok ? (wrapped ? off1 : off0) + so : 0x80000000u), BMV_ZP_SOFF(so), 0, 0);
        dst += NW * 1024;
        Q += dq, Rm += dr;
        if (Rm >= wcs) Rm -= wcs, ++Q;
      }
    }
  };

  // per-lane slice order: accumulator group q holds slice q ^ key of the 64-byte record
  const unsigned key = (unsigned)(lane >> 2) & 3u;
  unsigned cq[4];   // (byte offsets inside a record; the gather path uses them as they are)
#pragma unroll
  for (int q = 0; q < 4; ++q) cq[q] = ((unsigned)q ^ key) * 16u;

  float4 acc[4], acc2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = acc2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto blend4 = [&](int q, float4 t00, float4 t01, float4 t10, float4 t11, float a00, float a01, float a10, float a11) {
    float4 v;
    v.x = t00.x * a00 + t01.x * a01 + t10.x * a10 + t11.x * a11;
    v.y = t00.y * a00 + t01.y * a01 + t10.y * a10 + t11.y * a11;
    v.z = t00.z * a00 + t01.z * a01 + t10.z * a10 + t11.z * a11;
    v.w = t00.w * a00 + t01.w * a01 + t10.w * a10 + t11.w * a11;
    acc[q].x += v.x, acc[q].y += v.y, acc[q].z += v.z, acc[q].w += v.w;
    acc2[q].x += v.x * v.x, acc2[q].y += v.y * v.y, acc2[q].z += v.z * v.z, acc2[q].w += v.w * v.w;
  };

  // ---- 4. per view: geometry (under the latency of the view's fill), then the blend
  // staged window: the 4 weights and the two LDS row addresses per slice -- no validity logic, zeros are in the window
  float w00, w01, w10, w11;
  unsigned a0[4], a1[4];
#if BMV_ZP_PHOIST
  float Pall[S][12];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int k = 0; k < 12; ++k) Pall[s][k] = a.proj[((size_t)b * S + s) * 12 + k];
#endif
  auto geometry = [&](int s) {
#if BMV_ZP_PHOIST
    const float* P = Pall[s];
#else
    const float* P = a.proj + ((size_t)b * S + s) * 12;   // wave-uniform: scalar loads
#endif
    // (each multiply-add takes ONE scalar operand: x * P0, then + y * P1, + id * P3, + P2)
    const float px = fmaf(P[3], inv_depth, fmaf(P[1], fy, P[0] * fx)) + P[2];
    const float py = fmaf(P[7], inv_depth, fmaf(P[5], fy, P[4] * fx)) + P[6];
    const float pz = fmaf(P[11], inv_depth, fmaf(P[9], fy, P[8] * fx)) + P[10];
    const float iz = __builtin_amdgcn_rcpf(pz);
    // uv / ((W-1)/2) - 1 followed by grid_sample's ((g+1)/2) (W-1) is the identity up to rounding
    const float ix = px * iz, iy = py * iz;
    const int tx0 = floor_to_int(ix), ty0 = floor_to_int(iy);
    const float ax = __builtin_amdgcn_fractf(ix), ay = __builtin_amdgcn_fractf(iy);
    const int wcs = wc[s];
    const unsigned rec = (unsigned)(__mul24(ty0, wcs) + tx0 - (wy[s] * wcs + wx[s]));
    const unsigned rowb = (unsigned)wcs * 64u;
    const float bx = 1.f - ax, by = 1.f - ay;
    w00 = bx * by, w01 = ax * by, w10 = bx * ay, w11 = ax * ay;
#pragma unroll
    for (int q = 0; q < 4; ++q) a0[q] = (rec << 6) + cq[q], a1[q] = a0[q] + rowb;
  };
  auto blend_staged = [&]() {
#if BMV_ZP_TAPBUF == 1
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      blend4(q, lds4(win, a0[q]), lds4(win, a0[q] + 64u), lds4(win, a1[q]), lds4(win, a1[q] + 64u), w00, w01, w10, w11);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
#endif
    // two slices in flight: the reads of slice q + 1 are issued before the blend of slice q
    float4 t00 = lds4(win, a0[0]), t01 = lds4(win, a0[0] + 64u), t10 = lds4(win, a1[0]), t11 = lds4(win, a1[0] + 64u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 n00, n01, n10, n11;
      if (q < 3) n00 = lds4(win, a0[q + 1]), n01 = lds4(win, a0[q + 1] + 64u), n10 = lds4(win, a1[q + 1]), n11 = lds4(win, a1[q + 1] + 64u);
      __builtin_amdgcn_sched_barrier(0);
      blend4(q, t00, t01, t10, t11, w00, w01, w10, w11);
      __builtin_amdgcn_sched_barrier(0);
      if (q < 3) t00 = n00, t01 = n01, t10 = n10, t11 = n11;
    }
  };
  // the view gathered from global memory with the full zero-padding logic (rare)
  auto blend_gather = [&](int s) {
    const float* P = a.proj + ((size_t)b * S + s) * 12;
    const float px = P[0] * fx + P[1] * fy + P[2] + P[3] * inv_depth;
    const float py = P[4] * fx + P[5] * fy + P[6] + P[7] * inv_depth;
    const float pz = P[8] * fx + P[9] * fy + P[10] + P[11] * inv_depth;
    const float iz = __builtin_amdgcn_rcpf(fmaxf(pz, 1e-6f));
    const float ix = px * iz, iy = py * iz;
    const float flx = floorf(ix), fly = floorf(iy);
    // clamp before the int conversion (also maps NaN into range): anything outside ends with both taps invalid
    const int tx0 = (int)__builtin_amdgcn_fmed3f(flx, -2.f, (float)Ws), ty0 = (int)__builtin_amdgcn_fmed3f(fly, -2.f, (float)Hs);
    const bool vx0 = (unsigned)tx0 < (unsigned)Ws, vx1 = (unsigned)(tx0 + 1) < (unsigned)Ws;
    const bool vy0 = (unsigned)ty0 < (unsigned)Hs, vy1 = (unsigned)(ty0 + 1) < (unsigned)Hs;
    const float ax = ix - flx, ay = iy - fly;
    const bool any = (vx0 | vx1) & (vy0 | vy1);
    if (!__any(any)) return;
    const float wx0 = vx0 ? 1.f - ax : 0.f, wx1 = vx1 ? ax : 0.f;
    const float wy0 = (vy0 & any) ? 1.f - ay : 0.f, wy1 = (vy1 & any) ? ay : 0.f;
    const float g00w = wx0 * wy0, g01w = wx1 * wy0, g10w = wx0 * wy1, g11w = wx1 * wy1;
    // a tap outside the image has weight 0 and is parked on its in-image neighbour
    const int gx = any ? (vx0 ? tx0 : tx0 + 1) : 0, gy = any ? (vy0 ? ty0 : ty0 + 1) : 0;
    const unsigned g00 = vbase[s] + hoff + (unsigned)(gy * Ws + gx) * REC;
    const unsigned gdx = (any && vx0 && vx1) ? REC : 0u, gdy = (any && vy0 && vy1) ? (unsigned)Ws * REC : 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned o = g00 + cq[q];
      i32x4 ra = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
      i32x4 rb = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + gdx, 0, 0);
      i32x4 rc = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + gdy, 0, 0);
      i32x4 rd = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + gdx + gdy, 0, 0);
      blend4(q, *reinterpret_cast<float4*>(&ra), *reinterpret_cast<float4*>(&rb), *reinterpret_cast<float4*>(&rc),
             *reinterpret_cast<float4*>(&rd), g00w, g01w, g10w, g11w);
      __builtin_amdgcn_sched_barrier(0);   // one slice in flight: this path is rare, registers matter more
    }
  };

  issue_fill(0);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int mode = wmode[s];
    const bool staged = (mode == 1) | (mode == 2);
    if (staged) geometry(s);   // under the latency of this view's fill
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();           // the fill of view s has landed in every wave's pieces
    if (!(a.flags & 2)) {
      if (staged)
        blend_staged();
      else if (mode == 3)
        blend_gather(s);
    }
    if (s + 1 < S) {
      __syncthreads();         // every wave is done with the window
      issue_fill(s + 1);
    }
  }

  // ---- 5. variance back into channel order (2 x 16 v_cndmask), one dword per lane and channel, scalar channel offsets
  {
    const float inv_s = 1.f / (float)S;
    float V[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float m;
      m = acc[q].x * inv_s, V[q][0] = acc2[q].x * inv_s - m * m;
      m = acc[q].y * inv_s, V[q][1] = acc2[q].y * inv_s - m * m;
      m = acc[q].z * inv_s, V[q][2] = acc2[q].z * inv_s - m * m;
      m = acc[q].w * inv_s, V[q][3] = acc2[q].w * inv_s - m * m;
    }
    // lane masks of key & 1 and key & 2 (key = (lane >> 2) & 3)
    const unsigned long long m1 = 0xf0f0f0f0f0f0f0f0ull, m2 = 0xff00ff00ff00ff00ull;
    float T[4][4], Wn[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) T[c][k] = lane_select(V[c][k], V[c ^ 1][k], m1);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) Wn[c][k] = lane_select(T[c][k], T[c ^ 2][k], m2);
    if (inb && !(a.flags & 4)) {
      const unsigned cstride = (unsigned)(D * hw) * 4u;
      __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
          a.out + (size_t)b * C * D * hw, 0, (int)((size_t)C * D * hw * 4), 0x00020000);
      const unsigned voff = (unsigned)((size_t)d * hw + (size_t)y * w + x) * 4u;
      unsigned soff = (unsigned)chh * 16u * cstride;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, Wn[c][k]), orsrc, (int)voff, (int)soff, kStoreAux);
          soff += cstride;
        }
    }
  }
}

}  // namespace bmv

using namespace bmv;

namespace {

struct ZpVariant {
  int txw, tyh, dp, wpe, cap;   // tile, waves per SIMD the registers are budgeted for, records of LDS
};
// tuning table (algo 200 + i); caps in 64-byte records, multiples of 16
const ZpVariant kZp[] = {
    {32, 8, 1, 5, 448},   // 0: level 1 (source at the volume's resolution): 28 KB, 5 workgroups per CU
    {32, 8, 1, 4, 448},   // 1
    {16, 2, 8, 5, 448},   // 2: level 0 (source at twice the resolution, 8 planes share a window)
    {16, 2, 8, 4, 448},   // 3
    {32, 8, 1, 6, 400},   // 4
    {16, 2, 8, 6, 400},   // 5
    {32, 4, 1, 5, 256},   // 6: 128-thread workgroups
    {32, 4, 1, 6, 256},   // 7
    {16, 4, 8, 4, 448},   // 8: 512 threads
    {16, 4, 4, 5, 448},   // 9
    {32, 4, 2, 5, 448},   // 10
    {16, 2, 8, 5, 320},   // 11
    {32, 8, 1, 5, 400},   // 12
};
constexpr int kNumZp = sizeof(kZp) / sizeof(kZp[0]);

template <int TXW, int TYH, int DP, int S, int WPE, bool PU>
int zp_launch_one(const ZpArgs& a, int B, hipStream_t stream) {
  auto kern = sweep_zp_kernel<TXW, TYH, DP, S, WPE, PU>;
  const size_t lds = (size_t)a.cap * 64 + 64;   // window + the depth-range slots
  static size_t allowed = 0;
  if (lds > allowed) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      (void)hipGetLastError();
      return BMV_ERR_UNSUPPORTED;
    }
    allowed = lds;
  }
  dim3 grid(8u * (unsigned)(a.chalves * a.pgroups), (unsigned)(a.tiles_x * a.tyb), B), block(TXW * TYH * DP);
  const LaunchEvents ev = take_launch_events();
  if (ev.start)   // bench.py's roofline bracket: events bound to this dispatch (bmv_bind_next_launch)
    hipExtLaunchKernelGGL(kern, grid, block, lds, stream, ev.start, ev.stop, 0, a);
  else
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  return BMV_OK;
}

template <int TXW, int TYH, int DP, int WPE>
int zp_launch_s(const ZpArgs& a, int B, int S, bool pu, hipStream_t stream) {
  if (pu) {
    if (S == 3) return zp_launch_one<TXW, TYH, DP, 3, WPE, true>(a, B, stream);
    if (S == 2) return zp_launch_one<TXW, TYH, DP, 2, WPE, true>(a, B, stream);
    if (S == 4) return zp_launch_one<TXW, TYH, DP, 4, WPE, true>(a, B, stream);
  } else {
    if (S == 3) return zp_launch_one<TXW, TYH, DP, 3, WPE, false>(a, B, stream);
    if (S == 2) return zp_launch_one<TXW, TYH, DP, 2, WPE, false>(a, B, stream);
    if (S == 4) return zp_launch_one<TXW, TYH, DP, 4, WPE, false>(a, B, stream);
  }
  return BMV_ERR_UNSUPPORTED;
}

}  // namespace

// dv_plane_uniform: 0 = depth_values is (B,D,h,w), one hypothesis per voxel; 1 = (B,D), one hypothesis per plane
// (cascade level 0); 2 = a (B,D,h,w) tensor whose planes are constant (element [b,d,0,0] is read)
extern "C" int bmv_sweep_zp_launch(const float* feats, const float* proj, const float* dv, int dv_plane_uniform, int B,
                                   int S, int C, int Hs, int Ws, int D, int h, int w, float* out, const int* view_ids,
                                   int n_all, int variant, hipStream_t stream) {
  if ((C != 16 && C != 32) || S < 2 || S > 4) return BMV_ERR_UNSUPPORTED;
  if ((size_t)(view_ids ? n_all : S) * Hs * Ws * C * 4 >= ((size_t)1 << 31)) return BMV_ERR_UNSUPPORTED;
  if (Hs >= (1 << 14) - 2 || Ws >= (1 << 14) - 2) return BMV_ERR_UNSUPPORTED;
  if ((size_t)C * D * h * w * 4 >= ((size_t)1 << 31)) return BMV_ERR_UNSUPPORTED;   // 32-bit volume offsets
  if (variant < 0) variant = (float)Ws / (float)w <= 1.5f ? 0 : 2;
  if (variant >= kNumZp) return BMV_ERR_UNSUPPORTED;
  ZpVariant v = kZp[variant];
  if (const char* e = getenv("BMV_SWEEP_ZP_CAP")) {
    int c = atoi(e);
    if (c >= 16) v.cap = (c + 15) & ~15;
  }
  ZpArgs a;
  a.feats = feats, a.proj = proj, a.dv = dv, a.out = out, a.view_ids = view_ids, a.n_all = n_all;
  a.C = C, a.Hs = Hs, a.Ws = Ws, a.D = D, a.h = h, a.w = w;
  a.tiles_x = (w + v.txw - 1) / v.txw;
  const int tiles_y = (h + v.tyh - 1) / v.tyh;
  a.tyb = (tiles_y + 7) / 8;
  a.pgroups = (D + v.dp - 1) / v.dp;
  a.chalves = C / 16;
  a.cap = v.cap;
  a.tiles_x_magic = a.tiles_x == 1 ? 0u : (unsigned)(((unsigned long long)1 << 32) / (unsigned)a.tiles_x) + 1u;
  if (a.tiles_x * a.tyb >= 65536) return BMV_ERR_UNSUPPORTED;
  if (dv_plane_uniform == 1)
    a.dv_ps = 1, a.dv_rs = 0, a.dv_cs = 0, a.dv_bs = D;
  else if (dv_plane_uniform == 2)
    a.dv_ps = h * w, a.dv_rs = 0, a.dv_cs = 0, a.dv_bs = (long long)D * h * w;
  else
    a.dv_ps = h * w, a.dv_rs = w, a.dv_cs = 1, a.dv_bs = (long long)D * h * w;
  if ((size_t)D * h * w >= ((size_t)1 << 23)) return BMV_ERR_UNSUPPORTED;   // 24-bit multiplies of the hypothesis offsets
  a.flags = 0;
  if (const char* e = getenv("BMV_SWEEP_ZP_FLAGS")) a.flags = atoi(e);
  const bool pu = dv_plane_uniform != 0;
  int rc = BMV_ERR_UNSUPPORTED;
#define V(TXW, TYH, DP, WPE) \
  if (v.txw == TXW && v.tyh == TYH && v.dp == DP && v.wpe == WPE) rc = zp_launch_s<TXW, TYH, DP, WPE>(a, B, S, pu, stream);
  V(32, 8, 1, 5)
  V(32, 8, 1, 4)
  V(16, 2, 8, 5)
  V(16, 2, 8, 4)
  V(32, 8, 1, 6)
  V(16, 2, 8, 6)
  V(32, 4, 1, 5)
  V(32, 4, 1, 6)
  V(16, 4, 8, 4)
  V(16, 4, 4, 5)
  V(32, 4, 2, 5)
#undef V
  if (rc != BMV_OK) return rc;
  BMV_LAUNCH_END("bmv_sweep_variance_fwd(zp)");
}
