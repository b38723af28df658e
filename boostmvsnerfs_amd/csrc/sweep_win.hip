// Plane sweep with LDS-staged source windows (a3+a4, channel-last features, C in {16, 32}).
// Reference: lib/networks/enerf/utils.py:57-95 (homo_warp), :324-351 (build_feature_volume).
//
// Why: the gather kernels (sweep_tiled.hip, sweep_split.hip) fetch every bilinear tap as its own 16-byte-per-lane
// request through the vector L1 -- 503 MB of tap traffic per launch at config 2, which the texture path cannot move in
// less than ~13 us whatever HBM does.  Here the texels a tile of voxels needs are copied into LDS ONCE per view by
// LDS-DMA (coalesced row pieces of 1 KB, no VGPR round trip) and every tap is a ds_read_b128 (256 B/clk/CU).
//
// Decomposition.  A workgroup owns TXW x TYH target pixels x DP depth planes x 16 channels (a 32-channel level is
// two workgroups, one per channel half: a record is then 64 of the 128 bytes of a source pixel).
//   lane = one voxel with its 16 channels (sum / sum of squares: 32 accumulators); a wave is 64 consecutive voxels
//          of one plane (TXW = 32: two rows of 32), so every store instruction writes whole 128-byte lines of the
//          (B,C,D,h,w) volume and the 16-lane groups of a ds_read_b128 read neighbouring records;
//   1. geometry: every lane projects its voxel into the S views once and keeps (first tap | share bits, 4 axis
//      weights with the zero-padding validity folded in) per view;
//   2. window:   the EXACT bounding box of the taps that carry weight, per view, by a DPP wave reduction + one LDS
//      exchange (the depth hypotheses vary per pixel, so corner voxels do not bound it);
//   3. fill:     the box goes to LDS as 16-record pieces, NB windows in flight (views are processed in turn and
//      the fill of view s+1 runs under the blend of view s);
//   4. blend:    16 ds_read_b128 per voxel and view; a wave with a tap outside the staged box (box larger than the
//      LDS budget) gathers that view from global memory instead -- correctness never depends on the window;
//   5. variance written once.
// Bank conflicts: 16 lanes reading the same 16-byte slice of 16 neighbouring 64-byte records would hit 4 of the 16
// four-bank groups, so slice q of record r sits at slot q ^ ((r >> 2) & 3); the DMA writes LDS linearly, so the
// permutation is applied to the per-lane SOURCE address.
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "bmv_common.hpp"

namespace bmv {

namespace {

__device__ __forceinline__ float4 ld4_lds(const char* base, unsigned byte) {
  return *reinterpret_cast<const float4*>(base + byte);
}

// min and max over every 16-lane row (4 DPP steps each; the two chains fill each other's DPP wait states)
__device__ __forceinline__ void row_min_max(float& lo, float& hi) {
  asm volatile(
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(lo), "+v"(hi));
}
// min and max over every group of 8 lanes (3 DPP steps), two pairs at once
__device__ __forceinline__ void oct_min_max2(float& lo0, float& hi0, float& lo1, float& hi1) {
  asm volatile(
      "s_nop 1\n"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_min_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_max_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1));
}

}  // namespace

struct WinArgs {
  const float* feats;
  const float* proj;
  const float* dv;
  float* out;
  const int* view_ids;
  int n_all, C, Hs, Ws, D, h, w;
  int tiles_x, tyb, pgroups, chalves, cap;
  unsigned tiles_x_magic;   // floor(2^32 / tiles_x) + 1: n / tiles_x = umulhi(n, magic) for n < 2^16
  int flags;   // ablation switches (BMV_SWEEP_WIN_FLAGS, tuning only): 1 no fill, 2 no blend, 4 no store
  int nh;      // channel halves per workgroup (host side: 2 only where the kernel is instantiated)
};

// waves per SIMD the register allocator must leave room for: 256-thread workgroups (4 waves, 29 KB of LDS) fit 5 per
// CU = 5 waves per SIMD at 96 VGPRs (13 spilled: level 1 22.9 -> 21.9 us, scripts/ab_sweep_wpe.sh); the 512-thread
// ones are 2 per CU whatever the register count, so they keep 128 registers
#ifndef BMV_WIN_WPE
#define BMV_WIN_WPE(threads) ((threads) <= 256 ? 5 : 4)
#else
#define BMV_WIN_WPE_FIXED BMV_WIN_WPE
#undef BMV_WIN_WPE
#define BMV_WIN_WPE(threads) BMV_WIN_WPE_FIXED
#endif
// Cache-policy bits of the variance stores.  2 = non-temporal: the volume (42 MB per level, more than the 32 MB of L2)
// is written once and streams out -- level 1 (rows of 32 voxels = whole lines) 22.9 -> 21.35 us stand-alone (1 = sc0:
// no change; 17 / 19 = sc1: slower).  The consumer (the regulariser's first layer) then finds less of it in L2: the
// frame as a whole is unchanged (scripts/ab_sweep_nt.sh); -DBMV_WIN_STORE_AUX=0 restores cached stores.  Applied only
// to tiles whose rows are whole 128-byte lines (kStoreAux in the kernel).
#ifndef BMV_WIN_STORE_AUX
#define BMV_WIN_STORE_AUX 2
#endif
#ifndef BMV_WIN_TAPBUF
#define BMV_WIN_TAPBUF 2
#endif

// NH = channel halves a workgroup does one after the other (C = 32 with NH = 2: the prologue -- depth range, window
// boxes -- and the per-view tap geometry are computed once and reused for the second 16 channels; with NH = 1 the two
// halves are two workgroups that each redo them)
template <int TXW, int TYH, int DP, int S, int NB, int NH>
__global__ void __launch_bounds__(TXW* TYH* DP) __attribute__((amdgpu_waves_per_eu(BMV_WIN_WPE(TXW* TYH* DP), 8)))
sweep_win_kernel(const WinArgs a) {
  constexpr int NT = TXW * TYH * DP, NW = NT / 64;
  // non-temporal stores only where a wave row is a whole 128-byte line: for 16-pixel rows (64-byte half lines) the
  // streaming stores are not merged in L2 any more and WRITE_SIZE grows from 43.5 to 52.6 MB for the 41.9 MB volume
  constexpr int kStoreAux = TXW >= 32 ? BMV_WIN_STORE_AUX : 0;
  static_assert(NT % 64 == 0 && NT <= 1024, "workgroup size");
  static_assert((TXW * TYH) % 32 == 0, "a half wave covers 32 voxels of ONE plane");
  static_assert(8 * S <= 64, "corner lanes");
  __shared__ float2 slots[NW];
  extern __shared__ __attribute__((aligned(64))) char win[];  // NB windows of cap records (64 B each)

  unsigned long long st[12];   // tuning: phase time stamps (flags & 64), shader clock
#ifdef BMV_WIN_STAMPS
#define BMV_STAMP(i)                                       \
  if (a.flags & 64) {                                      \
    __builtin_amdgcn_sched_barrier(0);                     \
    st[i] = __builtin_amdgcn_s_memtime();                  \
    __builtin_amdgcn_sched_barrier(0);                     \
  }
#else
#define BMV_STAMP(i)
#endif
#pragma unroll
  for (int i = 0; i < 12; ++i) st[i] = 0;
  BMV_STAMP(0)
  // grid = (8 bands x channel halves x plane groups, tile columns x tile rows of a band, batch); blockIdx.x % 8 = the
  // band = the XCD whose L2 holds that band's source rows.  The scalar unit is shared by the whole CU: no integer
  // divisions here (a runtime s_div is ~40 scalar instructions), only shifts and one multiply-high
  const int b = blockIdx.z;
  const int band = blockIdx.x & 7;
  const int kx = blockIdx.x >> 3;
  const int chh0 = kx & (a.chalves - 1);         // chalves is 1 or 2 (1 when the halves are done in-kernel)
  const int pg = kx >> (a.chalves - 1);
  // blockIdx.y / tiles_x (magic 0 = one tile column: 2^32 / 1 + 1 does not fit the multiplier)
  const int j = a.tiles_x_magic ? (int)__umulhi((unsigned)blockIdx.y, (unsigned)a.tiles_x_magic) : (int)blockIdx.y;
  const int tx = blockIdx.y - j * a.tiles_x;
  const int ty = band * a.tyb + j;
  if (ty * TYH >= a.h) return;  // whole workgroup, before any barrier
  const int C = a.C, Hs = a.Hs, Ws = a.Ws, D = a.D, h = a.h, w = a.w;
  const unsigned REC = (unsigned)C * 4u;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = tid % TXW, ly = (tid / TXW) % TYH, ld = tid / (TXW * TYH);
  const int x = tx * TXW + lx, y = ty * TYH + ly, d = pg * DP + ld;
  const bool inb = (x < w) & (y < h) & (d < D);
  const int xc = min(x, w - 1), yc = min(y, h - 1), dc = min(d, D - 1);
  const size_t hw = (size_t)h * w;

  // corner lanes (lane < 8 S): corner (lane & 7) of the tile box in (x, y, 1/depth), view lane >> 3; their
  // projection rows are fetched now, under the latency of the depth load
  const int cview = min(lane >> 3, S - 1);
  float cP[12];
  {
    const float* Pc = a.proj + ((size_t)b * S + cview) * 12;
#pragma unroll
    for (int j = 0; j < 12; ++j) cP[j] = Pc[j];
  }
  const float inv_depth = __builtin_amdgcn_rcpf(a.dv[((size_t)b * D + dc) * hw + (size_t)yc * w + xc]);

  // ---- 1. range of 1/depth over the workgroup
  float ilo = inv_depth, ihi = inv_depth;
#ifdef BMV_WIN_STAMPS
  if (a.flags & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  BMV_STAMP(1)
  row_min_max(ilo, ihi);
  {
    auto rl = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    const float l0 = rl(ilo, 0), l1 = rl(ilo, 16), l2 = rl(ilo, 32), l3 = rl(ilo, 48);
    const float h0 = rl(ihi, 0), h1 = rl(ihi, 16), h2 = rl(ihi, 32), h3 = rl(ihi, 48);
    ilo = fminf(fminf(l0, l1), fminf(l2, l3)), ihi = fmaxf(fmaxf(h0, h1), fmaxf(h2, h3));
  }
  if (NW > 1) {
    if (lane == 0) slots[wave] = make_float2(ilo, ihi);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const float2 v = slots[k];
      ilo = fminf(ilo, v.x), ihi = fmaxf(ihi, v.y);
    }
  }

  BMV_STAMP(2)
  // ---- 2. tap window per view: a projected coordinate is a ratio of affine functions of (x, y, 1/depth), so over the
  // tile's box its extremes sit on the 8 corners (as long as the box stays in front of the camera: otherwise the
  // window is "everything", gets clipped to the LDS budget and the waves fall back to global gathers)
  int wx[S], wy[S], wcols[S], wrows[S], wdrow[S], wdcol[S];
  {
    const int bx0 = tx * TXW, bx1 = min(bx0 + TXW, w) - 1, by0 = ty * TYH, by1 = min(by0 + TYH, h) - 1;
    const float X = (float)((lane & 1) ? bx1 : bx0), Y = (float)((lane & 2) ? by1 : by0), I = (lane & 4) ? ihi : ilo;
    const float px = cP[0] * X + cP[1] * Y + cP[2] + cP[3] * I;
    const float py = cP[4] * X + cP[5] * Y + cP[6] + cP[7] * I;
    const float pz = cP[8] * X + cP[9] * Y + cP[10] + cP[11] * I;
    const float iz = __builtin_amdgcn_rcpf(fmaxf(pz, 1e-6f));
    const bool bad = !(pz > 1e-6f);
    const float u = px * iz, v = py * iz;
    float ulo = bad ? -INFINITY : u, uhi = bad ? INFINITY : u, vlo = bad ? -INFINITY : v, vhi = bad ? INFINITY : v;
    oct_min_max2(ulo, uhi, vlo, vhi);
    // texel range [floor(lo), floor(hi) + 1] with a rounding margin, clipped to the image; clipped to the LDS budget;
    // all on the corner lanes (vector ALU), the scalar unit only receives the results
    const float fWs = (float)Ws, fHs = (float)Hs;
    const int x_lo = (int)fminf(fmaxf(floorf(ulo - 0.01f), 0.f), fWs), x_hi = (int)fminf(fmaxf(floorf(uhi + 0.01f) + 1.f, -1.f), fWs - 1.f);
    const int y_lo = (int)fminf(fmaxf(floorf(vlo - 0.01f), 0.f), fHs), y_hi = (int)fminf(fmaxf(floorf(vhi + 0.01f) + 1.f, -1.f), fHs - 1.f);
    const bool empty = (x_hi < x_lo) | (y_hi < y_lo);
    const int wc_l = empty ? 0 : min(x_hi - x_lo + 1, a.cap);
    int fit = (int)((float)a.cap * __builtin_amdgcn_rcpf((float)max(wc_l, 1)));   // cap / wc, fixed up below
    fit += ((fit + 1) * wc_l <= a.cap) ? 1 : 0;
    fit -= (fit * wc_l > a.cap) ? 1 : 0;
    const int wr_l = empty ? 0 : min(y_hi - y_lo + 1, fit);
    // per piece a wave's records advance by 16 NW: (rows, columns) of that step
    int drow_l = (int)((float)(16 * NW) * __builtin_amdgcn_rcpf((float)max(wc_l, 1)));
    drow_l += ((drow_l + 1) * wc_l <= 16 * NW) ? 1 : 0;
    drow_l -= (drow_l * wc_l > 16 * NW) ? 1 : 0;
    const int dcol_l = 16 * NW - drow_l * wc_l;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      wx[s] = __builtin_amdgcn_readlane(empty ? 0 : x_lo, 8 * s), wy[s] = __builtin_amdgcn_readlane(empty ? 0 : y_lo, 8 * s);
      wcols[s] = __builtin_amdgcn_readlane(wc_l, 8 * s), wrows[s] = __builtin_amdgcn_readlane(wr_l, 8 * s);
      wdrow[s] = __builtin_amdgcn_readlane(drow_l, 8 * s), wdcol[s] = __builtin_amdgcn_readlane(dcol_l, 8 * s);
    }
  }

  BMV_STAMP(3)
  const int item_views = a.view_ids ? a.n_all : S;
  const char* fbytes = reinterpret_cast<const char*>(a.feats + (size_t)b * item_views * Hs * Ws * C);
  unsigned vbase[S];
#pragma unroll
  for (int s = 0; s < S; ++s)
    vbase[s] = (unsigned)(a.view_ids ? a.view_ids[b * S + s] : s) * (unsigned)(Hs * Ws) * REC;
  unsigned hoff = 0;   // byte offset of the channel half inside a source record
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(fbytes), 0, (int)((size_t)item_views * Hs * Ws * REC), 0x00020000);

  // ---- 3. fill: piece p = records [16p, 16p + 16) of the row-major window; lane = (record, slot).  Reads past the
  // window stay inside the buffer descriptor (or return 0) and land in LDS nobody reads.
  const int lrec = lane >> 2;
  // slice q of window record r sits at slot q ^ ((r >> 2) & 3); a piece starts at a multiple of 16 records, so the
  // slice this lane fetches for its slot (lane & 3) is the same for every piece
  const unsigned lslice = (unsigned)((lane ^ (lane >> 4)) & 3) * 16u;
  auto issue_fill = [&](int s, int buf) {
    const int wc = wcols[s], ntex = wc * wrows[s];
    if (ntex == 0 || (a.flags & 1)) return;
    const int npieces = (ntex + 15) >> 4;
    char* dst = win + (size_t)buf * a.cap * 64;
    // (row, col) of this lane's record in the wave's first piece; per piece the record index advances by 16 NW =
    // (drow, dcol) of the window, i.e. by a constant byte step plus one extra row whenever the column wraps
    const int L = wave * 16 + lrec;
    int row = (int)((float)L * (1.f / (float)wc));
    int col = L - row * wc;
    if (col >= wc) col -= wc, ++row;
    if (col < 0) col += wc, --row;
    unsigned off = vbase[s] + hoff + (unsigned)((wy[s] + row) * Ws + wx[s] + col) * REC + lslice;
    const unsigned step = (unsigned)(wdrow[s] * Ws + wdcol[s]) * REC, wrap = (unsigned)(Ws - wc) * REC;
    const int dcol = wdcol[s];
    for (int p = wave; p < npieces; p += NW) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, (int)off, 0,
                                               0, 0);
      col += dcol, off += step;
      if (col >= wc) col -= wc, off += wrap;
    }
  };
  int issued = 0, landed = 0;

  // ---- 4. geometry of this voxel in every view, reduced to what the blend needs: the LDS byte addresses of the two
  // upper taps (flags in the 4 free low bits of the first: 1 = lower row exists, 2 = a tap lies outside the staged
  // window, 4 = some tap carries weight) and the 4 bilinear weights with the zero padding folded in
  struct Taps {
    float ix, iy, ax, ay;
    int cx, cy, shx, shy;
    bool vx0, vx1, vy0, vy1, any;
  };
  const float fxc = (float)xc, fyc = (float)yc;
  auto project = [&](int s, float inv_depth) {
    Taps t;
    const float* P = a.proj + ((size_t)b * S + s) * 12;
    const float px = P[0] * fxc + P[1] * fyc + P[2] + P[3] * inv_depth;
    const float py = P[4] * fxc + P[5] * fyc + P[6] + P[7] * inv_depth;
    const float pz = P[8] * fxc + P[9] * fyc + P[10] + P[11] * inv_depth;
    const float iz = __builtin_amdgcn_rcpf(fmaxf(pz, 1e-6f));
    // uv / ((W-1)/2) - 1 followed by grid_sample's ((g+1)/2) (W-1) is the identity up to rounding
    t.ix = px * iz, t.iy = py * iz;
    const float flx = floorf(t.ix), fly = floorf(t.iy);
    // clamp before the int conversion (also maps NaN into range): anything outside ends with both taps invalid
    const int tx0 = (int)__builtin_amdgcn_fmed3f(flx, -2.f, (float)Ws), ty0 = (int)__builtin_amdgcn_fmed3f(fly, -2.f, (float)Hs);
    t.vx0 = ((unsigned)tx0 < (unsigned)Ws) & inb, t.vx1 = ((unsigned)(tx0 + 1) < (unsigned)Ws) & inb;
    t.vy0 = (unsigned)ty0 < (unsigned)Hs, t.vy1 = (unsigned)(ty0 + 1) < (unsigned)Hs;
    t.ax = t.ix - flx, t.ay = t.iy - fly;
    t.any = (t.vx0 | t.vx1) & (t.vy0 | t.vy1);
    // a tap outside the image has weight 0 and is parked on its in-image neighbour
    t.cx = t.vx0 ? tx0 : tx0 + 1, t.cy = t.vy0 ? ty0 : ty0 + 1;
    t.shx = (t.vx0 & t.vx1) ? 1 : 0, t.shy = (t.vy0 & t.vy1) ? 1 : 0;
    return t;
  };
  unsigned b00, b01, b10, b11;
  float w00, w01, w10, w11;
  auto geometry = [&](int s) {
    const Taps t = project(s, inv_depth);
    const float wx0 = t.vx0 ? 1.f - t.ax : 0.f, wx1 = t.vx1 ? t.ax : 0.f;
    const float wy0 = t.vy0 ? 1.f - t.ay : 0.f, wy1 = t.vy1 ? t.ay : 0.f;
    w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
    int rx = t.cx - wx[s], ry = t.cy - wy[s];
    const bool inwin = ((rx | ry) >= 0) & (rx + t.shx < wcols[s]) & (ry + t.shy < wrows[s]);
    const bool slow = t.any & !inwin;
    if (!t.any | slow) rx = ry = 0;   // parked on the window origin (a slow wave never reads LDS for this view)
    const int shx = (t.any & !slow) ? t.shx : 0;
    const int shy = (t.any & !slow) ? t.shy : 0;
    const unsigned boff = (unsigned)(s % NB) * (unsigned)a.cap * 64u;
    const unsigned L00 = (unsigned)(ry * wcols[s] + rx), L01 = L00 + (unsigned)shx;
    const unsigned L10 = L00 + (shy ? (unsigned)wcols[s] : 0u), L11 = L10 + (unsigned)shx;
    b00 = (boff + (L00 << 6)) | (((L00 >> 2) & 3u) << 4) | (slow ? 2u : 0u) | (t.any ? 4u : 0u);
    b01 = (boff + (L01 << 6)) | (((L01 >> 2) & 3u) << 4);
    b10 = (boff + (L10 << 6)) | (((L10 >> 2) & 3u) << 4);
    b11 = (boff + (L11 << 6)) | (((L11 >> 2) & 3u) << 4);
  };
  BMV_STAMP(4)

  float4 acc[4], acc2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = acc2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  using i32x4 = __attribute__((ext_vector_type(4))) int;

  auto blend = [&](int q, float4 t00, float4 t01, float4 t10, float4 t11, float a00, float a01, float a10,
                   float a11) {
    float4 v;
    v.x = t00.x * a00 + t01.x * a01 + t10.x * a10 + t11.x * a11;
    v.y = t00.y * a00 + t01.y * a01 + t10.y * a10 + t11.y * a11;
    v.z = t00.z * a00 + t01.z * a01 + t10.z * a10 + t11.z * a11;
    v.w = t00.w * a00 + t01.w * a01 + t10.w * a10 + t11.w * a11;
    acc[q].x += v.x, acc[q].y += v.y, acc[q].z += v.z, acc[q].w += v.w;
    acc2[q].x += v.x * v.x, acc2[q].y += v.y * v.y, acc2[q].z += v.z * v.z, acc2[q].w += v.w * v.w;
  };

  auto compute = [&](int s) {
    if (a.flags & 2) return;
    const unsigned f = b00;
    if (!__any((f & 4u) != 0)) return;   // no voxel of the wave sees this view: it contributes 0 to both sums
    if (__builtin_expect(__any((f & 2u) != 0), 0)) {
      if (a.flags & 16) return;
      // a tap outside the staged box (box clipped by the LDS budget, or a view partly behind the camera): this wave
      // gathers the view from global memory, from its own geometry
      float inv_depth2 = inv_depth;
      asm volatile("" : "+v"(inv_depth2));   // not a common subexpression of the first projection: nothing stays live
      const Taps t = project(s, inv_depth2);
      const int gx = t.any ? t.cx : 0, gy = t.any ? t.cy : 0;
      const unsigned g00 = vbase[s] + hoff + (unsigned)(gy * Ws + gx) * REC;
      const unsigned gdx = (t.any && t.shx) ? REC : 0u, gdy = (t.any && t.shy) ? (unsigned)Ws * REC : 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        i32x4 ra = __builtin_amdgcn_raw_buffer_load_b128(rsrc, g00 + 16u * q, 0, 0);
        i32x4 rb = __builtin_amdgcn_raw_buffer_load_b128(rsrc, g00 + gdx + 16u * q, 0, 0);
        i32x4 rc = __builtin_amdgcn_raw_buffer_load_b128(rsrc, g00 + gdy + 16u * q, 0, 0);
        i32x4 rd = __builtin_amdgcn_raw_buffer_load_b128(rsrc, g00 + gdx + gdy + 16u * q, 0, 0);
        blend(q, *reinterpret_cast<float4*>(&ra), *reinterpret_cast<float4*>(&rb), *reinterpret_cast<float4*>(&rc),
              *reinterpret_cast<float4*>(&rd), w00, w01, w10, w11);
        __builtin_amdgcn_sched_barrier(0);   // one slice in flight: this path is rare, registers matter more
      }
    } else {
      const unsigned a00 = f & ~15u, a01 = b01, a10 = b10, a11 = b11;
#if BMV_WIN_TAPBUF == 2
      // two slices in flight: the reads of slice q + 1 are issued before the blend of slice q
      float4 t00 = ld4_lds(win, a00), t01 = ld4_lds(win, a01), t10 = ld4_lds(win, a10), t11 = ld4_lds(win, a11);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 n00, n01, n10, n11;
        if (q < 3) {
          const unsigned m = (unsigned)(q + 1) << 4;
          n00 = ld4_lds(win, a00 ^ m), n01 = ld4_lds(win, a01 ^ m), n10 = ld4_lds(win, a10 ^ m), n11 = ld4_lds(win, a11 ^ m);
        }
        __builtin_amdgcn_sched_barrier(0);
        blend(q, t00, t01, t10, t11, w00, w01, w10, w11);
        __builtin_amdgcn_sched_barrier(0);
        if (q < 3) t00 = n00, t01 = n01, t10 = n10, t11 = n11;
      }
#else
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned m = (unsigned)q << 4;
        blend(q, ld4_lds(win, a00 ^ m), ld4_lds(win, a01 ^ m), ld4_lds(win, a10 ^ m), ld4_lds(win, a11 ^ m), w00, w01, w10, w11);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
  };

  // ---- 5. views in turn, NB windows in flight; per channel half
  unsigned gB[NH > 1 ? S : 1][4];
  float gW[NH > 1 ? S : 1][4];
  auto store_variance = [&](int chh) {
    if (inb && !(a.flags & 4)) {
      const float inv_s = 1.f / (float)S;
      const unsigned cstride = (unsigned)(D * hw) * 4u;
      __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
          a.out + (size_t)b * C * D * hw, 0, (int)((size_t)C * D * hw * 4), 0x00020000);
      const unsigned voff = (unsigned)((size_t)d * hw + (size_t)y * w + x) * 4u;
      unsigned soff = (unsigned)chh * 16u * cstride;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float m;
        m = acc[q].x * inv_s;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[q].x * inv_s - m * m), orsrc, (int)voff, (int)soff, kStoreAux);
        soff += cstride;
        m = acc[q].y * inv_s;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[q].y * inv_s - m * m), orsrc, (int)voff, (int)soff, kStoreAux);
        soff += cstride;
        m = acc[q].z * inv_s;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[q].z * inv_s - m * m), orsrc, (int)voff, (int)soff, kStoreAux);
        soff += cstride;
        m = acc[q].w * inv_s;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[q].w * inv_s - m * m), orsrc, (int)voff, (int)soff, kStoreAux);
        soff += cstride;
      }
    }
  };
#pragma unroll
  for (int hh = 0; hh < NH; ++hh) {
    const int chh = chh0 + hh;
    hoff = (unsigned)chh * 64u;
    if (hh > 0) {
      __syncthreads();   // every wave is done with the windows of the previous half
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = acc2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    issued = 0, landed = 0;
#pragma unroll
    for (; issued < (NB < S ? NB : S); ++issued) issue_fill(issued, issued);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      if (hh == 0) {
        geometry(s);   // under the latency of this view's fill
        if (NH > 1) {
          gB[s][0] = b00, gB[s][1] = b01, gB[s][2] = b10, gB[s][3] = b11;
          gW[s][0] = w00, gW[s][1] = w01, gW[s][2] = w10, gW[s][3] = w11;
        }
      } else {
        b00 = gB[s][0], b01 = gB[s][1], b10 = gB[s][2], b11 = gB[s][3];
        w00 = gW[s][0], w01 = gW[s][1], w10 = gW[s][2], w11 = gW[s][3];
      }
      if (s >= landed) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        landed = issued;
      }
      if (s == 0) { BMV_STAMP(5) } else if (s == 1) { BMV_STAMP(7) } else if (s == 2) { BMV_STAMP(9) }
      compute(s);
      if (s == 0) { BMV_STAMP(6) } else if (s == 1) { BMV_STAMP(8) } else if (s == 2) { BMV_STAMP(10) }
      if (issued < S && issued == s + NB) {
        __syncthreads();  // every wave is done with window s % NB
        issue_fill(issued, s % NB);
        ++issued;
      }
    }
    if (NH > 1 && !(a.flags & (32 | 64))) store_variance(chh);
  }
  if (NH > 1 && !(a.flags & (32 | 64))) return;
  const int chh = chh0;

#ifdef BMV_WIN_STAMPS
  if (a.flags & 64) {
    if (tid == 0) {
      float* o = a.out + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
      o[0] = (float)(unsigned)(st[0] & 0xffffffu), o[1] = (float)(unsigned)((st[0] >> 24) & 0xffffffu);
#pragma unroll
      for (int i = 1; i < 11; ++i) o[i + 1] = (float)(unsigned)(st[i] - st[0]);
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      o[12] = (float)(xcc & 0xf);
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      o[13] = (float)((hwid >> 8) & 0xf);   // CU id within the shader array
      o[14] = (float)((hwid >> 13) & 0x7);  // SE id
    }
    return;
  }
#endif
  if (a.flags & 32) {   // debug: window of view 0 / 1 and the depth range instead of the variance
    acc2[0] = make_float4((float)wcols[0] * S, (float)wrows[0] * S, (float)wx[0] * S, (float)wy[0] * S);
    acc2[1] = make_float4((float)wcols[1] * S, (float)wrows[1] * S, (float)wx[1] * S, (float)wy[1] * S);
    acc2[2] = make_float4(ilo * S, ihi * S, inv_depth * S, (float)(b00 >> 6) * S);
    acc2[3] = make_float4((float)(b00 & 63u) * S, (float)(b01 >> 6) * S, w00 * S, w11 * S);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ---- 6. variance: scalar channel offsets, one dword per lane and channel (a wave writes whole 128-byte lines)
  store_variance(chh);
}

}  // namespace bmv

using namespace bmv;

namespace {

struct Variant {
  int txw, tyh, dp, nb, cap;
};
// tuning table (algo 40 + i); caps in 64-byte records, multiples of 16
const Variant kVariants[] = {
    {32, 8, 1, 1, 448},   // 0
    {32, 8, 2, 1, 640},   // 1
    {32, 8, 2, 2, 640},   // 2
    {32, 4, 2, 1, 448},   // 3
    {32, 4, 2, 2, 448},   // 4
    {32, 8, 4, 1, 1024},  // 5
    {32, 8, 4, 2, 1024},  // 6
    {32, 2, 4, 2, 640},   // 7
    {16, 4, 8, 1, 448},   // 8
    {16, 4, 8, 2, 448},   // 9
    {16, 4, 16, 1, 608},  // 10
    {16, 4, 16, 2, 608},  // 11
    {16, 4, 4, 2, 320},   // 12
    {32, 2, 8, 2, 640},   // 13
    {32, 4, 1, 1, 256},   // 14
    {32, 2, 2, 2, 320},   // 15
    {16, 2, 8, 2, 256},   // 16: 256 threads x 8 planes (a wave = two planes of 32 voxels): 5 workgroups per CU
    {16, 2, 8, 1, 448},   // 17: the level-0 default (23.3 us against 27.4 for the 512-thread tile 9)
    {16, 2, 8, 1, 320},   // 18
    {32, 1, 8, 1, 448},   // 19: rows of 32 voxels (whole 128-byte lines): 26.9 us
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

template <int TXW, int TYH, int DP, int S, int NB, int NH = 1>
int launch_one(const WinArgs& a, int B, hipStream_t stream) {
  auto kern = sweep_win_kernel<TXW, TYH, DP, S, NB, NH>;
  const size_t lds = (size_t)NB * a.cap * 64;
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

template <int TXW, int TYH, int DP, int NB>
int launch_s(const WinArgs& a, int B, int S, hipStream_t stream) {
  if (a.nh == 2) {   // both channel halves in one workgroup (built for the level-0 shapes only)
    if constexpr (TXW == 16 && TYH == 4 && DP == 8) {
      if (S == 3) return launch_one<TXW, TYH, DP, 3, NB, 2>(a, B, stream);
    }
    return BMV_ERR_UNSUPPORTED;
  }
  if (S == 3) return launch_one<TXW, TYH, DP, 3, NB>(a, B, stream);
  if (S == 2) return launch_one<TXW, TYH, DP, 2, NB>(a, B, stream);
  if (S == 4) return launch_one<TXW, TYH, DP, 4, NB>(a, B, stream);
  return BMV_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int bmv_sweep_win_launch(const float* feats, const float* proj, const float* dv, int B, int S, int C, int Hs,
                                    int Ws, int D, int h, int w, float* out, const int* view_ids, int n_all, int variant,
                                    hipStream_t stream) {
  if ((C != 16 && C != 32) || S < 2 || S > 4) return BMV_ERR_UNSUPPORTED;
  if ((size_t)(view_ids ? n_all : S) * Hs * Ws * C * 4 >= ((size_t)1 << 31)) return BMV_ERR_UNSUPPORTED;
  if (Hs >= (1 << 14) - 2 || Ws >= (1 << 14) - 2) return BMV_ERR_UNSUPPORTED;  // 14-bit tap coordinates
  if ((size_t)C * D * h * w * 4 >= ((size_t)1 << 31)) return BMV_ERR_UNSUPPORTED;        // 32-bit volume offsets
  if (variant < 0) {
    // default by the source / volume scale: same resolution (cascade level 1: neighbours in a plane share texels)
    // or finer source (level 0: the planes of a pixel share texels along its epipolar line: 8 planes per workgroup, and
    // 256-thread workgroups -- a wave is two planes of 16 x 2 voxels -- so that 5 are resident per CU: 2560 workgroups
    // are two rounds of the chip where the 512-thread tile 9 needed 2.5)
    variant = (float)Ws / (float)w <= 1.5f ? 0 : 17;
  }
  if (variant >= kNumVariants) return BMV_ERR_UNSUPPORTED;
  Variant v = kVariants[variant];
  if (int c = tuning("BMV_SWEEP_WIN_CAP", 0)) {
    if (c >= 16) v.cap = (c + 15) & ~15;
  }
  WinArgs a;
  a.feats = feats, a.proj = proj, a.dv = dv, a.out = out, a.view_ids = view_ids, a.n_all = n_all;
  a.C = C, a.Hs = Hs, a.Ws = Ws, a.D = D, a.h = h, a.w = w;
  a.tiles_x = (w + v.txw - 1) / v.txw;
  const int tiles_y = (h + v.tyh - 1) / v.tyh;
  a.tyb = (tiles_y + 7) / 8;
  a.pgroups = (D + v.dp - 1) / v.dp;
  a.chalves = C / 16;
  a.nh = 1;
  // C = 32, opt-in (BMV_SWEEP_WIN_NH=2): the two channel halves in ONE workgroup where that kernel exists (tile
  // 16 x 4 x 8, S = 3), so that the prologue and the tap geometry are not done twice.  Measured: 27.6 -> 27.1 us
  // stand-alone, no difference in the frame -- those phases already hide under the fills -- so two workgroups stay
  // the default.
  if (C == 32 && S == 3 && v.txw == 16 && v.tyh == 4 && v.dp == 8 && tuning("BMV_SWEEP_WIN_NH", 0) == 2)
    a.nh = 2, a.chalves = 1;
  a.cap = v.cap;
  a.tiles_x_magic = a.tiles_x == 1 ? 0u : (unsigned)(((unsigned long long)1 << 32) / (unsigned)a.tiles_x) + 1u;
  if (a.tiles_x * a.tyb >= 65536) return BMV_ERR_UNSUPPORTED;
  a.flags = 0;
  a.flags = tuning("BMV_SWEEP_WIN_FLAGS", a.flags);
  int rc = BMV_ERR_UNSUPPORTED;
#define V(TXW, TYH, DP, NB) \
  if (v.txw == TXW && v.tyh == TYH && v.dp == DP && v.nb == NB) rc = launch_s<TXW, TYH, DP, NB>(a, B, S, stream);
  V(32, 8, 1, 1)
  V(32, 8, 2, 1)
  V(32, 8, 2, 2)
  V(32, 4, 2, 1)
  V(32, 4, 2, 2)
  V(32, 8, 4, 1)
  V(32, 8, 4, 2)
  V(32, 2, 4, 2)
  V(16, 4, 8, 1)
  V(16, 4, 8, 2)
  V(16, 4, 16, 1)
  V(16, 4, 16, 2)
  V(16, 4, 4, 2)
  V(32, 2, 8, 2)
  V(32, 4, 1, 1)
  V(32, 2, 2, 2)
  V(16, 2, 8, 2)
  V(16, 2, 8, 1)
  V(32, 1, 8, 1)
#undef V
  if (rc != BMV_OK) return rc;
  BMV_LAUNCH_END("bmv_sweep_variance_fwd(win)");
}
