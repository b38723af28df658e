// FeatureNet's last top-down step + smooth0 (feature_net.py:24-36: smooth0(bilinear_x2(p1) + lat0(c0))) on the BF16 matrix
// cores with three-piece fp32 operands (round 6).  Replaces csrc/conv.hip's fpn_smooth_kernel (fp32 16x16x4 MFMAs, the
// lateral 1x1 convolution and the bilinear term on the vector ALU: 115 us of a 0.77 ms frame, matrix pipe 43 % busy for
// three rounds) as the inference default.
//
// 1. The lateral convolution is FOLDED INTO THE SMOOTHING WEIGHTS on the host (float64): smooth0 is linear, so
//        smooth0(up(p1) + lat0(c0) + b_l) = conv3x3(up(p1); Ws) + conv3x3(c0; Ws . Wl) + sum over the taps INSIDE the image of Ws b_l
//    -- 8 more input channels for the matrix cores (32 + 8 = five octets) instead of 8 FMAs per value of a 32-channel
//    full-resolution map on the vector ALU; the bias term depends only on which image border a pixel touches (3 x 3
//    cases, a 288-byte table).  (The same kind of exact refactoring as folding eval-mode batch norm into a convolution.)
// 2. The matrix work is six v_mfma_f32_16x16x32_bf16 per product group on hi + mid + lo bf16 pieces of both operands (the
//    fp32 values exactly; what is dropped is <= 3 x 2^-24 of a product): csrc/conv_c4s.hip's arithmetic.
// 3. x-PAIRING: the 16 matrix rows are 8 output channels x 2 adjacent output columns; one k-step of 32 = the 4 input
//    columns a column pair touches x the 8 channels of an octet, i.e. exactly ONE step per (input row, filter row).
// 4. ROW WALK, input-stationary: a wave owns a strip of 62 output columns x TY rows and walks the TY + 2 input rows once per
//    octet; a staged row serves its (up to) three output rows from the SAME B operands: one ds_read_b128 per piece feeds 18
//    matrix instructions; an octet's A operands (3 filter rows x 3 pieces) stay in registers for the whole strip.
// 5. WAVES ARE INDEPENDENT: every wave stages its own rows in its own 6 KB of LDS (64 positions x 3 pieces x 2 buffers:
//    the two 16-pair tiles of a strip overlap by one pair, so 64 staged positions -- lane = position -- cover both).
//    No workgroup barrier anywhere: a wave's vector work (bilinear term, split) overlaps the matrix instructions of the
//    other waves on its SIMD instead of meeting them at a barrier.
// 6. The bilinear term costs 16 loads per row and lane: the horizontally interpolated coarse rows are kept in registers
//    across fine rows (a fine row needs coarse rows i0, i0 + 1; i0 advances by at most one per fine row).
#include <stdlib.h>

#include "bmv_common.hpp"

namespace bmv {

using f32x4p = __attribute__((ext_vector_type(4))) float;
using i32x4p = __attribute__((ext_vector_type(4))) int;
using bf16x8p = __attribute__((ext_vector_type(8))) __bf16;

struct FpnSArgs {
  const float* fine;     // c0 (B, 8, H, W)
  const float* coarse;   // p1 (B, 32, H/2, W/2)
  const int* wsplit;     // [octet 5][filter row 3][piece 3][lane 64][4]: octets 0..3 = smooth0 on up(p1), 4 = smooth0 . lat0 on c0
  const float* btab;     // (3 y cases, 3 x cases, 8): smooth0's bias + the lateral bias through the taps inside the image
  float* out;            // (B, 8, H, W), or null when `packed` is written instead
  const float* rgb;      // (B, 3, H, W): with `packed`
  float* packed;         // (B, H, W, 12) lookup records of the fused renderer: [ch 0 2 4 6 | ch 1 3 5 7 | r b | g 0]
  int B, H, W;
  float slope;
  int strips, tiles_y, ntiles;
  const void* const* table;   // deferred `rgb` (bmv_defer_pointer)
  int rgb_slot;
};

// ablation builds (scripts/ablate_fpn_s.py: timing only, wrong results): 1 no matrix instructions, 2 no tap / c0 loads,
// 4 no split + LDS writes, 8 no stores
#ifndef BMV_FPN_S_ABLATE
#define BMV_FPN_S_ABLATE 0
#endif
constexpr int kFSAblate = BMV_FPN_S_ABLATE;

constexpr int kFS_STRIP = 62;      // output columns of a strip (two 16-pair tiles overlapping by one pair)

__device__ __forceinline__ unsigned fs_pack_hi(float a, float b) {   // [bf16(a) | bf16(b) << 16] by truncation
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ unsigned fs_pack_rne(float a, float b) {  // round to nearest even (exact here: <= 8 bits left)
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fs_trunc(float v) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u);
}

template <int TY>
__global__ void __launch_bounds__(256, (TY <= 6 ? 3 : 2)) fpn_smooth_s_kernel(FpnSArgs a) {
  constexpr int NP = TY + 2;
  extern __shared__ i32x4p fs_lds[];                       // [wave 4][buffer 2][piece 3][64 positions]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wave;   // an XCD = a band of rows: the chain reads what it left in its L2
  if (tile >= a.ntiles) return;                             // (no barrier in this kernel: a wave may leave)
  const int strip = tile % a.strips, ty = (tile / a.strips) % a.tiles_y, b = tile / (a.strips * a.tiles_y);
  const int x0 = strip * kFS_STRIP, y0 = ty * TY;
  const int H = a.H, W = a.W, Hc = H >> 1, Wc = W >> 1, hw = H * W, hwc = Hc * Wc;
  i32x4p* my = fs_lds + wave * (2 * 3 * 64);
  const int n = lane & 15, kk = lane >> 4;

  // producer role: lane = staged position, fine column gx = x0 - 1 + lane.  Outside the image (zero padding of p0) the
  // horizontal weights are zero and the c0 loads go out of range: no select anywhere
  const int gx = x0 - 1 + lane;
  const bool xin = (gx >= 0) & (gx < W);
  Lerp1 lx = upsample_axis(min(max(gx, 0), W - 1), Wc, W);
  if (!xin) lx.l0 = lx.l1 = 0.f;
  const int coff0 = 4 * lx.i0, coff1 = 4 * lx.i1;                   // byte offsets inside a coarse row
  const int foff = xin ? 4 * gx : (int)0x80000000u;                 // ... inside a fine row
  // positions are parked de-interleaved (even columns | odd columns): the B reads of a lane group are then consecutive
  const int wpos = (lane & 1) * 32 + (lane >> 1);
  // consumer role: lane (n, kk) of tile q reads position 30 q + 2 n + kk = parity kk & 1, index 15 q + n + (kk >> 1)
  const int rpos = (kk & 1) * 32 + n + (kk >> 1);

  __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.coarse + (size_t)b * 32 * hwc), 0,
                                                                 (int)(4u * 32u * (unsigned)hwc), 0x00020000);
  __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fine + (size_t)b * 8 * hw), 0,
                                                                 (int)(4u * 8u * (unsigned)hw), 0x00020000);
  auto gyc = [&](int p) { return min(max(y0 - 1 + p, 0), H - 1); };
  auto rowok = [&](int p) { return (y0 - 1 + p >= 0) & (y0 - 1 + p < H); };
  // the vertical interpolation of every input row of the tile, once: lane p works out row p, a row reads it back into
  // scalar registers (the rows are wave-uniform, and gfx950 has no scalar float unit to compute them there)
  const Lerp1 lyv = upsample_axis(gyc(min(lane, NP)), Hc, H);
  auto row_lerp = [&](int p) {
    Lerp1 r;
    r.i0 = __builtin_amdgcn_readlane(lyv.i0, p), r.i1 = __builtin_amdgcn_readlane(lyv.i1, p);
    r.l0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lyv.l0), p));
    r.l1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lyv.l1), p));
    return r;
  };

  f32x4p acc[TY][2];
#pragma unroll
  for (int z = 0; z < TY; ++z) acc[z][0] = acc[z][1] = f32x4p{0.f, 0.f, 0.f, 0.f};

  // split 8 channel values of this lane's position into three bf16 pieces and park them in buffer `buf`
  auto park = [&](const float* v, int buf) {
    if (kFSAblate & 4) {
      if (v[0] == 12345.f) my[wpos] = i32x4p{1, 2, 3, 4};
      return;
    }
    i32x4p pc[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v0 = v[2 * e], v1 = v[2 * e + 1];
      pc[0][e] = (int)fs_pack_hi(v0, v1);
      const float r0 = v0 - fs_trunc(v0), r1 = v1 - fs_trunc(v1);
      pc[1][e] = (int)fs_pack_hi(r0, r1);
      pc[2][e] = (int)fs_pack_rne(r0 - fs_trunc(r0), r1 - fs_trunc(r1));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) my[(buf * 3 + q) * 64 + wpos] = pc[q];
  };
  // the matrix instructions of input row p (in buffer p & 1) for every output row it touches
  i32x4p A[3][3];       // [filter row][piece]
  auto multiply = [&](int p) {
    const i32x4p* bp = my + (p & 1) * 3 * 64 + rpos;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bf16x8p bx[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) bx[r] = __builtin_bit_cast(bf16x8p, bp[r * 64 + 15 * q]);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int zo = p - ky;               // input row p = output row zo + ky (known after unrolling)
        if (zo < 0 || zo >= TY) continue;
        if (kFSAblate & 1) {
          acc[zo][q] += __builtin_bit_cast(f32x4p, A[ky][0]) + __builtin_bit_cast(f32x4p, bx[ky]);
          continue;
        }
        // smallest terms first: (lo, hi), (mid, mid), (hi, lo), (mid, hi), (hi, mid), (hi, hi)
#pragma unroll
        for (int sum = 2; sum >= 0; --sum)
#pragma unroll
          for (int i = 0; i <= sum; ++i)
            acc[zo][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8p, A[ky][i]), bx[sum - i], acc[zo][q], 0, 0, 0);
      }
    }
  };
  const i32x4p* __restrict__ wp = reinterpret_cast<const i32x4p*>(a.wsplit) + lane;
  auto load_A = [&](int oct) {
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i / 3][i % 3] = wp[((size_t)oct * 9 + i) * 64];
  };

  // ---- octets 0..3: up(p1).  hA / hB = the horizontally interpolated coarse rows cA, cB of this octet at this lane
#pragma unroll 1
  for (int oct = 0; oct < 4; ++oct) {
    load_A(oct);
    // (scalar offset = channel + coarse row, vector offset = the lane's two columns: no address arithmetic per load)
    auto taps = [&](int crow, float (&t)[8][2]) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int so = 4 * (((oct * 8 + c) * Hc + crow) * Wc);
        if (kFSAblate & 2) {
          t[c][0] = (float)(so + lane), t[c][1] = (float)(c - lane);
          continue;
        }
        t[c][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, coff0, so, 0));
        t[c][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, coff1, so, 0));
      }
    };
    // nt[k & 1] = the taps of the coarse row that input row k brings in (its i1), requested TWO rows ahead: with one row
    // (36 matrix instructions) between request and use a wave sat ~0.8 us per row waiting for L2 (62 us for the launch)
    float hA[8], hB[8], nt[2][8][2];
    int cB;
    {   // state for input row 0, and its pieces
      const Lerp1 ly = row_lerp(0);
      float t0[8][2];
      taps(ly.i0, t0);
      taps(ly.i1, nt[0]);
      taps(row_lerp(1).i1, nt[1]);
#pragma unroll
      for (int c = 0; c < 8; ++c) hA[c] = lx.l0 * t0[c][0] + lx.l1 * t0[c][1], hB[c] = lx.l0 * nt[0][c][0] + lx.l1 * nt[0][c][1];
      cB = ly.i1;
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = ly.l0 * hA[c] + ly.l1 * hB[c];
      park(v, 0);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p + 2 < NP) taps(row_lerp(p + 2).i1, nt[p & 1]);       // (row p's taps were consumed one iteration ago)
      if (rowok(p)) multiply(p);
      if (p + 1 < NP) {
        const Lerp1 lyn = row_lerp(p + 1);
        const bool adv = lyn.i0 == cB;      // (wave-uniform) the next row's upper coarse row is this row's lower one
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          hA[c] = adv ? hB[c] : hA[c];
          hB[c] = lx.l0 * nt[(p + 1) & 1][c][0] + lx.l1 * nt[(p + 1) & 1][c][1];
        }
        cB = lyn.i1;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = lyn.l0 * hA[c] + lyn.l1 * hB[c];
        park(v, (p + 1) & 1);
      }
      // (one row's loads, matrix instructions and vector work per scheduling region)
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- octet 4: c0 through the folded weights; the finished output rows leave as soon as their last row is in
  const int cgq = kk & 1, rr = kk >> 1;
  const float* rgbp = a.packed ? deferred_load(a.table, a.rgb_slot, a.rgb) + (size_t)b * 3 * hw : nullptr;
  auto store_row = [&](int zo) {
    const int y = y0 + zo;
    if (y >= H) return;
    const int yc = y == 0 ? 0 : (y == H - 1 ? 2 : 1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int x = x0 + 30 * q + 2 * n + rr;
      if (x >= W || (q == 1 && n == 0)) continue;       // (tile 1's first pair is tile 0's last)
      if ((kFSAblate & 8) && acc[0][0][0] != 12345.f) continue;
      const int xc = x == 0 ? 0 : (x == W - 1 ? 2 : 1);
      const f32x4p bs = *reinterpret_cast<const f32x4p*>(a.btab + (yc * 3 + xc) * 8 + 4 * cgq);
      f32x4p v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = acc[zo][q][j] + bs[j];
        v[j] = fmaxf(t, 0.f) + a.slope * fminf(t, 0.f);
      }
      const size_t pix = (size_t)y * W + x;
      if (a.packed) {
        float* rec = a.packed + ((size_t)b * hw + pix) * 12;
        *reinterpret_cast<f32x4p*>(rec + 4 * cgq) = v;
        float2 c;
        if (cgq == 0)
          c.x = rgbp[pix], c.y = rgbp[2 * (size_t)hw + pix];
        else
          c.x = rgbp[(size_t)hw + pix], c.y = 0.f;
        *reinterpret_cast<float2*>(rec + 8 + 2 * cgq) = c;
      } else {
        float* o = a.out + ((size_t)b * 8 + 4 * cgq) * hw + pix;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[(size_t)j * hw] = v[j];
      }
    }
  };
  {
    load_A(4);
    auto fetch = [&](int p, float (&t)[8]) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        t[c] = (kFSAblate & 2) ? (float)(c + lane + p)
                               : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frs, foff, 4 * ((c * H + gyc(p)) * W), 0));
    };
    float nx[2][8];
    fetch(0, nx[0]);
    fetch(1, nx[1]);
    park(nx[0], 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p + 2 < NP) fetch(p + 2, nx[p & 1]);       // (two rows ahead, as above)
      if (rowok(p)) multiply(p);
      if (p >= 2) store_row(p - 2);          // output row p - 2 has its last contribution
      if (p + 1 < NP) park(nx[(p + 1) & 1], (p + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// FeatureNet's first block (feature_net.py:8-10: conv0 = ConvBnReLU(3, 8) + ConvBnReLU(8, 8), batch norm folded) in the
// same form: independent waves walk strips of 62 columns; the producer computes the FIRST layer (3 input channels: 27
// FMAs per value, weights wave-uniform) at its position from a rolling window of image rows kept in registers, splits the
// 8 values into three bf16 pieces and parks them; the second layer is ONE octet on the bf16 matrix cores (x-paired rows,
// 18 matrix instructions per staged row and tile).  Replaces csrc/conv.hip's conv0_fused_kernel (fp32 MFMAs, image tile
// in LDS, two barriers per 4-channel chunk: 37 us at the head of the frame, where nothing runs beside it).
// ---------------------------------------------------------------------------------------------------------------
struct Conv0SArgs {
  const float* in;       // (B, 3, H, W)
  const float* w0b0;     // first layer, batch norm folded: [channel 8][28] = the 27 weights (ci, ky, kx) + the bias
  const int* wsplit;     // second layer: [filter row 3][piece 3][lane 64][4] (one octet, x-paired rows as fpn_smooth_s)
  const float* bias;     // (8)
  float* out;            // (B, 8, H, W)
  int B, H, W;
  float slope1;
  int strips, tiles_y, ntiles;
  const void* const* table;   // deferred `in` (bmv_defer_pointer)
  int in_slot;
};

template <int TY>
__global__ void __launch_bounds__(256, 2) conv0_s_kernel(Conv0SArgs a) {
  constexpr int NP = TY + 2;
  extern __shared__ i32x4p fs_lds[];                       // [wave 4][buffer 2][piece 3][64 positions]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wave;   // an XCD = a band of rows: the chain reads what it left in its L2
  if (tile >= a.ntiles) return;
  const int strip = tile % a.strips, ty = (tile / a.strips) % a.tiles_y, b = tile / (a.strips * a.tiles_y);
  const int x0 = strip * kFS_STRIP, y0 = ty * TY;
  const int H = a.H, W = a.W, hw = H * W;
  i32x4p* my = fs_lds + wave * (2 * 3 * 64 + 56);
  // the first layer's 8 x 28 weights in this wave's LDS (wave-uniform 16-byte reads broadcast: through a pointer hipcc
  // fetched them with 442 uniform VECTOR loads per strip -- it cannot prove nobody stores there --, as kernel arguments
  // it parked all 216 in SGPRs spilled to lanes: one v_readlane per FMA)
  f32x4p* wl = reinterpret_cast<f32x4p*>(my + 2 * 3 * 64);
  if (lane < 56) wl[lane] = reinterpret_cast<const f32x4p*>(a.w0b0)[lane];
  const int n = lane & 15, kk = lane >> 4;
  const int gx = x0 - 1 + lane;                             // column of this lane's intermediate position
  const bool xin = (gx >= 0) & (gx < W);
  const int wpos = (lane & 1) * 32 + (lane >> 1), rpos = (kk & 1) * 32 + n + (kk >> 1);
  // the three image columns under the position (zero padding of the FIRST layer: out of range -> 0)
  int voff[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) voff[dx] = ((gx + dx - 1 >= 0) & (gx + dx - 1 < W)) ? 4 * (gx + dx - 1) : (int)0x80000000u;
  const float* in = deferred_load(a.table, a.in_slot, a.in);
  __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)b * 3 * hw), 0,
                                                                 (int)(4u * 3u * (unsigned)hw), 0x00020000);
  // image row q of the window is image row y0 - 2 + q; intermediate row p (image row y0 - 1 + p) reads q = p, p + 1, p + 2
  auto rowok = [&](int p) { return (y0 - 1 + p >= 0) & (y0 - 1 + p < H); };
  auto load_row = [&](int q, float (&t)[3][3]) {
    const int gy = y0 - 2 + q;
    const int dead = ((gy >= 0) & (gy < H)) ? 0 : (int)0x80000000u;
    const int gyc = min(max(gy, 0), H - 1);
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        t[ci][dx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, voff[dx] | dead, 4 * ((ci * H + gyc) * W), 0));
  };
  // zero padding of the SECOND layer: an intermediate position outside the image is 0, not relu(b0 + ...): its bias is
  // -1e30 there (ReLU then gives 0; rows outside the image are skipped altogether)
  const float xmask = xin ? 0.f : -1e30f;

  f32x4p acc[TY][2];
#pragma unroll
  for (int z = 0; z < TY; ++z) acc[z][0] = acc[z][1] = f32x4p{0.f, 0.f, 0.f, 0.f};
  i32x4p A[3][3];
  const i32x4p* __restrict__ wp = reinterpret_cast<const i32x4p*>(a.wsplit) + lane;
#pragma unroll
  for (int i = 0; i < 9; ++i) A[i / 3][i % 3] = wp[(size_t)i * 64];

  auto park = [&](const float* v, int buf) {
    i32x4p pc[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v0 = v[2 * e], v1 = v[2 * e + 1];
      pc[0][e] = (int)fs_pack_hi(v0, v1);
      const float r0 = v0 - fs_trunc(v0), r1 = v1 - fs_trunc(v1);
      pc[1][e] = (int)fs_pack_hi(r0, r1);
      pc[2][e] = (int)fs_pack_rne(r0 - fs_trunc(r0), r1 - fs_trunc(r1));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) my[(buf * 3 + q) * 64 + wpos] = pc[q];
  };
  auto multiply = [&](int p) {
    const i32x4p* bp = my + (p & 1) * 3 * 64 + rpos;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bf16x8p bx[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) bx[r] = __builtin_bit_cast(bf16x8p, bp[r * 64 + 15 * q]);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int zo = p - ky;
        if (zo < 0 || zo >= TY) continue;
#pragma unroll
        for (int sum = 2; sum >= 0; --sum)
#pragma unroll
          for (int i = 0; i <= sum; ++i)
            acc[zo][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8p, A[ky][i]), bx[sum - i], acc[zo][q], 0, 0, 0);
      }
    }
  };
  // first layer at this lane's position from window rows r0, r1, r2 (filter rows 0, 1, 2)
  auto first_layer = [&](const float (&r0)[3][3], const float (&r1)[3][3], const float (&r2)[3][3], float (&v)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {          // (channel by channel: 28 wave-uniform values = seven broadcast LDS reads)
      float w[28];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const f32x4p t = wl[c * 7 + i];
        w[4 * i] = t[0], w[4 * i + 1] = t[1], w[4 * i + 2] = t[2], w[4 * i + 3] = t[3];
      }
      float s = w[27] + xmask;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          s = fmaf(w[ci * 9 + kx], r0[ci][kx], s);
          s = fmaf(w[ci * 9 + 3 + kx], r1[ci][kx], s);
          s = fmaf(w[ci * 9 + 6 + kx], r2[ci][kx], s);
        }
      v[c] = s;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
  };
  const int cgq = kk & 1, rr = kk >> 1;
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[4 * cgq + j];
  auto store_row = [&](int zo) {
    const int y = y0 + zo;
    if (y >= H) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int x = x0 + 30 * q + 2 * n + rr;
      if (x >= W || (q == 1 && n == 0)) continue;
      float* o = a.out + ((size_t)b * 8 + 4 * cgq) * hw + (size_t)y * W + x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = acc[zo][q][j] + bs[j];
        o[(size_t)j * hw] = fmaxf(t, 0.f) + a.slope1 * fminf(t, 0.f);
      }
    }
  };

  // window ring of 4 image rows: intermediate row p + 1 is produced in iteration p from rows p + 1, p + 2, p + 3; row
  // p + 4 is requested at the top of iteration p into the slot of row p (last read when row p was produced, one
  // iteration ago): one iteration = 216 FMAs + 36 matrix instructions ahead of its use
  float R[4][3][3];
  load_row(0, R[0]), load_row(1, R[1]), load_row(2, R[2]), load_row(3, R[3]);
  {
    float v[8];
    first_layer(R[0], R[1], R[2], v);
    park(v, 0);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (p + 1 < NP) load_row(p + 4, R[p % 4]);
    if (rowok(p)) multiply(p);
    if (p >= 2) store_row(p - 2);
    if (p + 1 < NP) {
      float v[8];
      first_layer(R[(p + 1) % 4], R[(p + 2) % 4], R[(p + 3) % 4], v);
      park(v, (p + 1) & 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace bmv

using namespace bmv;

extern "C" {

// int32 words of bmv_fpn_smooth_s_fwd's split weights: [octet 5][filter row 3][piece 3][lane 64][4]
int bmv_fpn_smooth_s_wsplit_ints(void) { return 5 * 9 * 64 * 4; }

// smooth0(bilinear_x2(coarse, align_corners) + lat0(fine)) with the lateral convolution folded into the weights, on the
// bf16 matrix cores with three-piece fp32 operands.  fine (B,8,H,W), coarse (B,32,H/2,W/2); wsplit / btab:
// boostmvsnerfs_amd/convnet.py pack_fpn_smooth_s.  out (B,8,H,W), or rgb (B,3,H,W) + packed_out (B,H,W,12) = the fused
// renderer's lookup records.
int bmv_fpn_smooth_s_fwd(const float* fine, const float* coarse, const int* wsplit, const float* btab, float* out,
                         const float* rgb, float* packed_out, int B, int H, int W, float act_slope, bmv_stream_t stream) {
  BMV_REQUIRE(fine && coarse && wsplit && btab && (out || packed_out), "fpn_smooth_s: null pointer");
  BMV_REQUIRE(!packed_out || rgb, "fpn_smooth_s: lookup records need the source colours");
  BMV_REQUIRE(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "fpn_smooth_s: bad shape (H, W even, >= 2)");
  BMV_REQUIRE((size_t)B * H * W * 12 * 4 < ((size_t)1 << 40), "fpn_smooth_s: shape");
  FpnSArgs a;
  a.fine = fine, a.coarse = coarse, a.wsplit = wsplit, a.btab = btab, a.out = out, a.rgb = rgb, a.packed = packed_out;
  a.B = B, a.H = H, a.W = W, a.slope = act_slope;
  const DeferredPtr drgb = rgb ? deferred_for(rgb) : DeferredPtr{};
  a.table = drgb.table, a.rgb_slot = drgb.slot;
  a.strips = (W + kFS_STRIP - 1) / kFS_STRIP;
  // rows per wave TY: a wave walks TY + 2 input rows per octet.  The kernel is bound by instruction issue per SIMD, so a
  // launch costs what its most loaded SIMD walks: ceil(waves / 1024) waves x (TY + 2) rows, as long as that many waves
  // are resident (TY 6 fits 168 registers = 3 per SIMD, the larger tiles 2); the cheapest candidate wins
  // (512 x 640 x 3 views: TY 9 = 1881 waves, 2 x 11 rows).  BMV_FPN_S_ROWS forces one of 6, 8, 9, 10, 12.
  int rows = bmv::tuning("BMV_FPN_S_ROWS", 0);
  const int cand[5] = {12, 10, 9, 8, 6};
  if (rows != 12 && rows != 10 && rows != 9 && rows != 8 && rows != 6) {
    long best = -1;
    for (int i = 0; i < 5; ++i) {
      const long waves = (long)a.strips * ((H + cand[i] - 1) / cand[i]) * B, occ = cand[i] <= 6 ? 3 : 2;
      const long per_simd = (waves + 1023) / 1024;
      const long cost = (per_simd <= occ ? per_simd : ((waves + 1024 * occ - 1) / (1024 * occ)) * occ) * (cand[i] + 2);
      if (best < 0 || cost < best) best = cost, rows = cand[i];
    }
  }
  a.tiles_y = (H + rows - 1) / rows;
  a.ntiles = a.strips * a.tiles_y * B;
  const size_t lds = (size_t)4 * 2 * 3 * 64 * sizeof(i32x4p);
  hipStream_t st = as_stream(stream);
  const dim3 grid((a.ntiles + 3) / 4), block(256);
  if (rows == 12) hipLaunchKernelGGL(fpn_smooth_s_kernel<12>, grid, block, lds, st, a);
  else if (rows == 10) hipLaunchKernelGGL(fpn_smooth_s_kernel<10>, grid, block, lds, st, a);
  else if (rows == 9) hipLaunchKernelGGL(fpn_smooth_s_kernel<9>, grid, block, lds, st, a);
  else if (rows == 8) hipLaunchKernelGGL(fpn_smooth_s_kernel<8>, grid, block, lds, st, a);
  else hipLaunchKernelGGL(fpn_smooth_s_kernel<6>, grid, block, lds, st, a);
  BMV_LAUNCH_END("bmv_fpn_smooth_s_fwd");
}

// int32 words of bmv_conv0_s_fwd's split second-layer weights: [filter row 3][piece 3][lane 64][4]
int bmv_conv0_s_wsplit_ints(void) { return 9 * 64 * 4; }

// relu(conv3x3(relu(conv3x3(in (B,3,H,W); w0 (8,3,3,3)) + b0); second layer) + bias) -> out (B,8,H,W): FeatureNet's first block
// with the second layer on the bf16 matrix cores (three-piece fp32 operands); wsplit: convnet.py pack_conv0_s.
// w0b0: the first layer as [channel 8][28] floats = 27 weights (ci, ky, kx) + the bias, batch norm folded
int bmv_conv0_s_fwd(const float* in, const float* w0b0, const int* wsplit, const float* bias, float* out,
                    int B, int H, int W, float slope1, bmv_stream_t stream) {
  BMV_REQUIRE(in && w0b0 && wsplit && bias && out, "conv0_s: null pointer");
  BMV_REQUIRE(B > 0 && H > 0 && W > 0 && (size_t)3 * H * W * 4 < ((size_t)1 << 31), "conv0_s: bad shape");
  Conv0SArgs a;
  a.in = in, a.w0b0 = w0b0, a.wsplit = wsplit, a.bias = bias, a.out = out, a.B = B, a.H = H, a.W = W, a.slope1 = slope1;
  const DeferredPtr din = deferred_for(in);
  a.table = din.table, a.in_slot = din.slot;
  a.strips = (W + kFS_STRIP - 1) / kFS_STRIP;
  int rows = bmv::tuning("BMV_CONV0_S_ROWS", 0);
  const int cand[3] = {12, 10, 9};
  if (rows != 12 && rows != 10 && rows != 9) {
    long best = -1;
    for (int i = 0; i < 3; ++i) {
      const long waves = (long)a.strips * ((H + cand[i] - 1) / cand[i]) * B, occ = 2;
      const long per_simd = (waves + 1023) / 1024;
      const long cost = (per_simd <= occ ? per_simd : ((waves + 1024 * occ - 1) / (1024 * occ)) * occ) * (cand[i] + 2);
      if (best < 0 || cost < best) best = cost, rows = cand[i];
    }
  }
  a.tiles_y = (H + rows - 1) / rows;
  a.ntiles = a.strips * a.tiles_y * B;
  const size_t lds = (size_t)4 * (2 * 3 * 64 + 56) * sizeof(i32x4p);
  hipStream_t st = as_stream(stream);
  const dim3 grid((a.ntiles + 3) / 4), block(256);
  if (rows == 12) hipLaunchKernelGGL(conv0_s_kernel<12>, grid, block, lds, st, a);
  else if (rows == 10) hipLaunchKernelGGL(conv0_s_kernel<10>, grid, block, lds, st, a);
  else hipLaunchKernelGGL(conv0_s_kernel<9>, grid, block, lds, st, a);
  BMV_LAUNCH_END("bmv_conv0_s_fwd");
}

}  // extern "C"
