// Convolution engine for the 2-D feature pyramid and the 3-D cost regularisers (SURVEY.md section 8f,
// ranks 1-2: the steps on either side of the plane sweep).  Inference only; fp32 throughout.
//
//   conv block of the reference          lib/networks/enerf/utils.py:10-33  (ConvBnReLU / ConvBnReLU3D)
//   FeatureNet                           lib/networks/enerf/feature_net.py:4-36
//   MinCostRegNet / CostRegNet           lib/networks/enerf/cost_reg_net.py:4-86
//
// One implicit-GEMM kernel on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain): the batch-norm scale is folded
// into the weights on the host, its shift / the conv bias, the ReLU and the U-Net skip add are the epilogue,
// so a conv block is ONE launch and activations are written once.
//
// Tried and rejected (measured): staging 2-4 k-steps (8-16 channels) per barrier pair instead of one -- slower for
// every layer class (small tilings 733 -> 778 us over all layers, big 2-D tilings 86 -> 111 us for 32 -> 8): the
// larger stage costs more in resident blocks per CU than it saves in exposed load latency.
//
// Orientation (as the MLP of render.hip): weights are the A operand (rows = 16 output channels), the
// activations are the B operand (columns = 16 consecutive output x), one k-step = 4 input channels of one
// filter tap.  Activations stay in the reference's planar layout (B,C,D,H,W): an input tile with its halo is
// staged in LDS as [channel][z][y][x] with a plane stride chosen so that the 2 x 16 lanes of one LDS cycle
// (2 channels x 16 x) fall on 32 distinct banks; the per-tap operand is `ds_read_b32 base + immediate`.
// Tile loads are raw buffer loads whose per-thread slot offsets are computed once per block; out-of-image
// slots (zero padding) and padded input channels use an out-of-range offset and come back as 0.
//
// Row pairing (Cout <= 8, stride 1): the 16 MFMA rows hold the 8 output channels of TWO adjacent output rows
// y, y+1.  Input row y+j (j = 0..K) meets filter row j for output y and filter row j-1 for output y+1, so K+1
// MFMAs per (kz, kx) produce two output rows: (K+1)/(2K) = 2/3 of the MFMAs of padding 8 channels to 16.
#include <stdlib.h>

#include "bmv_common.hpp"

namespace bmv {

using f32x4 = __attribute__((ext_vector_type(4))) float;


struct ConvArgs {
  const float* in;     // (B, Cin, D, H, W)
  const float* wpack;  // [cout tile][cin chunk of 4][tap][4][16]
  const float* bias;   // (16 * cout tiles)
  const float* skip;   // nullable, layout of out
  float* out;          // (B, Cout, Do, Ho, Wo) or channel-last (B, Do, Ho, Wo, Cout)
  int B, Cin, D, H, W, Cout, Do, Ho, Wo;
  float slope;  // activation: v > 0 ? v : slope * v  (1 = none, 0 = ReLU, 0.01 = InPlaceABN's leaky ReLU)
  int channels_last;   // 0 planar, 1 channel-last, 2 = the renderer's volume records (bmv_conv_heads_fwd),
                       // 3 = quad-planar (B, Cout/4, Do, Ho, Wo, 4): the plane sweep's source layout (csrc/sweep_quad.hip)
  float* out2;         // mode 2: channel 8 (the depth logits), planar (B, Do, Ho, Wo)
  int band_map = 0;    // 1: the tile rows are cut into 8 bands, workgroup id % 8 = band (the plane sweep's XCD bands: the
                       // map it reads is then produced in the L2 it is read from); 0: XCD-contiguous runs of tiles
  const float* w2;     // second stage (TOP): packed 1x1 weights [2 tiles][8 chunks][4][16] and bias (32)
  const float* b2;
};
// QIN (conv_mfma_kernel): `in` is QUAD RECORDS (B, Cin/4, D, H, W, 4) -- the regularisers' first layer writes them
// (csrc/conv_c4.hip, mode 8): a slot's 4-channel chunk is one 16-byte load instead of four dword loads

// MAP: 0 = 2-D (row groups along y), 1 = 3-D with the block's 4/NCT row groups along z, 2 = 3-D along y
template <int KD, int K, int S, int NCT, int R, int MAP, bool PAIR>
struct ConvTile {
  static constexpr int NRG = 4 / NCT;  // row groups (waves per cout tile)
  static constexpr int TZ = (MAP == 1) ? NRG : 1;
  static constexpr int TY = (MAP == 1) ? R : NRG * R;
  static constexpr int TZH = (TZ - 1) * S + KD;
  static constexpr int TYH = (TY - 1) * S + K;
  static constexpr int RS = 15 * S + K;  // 16 outputs along x + halo
  static constexpr int SLOTS = TZH * TYH * RS;
  // stride 1: plane stride = 16 (mod 32); stride 2: odd  -> conflict-free operand reads
  static constexpr int PS = (S == 1) ? ((SLOTS + 15) / 32 * 32 + 16) : (SLOTS | 1);
  static constexpr int NSLOT = (SLOTS + 255) / 256;
  static constexpr int TAPS = PAIR ? KD * (K + 1) * K : KD * K * K;
  static constexpr int ROWBASE = (MAP == 1) ? S * TYH * RS : R * S * RS;  // LDS offset of one row group
  static constexpr int NACC = PAIR ? R / 2 : R;
  static_assert(!PAIR || (S == 1 && NCT == 1 && R % 2 == 0), "row pairing: stride 1, one cout tile, even rows");
};

// Occupancy the register allocator must leave room for (workgroups per CU = waves per SIMD).  The default allocation
// of the full-resolution layers lands a few registers above a step of the occupancy table, and their grids are small
// multiples of the resident-workgroup count, so one more resident workgroup removes a mostly idle second round:
//   3-D row-paired 16|32 -> 8 (conv0, 112 VGPRs, 1280 workgroups at level 1): 5 per CU = one round instead of 1.25
//   3-D 8 -> 9 heads (112 VGPRs, 2560 / 1280 workgroups): 5 per CU = 2 / 1 rounds instead of 2.5 / 1.25
//   2-D row-paired 32 -> 8 (smooth0, 68 VGPRs, 1920 workgroups): 8 per CU = one round instead of 1.07
#ifndef BMV_CONV_WPE_TUNED
#define BMV_CONV_WPE_TUNED 1
#endif
constexpr int conv_wpe(int KD, int K, int S, int NCT, int R, int MAP, bool PAIR) {
  if (!BMV_CONV_WPE_TUNED) return 1;
  if (KD == 3 && K == 3 && S == 1 && NCT == 1 && MAP == 1 && ((PAIR && R == 8) || (!PAIR && R == 4))) return 5;
  if (KD == 1 && K == 3 && S == 1 && NCT == 1 && MAP == 0 && PAIR && R == 8) return 8;
  return 1;
}

// TOP (FeatureNet's conv2.1 + toplayer, feature_net.py:14-16): a 1x1 convolution 32 -> 32 on this layer's output as a
// second stage of the same workgroup -- the activated tile goes to LDS, each wave finishes two rows x both output
// tiles with 32 MFMAs and writes the channel-last map the level-0 sweep reads; the 32-channel intermediate is never
// written and one launch disappears from the head of the frame.  Needs both output tiles in the workgroup (NCT = 2).
template <int KD, int K, int S, int NCT, int R, int MAP, bool PAIR, bool TOP = false, bool QIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(conv_wpe(KD, K, S, NCT, R, MAP, PAIR), 8)))
void conv_mfma_kernel(ConvArgs a) {
  using T = ConvTile<KD, K, S, NCT, R, MAP, PAIR>;
  __shared__ float lds[4 * T::PS];
  constexpr int TPS = T::TY * 16 + 16;   // plane stride of the second stage's tile: 16 (mod 32) for TY = 8
  __shared__ float top[TOP ? 32 * TPS : 1];
  static_assert(!TOP || (NCT == 2 && MAP == 0 && !PAIR && T::TY == 8 && KD == 1), "second stage: 32 channels x 8 rows per block");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = wave % NCT, rg = wave / NCT;
  const int ntx = (a.Wo + 15) / 16, nty = (a.Ho + T::TY - 1) / T::TY, ntz = (a.Do + T::TZ - 1) / T::TZ;
  // consecutive workgroup ids go round-robin over the 8 XCDs: give each XCD a contiguous run of tiles so that the
  // halo rows / planes neighbouring tiles share are served by one L2
  int tx, ty, tz, b;
  if (a.band_map) {
    // ... or, for a map the plane sweep reads next (quad-planar output): XCD k = rows band k of EVERY batch item, the
    // bands of csrc/sweep_quad.hip (blockIdx.x % 8 there too)
    const int band = blockIdx.x & 7, per = (nty + 7) >> 3;
    int j = blockIdx.x >> 3;
    tx = j % ntx, j /= ntx;
    ty = band * per + j % per, j /= per;
    tz = j % ntz, b = j / ntz;
    if (ty >= nty) return;   // whole workgroup, before any barrier
  } else {
    int bid = xcd_contiguous(blockIdx.x, gridDim.x);
    tx = bid % ntx, bid /= ntx;
    ty = bid % nty, bid /= nty;
    tz = bid % ntz, b = bid / ntz;
  }
  const int x0 = tx * 16, y0 = ty * T::TY, z0 = tz * T::TZ;
  const int ix0 = x0 * S - K / 2, iy0 = y0 * S - K / 2, iz0 = z0 * S - KD / 2;
  const int plane = a.D * a.H * a.W;

  // tile slots of this thread (same for every channel): byte offset in the channel plane, or out of range
  unsigned goff[T::NSLOT];
#pragma unroll
  for (int j = 0; j < T::NSLOT; ++j) {
    const int slot = tid + 256 * j;
    const int sx = slot % T::RS, t = slot / T::RS, sy = t % T::TYH, sz = t / T::TYH;
    const int gx = ix0 + sx, gy = iy0 + sy, gz = iz0 + sz;
    const bool ok = (slot < T::SLOTS) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H) & (gz >= 0) & (gz < a.D);
    goff[j] = ok ? (QIN ? 16u : 4u) * (unsigned)((gz * a.H + gy) * a.W + gx) : 0x80000000u;
  }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (size_t)b * a.Cin * plane), 0, (int)(4u * (unsigned)(a.Cin * plane)), 0x00020000);

  const int nchunk = (a.Cin + 3) / 4;
  const int cot = blockIdx.y * NCT + ct;  // cout tile of this wave
  const float* wp = a.wpack + (size_t)cot * nchunk * (T::TAPS * 64) + lane;

  f32x4 acc[T::NACC];
#pragma unroll
  for (int r = 0; r < T::NACC; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  float pre[4][T::NSLOT];
  auto load_tile = [&](int chunk) {
    if constexpr (QIN) {
      const unsigned cb = 16u * (unsigned)(chunk * plane);
#pragma unroll
      for (int j = 0; j < T::NSLOT; ++j) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[j] + cb, 0, 0));
        pre[0][j] = v[0], pre[1][j] = v[1], pre[2][j] = v[2], pre[3][j] = v[3];
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned cb = 4u * (unsigned)((chunk * 4 + c) * plane);
#pragma unroll
      for (int j = 0; j < T::NSLOT; ++j)
        pre[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, goff[j] + cb, 0, 0));
    }
  };
  load_tile(0);

  float wnext[T::TAPS];
#pragma unroll
  for (int t = 0; t < T::TAPS; ++t) wnext[t] = wp[t * 64];

  const float* ap = lds + (lane >> 4) * T::PS + rg * T::ROWBASE + (lane & 15) * S;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();  // every wave is done with the previous tile
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < T::NSLOT; ++j)
        if ((j + 1) * 256 <= T::SLOTS || tid + 256 * j < T::SLOTS) lds[c * T::PS + tid + 256 * j] = pre[c][j];
    // Row-paired kernels (36 taps) read the k-step's weights straight from L2 / L1 -- issued before the barrier,
    // consumed after it: a register double-buffer for them costs 36 VGPRs = one resident workgroup per CU (measured
    // +1 % frames/s without it).  The 27-tap kernels keep the double-buffer (measured -0.3 % without).
    float wv[T::TAPS];
    if constexpr (PAIR) {
#pragma unroll
      for (int t = 0; t < T::TAPS; ++t) wv[t] = wp[(size_t)chunk * (T::TAPS * 64) + t * 64];
    } else {
#pragma unroll
      for (int t = 0; t < T::TAPS; ++t) wv[t] = wnext[t];
    }
    __syncthreads();
    if (chunk + 1 < nchunk) {  // the next tile (and weights) are in flight during the MFMAs below
      load_tile(chunk + 1);
      if constexpr (!PAIR) {
#pragma unroll
        for (int t = 0; t < T::TAPS; ++t) wnext[t] = wp[(size_t)(chunk + 1) * (T::TAPS * 64) + t * 64];
      }
    }
    if constexpr (PAIR) {
#pragma unroll
      for (int kd = 0; kd < KD; ++kd)
#pragma unroll
        for (int j = 0; j <= K; ++j)
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            const float w = wv[(kd * (K + 1) + j) * K + kw];
#pragma unroll
            for (int p = 0; p < R / 2; ++p)
              acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, ap[(kd * T::TYH + 2 * p + j) * T::RS + kw], acc[p], 0,
                                                            0, 0);
            // keep the accumulators rotating (the scheduler otherwise chains 3-4 MFMAs on one accumulator, each
            // waiting out the 40-cycle dependent latency); memory instructions may still move across
            __builtin_amdgcn_sched_barrier(0x0090);
          }
    } else {
#pragma unroll
      for (int kd = 0; kd < KD; ++kd)
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            const float w = wv[(kd * K + kh) * K + kw];
#pragma unroll
            for (int r = 0; r < R; ++r)
              acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, ap[(kd * T::TYH + r * S + kh) * T::RS + kw], acc[r],
                                                            0, 0, 0);
            if constexpr (R <= 4) __builtin_amdgcn_sched_barrier(0x0090);   // as above: few accumulators, keep them rotating
          }
    }
  }

  // epilogue: lane = output x (lane & 15), registers = 4 consecutive output channels
  const int x = x0 + (lane & 15);
  const int g = lane >> 4;
  const int co0 = PAIR ? 4 * (g & 1) : cot * 16 + 4 * g;
  const int z = (MAP == 1) ? z0 + rg : z0;
  if constexpr (TOP) {
    // stage 1 -> LDS: act(conv + bias), [32 channels][8 rows][16 x]
#pragma unroll
    for (int r = 0; r < T::NACC; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[r][j] + a.bias[co0 + j];
        v = fmaxf(v, 0.f) + a.slope * fminf(v, 0.f);
        top[(co0 + j) * TPS + (rg * R + r) * 16 + (lane & 15)] = v;
      }
    __syncthreads();
    // stage 2: wave w finishes rows 2w, 2w+1 for both output tiles; 8 k-steps of 4 channels each
    const float* w2 = a.w2 + lane;
    const float* bp = top + g * TPS + (lane & 15);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      float wv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) wv[c] = w2[(t2 * 8 + c) * 64];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = 2 * wave + rr;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) o = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c], bp[4 * c * TPS + row * 16], o, 0, 0, 0);
        const int y = y0 + row, oc = t2 * 16 + 4 * g;
        if (x < a.Wo && y < a.Ho) {
          f32x4 q = {o[0] + a.b2[oc], o[1] + a.b2[oc + 1], o[2] + a.b2[oc + 2], o[3] + a.b2[oc + 3]};
          if (a.channels_last == 3)   // quad-planar: (B, 8, Ho, Wo, 4)
            *reinterpret_cast<f32x4*>(a.out + ((((size_t)b * 8 + (oc >> 2)) * a.Ho + y) * a.Wo + x) * 4) = q;
          else
            *reinterpret_cast<f32x4*>(a.out + (((size_t)b * a.Ho + y) * a.Wo + x) * 32 + oc) = q;
        }
      }
    }
    return;
  }
  if (x >= a.Wo || co0 >= a.Cout || z >= a.Do) return;
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[co0 + j];
  const size_t cs = (size_t)a.Do * a.Ho * a.Wo;
  const int ybase = (MAP == 1) ? y0 : y0 + rg * R;
#pragma unroll
  for (int r = 0; r < T::NACC; ++r) {
    const int y = PAIR ? ybase + 2 * r + (g >> 1) : ybase + r;
    if (y >= a.Ho) continue;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = acc[r][j] + bs[j];
      v[j] = fmaxf(v[j], 0.f) + a.slope * fminf(v[j], 0.f);
    }
    if (a.channels_last == 2) {
      // 8 feature channels as one 32-byte record per voxel (MFMA rows 0-7: the caller packed the weights with the
      // output channels in the renderer's even | odd order), row 8 -- the depth logit -- planar
      const size_t vox = (((size_t)b * a.Do + z) * a.Ho + y) * a.Wo + x;
      if (co0 < 8) {
        f32x4 q = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(a.out + vox * 8 + co0) = q;
      } else {
        a.out2[vox] = v[0];
      }
    } else if (a.channels_last == 3) {
      // quad-planar (Cout % 4 == 0, no skip: checked by the launcher): 4 consecutive channels of a pixel = one 16-byte store,
      // neighbouring lanes (x) write neighbouring records
      const size_t o = (((((size_t)b * (a.Cout >> 2) + (co0 >> 2)) * a.Do + z) * a.Ho + y) * a.Wo + x) * 4;
      f32x4 q = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(a.out + o) = q;
    } else if (a.channels_last) {
      const size_t o = ((((size_t)b * a.Do + z) * a.Ho + y) * a.Wo + x) * a.Cout + co0;
      if ((a.Cout & 3) == 0) {
        f32x4 q = {v[0], v[1], v[2], v[3]};
        if (a.skip) q += *reinterpret_cast<const f32x4*>(a.skip + o);
        *reinterpret_cast<f32x4*>(a.out + o) = q;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (co0 + j < a.Cout) a.out[o + j] = v[j] + (a.skip ? a.skip[o + j] : 0.f);
      }
    } else {
      const size_t o = ((((size_t)b * a.Cout + co0) * a.Do + z) * a.Ho + y) * a.Wo + x;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (co0 + j < a.Cout) a.out[o + j * cs] = v[j] + (a.skip ? a.skip[o + j * cs] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1) (cost_reg_net.py:23-41 conv7/9/11): out = 2 x in.
// Output o = 2 m + p gets  p = 0: in[m] w[1];  p = 1: in[m + 1] w[0] + in[m] w[2]  per axis, so the 8 output
// parities of an input position are 1+2+2+4+2+4+4+8 = 27 taps: no MFMA on stuffed zeros.  A wave owns R input
// rows x 16 input x and keeps the 8 parity accumulators of each; the two x parities of a lane are adjacent in
// the output row and leave as one 8-byte store.  The U-Net skip add is the epilogue.
// MAP as above (1: row groups along z, 2: along y).
// ---------------------------------------------------------------------------------------------------
template <int NCT, int R, int MAP>
struct ConvTTile {
  static constexpr int NRG = 4 / NCT;
  static constexpr int TZ = (MAP == 1) ? NRG : 1;
  static constexpr int TY = (MAP == 1) ? R : NRG * R;
  static constexpr int TZH = TZ + 1, TYH = TY + 1, RS = 17;
  static constexpr int SLOTS = TZH * TYH * RS;
  static constexpr int PS = (SLOTS + 15) / 32 * 32 + 16;
  static constexpr int NSLOT = (SLOTS + 255) / 256;
  static constexpr int ROWBASE = (MAP == 1) ? TYH * RS : R * RS;
};

// R <= 2: the allocator takes 184 registers (2 workgroups per CU; conv11 at level 1 is 640 workgroups = 1.25 rounds)
// but is as happy with 140 (3 per CU, one round, no spills) when asked
template <int NCT, int R, int MAP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((BMV_CONV_WPE_TUNED && R <= 2) ? 3 : 1, 8)))
void convT3d_mfma_kernel(ConvArgs a) {
  using T = ConvTTile<NCT, R, MAP>;
  __shared__ float lds[4 * T::PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = wave % NCT, rg = wave / NCT;
  const int ntx = (a.W + 15) / 16, nty = (a.H + T::TY - 1) / T::TY, ntz = (a.D + T::TZ - 1) / T::TZ;
  // consecutive workgroup ids go round-robin over the 8 XCDs: give each XCD a contiguous run of tiles so that the
  // halo rows / planes neighbouring tiles share are served by one L2
  int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  bid /= nty;
  const int tz = bid % ntz;
  const int b = bid / ntz;
  const int x0 = tx * 16, y0 = ty * T::TY, z0 = tz * T::TZ;
  const int plane = a.D * a.H * a.W;

  unsigned goff[T::NSLOT];
#pragma unroll
  for (int j = 0; j < T::NSLOT; ++j) {
    const int slot = tid + 256 * j;
    const int sx = slot % T::RS, t = slot / T::RS, sy = t % T::TYH, sz = t / T::TYH;
    const int gx = x0 + sx, gy = y0 + sy, gz = z0 + sz;
    const bool ok = (slot < T::SLOTS) & (gx < a.W) & (gy < a.H) & (gz < a.D);
    goff[j] = ok ? 4u * (unsigned)((gz * a.H + gy) * a.W + gx) : 0x80000000u;
  }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (size_t)b * a.Cin * plane), 0, (int)(4u * (unsigned)(a.Cin * plane)), 0x00020000);
  const int nchunk = (a.Cin + 3) / 4;
  const int cot = blockIdx.y * NCT + ct;
  const float* wp = a.wpack + (size_t)cot * nchunk * (27 * 64) + lane;

  f32x4 acc[R][8];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  float pre[4][T::NSLOT];
  auto load_tile = [&](int chunk) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned cb = 4u * (unsigned)((chunk * 4 + c) * plane);
#pragma unroll
      for (int j = 0; j < T::NSLOT; ++j)
        pre[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, goff[j] + cb, 0, 0));
    }
  };
  load_tile(0);
  float wnext[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) wnext[t] = wp[t * 64];

  const float* ap = lds + (lane >> 4) * T::PS + rg * T::ROWBASE + (lane & 15);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < T::NSLOT; ++j)
        if ((j + 1) * 256 <= T::SLOTS || tid + 256 * j < T::SLOTS) lds[c * T::PS + tid + 256 * j] = pre[c][j];
    float wv[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wv[t] = wnext[t];
    __syncthreads();
    if (chunk + 1 < nchunk) {
      load_tile(chunk + 1);
#pragma unroll
      for (int t = 0; t < 27; ++t) wnext[t] = wp[(size_t)(chunk + 1) * (27 * 64) + t * 64];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int pz = q >> 2, py = (q >> 1) & 1, px = q & 1;
#pragma unroll
      for (int dz = 0; dz <= pz; ++dz)
#pragma unroll
        for (int dy = 0; dy <= py; ++dy)
#pragma unroll
          for (int dx = 0; dx <= px; ++dx) {
            // parity 0: tap 1 at offset 0;  parity 1: (tap 0, offset 1), (tap 2, offset 0)
            const int kz = pz ? 2 * dz : 1, oz = pz ? 1 - dz : 0;
            const int ky = py ? 2 * dy : 1, oy = py ? 1 - dy : 0;
            const int kx = px ? 2 * dx : 1, ox = px ? 1 - dx : 0;
            const float w = wv[(kz * 3 + ky) * 3 + kx];
#pragma unroll
            for (int r = 0; r < R; ++r)
              acc[r][q] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(w, ap[(oz * T::TYH + r + oy) * T::RS + ox], acc[r][q], 0, 0, 0);
          }
    }
  }

  const int m = x0 + (lane & 15);
  const int co0 = cot * 16 + 4 * (lane >> 4);
  const int mz = (MAP == 1) ? z0 + rg : z0;
  if (m >= a.W || co0 >= a.Cout || mz >= a.D) return;
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[co0 + j];
  const size_t cs = (size_t)a.Do * a.Ho * a.Wo;
  const int ybase = (MAP == 1) ? y0 : y0 + rg * R;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int my = ybase + r;
    if (my >= a.H) continue;
    // all skip values of this input row first: skip and out may alias as far as the compiler knows, and a
    // load issued after a store would wait for it
    float2 sk[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t o = ((((size_t)b * a.Cout + co0) * a.Do + 2 * mz + (q >> 1)) * a.Ho + 2 * my + (q & 1)) * a.Wo + 2 * m;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        sk[q][j] = (a.skip && co0 + j < a.Cout) ? *reinterpret_cast<const float2*>(a.skip + o + j * cs)
                                                 : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t o = ((((size_t)b * a.Cout + co0) * a.Do + 2 * mz + (q >> 1)) * a.Ho + 2 * my + (q & 1)) * a.Wo + 2 * m;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (co0 + j >= a.Cout) continue;
        float v0 = acc[r][2 * q][j] + bs[j], v1 = acc[r][2 * q + 1][j] + bs[j];
        v0 = fmaxf(v0, 0.f) + a.slope * fminf(v0, 0.f), v1 = fmaxf(v1, 0.f) + a.slope * fminf(v1, 0.f);
        *reinterpret_cast<float2*>(a.out + o + j * cs) = make_float2(v0 + sk[q][j].x, v1 + sk[q][j].y);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// FPN top-down step (feature_net.py:24-36): out = bilinear_x2(coarse, align_corners=True) + conv1x1(fine) + bias.
// Memory-bound (writes 32 channels per pixel); a thread owns one pixel and walks the output channels, the
// 1x1 weights are wave-uniform (scalar loads).
// ---------------------------------------------------------------------------------------------------
template <int CF, bool COARSE_CL>
#ifndef BMV_FPN_WPE
#define BMV_FPN_WPE 1
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BMV_FPN_WPE, 8))) void fpn_topdown_kernel(const float* __restrict__ fine, const float* __restrict__ coarse,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int C, int H, int W, int quad, int csplit) {
  // csplit: the C output channels of a pixel are shared by csplit workgroups (grid.z = B x csplit): more waves in flight
  // for a kernel whose waves otherwise sit in a chain of 8 dependent load -> FMA -> store rounds
  const int part = blockIdx.z % csplit, b = blockIdx.z / csplit;
  const int c_begin = part * (C / csplit), c_end = c_begin + C / csplit;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int Hc = H / 2, Wc = W / 2;
  const size_t hw = (size_t)H * W, hwc = (size_t)Hc * Wc;
  float f[CF];
#pragma unroll
  for (int i = 0; i < CF; ++i) f[i] = fine[((size_t)b * CF + i) * hw + (size_t)y * W + x];
  const Lerp1 ly = upsample_axis(y, Hc, H), lx = upsample_axis(x, Wc, W);
  float* op = out + (size_t)b * C * hw + (size_t)y * W + x;
  if constexpr (COARSE_CL) {
    // coarse is (B, Hc, Wc, C), the channel-last map the plane sweep reads: a bilinear tap of FOUR channels is one
    // 16-byte load (C % 4 == 0, checked by the launcher); two quads in flight
    // ... or quad-planar (B, C/4, Hc, Wc, 4) (`quad`): the same 16-byte taps, pixel stride 4 and quad stride Hc Wc 4
    const float* cp = coarse + (size_t)b * hwc * C;
    const size_t ps = quad ? 4 : (size_t)C;                 // floats between neighbouring pixels
    const size_t qs = quad ? hwc * 4 : 4;                   // floats between the channel quads of a pixel
    const size_t o00 = (size_t)(ly.i0 * Wc + lx.i0) * ps, o01 = (size_t)(ly.i0 * Wc + lx.i1) * ps;
    const size_t o10 = (size_t)(ly.i1 * Wc + lx.i0) * ps, o11 = (size_t)(ly.i1 * Wc + lx.i1) * ps;
    const float w00 = ly.l0 * lx.l0, w01 = ly.l0 * lx.l1, w10 = ly.l1 * lx.l0, w11 = ly.l1 * lx.l1;
    for (int c = c_begin; c < c_end; c += 4) {
      const size_t co = (size_t)(c >> 2) * qs;
      const float4 t00 = *reinterpret_cast<const float4*>(cp + o00 + co), t01 = *reinterpret_cast<const float4*>(cp + o01 + co);
      const float4 t10 = *reinterpret_cast<const float4*>(cp + o10 + co), t11 = *reinterpret_cast<const float4*>(cp + o11 + co);
      float v[4] = {bias[c], bias[c + 1], bias[c + 2], bias[c + 3]};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < CF; ++i) v[j] = fmaf(w[(c + j) * CF + i], f[i], v[j]);
      // same association as the planar path: l0y (l0x a + l1x b) + l1y (l0x c + l1x d)
      v[0] += ly.l0 * (lx.l0 * t00.x + lx.l1 * t01.x) + ly.l1 * (lx.l0 * t10.x + lx.l1 * t11.x);
      v[1] += ly.l0 * (lx.l0 * t00.y + lx.l1 * t01.y) + ly.l1 * (lx.l0 * t10.y + lx.l1 * t11.y);
      v[2] += ly.l0 * (lx.l0 * t00.z + lx.l1 * t01.z) + ly.l1 * (lx.l0 * t10.z + lx.l1 * t11.z);
      v[3] += ly.l0 * (lx.l0 * t00.w + lx.l1 * t01.w) + ly.l1 * (lx.l0 * t10.w + lx.l1 * t11.w);
      (void)w00, (void)w01, (void)w10, (void)w11;
#pragma unroll
      for (int j = 0; j < 4; ++j) op[(size_t)(c + j) * hw] = v[j];
    }
  } else {
    // planar coarse: four channels per iteration, their 16 taps issued before the first is consumed
    const int i00 = ly.i0 * Wc + lx.i0, i01 = ly.i0 * Wc + lx.i1, i10 = ly.i1 * Wc + lx.i0, i11 = ly.i1 * Wc + lx.i1;
    const float* cp = coarse + (size_t)b * C * hwc;
    int c = c_begin;
    for (; c + 4 <= c_end; c += 4) {
      float t[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* q = cp + (size_t)(c + j) * hwc;
        t[j][0] = q[i00], t[j][1] = q[i01], t[j][2] = q[i10], t[j][3] = q[i11];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = bias[c + j];
#pragma unroll
        for (int i = 0; i < CF; ++i) v = fmaf(w[(c + j) * CF + i], f[i], v);
        v += ly.l0 * (lx.l0 * t[j][0] + lx.l1 * t[j][1]) + ly.l1 * (lx.l0 * t[j][2] + lx.l1 * t[j][3]);
        op[(size_t)(c + j) * hw] = v;
      }
    }
    for (; c < c_end; ++c) {
      float v = bias[c];
#pragma unroll
      for (int i = 0; i < CF; ++i) v = fmaf(w[c * CF + i], f[i], v);
      v += upsample_fetch(cp + (size_t)c * hwc, Wc, ly, lx);
      op[(size_t)c * hw] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// FPN top-down step FUSED into the smoothing convolution that consumes it (feature_net.py:24-36: smooth0(p0) with
// p0 = bilinear_x2(p1) + lat0(c0)): p0 is 32 channels at full resolution -- 126 MB written by the top-down kernel and
// read back by the 3x3 convolution at 512x640x3 views -- and exists only between the two.  Here the convolution's
// tile producer builds the 4-channel chunk of p0 it is about to consume: the coarse tile of the chunk goes to LDS
// (double-buffered, ~13x tap reuse), each thread evaluates its tile slots with exactly the expression of
// fpn_topdown_kernel (bias, 8 lateral FMAs in order, the bilinear term) and writes them where the plain kernel's
// loads would have landed; slots outside the image are the convolution's zero padding.  The MFMA part is
// conv_mfma_kernel<1,3,1,1,R,0,PAIR> unchanged (row pairing, 8 output channels).
// ---------------------------------------------------------------------------------------------------
struct FpnSmoothArgs {
  const float* fine;    // (B, 8, H, W)
  const float* coarse;  // (B, C, H/2, W/2) planar
  const float* wlat;    // (C, 8)
  const float* blat;    // (C)
  const float* wpack;   // smoothing conv, row-paired pack
  const float* bias;    // (16)
  float* out;           // (B, Cout, H, W), or null when `packed` is written instead
  const float* rgb;     // (B, 3, H, W): with `packed`
  float* packed;        // (B, H, W, 12) lookup records of the fused renderer: [ch 0 2 4 6 | ch 1 3 5 7 | r b | g 0]
  int B, C, Cout, H, W;
  float slope;
  int ntiles;           // tiles in all; the grid may be smaller (persistent form: a workgroup walks tiles gridDim.x apart)
  const void* const* table;   // deferred `rgb` (bmv_defer_pointer)
  int rgb_slot;
};

template <int R>
__global__ __launch_bounds__(256) void fpn_smooth_kernel(FpnSmoothArgs a) {
  using T = ConvTile<1, 3, 1, 1, R, 0, true>;
  constexpr int CF = 8;
  constexpr int CH = T::TYH / 2 + 3, CW = T::RS / 2 + 3, CN = CH * CW;   // coarse rows / columns under a tile
  static_assert(CN <= 256, "one coarse texel per thread");
  __shared__ float lds[4 * T::PS];
  __shared__ float cl[2][4][CN];
  const int tid = threadIdx.x, lane = tid & 63, rg = tid >> 6;
  const int ntx = (a.W + 15) / 16, nty = (a.H + T::TY - 1) / T::TY;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
  int bid = xcd_contiguous(tile, a.ntiles);
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  const int b = bid / nty;
  const int x0 = tx * 16, y0 = ty * T::TY;
  const int ix0 = x0 - 1, iy0 = y0 - 1;
  const int Hc = a.H / 2, Wc = a.W / 2;
  const int hw = a.H * a.W, hwc = Hc * Wc;
  // first coarse row / column any slot of the tile touches
  const int cy0 = upsample_axis(max(iy0, 0), Hc, a.H).i0, cx0 = upsample_axis(max(ix0, 0), Wc, a.W).i0;

  // this thread's tile slots: the fine-map values, the bilinear weights and the four tap offsets in the LDS coarse tile
  float f[T::NSLOT][CF], l1y[T::NSLOT], l1x[T::NSLOT];
  int o00[T::NSLOT], o01[T::NSLOT], o10[T::NSLOT], o11[T::NSLOT];
  bool ok[T::NSLOT];
  const float* fp = a.fine + (size_t)b * CF * hw;
#pragma unroll
  for (int j = 0; j < T::NSLOT; ++j) {
    const int slot = tid + 256 * j;
    const int sx = slot % T::RS, sy = slot / T::RS;
    const int gx = ix0 + sx, gy = iy0 + sy;
    ok[j] = (slot < T::SLOTS) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H);
    const int qx = ok[j] ? gx : 0, qy = ok[j] ? gy : 0;
#pragma unroll
    for (int i = 0; i < CF; ++i) f[j][i] = fp[(size_t)i * hw + qy * a.W + qx];
    const Lerp1 ly = upsample_axis(qy, Hc, a.H), lx = upsample_axis(qx, Wc, a.W);
    l1y[j] = ly.l1, l1x[j] = lx.l1;
    const int ry0 = min(max(ly.i0 - cy0, 0), CH - 1), ry1 = min(max(ly.i1 - cy0, 0), CH - 1);
    const int rx0 = min(max(lx.i0 - cx0, 0), CW - 1), rx1 = min(max(lx.i1 - cx0, 0), CW - 1);
    o00[j] = ry0 * CW + rx0, o01[j] = ry0 * CW + rx1, o10[j] = ry1 * CW + rx0, o11[j] = ry1 * CW + rx1;
  }
  // coarse texel of this thread (one per channel of a chunk)
  unsigned goffc = 0x80000000u;
  if (tid < CN) {
    const int cy = cy0 + tid / CW, cx = cx0 + tid % CW;
    if (cy < Hc && cx < Wc) goffc = 4u * (unsigned)(cy * Wc + cx);
  }
  __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.coarse + (size_t)b * a.C * hwc), 0, (int)(4u * (unsigned)(a.C * hwc)), 0x00020000);
  const int nchunk = a.C / 4;
  const float* wp = a.wpack + lane;

  f32x4 acc[T::NACC];
#pragma unroll
  for (int r = 0; r < T::NACC; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  float cpre[4];
  auto load_coarse = [&](int chunk) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      cpre[c] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rsc, goffc + 4u * (unsigned)((chunk * 4 + c) * hwc), 0, 0));
  };
  auto store_coarse = [&](int buf) {
    if (tid < CN) {
#pragma unroll
      for (int c = 0; c < 4; ++c) cl[buf][c][tid] = cpre[c];
    }
  };
  load_coarse(0);
  store_coarse(0);
  if (nchunk > 1) load_coarse(1);
  __syncthreads();

  const float* ap = lds + (lane >> 4) * T::PS + rg * T::ROWBASE + (lane & 15);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    // weights of this k-step: requested before the producer, consumed after the barrier
    float wv[T::TAPS];
#pragma unroll
    for (int t = 0; t < T::TAPS; ++t) wv[t] = wp[(size_t)chunk * (T::TAPS * 64) + t * 64];
    // p0 chunk -> LDS tile (previous MFMAs are behind the barrier at the end of the last iteration)
    const int buf = chunk & 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ch = chunk * 4 + c;
      const float bl = a.blat[ch];
      const float* wl = a.wlat + ch * CF;
#pragma unroll
      for (int j = 0; j < T::NSLOT; ++j) {
        if ((j + 1) * 256 > T::SLOTS && tid + 256 * j >= T::SLOTS) continue;
        float v = bl;
#pragma unroll
        for (int i = 0; i < CF; ++i) v = fmaf(wl[i], f[j][i], v);
        const float t00 = cl[buf][c][o00[j]], t01 = cl[buf][c][o01[j]], t10 = cl[buf][c][o10[j]], t11 = cl[buf][c][o11[j]];
        const float l0x = 1.f - l1x[j], l0y = 1.f - l1y[j];
        v += l0y * (l0x * t00 + l1x[j] * t01) + l1y[j] * (l0x * t10 + l1x[j] * t11);
        lds[c * T::PS + tid + 256 * j] = ok[j] ? v : 0.f;
      }
    }
    // next chunk's coarse tile into the other buffer (last read by the producer of chunk - 1, two barriers ago)
    if (chunk + 1 < nchunk) {
      store_coarse(buf ^ 1);
      if (chunk + 2 < nchunk) load_coarse(chunk + 2);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j <= 3; ++j)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float w = wv[j * 3 + kw];
#pragma unroll
        for (int p = 0; p < R / 2; ++p)
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, ap[(2 * p + j) * T::RS + kw], acc[p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x0090);
      }
    __syncthreads();
  }

  // epilogue of the row-paired kernel: lane = output x, registers = 4 consecutive output channels of row y
  const int x = x0 + (lane & 15);
  const int g = lane >> 4;
  const int co0 = 4 * (g & 1);
  const bool live = x < a.W && co0 < a.Cout;
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[live ? co0 + j : 0];
  const int ybase = y0 + rg * R;
  if (live && a.packed) {
    // one record per pixel, everything a bilinear tap of the renderer needs from this view: the wpack rows were
    // permuted on the host so that MFMA rows 0-3 are channels 0 2 4 6 and rows 4-7 channels 1 3 5 7 (the two lane
    // halves of the renderer's MLP take even / odd channels); the source colours ride along
    const int q = g & 1;
    const float* cp = deferred_load(a.table, a.rgb_slot, a.rgb) + (size_t)b * 3 * hw;
#pragma unroll
    for (int r = 0; r < T::NACC; ++r) {
      const int y = ybase + 2 * r + (g >> 1);
      if (y >= a.H) continue;
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = acc[r][j] + bs[j];
        v[j] = fmaxf(t, 0.f) + a.slope * fminf(t, 0.f);
      }
      const size_t pix = (size_t)y * a.W + x;
      float* rec = a.packed + ((size_t)b * hw + pix) * 12;
      *reinterpret_cast<f32x4*>(rec + 4 * q) = v;
      float2 c;
      if (q == 0)
        c.x = cp[pix], c.y = cp[2 * (size_t)hw + pix];
      else
        c.x = cp[(size_t)hw + pix], c.y = 0.f;
      *reinterpret_cast<float2*>(rec + 8 + 2 * q) = c;
    }
  } else if (live) {
#pragma unroll
  for (int r = 0; r < T::NACC; ++r) {
    const int y = ybase + 2 * r + (g >> 1);
    if (y >= a.H) continue;
    const size_t o = (((size_t)b * a.Cout + co0) * a.H + y) * a.W + x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[r][j] + bs[j];
      v = fmaxf(v, 0.f) + a.slope * fminf(v, 0.f);
      if (co0 + j < a.Cout) a.out[o + (size_t)j * hw] = v;
    }
  }
  }
  }   // tile loop (the k loop above ends with a barrier: the next tile may overwrite LDS)
}

// ---------------------------------------------------------------------------------------------------
// FeatureNet's first block as one launch (feature_net.py:8-10, conv0 = ConvBnReLU(3,8) + ConvBnReLU(8,8)): the first
// layer has 3 input channels -- 27 FMAs per output -- so the second layer's tile producer computes it instead of
// reading it: the 3-channel image tile (+2 pixels of halo) goes to LDS once, each thread evaluates its tile slots of
// the 8-channel intermediate 4 channels (one k-step) at a time, weights wave-uniform; slots outside the image are the
// second layer's zero padding.  The intermediate map (31 MB at 512x640x3 views, written and read back) and one launch
// disappear.  MFMA part: the row-paired 8 -> 8 kernel unchanged.
// ---------------------------------------------------------------------------------------------------
struct Conv0Args {
  const float* in;      // (B, 3, H, W)
  const float* w0;      // (8, 3, 3, 3) first layer, batch norm folded
  const float* b0;      // (8)
  const float* wpack;   // second layer, row-paired pack (Cin = 8, Cout <= 8)
  const float* bias;    // (16)
  float* out;           // (B, Cout, H, W)
  int B, Cout, H, W;
  float slope0, slope1;
  const void* const* table;   // deferred `in` (bmv_defer_pointer): read from table[in_slot] when the kernel runs
  int in_slot;
};

template <int R>
__global__ __launch_bounds__(256) void conv0_fused_kernel(Conv0Args a) {
  using T = ConvTile<1, 3, 1, 1, R, 0, true>;
  constexpr int IH = T::TYH + 2, IW = T::RS + 2, IN = IH * IW;   // image tile under the intermediate tile
  constexpr int NIN = (3 * IN + 255) / 256;
  // row pitch of the image tile in LDS: a wave's lanes walk the RS-wide slot rows, so with pitch = RS (mod 32) the
  // rows a 32-lane half touches land on disjoint banks (pitch RS + 2 = 20 put lanes 30 / 31 on the banks of lanes
  // 0 / 1: 32 % of the kernel's LDS cycles were conflict cycles, round-2 PMC)
  constexpr int IWP = T::RS + 32, INP = IH * IWP;
  static_assert(IWP >= IW, "pitch");
  __shared__ float lds[4 * T::PS];
  __shared__ float img[3 * INP];
  const int tid = threadIdx.x, lane = tid & 63, rg = tid >> 6;
  const int ntx = (a.W + 15) / 16, nty = (a.H + T::TY - 1) / T::TY;
  int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  const int b = bid / nty;
  const int x0 = tx * 16, y0 = ty * T::TY;
  const int ix0 = x0 - 1, iy0 = y0 - 1;          // origin of the intermediate tile (second layer's halo)
  const int hw = a.H * a.W;

  const float* in = deferred_load(a.table, a.in_slot, a.in);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)b * 3 * hw), 0,
                                                                  (int)(4u * (unsigned)(3 * hw)), 0x00020000);
#pragma unroll
  for (int j = 0; j < NIN; ++j) {
    const int e = tid + 256 * j;
    if (e < 3 * IN) {
      const int c = e / IN, r = e - c * IN, sy = r / IW, sx = r - sy * IW;
      const int gx = ix0 - 1 + sx, gy = iy0 - 1 + sy;
      const bool ok = (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H);
      img[c * INP + sy * IWP + sx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                             rsrc, ok ? 4u * (unsigned)(c * hw + gy * a.W + gx) : 0x80000000u, 0, 0));
    }
  }
  bool ok[T::NSLOT];
  int ibase[T::NSLOT];
#pragma unroll
  for (int j = 0; j < T::NSLOT; ++j) {
    const int slot = tid + 256 * j;
    const int sx = slot % T::RS, sy = slot / T::RS;
    const int gx = ix0 + sx, gy = iy0 + sy;
    ok[j] = (slot < T::SLOTS) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H);
    ibase[j] = (slot < T::SLOTS) ? sy * IWP + sx : 0;   // top-left input of the slot's 3x3 window
  }
  const float* wp = a.wpack + lane;
  f32x4 acc[T::NACC];
#pragma unroll
  for (int r = 0; r < T::NACC; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = lds + (lane >> 4) * T::PS + rg * T::ROWBASE + (lane & 15);
  __syncthreads();   // image tile complete
#pragma unroll 1
  for (int chunk = 0; chunk < 2; ++chunk) {
    float wv[T::TAPS];
#pragma unroll
    for (int t = 0; t < T::TAPS; ++t) wv[t] = wp[(size_t)chunk * (T::TAPS * 64) + t * 64];
#pragma unroll
    for (int j = 0; j < T::NSLOT; ++j) {
      if ((j + 1) * 256 > T::SLOTS && tid + 256 * j >= T::SLOTS) continue;
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = a.b0[chunk * 4 + c];
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float x = img[ci * INP + ibase[j] + ky * IWP + kx];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = fmaf(a.w0[((chunk * 4 + c) * 3 + ci) * 9 + ky * 3 + kx], x, v[c]);
          }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float r = fmaxf(v[c], 0.f) + a.slope0 * fminf(v[c], 0.f);
        lds[c * T::PS + tid + 256 * j] = ok[j] ? r : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j <= 3; ++j)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float w = wv[j * 3 + kw];
#pragma unroll
        for (int p = 0; p < R / 2; ++p)
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, ap[(2 * p + j) * T::RS + kw], acc[p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x0090);
      }
    __syncthreads();
  }
  const int x = x0 + (lane & 15);
  const int g = lane >> 4;
  const int co0 = 4 * (g & 1);
  if (x >= a.W || co0 >= a.Cout) return;
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[co0 + j];
  const int ybase = y0 + rg * R;
#pragma unroll
  for (int r = 0; r < T::NACC; ++r) {
    const int y = ybase + 2 * r + (g >> 1);
    if (y >= a.H) continue;
    const size_t o = (((size_t)b * a.Cout + co0) * a.H + y) * a.W + x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[r][j] + bs[j];
      v = fmaxf(v, 0.f) + a.slope1 * fminf(v, 0.f);
      if (co0 + j < a.Cout) a.out[o + (size_t)j * hw] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Split-K tiling for the deep U-Net levels (a few thousand voxels, 32-64 channels): with one 4-row x 16-column
// tile per workgroup those launches are a few hundred workgroups whose waves each walk ALL k-steps in sequence,
// exposing one global-load latency per k-step (measured 16-30 % MFMA-pipe occupancy).  Here every wave covers all
// 4 rows of the tile but only every 4th k-step (4 input channels) of a 16-channel LDS stage; the 4 partial sums meet
// in LDS and wave r finishes row r.  The dependent chain per wave is 4x shorter.
// ---------------------------------------------------------------------------------------------------
// TYV: output rows per workgroup (4, 2 or 1).  Every wave runs all TYV rows of its k-steps, so a wave's dependent chain
// is TYV x taps x (k-steps / 4) matrix instructions: 432 x 32 cycles = 6.5 us for a 64 -> 64 layer with 4 rows -- most of
// what such a launch takes, on a grid (96 workgroups for 1 x 32 x 40) that fills a third of the chip.  Fewer rows per
// workgroup = more workgroups with shorter chains (round 5).
template <int KD, int K, int S, bool IS3D, int TYV = 4>
struct SplitKTile {
  static constexpr int TY = TYV;
  static constexpr int TZH = KD, TYH = (TY - 1) * S + K, RS = 15 * S + K;
  static constexpr int SLOTS = TZH * TYH * RS;
  static constexpr int PS = (S == 1) ? ((SLOTS + 15) / 32 * 32 + 16) : (SLOTS | 1);
  static constexpr int NSLOT = (SLOTS + 255) / 256;
  static constexpr int TAPS = KD * K * K;
};

template <int KD, int K, int S, bool IS3D, int TYV = 4>
// stride 1: 4 workgroups per CU fit the LDS (38 KB each); the allocator needs 121 instead of 160 registers for that
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((BMV_CONV_WPE_TUNED && S == 1) ? 4 : 1, 8)))
void conv_splitk_kernel(ConvArgs a) {
  using T = SplitKTile<KD, K, S, IS3D, TYV>;
  constexpr int TY = TYV;
  __shared__ float lds[16 * T::PS];
  __shared__ f32x4 red[4][TY][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntx = (a.Wo + 15) / 16, nty = (a.Ho + TY - 1) / TY;
  int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  bid /= nty;
  const int z0 = bid % a.Do;
  const int b = bid / a.Do;
  const int x0 = tx * 16, y0 = ty * TY;
  const int ix0 = x0 * S - K / 2, iy0 = y0 * S - K / 2, iz0 = z0 * S - KD / 2;
  const int plane = a.D * a.H * a.W;

  unsigned goff[T::NSLOT];
#pragma unroll
  for (int j = 0; j < T::NSLOT; ++j) {
    const int slot = tid + 256 * j;
    const int sx = slot % T::RS, t = slot / T::RS, sy = t % T::TYH, sz = t / T::TYH;
    const int gx = ix0 + sx, gy = iy0 + sy, gz = iz0 + sz;
    const bool ok = (slot < T::SLOTS) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H) & (gz >= 0) & (gz < a.D);
    goff[j] = ok ? 4u * (unsigned)((gz * a.H + gy) * a.W + gx) : 0x80000000u;
  }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (size_t)b * a.Cin * plane), 0, (int)(4u * (unsigned)(a.Cin * plane)), 0x00020000);
  const int nk = (a.Cin + 3) / 4, nstage = (nk + 3) / 4;
  const int cot = blockIdx.y;
  const float* wp = a.wpack + (size_t)cot * nk * (T::TAPS * 64) + lane;

  f32x4 acc[TY];
#pragma unroll
  for (int r = 0; r < TY; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* ap = lds + (wave * 4 + (lane >> 4)) * T::PS + (lane & 15) * S;
  // 16 channels of the tile -> registers; channels past Cin (last stage) must read as 0: the offsets stay below 2^31
  // because (Cin + 15) * plane * 4 is checked on the host
  float pre[16][T::NSLOT];
  auto load_stage = [&](int stage) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int ch = stage * 16 + c;
      const unsigned cb = 4u * (unsigned)(ch * plane);
#pragma unroll
      for (int j = 0; j < T::NSLOT; ++j)
        pre[c][j] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ch < a.Cin ? goff[j] + cb : 0x80000000u, 0, 0));
    }
  };
  // taps whose input plane lies outside the volume multiply zeros: the deep levels are 1-2 planes thick, so up to 2/3
  // of a 3x3x3 filter (wave-uniform: a workgroup is one output plane)
  bool zin[KD];
#pragma unroll
  for (int kd = 0; kd < KD; ++kd) zin[kd] = !IS3D || (iz0 + kd >= 0 && iz0 + kd < a.D);
  // stride 1: the loads of stage s + 1 are in flight during the MFMAs of stage s (the tile registers are free once
  // they are in LDS); the stride-2 tile is 64 registers in a 256-register kernel and stays single-buffered
  constexpr bool PREFETCH = false;   // tried twice (tile only; tile + weights double-buffered): slower, see DESIGN 4.7
  float wv[T::TAPS], wnext[T::TAPS];
  auto load_w = [&](int stage, float (&w)[T::TAPS]) {   // this wave's k-step of the stage
    const int ks = stage * 4 + wave;
    if (ks < nk) {
#pragma unroll
      for (int t = 0; t < T::TAPS; ++t)
        if (zin[t / (K * K)]) w[t] = wp[(size_t)ks * (T::TAPS * 64) + t * 64];
    }
  };
  if (PREFETCH) load_stage(0), load_w(0, wv);
  for (int stage = 0; stage < nstage; ++stage) {
    if (!PREFETCH) load_stage(stage), load_w(stage, wv);
    const int ks = stage * 4 + wave;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
      for (int j = 0; j < T::NSLOT; ++j)
        if ((j + 1) * 256 <= T::SLOTS || tid + 256 * j < T::SLOTS) lds[c * T::PS + tid + 256 * j] = pre[c][j];
    __syncthreads();
    // the weights of a deep layer are cold in L2 when it starts (a first-touch read is ~2 us): next stage's are
    // requested a whole stage ahead, like the tile
    if (PREFETCH && stage + 1 < nstage) load_stage(stage + 1), load_w(stage + 1, wnext);
    if (ks < nk) {
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) {
        if (!zin[kd]) continue;
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            const float w = wv[(kd * K + kh) * K + kw];
#pragma unroll
            for (int r = 0; r < TY; ++r)
              acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, ap[(kd * T::TYH + r * S + kh) * T::RS + kw], acc[r], 0,
                                                            0, 0);
          }
      }
    }
    if (PREFETCH) {
#pragma unroll
      for (int t = 0; t < T::TAPS; ++t) wv[t] = wnext[t];
    }
  }
#pragma unroll
  for (int r = 0; r < TY; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave >= TY) return;
  const f32x4 sum = red[0][wave][lane] + red[1][wave][lane] + red[2][wave][lane] + red[3][wave][lane];

  // wave r finishes row r: lane = output x, registers = 4 consecutive output channels
  const int x = x0 + (lane & 15), y = y0 + wave;
  const int co0 = cot * 16 + 4 * (lane >> 4);
  if (x >= a.Wo || y >= a.Ho || co0 >= a.Cout) return;
  const size_t cs = (size_t)a.Do * a.Ho * a.Wo;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = sum[j] + a.bias[co0 + j];
    v[j] = fmaxf(v[j], 0.f) + a.slope * fminf(v[j], 0.f);
  }
  if (a.channels_last == 3) {
    const size_t o = (((((size_t)b * (a.Cout >> 2) + (co0 >> 2)) * a.Do + z0) * a.Ho + y) * a.Wo + x) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) a.out[o + j] = v[j];
  } else if (a.channels_last) {
    const size_t o = ((((size_t)b * a.Do + z0) * a.Ho + y) * a.Wo + x) * a.Cout + co0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (co0 + j < a.Cout) a.out[o + j] = v[j] + (a.skip ? a.skip[o + j] : 0.f);
  } else {
    const size_t o = ((((size_t)b * a.Cout + co0) * a.Do + z0) * a.Ho + y) * a.Wo + x;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (co0 + j < a.Cout) a.out[o + j * cs] = v[j] + (a.skip ? a.skip[o + j * cs] : 0.f);
  }
}

template <int KD, int K, int S, bool IS3D>
static void launch_splitk(const ConvArgs& a, hipStream_t st) {
  // rows per workgroup: 4 while that already gives the chip >= 1024 workgroups, else 2, else 1 (BMV_CONV_SPLITK_ROWS forces)
  const unsigned base = cdiv(a.Wo, 16) * a.Do * a.B * cdiv(a.Cout, 16);
  int rows = bmv::tuning("BMV_CONV_SPLITK_ROWS", 0);
  if (rows != 1 && rows != 2 && rows != 4) rows = base * cdiv(a.Ho, 4) >= 1024 ? 4 : base * cdiv(a.Ho, 2) >= 1024 ? 2 : 1;
  dim3 grid(cdiv(a.Wo, 16) * cdiv(a.Ho, rows) * a.Do * a.B, cdiv(a.Cout, 16));
  if (rows == 4)
    hipLaunchKernelGGL((conv_splitk_kernel<KD, K, S, IS3D, 4>), grid, dim3(256), 0, st, a);
  else if (rows == 2)
    hipLaunchKernelGGL((conv_splitk_kernel<KD, K, S, IS3D, 2>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_splitk_kernel<KD, K, S, IS3D, 1>), grid, dim3(256), 0, st, a);
}

// Split-K form of the transposed convolution for the deep levels: one input row x 16 input x per workgroup, wave w takes
// every 4th k-step of a 16-channel stage with all 8 parity accumulators, the partial sums meet in LDS and wave w
// finishes output row pair (pz, py) = (w >> 1, w & 1).
// 4 workgroups per CU (LDS 38 KB each): 90 instead of 144 registers, no spills; conv9 at level 1 is 3072 workgroups
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BMV_CONV_WPE_TUNED ? 4 : 1, 8)))
void convT3d_splitk_kernel(ConvArgs a) {
  constexpr int RS = 17, SLOTS = 2 * 2 * RS, PS = (SLOTS + 15) / 32 * 32 + 16;
  __shared__ float lds[16 * PS];
  __shared__ f32x4 red[4][8][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntx = (a.W + 15) / 16;
  int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int tx = bid % ntx;
  bid /= ntx;
  const int my = bid % a.H;
  bid /= a.H;
  const int mz = bid % a.D;
  const int b = bid / a.D;
  const int x0 = tx * 16;
  const int plane = a.D * a.H * a.W;
  unsigned goff = 0x80000000u;
  if (tid < SLOTS) {
    const int sx = tid % RS, t = tid / RS, sy = t & 1, sz = t >> 1;
    const int gx = x0 + sx, gy = my + sy, gz = mz + sz;
    if (gx < a.W && gy < a.H && gz < a.D) goff = 4u * (unsigned)((gz * a.H + gy) * a.W + gx);
  }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (size_t)b * a.Cin * plane), 0, (int)(4u * (unsigned)(a.Cin * plane)), 0x00020000);
  const int nk = (a.Cin + 3) / 4, nstage = (nk + 3) / 4;
  const int cot = blockIdx.y;
  const float* wp = a.wpack + (size_t)cot * nk * (27 * 64) + lane;
  f32x4 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = lds + (wave * 4 + (lane >> 4)) * PS + (lane & 15);
  const bool z1 = mz + 1 < a.D;   // the second input plane exists (else its taps multiply zeros)
  for (int stage = 0; stage < nstage; ++stage) {
    float pre[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int ch = stage * 16 + c;
      pre[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                             rsrc, ch < a.Cin ? goff + 4u * (unsigned)(ch * plane) : 0x80000000u, 0, 0));
    }
    const int ks = stage * 4 + wave;
    float wv[27];
    if (ks < nk) {
#pragma unroll
      for (int t = 0; t < 27; ++t) wv[t] = wp[(size_t)ks * (27 * 64) + t * 64];
    }
    __syncthreads();
    if (tid < SLOTS) {
#pragma unroll
      for (int c = 0; c < 16; ++c) lds[c * PS + tid] = pre[c];
    }
    __syncthreads();
    if (ks < nk) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int pz = q >> 2, py = (q >> 1) & 1, px = q & 1;
#pragma unroll
        for (int dz = 0; dz <= pz; ++dz) {
          const int kz = pz ? 2 * dz : 1, oz = pz ? 1 - dz : 0;
          if (oz == 1 && !z1) continue;
#pragma unroll
          for (int dy = 0; dy <= py; ++dy)
#pragma unroll
            for (int dx = 0; dx <= px; ++dx) {
              const int ky = py ? 2 * dy : 1, oy = py ? 1 - dy : 0;
              const int kx = px ? 2 * dx : 1, ox = px ? 1 - dx : 0;
              acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[(kz * 3 + ky) * 3 + kx], ap[(oz * 2 + oy) * RS + ox],
                                                            acc[q], 0, 0, 0);
            }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) red[wave][q][lane] = acc[q];
  __syncthreads();
  const int q0 = 2 * wave;  // this wave finishes parities (pz, py) = (wave >> 1, wave & 1), px = 0 and 1
  const f32x4 s0 = red[0][q0][lane] + red[1][q0][lane] + red[2][q0][lane] + red[3][q0][lane];
  const f32x4 s1 = red[0][q0 + 1][lane] + red[1][q0 + 1][lane] + red[2][q0 + 1][lane] + red[3][q0 + 1][lane];
  const int m = x0 + (lane & 15);
  const int co0 = cot * 16 + 4 * (lane >> 4);
  if (m >= a.W || co0 >= a.Cout) return;
  const size_t cs = (size_t)a.Do * a.Ho * a.Wo;
  const size_t o = ((((size_t)b * a.Cout + co0) * a.Do + 2 * mz + (wave >> 1)) * a.Ho + 2 * my + (wave & 1)) * a.Wo + 2 * m;
  float2 sk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    sk[j] = (a.skip && co0 + j < a.Cout) ? *reinterpret_cast<const float2*>(a.skip + o + j * cs) : make_float2(0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (co0 + j >= a.Cout) continue;
    float v0 = s0[j] + a.bias[co0 + j], v1 = s1[j] + a.bias[co0 + j];
    v0 = fmaxf(v0, 0.f) + a.slope * fminf(v0, 0.f), v1 = fmaxf(v1, 0.f) + a.slope * fminf(v1, 0.f);
    *reinterpret_cast<float2*>(a.out + o + j * cs) = make_float2(v0 + sk[j].x, v1 + sk[j].y);
  }
}

// ---------------------------------------------------------------------------------------------------
// Launch selection.  Big tiles (R rows per wave, all cout tiles of a layer in one block so the LDS tile is
// shared) when they still give every CU a few blocks; otherwise small tiles, one cout tile per block
// (grid.y walks the cout tiles): the deep U-Net levels have a few thousand voxels only.
// ---------------------------------------------------------------------------------------------------
template <int KD, int K, int S, int NCT, int R, int MAP, bool PAIR>
static unsigned conv_blocks(const ConvArgs& a) {
  using T = ConvTile<KD, K, S, NCT, R, MAP, PAIR>;
  return cdiv(a.Wo, 16) * cdiv(a.Ho, T::TY) * cdiv(a.Do, T::TZ) * a.B * cdiv(cdiv(a.Cout, 16), NCT);
}
template <int KD, int K, int S, int NCT, int R, int MAP, bool PAIR, bool QIN = false>
static void launch_conv(const ConvArgs& a, hipStream_t st) {
  using T = ConvTile<KD, K, S, NCT, R, MAP, PAIR>;
  dim3 grid(cdiv(a.Wo, 16) * cdiv(a.Ho, T::TY) * cdiv(a.Do, T::TZ) * a.B, cdiv(cdiv(a.Cout, 16), NCT));
  if (a.band_map) grid.x = 8u * cdiv(a.Wo, 16) * cdiv(cdiv(a.Ho, T::TY), 8) * cdiv(a.Do, T::TZ) * a.B;
  hipLaunchKernelGGL((conv_mfma_kernel<KD, K, S, NCT, R, MAP, PAIR, false, QIN>), grid, dim3(256), 0, st, a);
}

// the regularisers' stride-2 layer behind a first layer that wrote quad records (conv1: 8 -> 16): the tilings
// dispatch_conv<3, 3, 2, 4, 2, true> picks for one 16-channel output tile, with 16-byte staging
static bool dispatch_conv_s2_quad(const ConvArgs& a, hipStream_t st) {
  if (cdiv(a.Cout, 16) != 1) return false;
  if (conv_blocks<3, 3, 2, 1, 4, 1, false>(a) >= 512) return launch_conv<3, 3, 2, 1, 4, 1, false, true>(a, st), true;
  if (conv_blocks<3, 3, 2, 1, 2, 1, false>(a) >= 512) return launch_conv<3, 3, 2, 1, 2, 1, false, true>(a, st), true;
  return launch_conv<3, 3, 2, 1, 1, 2, false, true>(a, st), true;
}

constexpr unsigned kEnoughBlocks = 512;  // 2 per CU
static bool splitk_enabled() {
  const bool v = bmv::tuning("BMV_CONV_SPLITK", 1) != 0;
  return v;
}

// RB / RS: rows per wave of the big / small tiling
template <int KD, int K, int S, int RB, int RS_, bool IS3D>
static void dispatch_conv(const ConvArgs& a, hipStream_t st) {
  constexpr int MB = IS3D ? 1 : 0;   // big tiles: row groups along z for volumes
  constexpr int MS = IS3D ? 2 : 0;   // small tiles: one z slice
  const unsigned ncot = cdiv(a.Cout, 16);
  if constexpr (S == 1 && K == 3) {
    if (a.Cout <= 8) {               // row pairing
      if (conv_blocks<KD, K, S, 1, RB, MB, true>(a) >= kEnoughBlocks) {
        // measured inside the frame (bench.py, graph replay): 4 rows per wave win for the 3 -> 8 / 8 -> 8 full-resolution
        // 2-D layers (more resident blocks), 8 rows for 32-channel inputs and the volumes (fewer halo loads)
        if (!IS3D && a.Cin <= 8) return launch_conv<KD, K, S, 1, 4, MB, true>(a, st);
        const int r3d = bmv::tuning("BMV_CONV_PAIR_ROWS", 0);   // tuning
        if (r3d == 4) return launch_conv<KD, K, S, 1, 4, MB, true>(a, st);
        // 5: half-height tiles only where the full-height grid is below 4 workgroups per CU (level 0's 640)
        if (r3d == 5 && conv_blocks<KD, K, S, 1, RB, MB, true>(a) < 1024) return launch_conv<KD, K, S, 1, 4, MB, true>(a, st);
        return launch_conv<KD, K, S, 1, RB, MB, true>(a, st);
      }
      return launch_conv<KD, K, S, 1, 2, MS, true>(a, st);
    }
  }
  constexpr int RH = RB == 8 ? 4 : RB;  // feat_conv + depth_conv (9 channels on a full-size volume): 4 rows per wave
  if (IS3D && ncot == 1 && conv_blocks<KD, K, S, 1, RH, MB, false>(a) >= kEnoughBlocks)
    return launch_conv<KD, K, S, 1, RH, MB, false>(a, st);
  if (ncot == 1 && conv_blocks<KD, K, S, 1, RB, MB, false>(a) >= kEnoughBlocks)
    return launch_conv<KD, K, S, 1, RB, MB, false>(a, st);
  if (ncot == 2 && conv_blocks<KD, K, S, 2, RB, MB, false>(a) >= kEnoughBlocks)
    return launch_conv<KD, K, S, 2, RB, MB, false>(a, st);
  if (ncot >= 3 && conv_blocks<KD, K, S, 4, RB, MB, false>(a) >= kEnoughBlocks)
    return launch_conv<KD, K, S, 4, RB, MB, false>(a, st);
  if (conv_blocks<KD, K, S, 1, RS_, MB, false>(a) >= kEnoughBlocks)
    return launch_conv<KD, K, S, 1, RS_, MB, false>(a, st);
  // measured inside the frame (bench.py): 292.4 -> 296.8 Mray/s with Cin >= 16, 295.3 with Cin >= 32
  if (a.Cin >= 16 && splitk_enabled()) return launch_splitk<KD, K, S, IS3D>(a, st);
  launch_conv<KD, K, S, 1, 1, MS, false>(a, st);
}

template <int NCT, int R, int MAP>
static unsigned convT_blocks(const ConvArgs& a) {
  using T = ConvTTile<NCT, R, MAP>;
  return cdiv(a.W, 16) * cdiv(a.H, T::TY) * cdiv(a.D, T::TZ) * a.B * cdiv(cdiv(a.Cout, 16), NCT);
}
template <int NCT, int R, int MAP>
static void launch_convT(const ConvArgs& a, hipStream_t st) {
  using T = ConvTTile<NCT, R, MAP>;
  dim3 grid(cdiv(a.W, 16) * cdiv(a.H, T::TY) * cdiv(a.D, T::TZ) * a.B, cdiv(cdiv(a.Cout, 16), NCT));
  hipLaunchKernelGGL((convT3d_mfma_kernel<NCT, R, MAP>), grid, dim3(256), 0, st, a);
}

}  // namespace bmv

extern "C" {

int bmv_conv_pairs_rows(int Cout, int kd, int k, int stride) {
  return (Cout <= 8 && stride == 1 && k == 3 && (kd == 1 || kd == 3)) ? 1 : 0;
}

int bmv_conv_wpack_floats(int Cin, int Cout, int kd, int k, int stride) {
  const int taps = bmv_conv_pairs_rows(Cout, kd, k, stride) ? kd * (k + 1) * k : kd * k * k;
  return ((Cout + 15) / 16) * ((Cin + 3) / 4) * taps * 64;
}

// One thread per wpack element: the layouts documented in include/bmv.h, straight from torch's weight tensor
// ((Cout,Cin,taps), or (Cin,Cout,taps) for a transposed convolution) -- the training forward repacks every step.
namespace bmv {
__global__ void conv_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int kd, int k, int pair, int transposed,
                                 int flip, int total, float* __restrict__ wpack) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int nc = (Cin + 3) / 4, ntap = kd * k * k;
  // weight of the convolution being packed at (output channel, input channel, tap), from the source tensor
  auto src = [&](int co, int ci, int tap) {
    const int t = flip ? ntap - 1 - tap : tap;
    return transposed ? w[((size_t)ci * Cout + co) * ntap + t] : w[((size_t)co * Cin + ci) * ntap + t];
  };
  const int o = idx & 15, kk = (idx >> 4) & 3;
  int r = idx >> 6;
  float v = 0.f;
  if (pair) {
    const int taps = kd * (k + 1) * k;
    const int tap = r % taps, c = r / taps;
    const int kx = tap % k, j = (tap / k) % (k + 1), kz = tap / (k * (k + 1));
    const int ci = 4 * c + kk, co = o & 7, ky = o < 8 ? j : j - 1;
    if (co < Cout && ci < Cin && ky >= 0 && ky < k) v = src(co, ci, (kz * k + ky) * k + kx);
  } else {
    const int tap = r % ntap;
    r /= ntap;
    const int c = r % nc, t = r / nc;
    const int co = 16 * t + o, ci = 4 * c + kk;
    if (co < Cout && ci < Cin) v = src(co, ci, tap);
  }
  wpack[idx] = v;
}
}  // namespace bmv

int bmv_conv_pack_weights(const float* weight, int Cin, int Cout, int kd, int k, int stride, int transposed, int flip,
                          int for_transpose_kernel, float* wpack, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(weight && wpack, "conv_pack: null pointer");
  BMV_REQUIRE(Cin > 0 && Cout > 0 && kd > 0 && k > 0, "conv_pack: bad shape");
  const int pair = for_transpose_kernel ? 0 : bmv_conv_pairs_rows(Cout, kd, k, stride);
  const int total = for_transpose_kernel ? ((Cout + 15) / 16) * ((Cin + 3) / 4) * kd * k * k * 64
                                         : bmv_conv_wpack_floats(Cin, Cout, kd, k, stride);
  hipLaunchKernelGGL(conv_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), weight, Cout, Cin, kd, k,
                     pair, transposed, flip, total, wpack);
  BMV_LAUNCH_END("bmv_conv_pack_weights");
}

int bmv_conv_fwd(const float* in, const float* wpack, const float* bias, const float* skip, float* out, int B, int Cin,
                 int D, int H, int W, int Cout, int kd, int k, int stride, float act_slope, int out_channels_last,
                 bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(in && wpack && bias && out, "conv: null pointer");
  BMV_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "conv: bad shape");
  BMV_REQUIRE(stride == 1 || stride == 2, "conv: stride %d unsupported", stride);
  BMV_REQUIRE((size_t)Cin * D * H * W < (1u << 29), "conv: one batch item must stay below 2 GiB");
  ConvArgs a;
  a.in = in, a.wpack = wpack, a.bias = bias, a.skip = skip, a.out = out;
  a.B = B, a.Cin = Cin, a.D = D, a.H = H, a.W = W, a.Cout = Cout;
  const int p = k / 2, pd = kd / 2;
  a.Do = (D + 2 * pd - kd) / stride + 1, a.Ho = (H + 2 * p - k) / stride + 1, a.Wo = (W + 2 * p - k) / stride + 1;
  const bool in_quad = (out_channels_last & 16) != 0;      // `in` as quad records (B, Cin/4, D, H, W, 4)
  out_channels_last &= ~16;
  BMV_REQUIRE(!in_quad || (kd == 3 && k == 3 && stride == 2 && Cin % 4 == 0 && Cin < 16 && Cout <= 16),
              "conv: quad-record input is built for the regularisers' stride-2 layer (3x3x3, Cin 4 / 8 / 12, Cout <= 16)");
  BMV_REQUIRE(out_channels_last == 0 || out_channels_last == 1 || out_channels_last == 3, "conv: out layout %d", out_channels_last);
  BMV_REQUIRE(out_channels_last != 3 || ((Cout & 3) == 0 && !skip), "conv: quad-planar output needs Cout %% 4 == 0 and no skip");
  a.slope = act_slope, a.channels_last = out_channels_last, a.out2 = nullptr;
  a.band_map = out_channels_last == 3 && kd == 1;
  hipStream_t st = as_stream(stream);
  if (kd == 1 && k == 3 && stride == 1)
    dispatch_conv<1, 3, 1, 8, 2, false>(a, st);
  else if (kd == 1 && k == 5 && stride == 2)
    dispatch_conv<1, 5, 2, 4, 2, false>(a, st);
  else if (kd == 1 && k == 1 && stride == 1)
    dispatch_conv<1, 1, 1, 8, 2, false>(a, st);
  else if (kd == 3 && k == 3 && stride == 1)
    dispatch_conv<3, 3, 1, 8, 2, true>(a, st);
  else if (kd == 3 && k == 3 && stride == 2 && in_quad)
    dispatch_conv_s2_quad(a, st);
  else if (kd == 3 && k == 3 && stride == 2)
    dispatch_conv<3, 3, 2, 4, 2, true>(a, st);
  else
    BMV_REQUIRE(false, "conv: kernel (%d,%d,%d) stride %d is not one of the shapes of FeatureNet / CostRegNet", kd, k,
                k, stride);
  BMV_LAUNCH_END("conv_fwd");
}

int bmv_conv_heads_fwd(const float* in, const float* wpack, const float* bias, float* records_out, float* depth_out,
                       int B, int Cin, int D, int H, int W, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(in && wpack && bias && records_out && depth_out, "conv_heads: null pointer");
  BMV_REQUIRE(B > 0 && Cin > 0 && Cin < 16 && D > 0 && H > 0 && W > 0, "conv_heads: bad shape (Cin=%d)", Cin);
  BMV_REQUIRE((size_t)Cin * D * H * W < (1u << 29), "conv_heads: one batch item must stay below 2 GiB");
  ConvArgs a;
  a.in = in, a.wpack = wpack, a.bias = bias, a.skip = nullptr, a.out = records_out, a.out2 = depth_out;
  a.B = B, a.Cin = Cin, a.D = D, a.H = H, a.W = W, a.Cout = 9;
  a.Do = D, a.Ho = H, a.Wo = W;
  a.slope = 1.f, a.channels_last = 2;
  dispatch_conv<3, 3, 1, 8, 2, true>(a, as_stream(stream));     // Cin < 16: never the split-K form
  BMV_LAUNCH_END("conv_heads_fwd");
}

int bmv_conv3d_transpose_fwd(const float* in, const float* wpack, const float* bias, const float* skip, float* out, int B,
                             int Cin, int D, int H, int W, int Cout, float act_slope, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(in && wpack && bias && out, "convT3d: null pointer");
  BMV_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "convT3d: bad shape");
  BMV_REQUIRE((size_t)Cin * D * H * W < (1u << 29), "convT3d: one batch item must stay below 2 GiB");
  ConvArgs a;
  a.in = in, a.wpack = wpack, a.bias = bias, a.skip = skip, a.out = out;
  a.B = B, a.Cin = Cin, a.D = D, a.H = H, a.W = W, a.Cout = Cout;
  a.Do = 2 * D, a.Ho = 2 * H, a.Wo = 2 * W;
  a.slope = act_slope, a.channels_last = 0;
  hipStream_t st = as_stream(stream);
  if (convT_blocks<1, 4, 1>(a) >= kEnoughBlocks)
    launch_convT<1, 4, 1>(a, st);
  else if (convT_blocks<1, 2, 1>(a) >= kEnoughBlocks)
    launch_convT<1, 2, 1>(a, st);
  else if (Cin >= 16 && splitk_enabled()) {
    dim3 grid(cdiv(W, 16) * H * D * B, cdiv(Cout, 16));
    hipLaunchKernelGGL(convT3d_splitk_kernel, grid, dim3(256), 0, st, a);
  } else
    launch_convT<1, 1, 2>(a, st);
  BMV_LAUNCH_END("convT3d_fwd");
}

int bmv_fpn_topdown_fwd(const float* fine, const float* coarse, const float* w, const float* bias, float* out, int B,
                        int Cf, int C, int H, int W, int coarse_channels_last, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(fine && coarse && w && bias && out, "fpn_topdown: null pointer");
  BMV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "fpn_topdown: bad shape");
  BMV_REQUIRE(!coarse_channels_last || C % 4 == 0, "fpn_topdown: a channel-last coarse map needs C %% 4 == 0 (C=%d)", C);
  int csplit = bmv::tuning("BMV_FPN_TOPDOWN_SPLIT", 2);   // measured in the frame: 1: 894 us, 2: 886, 4: 887
  if (csplit < 1 || C % (4 * csplit) != 0) csplit = 1;
  dim3 grid(cdiv(W, 64), cdiv(H, 4), B * csplit);
  hipStream_t st = as_stream(stream);
  const int quad = coarse_channels_last == 3;   // 0 planar, 1 channel-last, 3 quad-planar
  if (Cf == 8 && !coarse_channels_last)
    hipLaunchKernelGGL((fpn_topdown_kernel<8, false>), grid, dim3(256), 0, st, fine, coarse, w, bias, out, C, H, W, quad, csplit);
  else if (Cf == 16 && !coarse_channels_last)
    hipLaunchKernelGGL((fpn_topdown_kernel<16, false>), grid, dim3(256), 0, st, fine, coarse, w, bias, out, C, H, W, quad, csplit);
  else if (Cf == 8)
    hipLaunchKernelGGL((fpn_topdown_kernel<8, true>), grid, dim3(256), 0, st, fine, coarse, w, bias, out, C, H, W, quad, csplit);
  else if (Cf == 16)
    hipLaunchKernelGGL((fpn_topdown_kernel<16, true>), grid, dim3(256), 0, st, fine, coarse, w, bias, out, C, H, W, quad, csplit);
  else
    BMV_REQUIRE(false, "fpn_topdown: %d lateral input channels unsupported (FeatureNet has 8 and 16)", Cf);
  BMV_LAUNCH_END("fpn_topdown_fwd");
}

int bmv_conv_top_fwd(const float* in, const float* wpack, const float* bias, const float* wpack_top,
                     const float* bias_top, float* out, int B, int H, int W, float act_slope, int out_layout,
                     bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(in && wpack && bias && wpack_top && bias_top && out, "conv_top: null pointer");
  BMV_REQUIRE(out_layout == 1 || out_layout == 3, "conv_top: out layout %d (1 channel-last, 3 quad-planar)", out_layout);
  BMV_REQUIRE(B > 0 && H > 0 && W > 0, "conv_top: bad shape");
  BMV_REQUIRE((size_t)32 * H * W < (1u << 29), "conv_top: one batch item must stay below 2 GiB");
  ConvArgs a;
  a.in = in, a.wpack = wpack, a.bias = bias, a.skip = nullptr, a.out = out, a.out2 = nullptr;
  a.w2 = wpack_top, a.b2 = bias_top;
  a.B = B, a.Cin = 32, a.D = 1, a.H = H, a.W = W, a.Cout = 32, a.Do = 1, a.Ho = H, a.Wo = W;
  a.slope = act_slope, a.channels_last = out_layout, a.band_map = out_layout == 3;
  using T = ConvTile<1, 3, 1, 2, 4, 0, false>;
  dim3 grid(a.band_map ? 8u * cdiv(W, 16) * cdiv(cdiv(H, T::TY), 8) * B : cdiv(W, 16) * cdiv(H, T::TY) * B, 1);
  hipLaunchKernelGGL((conv_mfma_kernel<1, 3, 1, 2, 4, 0, false, true>), grid, dim3(256), 0, as_stream(stream), a);
  BMV_LAUNCH_END("conv_top_fwd");
}

int bmv_conv0_fused_fwd(const float* in, const float* w0, const float* b0, const float* wpack, const float* bias,
                        float* out, int B, int Cout, int H, int W, float slope0, float slope1, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(in && w0 && b0 && wpack && bias && out, "conv0_fused: null pointer");
  BMV_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout <= 8, "conv0_fused: bad shape (Cout=%d)", Cout);
  BMV_REQUIRE((size_t)3 * H * W < (1u << 29), "conv0_fused: one batch item must stay below 2 GiB");
  Conv0Args a;
  a.in = in, a.w0 = w0, a.b0 = b0, a.wpack = wpack, a.bias = bias, a.out = out;
  a.B = B, a.Cout = Cout, a.H = H, a.W = W, a.slope0 = slope0, a.slope1 = slope1;
  const DeferredPtr din = deferred_for(in);
  a.table = din.table, a.in_slot = din.slot;
  const int rows = bmv::tuning("BMV_CONV0_R", 4);
  hipStream_t st = as_stream(stream);
  if (rows == 8) {
    using T = ConvTile<1, 3, 1, 1, 8, 0, true>;
    hipLaunchKernelGGL(conv0_fused_kernel<8>, dim3(cdiv(W, 16) * cdiv(H, T::TY) * B), dim3(256), 0, st, a);
  } else {
    using T = ConvTile<1, 3, 1, 1, 4, 0, true>;
    hipLaunchKernelGGL(conv0_fused_kernel<4>, dim3(cdiv(W, 16) * cdiv(H, T::TY) * B), dim3(256), 0, st, a);
  }
  BMV_LAUNCH_END("conv0_fused_fwd");
}

int bmv_fpn_smooth_fwd(const float* fine, const float* coarse, const float* w_lat, const float* b_lat,
                       const float* wpack, const float* bias, float* out, const float* rgb, float* packed_out, int B,
                       int Cf, int C, int Cout, int H, int W, float act_slope, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(fine && coarse && w_lat && b_lat && wpack && bias && (out || packed_out), "fpn_smooth: null pointer");
  BMV_REQUIRE(!packed_out || (rgb && Cout == 8), "fpn_smooth: lookup records need the source colours and 8 channels");
  BMV_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "fpn_smooth: bad shape");
  BMV_REQUIRE(Cf == 8 && C % 4 == 0 && C > 0 && Cout > 0 && Cout <= 8,
              "fpn_smooth: built for 8 lateral channels, C %% 4 == 0 and <= 8 output channels (Cf=%d, C=%d, Cout=%d)", Cf,
              C, Cout);
  BMV_REQUIRE((size_t)C * (H / 2) * (W / 2) < (1u << 29), "fpn_smooth: one coarse batch item must stay below 2 GiB");
  FpnSmoothArgs a;
  a.fine = fine, a.coarse = coarse, a.wlat = w_lat, a.blat = b_lat, a.wpack = wpack, a.bias = bias, a.out = out;
  a.rgb = rgb, a.packed = packed_out;
  a.B = B, a.C = C, a.Cout = Cout, a.H = H, a.W = W, a.slope = act_slope;
  const DeferredPtr drgb = rgb ? deferred_for(rgb) : DeferredPtr{};
  a.table = drgb.table, a.rgb_slot = drgb.slot;
  const int rows = bmv::tuning("BMV_FPN_SMOOTH_R", 8);
  // BMV_FPN_SMOOTH_PERSIST=n: at most n workgroups per CU, each walking several tiles -- leaves registers for the short
  // launches of the level-0 regulariser that run beside this kernel on the second stream
  const int persist = bmv::tuning("BMV_FPN_SMOOTH_PERSIST", 0);
  hipStream_t st = as_stream(stream);
  if (rows == 4) {
    using T = ConvTile<1, 3, 1, 1, 4, 0, true>;
    a.ntiles = cdiv(W, 16) * cdiv(H, T::TY) * B;
    const int grid = persist > 0 ? min(a.ntiles, persist * 256) : a.ntiles;
    hipLaunchKernelGGL(fpn_smooth_kernel<4>, dim3(grid), dim3(256), 0, st, a);
  } else {
    using T = ConvTile<1, 3, 1, 1, 8, 0, true>;
    a.ntiles = cdiv(W, 16) * cdiv(H, T::TY) * B;
    const int grid = persist > 0 ? min(a.ntiles, persist * 256) : a.ntiles;
    hipLaunchKernelGGL(fpn_smooth_kernel<8>, dim3(grid), dim3(256), 0, st, a);
  }
  BMV_LAUNCH_END("fpn_smooth_fwd");
}

}  // extern "C"
