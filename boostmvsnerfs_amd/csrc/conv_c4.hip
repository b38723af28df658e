// 3x3x3 / 3x3 stride-1 convolutions with FEW output channels (Cout <= 12: the regularisers' first layers 32|16 -> 8,
// their 8 -> 8 + 1 heads, FeatureNet's smooth layers) on v_mfma_f32_4x4x1_16b_f32.
//   conv block of the reference          lib/networks/enerf/utils.py:10-33  (ConvBnReLU / ConvBnReLU3D)
//   MinCostRegNet / CostRegNet           lib/networks/enerf/cost_reg_net.py:4-86  (conv0, feat_conv, depth_conv)
//
// Why another kernel (round 5).  csrc/conv.hip computes every layer on 16-row MFMA tiles (v_mfma_f32_16x16x4_f32: rows =
// output channels).  A layer with 8 output channels fills half of a tile; ROW PAIRING (rows = 8 channels x 2 output rows,
// K + 1 input rows per filter column) recovers 2K / (K + 1) = 75 % for K = 3 -- and no pairing can do better (two shifted
// copies of the 27-tap cube cover at least 36 positions).  The 9-channel heads fill 9 / 16 = 56 %, the 8-channel transposed
// convolution 50 %.  These layers were ~315 us of a ~930 us frame and run at the fp32 matrix rate.
// v_mfma_f32_4x4x1_16b_f32 is 16 independent 4 x 4 x 1 outer products: 4 output channels x 4 positions per block, 64
// positions per wave-instruction, K = 1 = ONE (input channel, tap) pair, every row useful for Cout = 4 g.  Measured issue
// rates on MI355X (scripts/ubench/mfma_f32_shapes.hip -> profiles/r5/mfma_f32_shapes_ubench.txt, 4 waves per SIMD,
// independent accumulators): 32x32x2 65 cycles = 154 TFLOP/s, 16x16x4 36 cycles = 139, 4x4x1 10.3 cycles = 122 -- the small
// shape pays 12 % in issue rate and wins 1 / 0.75 (8 channels) or 0.75 / 0.56 (9 channels) in useful rows: 122 effective
// TFLOP/s against 104 / 78.  Stand-alone on the frame's layers (profiles/r5/conv_c4_layers.txt): 71 -> 59, 64 -> 56, 51 ->
// 39, 28 -> 22 us, the transposed layer 26 -> 19 and 17.5 -> 13; in the frame 356 -> 375 Mray/s.
//
// Operands.  lane l = (block l / 4, index l % 4): A[l] = W[cout 4 g + l % 4][cin][tap] (the same in all blocks), B[l] = the
// input at lane l's OWN position shifted by the tap, D[r] at lane l = out[cout 4 g + r] at lane l's position.  A k-step
// needs one new A and one new B register per lane, so operands are fetched four input channels at a time: the input tile
// sits in LDS CHANNEL-LAST per 4-channel chunk (`tile[pos][4]`: one ds_read_b128 = the B operands of 4 k-steps), the
// weights as `w[chunk][tap][g][cout 4][cin 4]` (one ds_read_b128, 4 distinct addresses per wave: broadcast).  Per tap and
// wave: 2 A reads + TZ B reads feed 8 TZ MFMAs (64 TZ matrix cycles per SIMD against 4 (2 + TZ) LDS cycles per CU).
// Lanes are mapped to positions so that every 16-lane group of a ds_read_b128 (MI355X_MICROARCH.md, LDS: {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} and the same + 32) reads 16 CONSECUTIVE x of one row: conflict-free on any row pitch.
#include <stdlib.h>

#include "bmv_common.hpp"

namespace bmv {

using f32x4c = __attribute__((ext_vector_type(4))) float;

struct C4Args {
  const float* in;      // (B, Cin, D, H, W)
  const float* wpack;   // [chunk][tap][g][cout 4][cin 4], batch norm folded in, zero padded
  const float* bias;    // (4 NG)
  float* out;           // mode 0: planar (B, Cout, D, H, W); mode 2: the renderer's volume records (B, D, H, W, 8)
  float* out2;          // mode 2: channel 8 (the depth logits), planar (B, D, H, W)
  int B, Cin, D, H, W, Cout;
  float slope;          // activation: v > 0 ? v : slope * v
  int mode;
};

// ablation builds (scripts/ablate_conv_c4.py: BMV_C4_DEFS=-DBMV_C4_ABLATE=n; timing only, wrong results): 1 no matrix
// instructions, 2 no tile loads, 4 no LDS reads per tap (tap 0's operands), 8 no weight copy, 16 no stores.  Compile-time:
// the same switches as run-time flags cost the full kernel 59 -> 76 us (the tap loop's LDS offsets stopped being immediates)
#ifndef BMV_C4_ABLATE
#define BMV_C4_ABLATE 0
#endif
constexpr int kC4Ablate = BMV_C4_ABLATE;

constexpr int kC4RS = 18;   // tile row pitch in positions: 16 outputs + halo

// NG: groups of 4 output channels; NW: waves per workgroup (a wave = 16 x by 4 y); TZ: output planes per wave; KD: 3 / 1
// VD (NG == 3, Cout == 9: feat_conv + depth_conv): the ninth channel -- the depth logit -- is 4 vector FMAs per (tap,
// chunk) on the B operands the matrix instructions use anyway, with its weights as a wave-uniform LDS read, instead of a
// third matrix group that would be 1 / 4 full: 8 instead of 12 matrix instructions per tap, the same fp32 FMA chain.
// QIN: the input is QUAD RECORDS (B, Cin/4, D, H, W, 4) -- what the plane sweep writes with OQ (csrc/sweep_quad.hip): a
// position's 4-channel chunk is ONE 16-byte load instead of four dword loads from four channel planes
template <int NG, int NW, int TZ, int KD, bool VD = false, bool QIN = false>
__global__ void __launch_bounds__(64 * NW) conv_c4_kernel(C4Args a) {
  constexpr int NGM = VD ? NG - 1 : NG;     // groups on the matrix cores
  static_assert(!VD || NG == 3, "the vector channel is channel 8 of a 9-channel layer");
  constexpr int NT = 64 * NW, TY = 4 * NW, TYH = TY + 2, TZH = TZ + KD - 1;
  constexpr int POS = TZH * TYH * kC4RS, NSLOT = (POS + NT - 1) / NT, TAPS = KD * 9;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nchunk = (a.Cin + 3) >> 2;
  f32x4c* wl = reinterpret_cast<f32x4c*>(lds);                               // [nchunk][TAPS][NG][4] x float4
  f32x4c* tile = reinterpret_cast<f32x4c*>(lds) + nchunk * TAPS * NG * 4;     // [POS] x float4 (4 input channels)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntx = (a.W + 15) >> 4, nty = (a.H + TY - 1) / TY, ntz = (a.D + TZ - 1) / TZ;
  int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  bid /= nty;
  const int tz = bid % ntz, b = bid / ntz;
  const int x0 = tx * 16, y0 = ty * TY, z0 = tz * TZ;
  const int plane = a.D * a.H * a.W;

  // tile slots of this thread: byte offset inside a channel plane, or out of range (zero padding)
  unsigned goff[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int slot = tid + NT * j;
    const int sx = slot % kC4RS, t = slot / kC4RS, sy = t % TYH, sz = t / TYH;
    const int gx = x0 - 1 + sx, gy = y0 - 1 + sy, gz = z0 - KD / 2 + sz;
    const bool ok = (slot < POS) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H) & (gz >= 0) & (gz < a.D) & !(kC4Ablate & 2);
    goff[j] = ok ? (QIN ? 16u : 4u) * (unsigned)((gz * a.H + gy) * a.W + gx) : 0x80000000u;
  }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (size_t)b * a.Cin * plane), 0, (int)(4u * (unsigned)(a.Cin * plane)), 0x00020000);
  f32x4c pre[NSLOT];
  auto load_tile = [&](int chunk) {
    if constexpr (QIN) {
      const unsigned cb = 16u * (unsigned)(chunk * plane);
#pragma unroll
      for (int j = 0; j < NSLOT; ++j)
        pre[j] = __builtin_bit_cast(f32x4c, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[j] + cb, 0, 0));
      return;
    }
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // (channels past Cin: the plane offset leaves the buffer -> 0)
        const unsigned cb = 4u * (unsigned)((chunk * 4 + c) * plane);
        pre[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, goff[j] + cb, 0, 0));
      }
  };
  load_tile(0);
  // weights of every chunk, once per workgroup -- BEHIND the first tile's loads: one exposed L2 round trip per workgroup
  // instead of two (the copy waits for its own loads before it can write LDS)
  {
    const f32x4c* src = reinterpret_cast<const f32x4c*>(a.wpack);
    if (!(kC4Ablate & 8))
      for (int i = tid; i < nchunk * TAPS * NG * 4; i += NT) wl[i] = src[i];
  }

  // lane -> position of the wave's 16 x 4 patch: the 16-lane groups of ds_read_b128 take 16 consecutive x of one row
  const unsigned lh = (unsigned)lane & 31u, odd = 0xF00F0FF0u;
  const int rb = (int)((odd >> lh) & 1u);
  const int xi = __popc((rb ? odd : ~odd) & ((1u << lh) - 1u));
  const int ry = 2 * (lane >> 5) + rb;
  const int ai = lane & 3;
  const f32x4c* bp = tile + (wave * 4 + ry) * kC4RS + xi;          // + ((z + kz) * TYH + ky) * RS + kx
  const f32x4c* ap = wl + ai;                                       // + ((chunk * TAPS + tap) * NG + g) * 4

  f32x4c acc[TZ][NG];
#pragma unroll
  for (int z = 0; z < TZ; ++z)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[z][g] = f32x4c{0.f, 0.f, 0.f, 0.f};
  const f32x4c* dp = wl + (NG - 1) * 4;     // VD: row 0 of the last group = the depth channel's 4 input-channel weights

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();   // every wave is done with the previous tile (first pass: the weights are being written)
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
      if ((j + 1) * NT <= POS || tid + NT * j < POS) tile[tid + NT * j] = pre[j];
    __syncthreads();
    if (chunk + 1 < nchunk) load_tile(chunk + 1);   // in flight under the MFMAs below
    const f32x4c* aw = ap + chunk * (TAPS * NG * 4);
#pragma unroll
    for (int kd = 0; kd < KD; ++kd)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int tap = (kd * 3 + ky) * 3 + kx;
          f32x4c A[NGM], Bv[TZ];
          constexpr bool kOneTap = (kC4Ablate & 4) != 0;
#pragma unroll
          for (int g = 0; g < NGM; ++g) A[g] = aw[((kOneTap ? 0 : tap) * NG + g) * 4];
#pragma unroll
          for (int z = 0; z < TZ; ++z) Bv[z] = bp[kOneTap ? z * TYH * kC4RS : ((z + kd) * TYH + ky) * kC4RS + kx];
          if (kC4Ablate & 1) {
#pragma unroll
            for (int z = 0; z < TZ; ++z)
#pragma unroll
              for (int g = 0; g < NGM; ++g) acc[z][g] += A[g] + Bv[z];
            continue;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int z = 0; z < TZ; ++z)
#pragma unroll
              for (int g = 0; g < NGM; ++g)
                acc[z][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[g][k], Bv[z][k], acc[z][g], 0, 0, 0);
          if constexpr (VD) {
            const f32x4c wd = dp[(chunk * TAPS + tap) * NG * 4];
#pragma unroll
            for (int z = 0; z < TZ; ++z)
#pragma unroll
              for (int k = 0; k < 4; ++k) acc[z][NG - 1][0] = fmaf(wd[k], Bv[z][k], acc[z][NG - 1][0]);
          }
        }
  }

  // epilogue: register r of group g = output channel 4 g + r at this lane's position
  const int x = x0 + xi, y = y0 + wave * 4 + ry;
  if (x >= a.W || y >= a.H) return;
  if ((kC4Ablate & 16) && acc[0][0][0] != 12345.f) return;
  const size_t cs = (size_t)plane;
#pragma unroll
  for (int z = 0; z < TZ; ++z) {
    const int zz = z0 + z;
    if (zz >= a.D) continue;
    const size_t vox = ((size_t)zz * a.H + y) * a.W + x;
    if (a.mode == 2) {
      // the renderer's volume records: 8 feature channels as one 32-byte record per voxel (the caller packed the
      // weights with the output channels in the record's even | odd order), channel 8 -- the depth logit -- planar
      float v[4 * NG];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = acc[z][g][r] + a.bias[4 * g + r];
          v[4 * g + r] = fmaxf(t, 0.f) + a.slope * fminf(t, 0.f);
        }
      f32x4c* rec = reinterpret_cast<f32x4c*>(a.out + ((size_t)b * cs + vox) * 8);
      rec[0] = f32x4c{v[0], v[1], v[2], v[3]};
      rec[1] = f32x4c{v[4], v[5], v[6], v[7]};
      if (NG > 2) a.out2[(size_t)b * cs + vox] = v[8];
    } else if (a.mode == 8) {
      // quad records (B, Cout/4, D, H, W, 4), Cout % 4 == 0: a group's 4 channels of a voxel are one 16-byte store
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        f32x4c v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = acc[z][g][r] + a.bias[4 * g + r];
          v[r] = fmaxf(t, 0.f) + a.slope * fminf(t, 0.f);
        }
        reinterpret_cast<f32x4c*>(a.out)[((size_t)b * NG + g) * cs + vox] = v;
      }
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 4 * g + r;
          if (co >= a.Cout) continue;
          const float t = acc[z][g][r] + a.bias[co];
          a.out[((size_t)b * a.Cout + co) * cs + vox] = fmaxf(t, 0.f) + a.slope * fminf(t, 0.f);
        }
    }
  }
}

template <int NG, int NW, int TZ, int KD, bool VD = false, bool QIN = false>
static int c4_launch(const C4Args& a, hipStream_t st) {
  constexpr int TY = 4 * NW, POS = (TZ + KD - 1) * (TY + 2) * kC4RS;
  const int nchunk = (a.Cin + 3) >> 2;
  const size_t lds = ((size_t)nchunk * KD * 9 * NG * 4 + POS) * 16;
  auto kern = conv_c4_kernel<NG, NW, TZ, KD, VD, QIN>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return BMV_ERR_UNSUPPORTED;
  }
  const unsigned ntx = (a.W + 15) / 16, nty = (a.H + TY - 1) / TY, ntz = (a.D + TZ - 1) / TZ;
  hipLaunchKernelGGL(kern, dim3(ntx * nty * ntz * a.B), dim3(64 * NW), lds, st, a);
  return BMV_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// ConvTranspose3d(k = 3, stride 2, padding 1, output_padding 1) with Cout <= 8 (cost_reg_net.py:23-41 conv11: 16 -> 8,
// the regularisers' last up-sampling step, + the U-Net skip) on the same 4 x 4 x 1 blocks.  Output o = 2 m + p per axis:
// p = 0: in[m] w[1];  p = 1: in[m] w[2] + in[m + 1] w[0] -- the 8 output parities of an input position take 1 + 2 + 2 +
// 4 + 2 + 4 + 4 + 8 = 27 taps, no product with a stuffed zero.  A wave owns 16 x 4 INPUT positions of one plane and ONE
// group of 4 output channels with the 8 parity accumulators (32 registers); the 8 shifted B operands of a 4-channel
// chunk are 8 ds_read_b128.  The two x parities of a lane are adjacent in the output row: one 8-byte store, rows of 16
// lanes = 128 contiguous bytes.  (The 16-row kernel of conv.hip ran this layer at 23 TF/s: half of every tile empty and
// a 144-register epilogue; 24.9 us at level 1.)
// ---------------------------------------------------------------------------------------------------------------
struct CT4Args {
  const float* in;      // (B, Cin, D, H, W)
  const float* wpack;   // [chunk][tap][g][cout 4][cin 4] (taps of the ConvTranspose3d weight, batch norm folded in)
  const float* bias;    // (4 NG)
  const float* skip;    // nullable, layout of out
  float* out;           // (B, Cout, 2 D, 2 H, 2 W)
  int B, Cin, D, H, W, Cout;
  float slope;
  int skip_quad;        // (quad-record output only) the skip is quad records too: (B, Cout/4, 2D, 2H, 2W, 4)
};

// OQ: the output as QUAD RECORDS (B, Cout/4, 2D, 2H, 2W, 4) -- a wave's 4 output channels of a voxel are one 16-byte
// store, the two x parities 32 contiguous bytes -- the layout the heads' kernel (conv_c4_kernel, QIN) stages with one
// 16-byte load per position; the skip stays planar (it is the first layer's output, which the stride-2 layer reads too)
template <int NG, int NYS, bool OQ = false>   // NG cout groups x NYS slabs of 4 input rows = waves per workgroup
__global__ void __launch_bounds__(64 * NG * NYS) convT_c4_kernel(CT4Args a) {
  constexpr int NT = 64 * NG * NYS, TY = 4 * NYS, TYH = TY + 1, RS = 17;
  constexpr int POS = 2 * TYH * RS, NSLOT = (POS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nchunk = (a.Cin + 3) >> 2;
  f32x4c* wl = reinterpret_cast<f32x4c*>(lds);                     // [nchunk][27][NG][4]
  f32x4c* tile = wl + nchunk * 27 * NG * 4;                         // [2][TYH][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = wave % NG, ys = wave / NG;
  const int ntx = (a.W + 15) >> 4, nty = (a.H + TY - 1) / TY;
  int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  bid /= nty;
  const int mz = bid % a.D, b = bid / a.D;
  const int x0 = tx * 16, y0 = ty * TY;
  const int plane = a.D * a.H * a.W;
  unsigned goff[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int slot = tid + NT * j;
    const int sx = slot % RS, t = slot / RS, sy = t % TYH, sz = t / TYH;
    const int gx = x0 + sx, gy = y0 + sy, gz = mz + sz;
    const bool ok = (slot < POS) & (gx < a.W) & (gy < a.H) & (gz < a.D);
    goff[j] = ok ? 4u * (unsigned)((gz * a.H + gy) * a.W + gx) : 0x80000000u;
  }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (size_t)b * a.Cin * plane), 0, (int)(4u * (unsigned)(a.Cin * plane)), 0x00020000);
  f32x4c pre[NSLOT];
  auto load_tile = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        pre[j][c] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, goff[j] + 4u * (unsigned)((chunk * 4 + c) * plane), 0, 0));
  };
  load_tile(0);
  {   // (the weights behind the first tile's loads, as in conv_c4_kernel)
    const f32x4c* src = reinterpret_cast<const f32x4c*>(a.wpack);
    for (int i = tid; i < nchunk * 27 * NG * 4; i += NT) wl[i] = src[i];
  }
  const unsigned lh = (unsigned)lane & 31u, odd = 0xF00F0FF0u;
  const int rb = (int)((odd >> lh) & 1u);
  const int xi = __popc((rb ? odd : ~odd) & ((1u << lh) - 1u));
  const int ry = 2 * (lane >> 5) + rb;
  const f32x4c* bp = tile + (ys * 4 + ry) * RS + xi;                // + (oz * TYH + oy) * RS + ox
  const f32x4c* ap = wl + g * 4 + (lane & 3);                       // + (chunk * 27 + tap) * NG * 4
  f32x4c acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = f32x4c{0.f, 0.f, 0.f, 0.f};
  const bool z1 = mz + 1 < a.D;    // the second input plane exists (else its taps multiply zeros: skipped, wave-uniform)
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
      if ((j + 1) * NT <= POS || tid + NT * j < POS) tile[tid + NT * j] = pre[j];
    __syncthreads();
    if (chunk + 1 < nchunk) load_tile(chunk + 1);
    const f32x4c* aw = ap + chunk * (27 * NG * 4);
    f32x4c Bv[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) Bv[o] = bp[((o >> 2) * TYH + ((o >> 1) & 1)) * RS + (o & 1)];
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
            const f32x4c A = aw[((kz * 3 + ky) * 3 + kx) * NG * 4];
            const f32x4c Bq = Bv[oz * 4 + oy * 2 + ox];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[k], Bq[k], acc[q], 0, 0, 0);
          }
      }
    }
  }
  const int mx = x0 + xi, my = y0 + ys * 4 + ry;
  if (mx >= a.W || my >= a.H) return;
  const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
  const size_t cs = (size_t)Do * Ho * Wo;
  if constexpr (OQ) {
    float bs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bs[r] = a.bias[4 * g + r];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const size_t sp = (size_t)(2 * mz + (qq >> 1)) * Ho * Wo + (size_t)(2 * my + (qq & 1)) * Wo + 2 * mx;
      float2 sk[4];
      if (a.skip && a.skip_quad) {
        const f32x4c* sq = reinterpret_cast<const f32x4c*>(a.skip) + ((size_t)b * NG + g) * cs + sp;
        const f32x4c s0 = sq[0], s1 = sq[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) sk[r] = make_float2(s0[r], s1[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sk[r] = a.skip ? *reinterpret_cast<const float2*>(a.skip + ((size_t)b * a.Cout + 4 * g + r) * cs + sp) : make_float2(0.f, 0.f);
      }
      f32x4c v0, v1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t0 = acc[2 * qq][r] + bs[r], t1 = acc[2 * qq + 1][r] + bs[r];
        v0[r] = fmaxf(t0, 0.f) + a.slope * fminf(t0, 0.f) + sk[r].x;
        v1[r] = fmaxf(t1, 0.f) + a.slope * fminf(t1, 0.f) + sk[r].y;
      }
      f32x4c* rec = reinterpret_cast<f32x4c*>(a.out) + ((size_t)b * NG + g) * cs + sp;
      rec[0] = v0, rec[1] = v1;
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = 4 * g + r;
    if (co >= a.Cout) continue;
    const float bs = a.bias[co];
    float2 sk[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const size_t o = (((size_t)b * a.Cout + co) * Do + 2 * mz + (qq >> 1)) * (size_t)Ho * Wo + (size_t)(2 * my + (qq & 1)) * Wo + 2 * mx;
      sk[qq] = a.skip ? *reinterpret_cast<const float2*>(a.skip + o) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const size_t o = (((size_t)b * a.Cout + co) * Do + 2 * mz + (qq >> 1)) * (size_t)Ho * Wo + (size_t)(2 * my + (qq & 1)) * Wo + 2 * mx;
      float v0 = acc[2 * qq][r] + bs, v1 = acc[2 * qq + 1][r] + bs;
      v0 = fmaxf(v0, 0.f) + a.slope * fminf(v0, 0.f), v1 = fmaxf(v1, 0.f) + a.slope * fminf(v1, 0.f);
      *reinterpret_cast<float2*>(a.out + o) = make_float2(v0 + sk[qq].x, v1 + sk[qq].y);
    }
  }
  (void)cs;
}

template <int NG, int NYS, bool OQ = false>
static int ct4_launch(const CT4Args& a, hipStream_t st) {
  constexpr int TY = 4 * NYS, POS = 2 * (TY + 1) * 17;
  const int nchunk = (a.Cin + 3) >> 2;
  const size_t lds = ((size_t)nchunk * 27 * NG * 4 + POS) * 16;
  auto kern = convT_c4_kernel<NG, NYS, OQ>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return BMV_ERR_UNSUPPORTED;
  }
  const unsigned ntx = (a.W + 15) / 16, nty = (a.H + TY - 1) / TY;
  hipLaunchKernelGGL(kern, dim3(ntx * nty * a.D * a.B), dim3(64 * NG * NYS), lds, st, a);
  return BMV_OK;
}

}  // namespace bmv

using namespace bmv;

extern "C" {

// floats of the packed weights for bmv_conv_c4_fwd: [chunk][tap][g][cout 4][cin 4]
int bmv_conv_c4_wpack_floats(int Cout, int Cin, int kd) {
  if (Cout < 1 || Cout > 12 || Cin < 1 || (kd != 1 && kd != 3)) {
    set_error("bmv_conv_c4_wpack_floats: Cout=%d (1..12), Cin=%d, kd=%d (1 or 3)", Cout, Cin, kd);
    return BMV_ERR_UNSUPPORTED;
  }
  return ((Cin + 3) / 4) * kd * 9 * ((Cout + 3) / 4) * 16;
}

int bmv_conv_c4_fwd(const float* in, const float* wpack, const float* bias, float* out, float* out2, int B, int Cin, int D,
                    int H, int W, int Cout, int kd, float slope, int mode, int variant, bmv_stream_t stream) {
  BMV_REQUIRE(in && wpack && bias && out, "bmv_conv_c4_fwd: null pointer");
  BMV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0, "bmv_conv_c4_fwd: bad shape");
  const bool qin = (mode & 4) != 0;      // input as quad records
  mode &= ~4;
  BMV_REQUIRE(mode == 0 || (mode == 2 && (Cout == 8 || (Cout == 9 && out2))) || (mode == 8 && Cout % 4 == 0),
              "bmv_conv_c4_fwd: mode=%d with Cout=%d", mode, Cout);
  BMV_REQUIRE(!qin || (Cin % 4 == 0), "bmv_conv_c4_fwd: quad-record input needs Cin %% 4 == 0 (Cin=%d)", Cin);
  if (Cout < 1 || Cout > 12 || (kd != 1 && kd != 3) || (kd == 1 && D != 1) ||
      (size_t)Cin * D * H * W * 4 >= ((size_t)1 << 31) || (size_t)((Cin + 3) / 4) * kd * 9 * ((Cout + 3) / 4) * 64 > 100 * 1024) {
    set_error("bmv_conv_c4_fwd: shape not covered (Cout=%d Cin=%d kd=%d %dx%dx%d)", Cout, Cin, kd, D, H, W);
    return BMV_ERR_UNSUPPORTED;
  }
  C4Args a{in, wpack, bias, out, out2, B, Cin, D, H, W, Cout, slope, mode};
  const int ng = (Cout + 3) / 4;
  int rc = BMV_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  // variant: 0 = default tile of the shape; 1.. = tuning (tests / scripts/bench_conv_c4.py)
#define C4(NGV, NWV, TZV, KDV) \
  if (ng == NGV && kd == KDV) rc = c4_launch<NGV, NWV, TZV, KDV>(a, st)
  if (kd == 3) {
    // (measured on the frame's four layers, profiles/r5/conv_c4_layers.txt: one plane per wave -- 58 registers,
    // 7-8 waves per SIMD -- beats two planes per wave at 106 registers everywhere: 56.8 / 54.2 / 38.3 / 22.1 us against
    // 60.0 / 54.5 / 41.7 / 25.0)
    if (qin) {
      // (the regularisers' first layer behind the plane sweep, the heads behind conv11: the default tiles only)
      if (variant == 0 && Cout == 9) rc = c4_launch<3, 4, 1, 3, true, true>(a, st);
      else if (variant == 0 && ng == 3) rc = c4_launch<3, 4, 1, 3, false, true>(a, st);
      if (variant == 0 && ng == 2) rc = c4_launch<2, 4, 1, 3, false, true>(a, st);
      if (variant == 0 && ng == 1) rc = c4_launch<1, 4, 1, 3, false, true>(a, st);
    } else if (variant == 0 && Cout == 9) {
      rc = c4_launch<3, 4, 1, 3, true>(a, st);       // the 9-channel heads: 8 channels on the matrix cores + 1 on the vector ALU
    } else if (variant == 0 || variant == 3) {
      C4(1, 4, 1, 3);
      C4(2, 4, 1, 3);
      C4(3, 4, 1, 3);
    } else if (variant == 1) {
      C4(1, 4, 2, 3);
      C4(2, 4, 2, 3);
      C4(3, 4, 2, 3);
    } else if (variant == 2) {
      C4(1, 2, 2, 3);
      C4(2, 2, 2, 3);
      C4(3, 2, 2, 3);
    } else if (variant == 4) {
      C4(2, 4, 4, 3);
      C4(3, 2, 4, 3);
    }
  } else if (!qin) {
    if (variant == 0 || variant == 1) {
      C4(1, 4, 1, 1);
      C4(2, 4, 1, 1);
      C4(3, 4, 1, 1);
    } else if (variant == 2) {
      C4(1, 2, 1, 1);
      C4(2, 2, 1, 1);
      C4(3, 2, 1, 1);
    }
  }
#undef C4
  if (rc == BMV_ERR_UNSUPPORTED) set_error("bmv_conv_c4_fwd: variant %d not instantiated for Cout=%d kd=%d", variant, Cout, kd);
  if (rc != BMV_OK) return rc;
  BMV_LAUNCH_END("bmv_conv_c4_fwd");
}

// ConvTranspose3d(k = 3, stride 2, padding 1, output_padding 1) for Cout <= 8 on the 4 x 4 x 1 blocks:
// in (B,Cin,D,H,W) -> out (B,Cout,2D,2H,2W) = act(convT(in) + bias) + skip; wpack as bmv_conv_c4_fwd's (kd = 3) with the
// taps of the transposed weight (Cin,Cout,3,3,3) -> [chunk][tap][g][cout][cin]
int bmv_conv3d_transpose_c4_fwd(const float* in, const float* wpack, const float* bias, const float* skip, float* out, int B, int Cin,
                     int D, int H, int W, int Cout, float slope, int variant, bmv_stream_t stream) {
  BMV_REQUIRE(in && wpack && bias && out, "bmv_conv3d_transpose_c4_fwd: null pointer");
  BMV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0, "bmv_conv3d_transpose_c4_fwd: bad shape");
  if (Cout < 1 || Cout > 8 || (size_t)Cin * D * H * W * 4 >= ((size_t)1 << 31) || (size_t)Cout * D * H * W * 32 >= ((size_t)1 << 33) ||
      (size_t)((Cin + 3) / 4) * 27 * ((Cout + 3) / 4) * 64 > 100 * 1024) {
    set_error("bmv_conv3d_transpose_c4_fwd: shape not covered (Cout=%d Cin=%d %dx%dx%d)", Cout, Cin, D, H, W);
    return BMV_ERR_UNSUPPORTED;
  }
  CT4Args a{in, wpack, bias, skip, out, B, Cin, D, H, W, Cout, slope, (variant & 32) ? 1 : 0};
  BMV_REQUIRE(!(variant & 32) || (variant & 16), "bmv_conv3d_transpose_c4_fwd: a quad-record skip goes with the quad-record output");
  variant &= ~32;
  const int ng = (Cout + 3) / 4;
  hipStream_t st = as_stream(stream);
  int rc = BMV_ERR_UNSUPPORTED;
  if (variant & 16) {      // out as quad records (default tiling)
    BMV_REQUIRE(Cout == 8 && (variant & 15) == 0, "bmv_conv3d_transpose_c4_fwd: quad-record output is built for Cout = 8, variant 0");
    rc = ct4_launch<2, 2, true>(a, st);
    if (rc == BMV_ERR_UNSUPPORTED) set_error("bmv_conv3d_transpose_c4_fwd: cannot reserve LDS");
    if (rc != BMV_OK) return rc;
    BMV_LAUNCH_END("bmv_conv3d_transpose_c4_fwd");
  }
  if (ng == 2) rc = variant == 1 ? ct4_launch<2, 1>(a, st) : variant == 2 ? ct4_launch<2, 4>(a, st) : ct4_launch<2, 2>(a, st);
  if (ng == 1) rc = variant == 1 ? ct4_launch<1, 1>(a, st) : variant == 2 ? ct4_launch<1, 4>(a, st) : ct4_launch<1, 2>(a, st);
  if (rc != BMV_OK) {
    if (rc == BMV_ERR_UNSUPPORTED) set_error("bmv_conv3d_transpose_c4_fwd: cannot reserve LDS");
    return rc;
  }
  BMV_LAUNCH_END("bmv_conv3d_transpose_c4_fwd");
}

}  // extern "C"
