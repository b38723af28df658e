// 3x3x3 stride-1 convolutions with FEW output channels (the regularisers' first layers 32 | 16 -> 8 and their 8 -> 8 + 1
// heads) on the BF16 matrix cores with THREE-PIECE fp32 operands, staged from 16-byte QUAD RECORDS (round 6).
//   conv block of the reference          lib/networks/enerf/utils.py:10-33  (ConvBnReLU3D)
//   MinCostRegNet / CostRegNet           lib/networks/enerf/cost_reg_net.py:4-86  (conv0, feat_conv, depth_conv)
//
// Why.  csrc/conv_c4.hip runs these four layers of a frame on v_mfma_f32_4x4x1 at the shape's peak (36 of a first layer's
// 50 us are matrix instructions: 122 TFLOP/s); they are 152 us of the 0.79 ms frame's critical path.  An fp32 product
// can be evaluated on the bf16 pipe at fp32 accuracy: x = hi + mid + lo (three bf16 pieces = the 24 significand bits
// EXACTLY: hi = the upper 16 bits of x, mid = the upper 16 bits of x - hi, lo = x - hi - mid, which has at most 8
// significant bits left), x w = hh + hm + mh + hl + mm + lh with fp32 accumulation, smallest terms first; the dropped ml,
// lm, ll are <= 3 x 2^-24 of a product, the size of ONE fp32 rounding of it (csrc/conv_split.hip, round 3; csrc/mlp.hpp,
// round 5: the renderer's MLP).  Six v_mfma_f32_16x16x32_bf16 (96 cycles) do 32 k-values for a 16 x 16 tile where the
// 4-row fp32 blocks need 330: 3.4 x the rate.  conv_split.hip had that arithmetic but staged planar dword tiles without
// prefetch and lost to conv_c4 in the frame; this kernel has conv_c4's skeleton:
//   * the input is quad records (B, Cin/4, D, H, W, 4) -- what the plane sweep (conv0) and conv11 (heads) write: a position's
//     8-channel octet is TWO 16-byte loads; the NEXT octet's loads are in flight under the matrix instructions;
//   * every staged value is split ONCE (not per use: 27 taps x 8 output channels) and parked in LDS as three position-major
//     planes piece[pos][8 x bf16]: a B operand is one ds_read_b128 per piece;
//   * k = (tap slot, channel of the octet): a lane's 8 k-values are the octet at ONE tap slot, the four lane groups four
//     consecutive slots; the accumulator layout is the fp32 16x16x4 form's (row = 4 (lane >> 4) + j, column = lane & 15);
//   * ROW PAIRING for the 8-channel layers: the 16 matrix rows are 8 channels x 2 adjacent output rows, k walks 3 x 4 x 3
//     tap slots (4 input rows) = 36 = exactly 9 steps of 4: every row and every lane useful, 2 / 3 of the unpaired
//     instructions; the 9-channel heads run unpaired (9 of 16 rows, 27 + 1 slots = 7 steps);
//   * the A operands (weights, split and laid out in lane order on the host) are read from global memory (L2-resident:
//     27 KB per octet) three steps ahead, requested BEFORE the next tile's loads so that waiting for them (vmcnt is
//     in order) does not wait for the tile.
#include <stdlib.h>

#include "bmv_common.hpp"

namespace bmv {

using f32x4s = __attribute__((ext_vector_type(4))) float;
using i32x4s = __attribute__((ext_vector_type(4))) int;
using bf16x8s = __attribute__((ext_vector_type(8))) __bf16;

struct C4SArgs {
  const float* in;      // quad records (B, Cin/4, D, H, W, 4), Cin % 8 == 0
  const int* wsplit;    // [octet][step][piece 3][lane 64][4 dwords]
  const float* bias;    // (16)
  float* out;           // mode 0: planar (B, Cout, D, H, W); mode 8: quad records (B, Cout/4, D, H, W, 4); mode 2: the
                        // renderer's volume records (B, D, H, W, 8) of channels 0..7
  float* out2;          // mode 2: channel 8 (the depth logits), planar (B, D, H, W)
  int B, Cin, D, H, W, Cout;
  float slope;          // activation: v > 0 ? v : slope * v
  int mode;
};

constexpr int kS_RS = 18, kS_TYH = 18, kS_POS = 3 * kS_TYH * kS_RS, kS_NSLOT = (kS_POS + 255) / 256;

__device__ __forceinline__ unsigned s_pack_hi(float a, float b) {   // [bf16(a) | bf16(b) << 16] by truncation
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ unsigned s_pack_rne(float a, float b) {  // round to nearest even (exact here: <= 8 bits left)
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float s_trunc(float v) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u);
}

// PAIR: Cout == 8, matrix rows = (output row y + (m >> 3), channel m & 7); else rows = channel m (Cout <= 16)
// A workgroup = 4 waves = 16 x by 16 y outputs of ONE plane; a wave = 4 rows = 2 row pairs (PAIR) / 4 rows
template <bool PAIR>
__global__ void __launch_bounds__(256) conv_c4s_kernel(C4SArgs a) {
  constexpr int STEPS = PAIR ? 9 : 7, NQ = PAIR ? 2 : 4, WD = 3;   // WD: steps the A operands are requested ahead
  extern __shared__ i32x4s s_part[];                                // [3][kS_POS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kk = lane >> 4;
  const int ntx = (a.W + 15) >> 4, nty = (a.H + 15) >> 4;
  int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  bid /= nty;
  const int z0 = bid % a.D, b = bid / a.D;
  const int x0 = tx * 16, y0 = ty * 16;
  const int plane = a.D * a.H * a.W;
  const int nocts = a.Cin >> 3;

  // tile slots of this thread (a slot = one position of the 3 x 18 x 18 input box): byte offset of its record inside a
  // channel quad's block, or out of range (zero padding / past the box)
  unsigned goff[kS_NSLOT];
#pragma unroll
  for (int j = 0; j < kS_NSLOT; ++j) {
    const int slot = tid + 256 * j;
    const int sx = slot % kS_RS, t = slot / kS_RS, sy = t % kS_TYH, sz = t / kS_TYH;
    const int gx = x0 - 1 + sx, gy = y0 - 1 + sy, gz = z0 - 1 + sz;
    const bool ok = (slot < kS_POS) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H) & (gz >= 0) & (gz < a.D);
    goff[j] = ok ? 16u * (unsigned)((gz * a.H + gy) * a.W + gx) : 0x80000000u;
  }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (size_t)b * a.Cin * plane), 0, (int)(4u * (unsigned)(a.Cin * plane)), 0x00020000);
  f32x4s pre[kS_NSLOT][2];
  auto load_tile = [&](int oct) {
    const unsigned cb = 16u * (unsigned)(2 * oct * plane);
#pragma unroll
    for (int j = 0; j < kS_NSLOT; ++j) {
      pre[j][0] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[j] + cb, 0, 0));
      pre[j][1] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[j] + cb + 16u * (unsigned)plane, 0, 0));
    }
  };
  load_tile(0);

  // per-lane tap offsets (positions) of the steps: slot t = 4 g + kk
  int tapoff[STEPS];
#pragma unroll
  for (int g = 0; g < STEPS; ++g) {
    const int t = 4 * g + kk;
    if (PAIR)
      tapoff[g] = ((t / 12) * kS_TYH + (t / 3) % 4) * kS_RS + t % 3;      // (kz, ky' in 0..3, kx)
    else
      tapoff[g] = t < 27 ? ((t / 9) * kS_TYH + (t / 3) % 3) * kS_RS + t % 3 : 0;   // (slot 27: zero weights)
  }
  // origins of this wave's 16-wide output pieces: PAIR: row pairs (4 wave + 2 q, + 1); else rows 4 wave + q
  int pbase[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) pbase[q] = (4 * wave + (PAIR ? 2 * q : q)) * kS_RS + n;

  f32x4s acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = f32x4s{0.f, 0.f, 0.f, 0.f};

  const i32x4s* __restrict__ wp = reinterpret_cast<const i32x4s*>(a.wsplit) + lane;
  for (int oct = 0; oct < nocts; ++oct) {
    __syncthreads();      // every wave is done with the previous octet's pieces
#pragma unroll
    for (int j = 0; j < kS_NSLOT; ++j) {
      if ((j + 1) * 256 > kS_POS && tid + 256 * j >= kS_POS) continue;
      i32x4s pc[3];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float v0 = pre[j][h][2 * e], v1 = pre[j][h][2 * e + 1];
          pc[0][2 * h + e] = (int)s_pack_hi(v0, v1);
          const float r0 = v0 - s_trunc(v0), r1 = v1 - s_trunc(v1);
          pc[1][2 * h + e] = (int)s_pack_hi(r0, r1);
          pc[2][2 * h + e] = (int)s_pack_rne(r0 - s_trunc(r0), r1 - s_trunc(r1));
        }
#pragma unroll
      for (int p = 0; p < 3; ++p) s_part[p * kS_POS + tid + 256 * j] = pc[p];
    }
    __syncthreads();
    // A operands of the first WD steps, then the next octet's tile: waiting for step g's weights (vmcnt is in order)
    // meets the tile's loads only from step WD on
    const i32x4s* __restrict__ wo = wp + (size_t)oct * STEPS * 3 * 64;
    i32x4s wn[WD][3];
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
      for (int p = 0; p < 3; ++p) wn[d][p] = wo[(d * 3 + p) * 64];
    if (oct + 1 < nocts) load_tile(oct + 1);
#pragma unroll
    for (int g = 0; g < STEPS; ++g) {
      bf16x8s aw[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) aw[p] = __builtin_bit_cast(bf16x8s, wn[g % WD][p]);
      if (g + WD < STEPS) {
#pragma unroll
        for (int p = 0; p < 3; ++p) wn[g % WD][p] = wo[((g + WD) * 3 + p) * 64];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int pos = pbase[q] + tapoff[g];
        bf16x8s bx[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bx[p] = __builtin_bit_cast(bf16x8s, s_part[p * kS_POS + pos]);
        // smallest terms first: (lo, hi), (mid, mid), (hi, lo), (mid, hi), (hi, mid), (hi, hi)
#pragma unroll
        for (int sum = 2; sum >= 0; --sum)
#pragma unroll
          for (int i = 0; i <= sum; ++i) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[i], bx[sum - i], acc[q], 0, 0, 0);
      }
    }
  }

  // epilogue: accumulator j of lane (n, kk) = matrix row 4 kk + j at x = n of the piece
  const int x = x0 + n;
  if (x >= a.W) return;
  const size_t cs = (size_t)plane;
  const int cg = PAIR ? (kk & 1) : kk;                 // group of 4 output channels this lane holds
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[4 * cg + j];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int y = y0 + 4 * wave + (PAIR ? 2 * q + (kk >> 1) : q);
    if (y >= a.H) continue;
    const size_t vox = ((size_t)z0 * a.H + y) * a.W + x;
    f32x4s v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = acc[q][j] + bs[j];
      v[j] = fmaxf(t, 0.f) + a.slope * fminf(t, 0.f);
    }
    if (a.mode == 2) {          // the renderer's volume records (channels 0..7) + the planar depth logits (channel 8)
      if (cg < 2)
        *reinterpret_cast<f32x4s*>(a.out + ((size_t)b * cs + vox) * 8 + 4 * cg) = v;
      else if (cg == 2 && a.out2)
        a.out2[(size_t)b * cs + vox] = v[0];
    } else if (a.mode == 8) {   // quad records (B, Cout/4, D, H, W, 4)
      if (4 * cg < a.Cout) reinterpret_cast<f32x4s*>(a.out)[((size_t)b * (a.Cout >> 2) + cg) * cs + vox] = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = 4 * cg + j;
        if (co < a.Cout) a.out[((size_t)b * a.Cout + co) * cs + vox] = v[j];
      }
    }
  }
}

}  // namespace bmv

using namespace bmv;

extern "C" {

// int32 words of the split weights of bmv_conv_c4s_fwd: [octet][step][piece 3][lane 64][4]; pair = the row-paired form
// (Cout == 8: 9 steps), else 7 steps
int bmv_conv_c4s_wsplit_ints(int Cin, int pair) {
  if (Cin < 8 || (Cin & 7)) {
    set_error("bmv_conv_c4s_wsplit_ints: Cin=%d must be a multiple of 8", Cin);
    return BMV_ERR_UNSUPPORTED;
  }
  return (Cin / 8) * (pair ? 9 : 7) * 3 * 64 * 4;
}

// 3x3x3 convolution, stride 1, padding 1, on the bf16 matrix cores with three-piece fp32 operands (fp32 accuracy):
// in = quad records (B, Cin/4, D, H, W, 4); out by mode (0 planar, 8 quad records, 2 volume records + out2 = channel 8);
// pair != 0: Cout == 8, weights packed row-paired (boostmvsnerfs_amd/convnet.py pack_conv_c4s)
int bmv_conv_c4s_fwd(const float* in, const int* wsplit, const float* bias, float* out, float* out2, int B, int Cin, int D,
                     int H, int W, int Cout, int pair, float slope, int mode, bmv_stream_t stream) {
  BMV_REQUIRE(in && wsplit && bias && out, "bmv_conv_c4s_fwd: null pointer");
  BMV_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "bmv_conv_c4s_fwd: bad shape");
  BMV_REQUIRE(mode == 0 || (mode == 2 && (Cout == 8 || (Cout == 9 && out2))) || (mode == 8 && Cout % 4 == 0),
              "bmv_conv_c4s_fwd: mode=%d with Cout=%d", mode, Cout);
  if (Cin < 8 || (Cin & 7) || Cout < 1 || Cout > 16 || (pair && Cout != 8) ||
      (size_t)Cin * D * H * W * 4 >= ((size_t)1 << 31) || (size_t)Cout * D * H * W * 4 >= ((size_t)1 << 32)) {
    set_error("bmv_conv_c4s_fwd: shape not covered (Cin=%d (%%8), Cout=%d (<= 16; 8 when paired), %dx%dx%d)", Cin, Cout, D, H, W);
    return BMV_ERR_UNSUPPORTED;
  }
  C4SArgs a{in, wsplit, bias, out, out2, B, Cin, D, H, W, Cout, slope, mode};
  const size_t lds = (size_t)3 * kS_POS * sizeof(i32x4s);
  const unsigned grid = (unsigned)(((W + 15) / 16) * ((H + 15) / 16) * D * B);
  hipStream_t st = as_stream(stream);
  static_assert((size_t)3 * kS_POS * sizeof(i32x4s) <= 64 * 1024, "dynamic LDS within the default limit: no attribute call");
  if (pair)
    hipLaunchKernelGGL(conv_c4s_kernel<true>, dim3(grid), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL(conv_c4s_kernel<false>, dim3(grid), dim3(256), lds, st, a);
  BMV_LAUNCH_END("bmv_conv_c4s_fwd");
}

}  // extern "C"
