// EXPERIMENT (round 3, opt-in: BMV_CONV_SPLIT=2|3|auto): a 3x3x3 stride-1 convolution on the BF16 matrix cores with SPLIT
// fp32 operands.  Two pieces: x = hi + lo with hi = the upper 16 bits of x (a bf16 value) and lo = bf16(x - hi); a product
// is hi*hi + hi*lo + lo*hi with fp32 accumulation (relative error <= 2^-16; the dropped lo*lo term is 2^-16 of it).
// Three pieces: hi + mid + lo = the 24 mantissa bits exactly; hh, hm, mh, hl, mm, lh (what is dropped is <= 3 x 2^-24 of
// the product: fp32-equivalent).  tests/tools/probe_split_bf16.py emulates exactly this arithmetic in EVERY convolution
// of the network on the CPU oracle: the whole frame moves by <= 1.3e-5 relative with two pieces and 1e-6 with three (the
// bar is 1e-3; the fp32 engine's own distance to the oracle is 5e-6).  Why: v_mfma_f32_16x16x32_bf16 retires 32
// k-values in 16 cycles, the fp32 form 4 in 32: three (six) bf16 MFMAs replace eight fp32 ones (48 / 96 vs 256 cycles per
// 32 k-values), and the first / last layers of the regularisers are MFMA-bound at 0.61-0.64 busy.  The
// reference on an NVIDIA GPU runs these convolutions in TF32 (10-bit mantissas) by default.
//
// Layout.  k = (tap, channel): the 8 k-values of a lane are the 8 channels of one OCTET at one tap, the four lane groups
// (lane >> 4) are four consecutive taps -- so a B operand is ONE 16-byte LDS read per piece and lane, from position-major
// planes piece[pos][8 x bf16] (16-byte stride: conflict-free).  The split happens ONCE, when a tile is staged, not per
// use (27 taps x output tiles).  A operand: the weights, split and laid out on the host in exactly the lane order, one
// 16-byte load per piece and step.  Accumulators have the layout of the fp32 16x16x4 form (row = 4 (lane >> 4) + j =
// output channel, column = lane & 15 = x), so epilogues carry over.
// Tile: TZ x (4 RW) x TX outputs per workgroup of 4 waves (default 2 x 4 x 32), input tile (TZ + 2) x (4 RW + 2) x (TX + 2)
// positions per octet, staged with 16-byte loads of 4 consecutive pixels per channel.  The kernel is bound by the
// latency of that staging, and small tiles -- many workgroups taking turns on a CU -- hide it best.
#include "bmv_common.hpp"

namespace bmv {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

struct ConvSplitArgs {
  const float* in;      // (B, Cin, D, H, W)
  const int* wsplit;    // [octet][step 7][part 2][lane 64][4 dwords]
  const float* bias;    // (16)
  float* out;           // (B, Cout, D, H, W); or, with `depth_out`, the renderer's volume records (B, D, H, W, 8)
  float* depth_out;     // (B, D, H, W): heads form -- channels 0..7 -> one 32-byte record per voxel, channel 8 -> here
  int B, Cin, Cout, D, H, W;
  float slope;
};   // wsplit: [octet][step 7][part P][lane 64][4 dwords], P = 2 (hi, lo) or 3 (hi, mid, lo)

constexpr int kSteps = 7;                                                            // 28 tap slots

__device__ __forceinline__ unsigned pack_hi(float a, float b) {   // [bf16(a) | bf16(b) << 16] by truncation
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ unsigned pack_lo(float a, float b) {   // round to nearest even
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float trunc_bf16(float v) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u);
}

// TZ output planes per workgroup (input planes TZ + 2: the z halo is read (TZ + 2) / TZ times instead of 3 times)
// P parts per operand: 2 = hi + lo, products hh, hl, lh (2^-16 per product); 3 = hi + mid + lo (the 24 bits of an fp32
// mantissa in three bf16 pieces: x is represented EXACTLY), products hh, hm, mh, hl, mm, lh -- what is dropped (ml, lm, ll)
// is <= 3 x 2^-24 of the product, the size of an fp32 rounding
// (Tried and dropped: 4 MFMA waves + 4 staging waves per workgroup with a double-buffered tile -- 88 / 108 us for the
// level-1 first layer against 55 / 78: the kernel is bound by the latency of staging, and half the threads staging with
// one workgroup per CU is worse than three whole workgroups taking turns.)
// TX = 32 or 16 output columns per tile (16: half the LDS per workgroup -> more workgroups taking turns per CU)
// RW = 2 or 1 output rows per wave (tile height 8 or 4: smaller tiles, more workgroups per CU)
template <int TZ, int P, int TX, int RW>
__global__ void __launch_bounds__(256) conv3d_split_kernel(ConvSplitArgs a) {
  constexpr int kTY = 4 * RW, kIY = kTY + 2;
  constexpr int kTX = TX, kIX = kTX + 2, kPlane = kIY * kIX, kGX = (kTX + 8) / 4, XH = TX / 16;
  constexpr int NPL = TZ + 2, kPos = NPL * kPlane, NG = NPL * kIY * kGX, NSLOT = (NG + 255) / 256, NQ = RW * XH * TZ;
  extern __shared__ i32x4 part_raw[];                      // [P][kPos]
  auto part = [&](int q) { return part_raw + (long)q * kPos; };
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kk = lane >> 4;
  const int ntx = (a.W + kTX - 1) / kTX, nty = (a.H + kTY - 1) / kTY;
  int bid = blockIdx.x;
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  const int z0 = (bid / nty) * TZ;
  const int b = blockIdx.y;
  const int x0 = tx * kTX, y0 = ty * kTY;
  const long plane = (long)a.H * a.W, vol = (long)a.D * plane;
  const int nocts = a.Cin / 8;

  // staging slots: a slot = 4 consecutive pixels (one 16-byte load per channel) of an input row, aligned to 4 (W % 4 == 0,
  // x0 % 4 == 0): pixels x0 - 4 + 4 gi .. + 3; tile position px = 4 gi + e - 3 (positions outside [0, 34) are not kept)
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (long)b * a.Cin * vol), 0,
                                                                (int)((long)a.Cin * vol * 4), 0x00020000);
  unsigned goff[NSLOT];
  int pos0[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int s = tid + 256 * i;
    const int gi = s % kGX, py = (s / kGX) % kIY, pz = s / (kGX * kIY);
    const int gz = z0 - 1 + pz, gy = y0 - 1 + py, gx = x0 - 4 + 4 * gi;
    const bool ok = s < NG && gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    goff[i] = ok ? (unsigned)(((long)gz * a.H + gy) * a.W + gx) * 4u : 0x80000000u;
    pos0[i] = s < NG ? pz * kPlane + py * kIX + 4 * gi - 3 : -1000000;
  }
  const unsigned cstride = (unsigned)(vol * 4);
  // (Tried and dropped: the next octet's loads kept in registers under the MFMAs -- 176-188 registers, two workgroups per
  // CU instead of three: 75 us instead of 55 for the level-1 first layer.  Whole workgroups taking turns hide the
  // staging latency better than a deeper pipeline inside one.)
  auto stage = [&](int oct) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      f32x4 v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c)
        v[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] + (unsigned)(oct * 8 + c) * cstride, 0, 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int px = ((tid + 256 * i) % kGX) * 4 + e - 3;
        if (px < 0 || px >= kIX || pos0[i] <= -1000000) continue;
        i32x4 pc[P];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0 = v[2 * j][e], v1 = v[2 * j + 1][e];
          pc[0][j] = (int)pack_hi(v0, v1);
          const float r0 = v0 - trunc_bf16(v0), r1 = v1 - trunc_bf16(v1);
          if constexpr (P == 2) {
            pc[1][j] = (int)pack_lo(r0, r1);
          } else {
            pc[1][j] = (int)pack_hi(r0, r1);
            pc[2][j] = (int)pack_lo(r0 - trunc_bf16(r0), r1 - trunc_bf16(r1));
          }
        }
#pragma unroll
        for (int q = 0; q < P; ++q) part(q)[pos0[i] + e] = pc[q];
      }
    }
  };

  // per-lane tap offsets (positions) of the 7 steps: tap t = 4 g + kk -> (kz, ky, kx); slot 27 has zero weights
  int tapoff[kSteps];
#pragma unroll
  for (int g = 0; g < kSteps; ++g) {
    const int t = 4 * g + kk;
    tapoff[g] = t < 27 ? (t / 9) * kPlane + ((t / 3) % 3) * kIX + (t % 3) : 0;
  }
  // origin positions of this wave's 16-wide output pieces: q = (plane, row of the wave's two, 16-column piece)
  int pbase[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    pbase[q] = (q / (RW * XH)) * kPlane + (RW * wave + ((q / XH) % RW)) * kIX + 16 * (q % XH) + n;

  f32x4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};

  const i32x4* __restrict__ wp = reinterpret_cast<const i32x4*>(a.wsplit) + lane;
  for (int oct = 0; oct < nocts; ++oct) {
    stage(oct);
    __syncthreads();
    const i32x4* __restrict__ wo = wp + (long)oct * kSteps * P * 64;
    i32x4 wn[P];
#pragma unroll
    for (int q = 0; q < P; ++q) wn[q] = wo[q * 64];
#pragma unroll
    for (int g = 0; g < kSteps; ++g) {
      bf16x8 aw[P];
#pragma unroll
      for (int q = 0; q < P; ++q) aw[q] = __builtin_bit_cast(bf16x8, wn[q]);
      if (g + 1 < kSteps) {
#pragma unroll
        for (int q = 0; q < P; ++q) wn[q] = wo[((g + 1) * P + q) * 64];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int p = pbase[q] + tapoff[g];
        bf16x8 bx[P];
#pragma unroll
        for (int r = 0; r < P; ++r) bx[r] = __builtin_bit_cast(bf16x8, part(r)[p]);
        // smallest terms first
#pragma unroll
        for (int sum = P - 1; sum >= 0; --sum)
#pragma unroll
          for (int i = 0; i <= sum; ++i) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[i], bx[sum - i], acc[q], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue: accumulator j of lane (n, kk) = output channel 4 kk + j at x = n of its piece
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int z = z0 + q / (RW * XH), y = y0 + RW * wave + ((q / XH) % RW), x = x0 + 16 * (q % XH) + n;
    if (z >= a.D || y >= a.H || x >= a.W) continue;
    if (a.depth_out) {   // heads: no activation (cost_reg_net.py:30-31, 80-81 end in plain convolutions)
      const long vox = ((long)b * a.D + z) * plane + (long)y * a.W + x;
      if (kk < 2) {
        f32x4 v = acc[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += a.bias[4 * kk + j];
        *reinterpret_cast<f32x4*>(a.out + vox * 8 + 4 * kk) = v;
      } else if (kk == 2) {
        a.depth_out[vox] = acc[q][0] + a.bias[8];
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = 4 * kk + j;
      if (co >= a.Cout) continue;
      float v = acc[q][j] + a.bias[co];
      v = fmaxf(v, 0.f) + a.slope * fminf(v, 0.f);
      a.out[(((long)b * a.Cout + co) * a.D + z) * plane + (long)y * a.W + x] = v;
    }
  }
}

}  // namespace bmv

extern "C" {

int bmv_conv3d_split_wsplit_ints(int Cin, int parts) { return (Cin / 8) * bmv::kSteps * parts * 64 * 4; }

static int conv3d_split_launch(const float* in, const int* wsplit, int parts, const float* bias, float* out, float* depth_out,
                               int B, int Cin, int D, int H, int W, int Cout, float act_slope, bmv_stream_t stream);

int bmv_conv3d_split_fwd(const float* in, const int* wsplit, int parts, const float* bias, float* out, int B, int Cin, int D,
                         int H, int W, int Cout, float act_slope, bmv_stream_t stream) {
  return conv3d_split_launch(in, wsplit, parts, bias, out, nullptr, B, Cin, D, H, W, Cout, act_slope, stream);
}

int bmv_conv3d_split_heads_fwd(const float* in, const int* wsplit, int parts, const float* bias, float* records_out,
                               float* depth_out, int B, int Cin, int D, int H, int W, bmv_stream_t stream) {
  BMV_REQUIRE(depth_out, "conv3d_split_heads: null pointer");
  return conv3d_split_launch(in, wsplit, parts, bias, records_out, depth_out, B, Cin, D, H, W, 9, 1.f, stream);
}

static int conv3d_split_launch(const float* in, const int* wsplit, int parts, const float* bias, float* out, float* depth_out,
                               int B, int Cin, int D, int H, int W, int Cout, float act_slope, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(in && wsplit && bias && out, "conv3d_split: null pointer");
  BMV_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "conv3d_split: bad shape");
  BMV_REQUIRE(Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout <= 16, "conv3d_split: Cin %% 8 == 0 and Cout <= 16 (Cin=%d Cout=%d)", Cin, Cout);
  BMV_REQUIRE((long)Cin * D * H * W < (1L << 29), "conv3d_split: one batch item must stay below 2 GiB");
  ConvSplitArgs a;
  a.in = in, a.wsplit = wsplit, a.bias = bias, a.out = out, a.depth_out = depth_out;
  a.B = B, a.Cin = Cin, a.Cout = Cout, a.D = D, a.H = H, a.W = W, a.slope = act_slope;
  BMV_REQUIRE(W % 4 == 0, "conv3d_split: W %% 4 == 0 (16-byte staging loads; W=%d)", W);
  BMV_REQUIRE(parts == 2 || parts == 3, "conv3d_split: parts must be 2 or 3 (got %d)", parts);
  const int tz = bmv::tuning("BMV_CONV_SPLIT_TZ", 2);
  hipStream_t st = as_stream(stream);
#define BMV_SPLIT_LAUNCH(TZV, PV, TXV, RWV)                                                                              \
  do {                                                                                                                   \
    const size_t lds = (size_t)PV * (TZV + 2) * (4 * RWV + 2) * (TXV + 2) * sizeof(i32x4);                               \
    BMV_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_split_kernel<TZV, PV, TXV, RWV>),               \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,                 \
                "conv3d_split: cannot reserve %zu B of LDS", lds);                                                       \
    const dim3 grid(cdiv(W, TXV) * cdiv(H, 4 * RWV) * cdiv(D, TZV), B);                                                  \
    hipLaunchKernelGGL((conv3d_split_kernel<TZV, PV, TXV, RWV>), grid, dim3(256), lds, st, a);                           \
  } while (0)
#define BMV_SPLIT_TX_RW(TZV, PV)                                  \
  do {                                                            \
    if (txv == 16 && rwv == 1) BMV_SPLIT_LAUNCH(TZV, PV, 16, 1);  \
    else if (txv == 16) BMV_SPLIT_LAUNCH(TZV, PV, 16, 2);         \
    else if (rwv == 1) BMV_SPLIT_LAUNCH(TZV, PV, 32, 1);          \
    else BMV_SPLIT_LAUNCH(TZV, PV, 32, 2);                        \
  } while (0)
  const int txv = bmv::tuning("BMV_CONV_SPLIT_TX", 32);
  const int rwv = bmv::tuning("BMV_CONV_SPLIT_RW", 1);
  if (tz == 1 && parts == 2) BMV_SPLIT_TX_RW(1, 2);
  else if (tz == 1) BMV_SPLIT_TX_RW(1, 3);
  else if (parts == 2) BMV_SPLIT_TX_RW(2, 2);
  else BMV_SPLIT_TX_RW(2, 3);
#undef BMV_SPLIT_TX_RW
#undef BMV_SPLIT_LAUNCH
  BMV_LAUNCH_END("bmv_conv3d_split_fwd");
}

}  // extern "C"
