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
// Orientation (as the MLP of render.hip): weights are the A operand (rows = 16 output channels), the
// activations are the B operand (columns = 16 consecutive output x), one k-step = 4 input channels of one
// filter tap.  Activations stay in the reference's planar layout (B,C,D,H,W): an input tile with its halo is
// staged in LDS as [channel][z][y][x] with a plane stride chosen so that the 2 x 16 lanes of one LDS cycle
// (2 channels x 16 x) fall on 32 distinct banks; the per-tap operand is `ds_read_b32 base + immediate`.
// Tile loads are raw buffer loads whose per-thread slot offsets are computed once per block; out-of-image
// slots (zero padding) and padded input channels use an out-of-range offset and come back as 0.
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
  int relu, channels_last;
};

template <int KD, int K, int S, int NCT, int R, bool IS3D>
struct ConvTile {
  static constexpr int NRG = 4 / NCT;  // row groups (waves per cout tile)
  static constexpr int TZ = IS3D ? NRG : 1;
  static constexpr int TY = IS3D ? R : NRG * R;
  static constexpr int TZH = (TZ - 1) * S + KD;
  static constexpr int TYH = (TY - 1) * S + K;
  static constexpr int RS = 15 * S + K;  // 16 outputs along x + halo
  static constexpr int SLOTS = TZH * TYH * RS;
  // stride 1: plane stride = 16 (mod 32); stride 2: odd  -> conflict-free operand reads
  static constexpr int PS = (S == 1) ? ((SLOTS + 15) / 32 * 32 + 16) : (SLOTS | 1);
  static constexpr int NSLOT = (SLOTS + 255) / 256;
  static constexpr int TAPS = KD * K * K;
  static constexpr int ROWBASE = IS3D ? S * TYH * RS : R * S * RS;  // LDS offset of one row group
};

template <int KD, int K, int S, int NCT, int R, bool IS3D>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  using T = ConvTile<KD, K, S, NCT, R, IS3D>;
  __shared__ float lds[4 * T::PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = wave % NCT, rg = wave / NCT;
  const int ntx = (a.Wo + 15) / 16, nty = (a.Ho + T::TY - 1) / T::TY, ntz = (a.Do + T::TZ - 1) / T::TZ;
  int bid = blockIdx.x;
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  bid /= nty;
  const int tz = bid % ntz;
  const int b = bid / ntz;
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
    goff[j] = ok ? 4u * (unsigned)((gz * a.H + gy) * a.W + gx) : 0x80000000u;
  }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (size_t)b * a.Cin * plane), 0, (int)(4u * (unsigned)(a.Cin * plane)), 0x00020000);

  const int nchunk = (a.Cin + 3) / 4;
  const int cot = blockIdx.y * NCT + ct;  // cout tile of this wave
  const float* wp = a.wpack + (size_t)cot * nchunk * (T::TAPS * 64) + lane;

  f32x4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

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
    float wv[T::TAPS];
#pragma unroll
    for (int t = 0; t < T::TAPS; ++t) wv[t] = wnext[t];
    __syncthreads();
    if (chunk + 1 < nchunk) {  // next tile and next weights are in flight during the MFMAs below
      load_tile(chunk + 1);
#pragma unroll
      for (int t = 0; t < T::TAPS; ++t) wnext[t] = wp[(size_t)(chunk + 1) * (T::TAPS * 64) + t * 64];
    }
#pragma unroll
    for (int kd = 0; kd < KD; ++kd)
#pragma unroll
      for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const float w = wv[(kd * K + kh) * K + kw];
#pragma unroll
          for (int r = 0; r < R; ++r)
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, ap[(kd * T::TYH + r * S + kh) * T::RS + kw], acc[r], 0,
                                                          0, 0);
        }
  }

  // epilogue: lane = output x (lane & 15), registers = 4 consecutive output channels
  const int x = x0 + (lane & 15);
  const int co0 = cot * 16 + 4 * (lane >> 4);
  if (x >= a.Wo || co0 >= a.Cout) return;
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[co0 + j];
  const int z = IS3D ? z0 + rg : 0;
  if (z >= a.Do) return;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int y = IS3D ? y0 + r : y0 + rg * R + r;
    if (y >= a.Ho) break;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = acc[r][j] + bs[j];
      if (a.relu) v[j] = fmaxf(v[j], 0.f);
    }
    if (a.channels_last) {
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
      const size_t cs = (size_t)a.Do * a.Ho * a.Wo;
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
// ---------------------------------------------------------------------------------------------------
template <int NCT, int R>
struct ConvTTile {
  static constexpr int NRG = 4 / NCT;
  static constexpr int TZH = NRG + 1, TYH = R + 1, RS = 17;
  static constexpr int SLOTS = TZH * TYH * RS;
  static constexpr int PS = (SLOTS + 15) / 32 * 32 + 16;
  static constexpr int NSLOT = (SLOTS + 255) / 256;
};

template <int NCT, int R>
__global__ __launch_bounds__(256) void convT3d_mfma_kernel(ConvArgs a) {
  using T = ConvTTile<NCT, R>;
  __shared__ float lds[4 * T::PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = wave % NCT, rg = wave / NCT;
  const int ntx = (a.W + 15) / 16, nty = (a.H + R - 1) / R, ntz = (a.D + T::NRG - 1) / T::NRG;
  int bid = blockIdx.x;
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  bid /= nty;
  const int tz = bid % ntz;
  const int b = bid / ntz;
  const int x0 = tx * 16, y0 = ty * R, z0 = tz * T::NRG;
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

  const float* ap = lds + (lane >> 4) * T::PS + rg * (T::TYH * T::RS) + (lane & 15);
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
  const int mz = z0 + rg;
  if (m >= a.W || co0 >= a.Cout || mz >= a.D) return;
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[co0 + j];
  const size_t cs = (size_t)a.Do * a.Ho * a.Wo;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int my = y0 + r;
    if (my >= a.H) break;
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      const int pz = q >> 2, py = (q >> 1) & 1;
      const size_t o = ((((size_t)b * a.Cout + co0) * a.Do + 2 * mz + pz) * a.Ho + 2 * my + py) * a.Wo + 2 * m;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (co0 + j >= a.Cout) break;
        float v0 = acc[r][q][j] + bs[j], v1 = acc[r][q + 1][j] + bs[j];
        if (a.relu) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f);
        if (a.skip) {
          const float2 s = *reinterpret_cast<const float2*>(a.skip + o + j * cs);
          v0 += s.x, v1 += s.y;
        }
        *reinterpret_cast<float2*>(a.out + o + j * cs) = make_float2(v0, v1);
      }
    }
  }
}

template <int NCT, int R>
static void launch_convT(const ConvArgs& a, hipStream_t st) {
  using T = ConvTTile<NCT, R>;
  dim3 grid(cdiv(a.W, 16) * cdiv(a.H, R) * cdiv(a.D, T::NRG) * a.B, cdiv(cdiv(a.Cout, 16), NCT));
  hipLaunchKernelGGL((convT3d_mfma_kernel<NCT, R>), grid, dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------------------
// FPN top-down step (feature_net.py:24-36): out = bilinear_x2(coarse, align_corners=True) + conv1x1(fine) + bias.
// Memory-bound (writes 32 channels per pixel); a thread owns one pixel and walks the output channels, the
// 1x1 weights are wave-uniform (scalar loads).
// ---------------------------------------------------------------------------------------------------
template <int CF>
__global__ __launch_bounds__(256) void fpn_topdown_kernel(const float* __restrict__ fine, const float* __restrict__ coarse,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int C, int H, int W) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (x >= W || y >= H) return;
  const int Hc = H / 2, Wc = W / 2;
  const size_t hw = (size_t)H * W, hwc = (size_t)Hc * Wc;
  float f[CF];
#pragma unroll
  for (int i = 0; i < CF; ++i) f[i] = fine[((size_t)b * CF + i) * hw + (size_t)y * W + x];
  const Lerp1 ly = upsample_axis(y, Hc, H), lx = upsample_axis(x, Wc, W);
  for (int c = 0; c < C; ++c) {
    float v = bias[c];
#pragma unroll
    for (int i = 0; i < CF; ++i) v = fmaf(w[c * CF + i], f[i], v);
    v += upsample_fetch(coarse + ((size_t)b * C + c) * hwc, Wc, ly, lx);
    out[((size_t)b * C + c) * hw + (size_t)y * W + x] = v;
  }
}

template <int KD, int K, int S, int NCT, int R, bool IS3D>
static void launch_conv(const ConvArgs& a, hipStream_t st) {
  using T = ConvTile<KD, K, S, NCT, R, IS3D>;
  const unsigned ntx = cdiv(a.Wo, 16), nty = cdiv(a.Ho, T::TY), ntz = cdiv(a.Do, T::TZ);
  const unsigned ncot = cdiv(a.Cout, 16);
  dim3 grid(ntx * nty * ntz * a.B, cdiv(ncot, NCT));
  hipLaunchKernelGGL((conv_mfma_kernel<KD, K, S, NCT, R, IS3D>), grid, dim3(256), 0, st, a);
}

template <int KD, int K, int S, int R, bool IS3D>
static void launch_conv_nct(const ConvArgs& a, hipStream_t st) {
  const unsigned ncot = cdiv(a.Cout, 16);
  if (ncot == 1)
    launch_conv<KD, K, S, 1, R, IS3D>(a, st);
  else if (ncot == 2)
    launch_conv<KD, K, S, 2, R, IS3D>(a, st);
  else
    launch_conv<KD, K, S, 4, R, IS3D>(a, st);
}

}  // namespace bmv

extern "C" {

int bmv_conv_wpack_floats(int Cin, int Cout, int kd, int kh, int kw) {
  return ((Cout + 15) / 16) * ((Cin + 3) / 4) * kd * kh * kw * 64;
}

int bmv_conv_fwd(const float* in, const float* wpack, const float* bias, const float* skip, float* out, int B, int Cin,
                 int D, int H, int W, int Cout, int kd, int k, int stride, int relu, int out_channels_last,
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
  a.relu = relu, a.channels_last = out_channels_last;
  hipStream_t st = as_stream(stream);
  if (kd == 1 && k == 3 && stride == 1)
    launch_conv_nct<1, 3, 1, 8, false>(a, st);
  else if (kd == 1 && k == 5 && stride == 2)
    launch_conv_nct<1, 5, 2, 4, false>(a, st);
  else if (kd == 1 && k == 1 && stride == 1)
    launch_conv_nct<1, 1, 1, 8, false>(a, st);
  else if (kd == 3 && k == 3 && stride == 1)
    launch_conv_nct<3, 3, 1, 8, true>(a, st);
  else if (kd == 3 && k == 3 && stride == 2)
    launch_conv_nct<3, 3, 2, 4, true>(a, st);
  else
    BMV_REQUIRE(false, "conv: kernel (%d,%d,%d) stride %d is not one of the shapes of FeatureNet / CostRegNet", kd, k,
                k, stride);
  BMV_LAUNCH_END("conv_fwd");
}

int bmv_conv3d_transpose_fwd(const float* in, const float* wpack, const float* bias, const float* skip, float* out, int B, int Cin,
                    int D, int H, int W, int Cout, int relu, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(in && wpack && bias && out, "convT3d: null pointer");
  BMV_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "convT3d: bad shape");
  BMV_REQUIRE((size_t)Cin * D * H * W < (1u << 29), "convT3d: one batch item must stay below 2 GiB");
  ConvArgs a;
  a.in = in, a.wpack = wpack, a.bias = bias, a.skip = skip, a.out = out;
  a.B = B, a.Cin = Cin, a.D = D, a.H = H, a.W = W, a.Cout = Cout;
  a.Do = 2 * D, a.Ho = 2 * H, a.Wo = 2 * W;
  a.relu = relu, a.channels_last = 0;
  hipStream_t st = as_stream(stream);
  const unsigned ncot = cdiv(Cout, 16);
  if (ncot == 1)
    launch_convT<1, 4>(a, st);
  else
    launch_convT<2, 4>(a, st);
  BMV_LAUNCH_END("convT3d_fwd");
}

int bmv_fpn_topdown_fwd(const float* fine, const float* coarse, const float* w, const float* bias, float* out, int B,
                        int Cf, int C, int H, int W, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(fine && coarse && w && bias && out, "fpn_topdown: null pointer");
  BMV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "fpn_topdown: bad shape");
  dim3 grid(cdiv(W, 64), cdiv(H, 4), B);
  hipStream_t st = as_stream(stream);
  if (Cf == 8)
    hipLaunchKernelGGL(fpn_topdown_kernel<8>, grid, dim3(256), 0, st, fine, coarse, w, bias, out, C, H, W);
  else if (Cf == 16)
    hipLaunchKernelGGL(fpn_topdown_kernel<16>, grid, dim3(256), 0, st, fine, coarse, w, bias, out, C, H, W);
  else
    BMV_REQUIRE(false, "fpn_topdown: %d lateral input channels unsupported (FeatureNet has 8 and 16)", Cf);
  BMV_LAUNCH_END("fpn_topdown_fwd");
}

}  // extern "C"
