// MVSNeRF backbone kernels (a18-a25): reference-view projection matrices, the padded
// plane sweep with colour channels and in-frustum counting, and the fused
// sample -> 6x128 MLP kernel.
// Reference: lib/networks/mvsnerf/network.py:153-229, 887-1126; utils.py:112-146, 300-383,
//            580-630; renderer.py:111-137.
#include "mlp.hpp"          // f32x16, n16(), BMV_MFMA, fences, xhalf_sum
#include "render_geom.hpp"
#include "scatter.hpp"

namespace bmv {

// ---------------------------------------------------------------------------
// a18
// ---------------------------------------------------------------------------
__device__ inline void invert4x4(double m[4][8]) {  // Gauss-Jordan, partial pivoting, [A | I] -> [I | A^-1]
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    double best = fabs(m[col][col]);
    for (int r = col + 1; r < 4; ++r)
      if (fabs(m[r][col]) > best) best = fabs(m[r][col]), piv = r;
    if (piv != col)
      for (int c = 0; c < 8; ++c) {
        double t = m[col][c];
        m[col][c] = m[piv][c];
        m[piv][c] = t;
      }
    double inv = 1.0 / m[col][col];
    for (int c = 0; c < 8; ++c) m[col][c] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == col) continue;
      double f = m[r][col];
      for (int c = 0; c < 8; ++c) m[r][c] -= f * m[col][c];
    }
  }
}

__device__ inline void scaled_proj(const float* K, const float* E, float P[4][4]) {
  for (int r = 0; r < 3; ++r) {
    float k[3];
    for (int c = 0; c < 3; ++c) k[c] = r < 2 ? K[r * 3 + c] * 0.25f : K[r * 3 + c];
    for (int c = 0; c < 4; ++c) {
      float a = 0.f;
      for (int j = 0; j < 3; ++j) a += k[j] * E[j * 4 + c];
      P[r][c] = a;
    }
  }
  P[3][0] = P[3][1] = P[3][2] = 0.f, P[3][3] = 1.f;
}

__global__ void mvs_proj_mats_kernel(const float* __restrict__ exts, const float* __restrict__ ixts, int B, int S,
                                     float* __restrict__ proj) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * S) return;
  int b = idx / S, i = idx - b * S;
  float* out = proj + (size_t)idx * 12;
  if (i == 0) {
    for (int k = 0; k < 12; ++k) out[k] = (k == 0 || k == 5 || k == 10) ? 1.f : 0.f;
    return;
  }
  float Pr[4][4], Pi[4][4];
  scaled_proj(ixts + ((size_t)b * S) * 9, exts + ((size_t)b * S) * 16, Pr);
  scaled_proj(ixts + (size_t)idx * 9, exts + (size_t)idx * 16, Pi);
  double m[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) m[r][c] = (double)Pr[r][c], m[r][4 + c] = r == c ? 1.0 : 0.0;
  invert4x4(m);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      double a = 0.0;
      for (int k = 0; k < 4; ++k) a += (double)Pi[r][k] * (double)(float)m[k][4 + c];
      out[r * 4 + c] = (float)a;
    }
}

// ---------------------------------------------------------------------------
// F.interpolate(mode='bilinear', align_corners=False), size given
// ---------------------------------------------------------------------------
__device__ __forceinline__ Lerp1 half_pixel_axis(int dst, int in_size, int out_size) {
  Lerp1 r;
  float scale = (float)in_size / (float)out_size;
  float src = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
  r.i0 = min((int)src, in_size - 1);
  r.i1 = r.i0 + ((r.i0 < in_size - 1) ? 1 : 0);
  r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
  r.l0 = 1.f - r.l1;
  return r;
}

__global__ void resize_bilinear_kernel(const float* __restrict__ src, int H, int W, int h, int w,
                                       float* __restrict__ dst) {
  int plane = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h * w) return;
  int y = i / w, x = i - y * w;
  Lerp1 ly = half_pixel_axis(y, H, h), lx = half_pixel_axis(x, W, w);
  dst[(size_t)plane * h * w + i] = upsample_fetch(src + (size_t)plane * H * W, W, ly, lx);
}

// ---------------------------------------------------------------------------
// a19 + a20: one thread per padded voxel, all channels in registers.
// ---------------------------------------------------------------------------
template <int C, int S>
__global__ void __launch_bounds__(256) mvs_sweep_kernel(const float* __restrict__ imgs,
                                                         const float* __restrict__ feats,
                                                         const float* __restrict__ proj,
                                                         const float* __restrict__ depth_values, int h, int w, int D,
                                                         int pad, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int hp = h + 2 * pad, wp = w + 2 * pad;
  const size_t nvox = (size_t)D * hp * wp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox) return;
  const int xp = (int)(i % wp), yp = (int)((i / wp) % hp), d = (int)(i / ((size_t)hp * wp));
  const int x = xp - pad, y = yp - pad;
  const bool inside = x >= 0 && x < w && y >= 0 && y < h;
  const size_t plane = (size_t)h * w;
  const float depth = depth_values[b * D + d];
  float* o = out + (size_t)b * (3 * S + C) * nvox + i;

  float acc[C], acc2[C];
  {  // reference view: un-warped, zero-padded (network.py:907-918)
    const float* f0 = feats + ((size_t)b * S) * C * plane + (inside ? (size_t)y * w + x : 0);
    const float* c0 = imgs + ((size_t)b * S) * 3 * plane + (inside ? (size_t)y * w + x : 0);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float v = inside ? f0[c * plane] : 0.f;
      acc[c] = v, acc2[c] = v * v;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * nvox] = inside ? c0[c * plane] : 0.f;
  }
  float count = 1.f;
#pragma unroll
  for (int s = 1; s < S; ++s) {
    const float* P = proj + ((size_t)b * S + s) * 12;
    float px = P[0] * x + P[1] * y + P[2] + P[3] / depth;
    float py = P[4] * x + P[5] * y + P[6] + P[7] / depth;
    float pz = P[8] * x + P[9] * y + P[10] + P[11] / depth;
    float gx = (px / pz) / ((float)(w - 1) * 0.5f) - 1.f;  // no z clamp in this variant (utils.py:617)
    float gy = (py / pz) / ((float)(h - 1) * 0.5f) - 1.f;
    count += (gx > -1.f && gx < 1.f && gy > -1.f && gy < 1.f) ? 1.f : 0.f;
    Taps2 t = taps_zeros(unnorm(gx, w), unnorm(gy, h), w, h);
    const float* f = feats + ((size_t)b * S + s) * C * plane;
    const float* cimg = imgs + ((size_t)b * S + s) * 3 * plane;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[(3 * s + c) * nvox] = tap_fetch(cimg + c * plane, t);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float v = tap_fetch(f + c * plane, t);
      acc[c] += v, acc2[c] += v * v;
    }
  }
  const float inv = 1.f / count;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float m = acc[c] * inv;
    o[(3 * S + c) * nvox] = acc2[c] * inv - m * m;
  }
}

// ---------------------------------------------------------------------------
// a19 + a20 on CHANNEL-LAST features (round 3).  The kernel above gathers every tap of every channel as its own dword
// (35 channels x 4 taps x 2 warped views = 280 scattered loads per voxel): at 128 planes it runs 298 us for the 297 MB
// it writes -- bound by the texture addresser, not by the stores.  Here the features are channel-last (B,S,h,w,C) and
// FOUR lanes share a voxel: lane (voxel, sub) reads the 16-byte slices sub and 4 + sub of a tap's 128-byte record, so
// one load instruction of a wave covers 16 voxels x 64 contiguous bytes (16 half lines) instead of 64 lanes x 16 bytes
// of 64 different lines (the first channel-last version, lane = voxel: 182 us -- the addresser works per line
// touched).  A lane keeps 8 channels (16 accumulators): no register pressure, full occupancy.  The 9 colour channels
// are split over the 4 lanes of a voxel (planar dword gathers).  Stores: the workgroup's 64 voxels x 41 channels are
// staged in LDS and leave as whole channel rows (256 contiguous bytes per instruction: 132 -> 112 us; the direct form
// wrote 4 channels x 16 voxels = four 64-byte pieces per instruction); the volume (297 MB) is written once and read
// once by the regulariser: `nt`.  Geometry as above but with v_rcp_f32 for the three quotients (1 ulp; the tests hold both
// kernels to the reference's volume at the project tolerance).
// ---------------------------------------------------------------------------
template <int S, int AUX>
__global__ void __launch_bounds__(256) mvs_sweep_cl_kernel(const float* __restrict__ imgs,
                                                            const float* __restrict__ feats_cl,
                                                            const float* __restrict__ proj,
                                                            const float* __restrict__ depth_values, int h, int w, int D,
                                                            int pad, float* __restrict__ out) {
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  constexpr int C = 32;
  const int b = blockIdx.y;
  const int hp = h + 2 * pad, wp = w + 2 * pad;
  const unsigned nvox = (unsigned)D * hp * wp;
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned i = gid >> 2, sub = gid & 3u;
  const bool live = i < nvox;
  const unsigned ic = live ? i : nvox - 1;
  const int xp = (int)(ic % wp), yp = (int)((ic / wp) % hp), d = (int)(ic / ((unsigned)hp * wp));
  const int x = xp - pad, y = yp - pad;
  const bool inside = x >= 0 && x < w && y >= 0 && y < h;
  const size_t plane = (size_t)h * w;
  const float inv_depth = __builtin_amdgcn_rcpf(depth_values[b * D + d]);
  __amdgpu_buffer_rsrc_t frs = make_rsrc(feats_cl + (size_t)b * S * plane * C, (size_t)S * plane * C * 4);
  __amdgpu_buffer_rsrc_t ors = make_rsrc(out + (size_t)b * (3 * S + C) * nvox, (size_t)(3 * S + C) * nvox * 4);
  __amdgpu_buffer_rsrc_t irs = make_rsrc(imgs + (size_t)b * S * 3 * plane, (size_t)S * 3 * plane * 4);
  const unsigned cst = nvox * 4u;                      // byte stride of an output channel
  const unsigned ovoff = live ? i * 4u : 0x80000000u;  // lanes past the volume store out of range
  const unsigned sl = sub * 16u;                       // this lane's slices of a record: bytes [sl, sl+16) and [64+sl, ..)
  const float fx = (float)x, fy = (float)y;

  // accumulators: channels 4 sub .. 4 sub + 3 and 16 + 4 sub .. ; reference view first (un-warped, zero-padded,
  // network.py:907-918)
  float4 acc[2], acc2[2];
  {
    const unsigned r0 = inside ? (unsigned)(y * w + x) * (C * 4u) + sl : 0x80000000u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(frs, r0 + 64u * j, 0, 0);
      const float4 f = *reinterpret_cast<float4*>(&v);
      acc[j] = f;
      acc2[j] = make_float4(f.x * f.x, f.y * f.y, f.z * f.z, f.w * f.w);
    }
  }
  // the 9 colour channels (3 S) over the 4 lanes of a voxel: lane sub writes colours sub, sub + 4, sub + 8
  float col[3] = {0.f, 0.f, 0.f};
  if (sub < 3) {   // colour `sub` is channel sub of the reference view
    const unsigned o = inside ? (unsigned)(sub * plane + (size_t)y * w + x) * 4u : 0x80000000u;
    col[0] = ld_buf(irs, o, 0);
  }
  float count = 1.f;
#pragma unroll
  for (int s = 1; s < S; ++s) {
    const float* P = proj + ((size_t)b * S + s) * 12;
    const float px = P[0] * fx + P[1] * fy + P[2] + P[3] * inv_depth;
    const float py = P[4] * fx + P[5] * fy + P[6] + P[7] * inv_depth;
    const float pz = P[8] * fx + P[9] * fy + P[10] + P[11] * inv_depth;
    const float iz = __builtin_amdgcn_rcpf(pz);          // no z clamp in this variant (utils.py:617)
    const float gx = (px * iz) / ((float)(w - 1) * 0.5f) - 1.f;
    const float gy = (py * iz) / ((float)(h - 1) * 0.5f) - 1.f;
    count += (gx > -1.f && gx < 1.f && gy > -1.f && gy < 1.f) ? 1.f : 0.f;
    const Taps2 t = taps_zeros(unnorm(gx, w), unnorm(gy, h), w, h);
    // colours 3 s + c of this view: colour index cc = sub + 4 j -> (view cc / 3, channel cc % 3)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int cc = (int)sub + 4 * j;
      if (cc / 3 == s) {
        const unsigned cb = (unsigned)((s * 3 + cc % 3) * plane) * 4u;
        float v = ld_buf(irs, cb + (unsigned)t.o00 * 4u, 0) * t.w00;
        v += ld_buf(irs, cb + (unsigned)t.o01 * 4u, 0) * t.w01;
        v += ld_buf(irs, cb + (unsigned)t.o10 * 4u, 0) * t.w10;
        v += ld_buf(irs, cb + (unsigned)t.o11 * 4u, 0) * t.w11;
        col[j] = v;
      }
    }
    // the 4 taps as byte offsets of channel-last records (in-bounds by construction: taps_zeros parks invalid taps)
    const unsigned vb = (unsigned)s * (unsigned)plane * (C * 4u) + sl;
    const unsigned o00 = vb + (unsigned)t.o00 * (C * 4u), o01 = vb + (unsigned)t.o01 * (C * 4u);
    const unsigned o10 = vb + (unsigned)t.o10 * (C * 4u), o11 = vb + (unsigned)t.o11 * (C * 4u);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      i32x4 a00 = __builtin_amdgcn_raw_buffer_load_b128(frs, o00 + 64u * j, 0, 0);
      i32x4 a01 = __builtin_amdgcn_raw_buffer_load_b128(frs, o01 + 64u * j, 0, 0);
      i32x4 a10 = __builtin_amdgcn_raw_buffer_load_b128(frs, o10 + 64u * j, 0, 0);
      i32x4 a11 = __builtin_amdgcn_raw_buffer_load_b128(frs, o11 + 64u * j, 0, 0);
      const float4 f00 = *reinterpret_cast<float4*>(&a00), f01 = *reinterpret_cast<float4*>(&a01);
      const float4 f10 = *reinterpret_cast<float4*>(&a10), f11 = *reinterpret_cast<float4*>(&a11);
      // same accumulation order as tap_fetch / aten's grid_sampler_2d: nw, ne, sw, se
      float4 v;
      v.x = f00.x * t.w00, v.x += f01.x * t.w01, v.x += f10.x * t.w10, v.x += f11.x * t.w11;
      v.y = f00.y * t.w00, v.y += f01.y * t.w01, v.y += f10.y * t.w10, v.y += f11.y * t.w11;
      v.z = f00.z * t.w00, v.z += f01.z * t.w01, v.z += f10.z * t.w10, v.z += f11.z * t.w11;
      v.w = f00.w * t.w00, v.w += f01.w * t.w01, v.w += f10.w * t.w10, v.w += f11.w * t.w11;
      acc[j].x += v.x, acc[j].y += v.y, acc[j].z += v.z, acc[j].w += v.w;
      acc2[j].x += v.x * v.x, acc2[j].y += v.y * v.y, acc2[j].z += v.z * v.z, acc2[j].w += v.w * v.w;
    }
  }
  const float inv = 1.f / count;
  if constexpr (AUX >= 0x100) {
    // STAGED stores: the workgroup's 64 voxels x (3 S + C) channels go through LDS so that every store instruction is
    // one channel row of 64 consecutive voxels (256 contiguous bytes) instead of 4 channels x 16 voxels (4 x 64 bytes)
    constexpr int NCH = 3 * S + C;
    __shared__ float tile[NCH][64 + 1];
    const int vl = threadIdx.x >> 2;                       // voxel of the workgroup
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const unsigned cc = sub + 4u * j;
      if (cc < 3u * S) tile[cc][vl] = col[j];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c0 = 3 * S + 16 * j + 4 * (int)sub;
      float m;
      m = acc[j].x * inv, tile[c0 + 0][vl] = acc2[j].x * inv - m * m;
      m = acc[j].y * inv, tile[c0 + 1][vl] = acc2[j].y * inv - m * m;
      m = acc[j].z * inv, tile[c0 + 2][vl] = acc2[j].z * inv - m * m;
      m = acc[j].w * inv, tile[c0 + 3][vl] = acc2[j].w * inv - m * m;
    }
    __syncthreads();
    const unsigned i0 = blockIdx.x * 64u;                  // first voxel of the workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool ok = i0 + lane < nvox;
    for (int c = wv; c < NCH; c += 4)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, tile[c][lane]), ors,
                                            (int)(ok ? (i0 + lane) * 4u + (unsigned)c * cst : 0x80000000u), 0, AUX & 0xff);
    return;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const unsigned cc = sub + 4u * j;
    // (colour 9, 10, 11 do not exist for S = 3: those lanes store out of range)
    // (scalar offsets cannot differ per lane: the colour's channel offset goes into the vector offset)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, col[j]), ors,
                                          (int)((live && cc < 3u * S) ? i * 4u + cc * cst : 0x80000000u), 0, AUX & 0xff);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned cb = ovoff + (3u * S + 16u * j + 4u * sub) * cst;     // first of this lane's 4 channels
    float m;
    m = acc[j].x * inv;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[j].x * inv - m * m), ors, (int)(live ? cb : 0x80000000u), (int)(0 * cst), AUX & 0xff);
    m = acc[j].y * inv;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[j].y * inv - m * m), ors, (int)(live ? cb : 0x80000000u), (int)(1 * cst), AUX & 0xff);
    m = acc[j].z * inv;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[j].z * inv - m * m), ors, (int)(live ? cb : 0x80000000u), (int)(2 * cst), AUX & 0xff);
    m = acc[j].w * inv;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[j].w * inv - m * m), ors, (int)(live ? cb : 0x80000000u), (int)(3 * cst), AUX & 0xff);
  }
}

// ---------------------------------------------------------------------------
// a25: Renderer_ours (6 x 128) on fp32 MFMA, same transposed orientation as mlp.hpp
// (sample on the lane, neurons in accumulators, outputs of a layer are the B operands of
// the next).  The 0.5 MB of weights does not fit LDS: the packed blob is a sequence of
// CHUNKS (one per layer and pair of 32-neuron output tiles, [k-step][2 tiles][64 lanes]),
// streamed through one LDS buffer by the whole workgroup; the 4 waves of a workgroup walk
// the chunks in lockstep, each on its own 32-sample tile.
// ---------------------------------------------------------------------------
struct MvsMlp {
  static constexpr int N_CHUNKS = 17;
  static constexpr int FIRST_STREAMED = 2;   // chunks 0, 1 (pts_bias: 40 floats per lane) stay in registers for the whole launch
  // k-steps per chunk: bias x2, L0 x2, L1..L4 x2 each, L5 x2, feature x2, views x1
  __host__ __device__ static constexpr int steps(int c) {
    return c < 2 ? 10 : c < 4 ? 32 : c < 12 ? 64 : c < 14 ? 96 : c < 16 ? 64 : 66;
  }
  // (closed forms: the chunk counter may be a run-time value, BMV_MVS_LAYER_LOOP in mvs_mlp_forward)
  __host__ __device__ static constexpr int offset(int c) {
    return 128 * (c < 2 ? 10 * c : c < 4 ? 20 + 32 * (c - 2) : c < 12 ? 84 + 64 * (c - 4) : c < 14 ? 596 + 96 * (c - 12)
                  : c < 16 ? 788 + 64 * (c - 14) : 916 + 66 * (c - 16));
  }
  static constexpr int A_TOTAL = 2 * 10 * 128 + 2 * 32 * 128 + 8 * 64 * 128 + 2 * 96 * 128 + 2 * 64 * 128 + 66 * 128;
  // resident small tables, [tile][half][reg 16] (entry idx = tile * 16 + reg of a lane half: MVS_SMALL in the forward)
  static constexpr int S_BBIAS = 0;                   // pts_bias bias          [64]
  static constexpr int S_BL = S_BBIAS + 128;          // pts_linears.{0..5} bias [6][64]
  static constexpr int S_BF = S_BL + 6 * 128;         // feature_linear bias     [64]
  static constexpr int S_BV = S_BF + 128;             // views_linears.0 bias    [32]
  static constexpr int S_WA = S_BV + 64;              // alpha_linear weight     [64]
  static constexpr int S_WRGB = S_WA + 128;           // rgb_linear weight       [3][32]
  static constexpr int S_SC = S_WRGB + 3 * 64;        // alpha bias, rgb bias x3
  static constexpr int S_TOTAL = S_SC + 4;
  static constexpr int TOTAL = A_TOTAL + S_TOTAL;
  // The matrix chunks a second time behind the blob, pre-split into three bf16 pieces for v_mfma_f32_32x32x16_bf16 (the
  // bf16 x 3 form of mlp.hpp's BMV_SPLIT_CHAIN2): [piece 3][bf16 k-step][tile 2][lane 64] x 16 bytes, one bf16 k-step =
  // 8 fp32 k-steps.  Round 5 did this for the ten 128 -> 128 chunks (pts_linears.1-4, feature_linear: 8 k-steps, 48 KB);
  // round 6 adds pts_linears.0 (4 k-steps), pts_linears.5 (12: the skip's 63 embedded values + 128) and views_linears.0
  // (9: 128 + the 3 direction values; 66 -> 72 fp32 k-steps, zeros behind).  Only pts_bias (K = 20, 40 of the 1964 fp32
  // MFMAs of a tile) stays fp32.  A pts_linears.5 chunk is 72 KB in this form: that sets CHUNK_MAX, two LDS buffers.
  __host__ __device__ static constexpr int split_ks(int c) {          // bf16 k-steps of chunk c's split form (0: none)
    return c < 2 ? 0 : c < 4 ? 4 : c < 12 ? 8 : c < 14 ? 12 : c < 16 ? 8 : 9;
  }
  __host__ __device__ static constexpr bool is_split(int c) { return split_ks(c) > 0; }
  static constexpr int SPLIT_KSTEP = 3 * 2 * 64 * 4;                 // floats (dwords) of one bf16 k-step: 6 KB
  __host__ __device__ static constexpr int split_size(int c) { return split_ks(c) * SPLIT_KSTEP; }
  __host__ __device__ static constexpr int split_offset(int c) {
    return SPLIT_KSTEP * (c < 2 ? 0 : c < 4 ? 4 * (c - 2) : c < 12 ? 8 + 8 * (c - 4) : c < 14 ? 72 + 12 * (c - 12)
                          : c < 16 ? 96 + 8 * (c - 14) : 112 + 9 * (c - 16));
  }
  static constexpr int CHUNK_MAX = 12 * SPLIT_KSTEP;                 // floats: 72 KB (>= the 48 KB of a 96-step fp32 chunk)
  static constexpr int S_SPLIT = (TOTAL + 255) / 256 * 256;
  static constexpr int TOTAL_S = S_SPLIT + (2 * 4 + 8 * 8 + 2 * 12 + 2 * 8 + 9) * SPLIT_KSTEP;
};
static_assert(MvsMlp::S_SPLIT + MvsMlp::split_offset(MvsMlp::N_CHUNKS) == MvsMlp::TOTAL_S, "split chunk table");
static_assert(MvsMlp::split_size(12) == MvsMlp::CHUNK_MAX && MvsMlp::steps(12) * 128 <= MvsMlp::CHUNK_MAX, "largest chunk");
static_assert(MvsMlp::offset(MvsMlp::N_CHUNKS) == MvsMlp::A_TOTAL, "chunk table");
constexpr bool mvs_chunk_tables_consistent() {
  int o = 0, so = 0;
  for (int c = 0; c <= MvsMlp::N_CHUNKS; ++c) {
    if (MvsMlp::offset(c) != o || MvsMlp::split_offset(c) != so) return false;
    if (c < MvsMlp::N_CHUNKS) o += MvsMlp::steps(c) * 128, so += MvsMlp::split_size(c);
  }
  return true;
}
static_assert(mvs_chunk_tables_consistent(), "closed forms of the chunk offsets");
static constexpr int kMvsSmall = ((MvsMlp::S_TOTAL + 255) / 256) * 256;   // floats of LDS in front of the chunk buffers

__device__ __forceinline__ int hid_index(int u, int h) { return 32 * (u >> 4) + n16(u & 15, h); }  // k-step u of a 128-wide input

// A entry of chunk c for output 32 tl + i of the chunk's 64, fp32 k-step t, lane half h (zeros where the layer has no input)
__device__ __forceinline__ float mvs_chunk_weight(const bmv_mvs_mlp_params& p, int c, int tl, int i, int h, int t) {
  if (c < 2) {                                    // pts_bias: 20 -> 128
    const int n = 64 * c + 32 * tl + i, k = 2 * t + h;
    return k < 20 ? p.bias_w[n * 20 + k] : 0.f;
  }
  if (c < 4) {                                    // pts_linears.0: 63 -> 128
    const int n = 64 * (c - 2) + 32 * tl + i, k = 2 * t + h;
    return k < 63 ? p.pts_w[0][n * 63 + k] : 0.f;
  }
  if (c < 12) {                                   // pts_linears.1-4: 128 -> 128
    const int l = 1 + (c - 4) / 2, n = 64 * ((c - 4) & 1) + 32 * tl + i;
    return p.pts_w[l][n * 128 + hid_index(t, h)];
  }
  if (c < 14) {                                   // pts_linears.5: [pts 63 | h 128] -> 128
    const int n = 64 * (c - 12) + 32 * tl + i;
    if (t < 32) {
      const int k = 2 * t + h;
      return k < 63 ? p.pts_w[5][n * 191 + k] : 0.f;
    }
    return p.pts_w[5][n * 191 + 63 + hid_index(t - 32, h)];
  }
  if (c < 16) {                                   // feature_linear: 128 -> 128
    const int n = 64 * (c - 14) + 32 * tl + i;
    return p.feature_w[n * 128 + hid_index(t, h)];
  }
  const int n = 32 * tl + i;                      // views_linears.0: [feature 128 | dir 3] -> 64
  if (t < 64) return p.views_w[n * 131 + hid_index(t, h)];
  const int k = 2 * (t - 64) + h;
  return k < 3 ? p.views_w[n * 131 + 128 + k] : 0.f;
}

__global__ void mvs_mlp_pack_kernel(bmv_mvs_mlp_params p, float* __restrict__ blob) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= MvsMlp::TOTAL_S) return;
  if (idx >= MvsMlp::TOTAL) {
    if (idx < MvsMlp::S_SPLIT) {
      blob[idx] = 0.f;
      return;
    }
    // split chunks: dword q of (piece, bf16 k-step T, tile, lane (i, h)) = that piece (round to nearest; the last one is
    // exact) of the fp32 A entries of fp32 k-steps t = 8 T + 2 q, 8 T + 2 q + 1 for this tile and lane (low | high half)
    int e = idx - MvsMlp::S_SPLIT;
    int c = 0;
    while (e >= MvsMlp::split_size(c)) e -= MvsMlp::split_size(c), ++c;
    const int ks = MvsMlp::split_ks(c);
    const int q = e & 3, lane = (e >> 2) & 63, tl = (e >> 8) & 1, T = (e >> 9) % ks, pc = (e >> 9) / ks;
    const int i = lane & 31, h = lane >> 5;
    auto rn = [](float v) {
      const unsigned u = __float_as_uint(v);
      return __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
    };
    unsigned packed = 0;
    for (int jj = 0; jj < 2; ++jj) {
      const int t = 8 * T + 2 * q + jj;
      const float w = mvs_chunk_weight(p, c, tl, i, h, t);
      const float hi = rn(w), r1 = w - hi, mid = rn(r1), r2 = r1 - mid;
      const float piece = pc == 0 ? hi : pc == 1 ? mid : r2;
      packed |= (__float_as_uint(piece) >> 16) << (16 * jj);
    }
    blob[idx] = __uint_as_float(packed);
    return;
  }
  float v = 0.f;
  if (idx < MvsMlp::A_TOTAL) {
    int c = 0, base = 0;
    while (idx >= base + MvsMlp::steps(c) * 128) base += MvsMlp::steps(c) * 128, ++c;
    const int rel = idx - base;
    const int lane = rel & 63, tl = (rel >> 6) & 1, t = rel >> 7;
    v = mvs_chunk_weight(p, c, tl, lane & 31, lane >> 5, t);
  } else {
    int rel = idx - MvsMlp::A_TOTAL;
    if (rel >= MvsMlp::S_SC) {
      int k = rel - MvsMlp::S_SC;
      v = k == 0 ? p.alpha_b[0] : p.rgb_b[k - 1];
    } else {
      int h = (rel >> 4) & 1, e = (rel >> 5) * 16 + (rel & 15);   // [tile][half][16]: a lane reads its 16 as 4 x 16 bytes
      if (rel < MvsMlp::S_BL) {
        v = p.bias_b[hid_index(e, h)];
      } else if (rel < MvsMlp::S_BF) {
        int q = e - MvsMlp::S_BL / 2;
        v = p.pts_b[q / 64][hid_index(q % 64, h)];
      } else if (rel < MvsMlp::S_BV) {
        v = p.feature_b[hid_index(e - MvsMlp::S_BF / 2, h)];
      } else if (rel < MvsMlp::S_WA) {
        v = p.views_b[hid_index(e - MvsMlp::S_BV / 2, h)];
      } else if (rel < MvsMlp::S_WRGB) {
        v = p.alpha_w[hid_index(e - MvsMlp::S_WA / 2, h)];
      } else {
        int q = e - MvsMlp::S_WRGB / 2;
        v = p.rgb_w[(q / 32) * 64 + hid_index(q % 32, h)];
      }
    }
  }
  blob[idx] = v;
}

// Weight streaming.  The chunks go through TWO LDS buffers by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction,
// no VGPR round trip): the chunk behind chunk c is in flight while the MFMAs of c run, so a chunk costs one workgroup
// barrier and no exposed load latency.  The request is not issued as a burst behind the barrier but piece by piece between
// the MFMAs of chunk c (mvs_mlp_forward: dma_step), each piece hand-written in the scalar-base form (mvs_dma_piece).  The
// wait in front of a chunk is s_waitcnt vmcnt(0) + a bare s_barrier (__syncthreads() adds a fence nobody needs here).  The
// buffer slot is a running count over chunks AND tiles (15 streamed chunks per tile: the first one of the next tile is
// fetched under the last one of this one); pts_bias' two chunks are not streamed, their 40 floats per lane stay in
// registers (load_pts_bias_weights).  History: one buffer + a register-staged copy between two barriers per chunk left
// the matrix pipe idle half of the time (round 1); a third buffer with prefetch distance 2 and counted vmcnt waits was 3 %
// SLOWER (round 6, profiles/r6/mvs_pipeline.txt part 1: the stream's latency was never the stall, its issue was -- part 2)
// and no longer fits since a pts_linears.5 chunk is 72 KB.
struct ChunkPipe {
  int slot;    // LDS buffer (0 / 1) of the next chunk this workgroup consumes
  bool more;   // another tile follows this one: prefetch its first chunks
#ifdef BMV_MVS_STAMPS
  int tile_no = 0;
  unsigned long long prev_start = 0;
#endif
};
static constexpr int kMvsBuffers = 2;

template <bool SPLIT = false>
__device__ __forceinline__ int chunk_pieces(int c) {   // 1 KB pieces of chunk c (128 floats per k-step = 512 B)
  return (SPLIT && MvsMlp::is_split(c)) ? MvsMlp::split_size(c) / 256 : MvsMlp::steps(c) / 2;
}

// One 1 KB piece of a chunk by LDS-DMA: 16 bytes per lane from src + lane16 to LDS byte address dst + 16 lane, src and dst
// WAVE-UNIFORM.  Hand-written in the scalar-base form (global_load_lds_dwordx4 voffset, s[base:base+1]): through the builtin the
// compiler builds a 64-bit vector address per piece (a v_lshl_add_u64 each, ~200 per tile) and hoists the per-chunk bases out
// of the tile loop -- up to 300 bytes of scratch once the split around it changes.  M0 (the LDS address) is put back.
__device__ __forceinline__ void mvs_dma_piece(const char* src, unsigned dst, unsigned lane16) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane16), "s"(src), "s"(dst) : "memory");
}
__device__ __forceinline__ unsigned mvs_lds_address(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(p);
}

template <bool SPLIT = false>
__device__ __forceinline__ void issue_chunk(const float* __restrict__ blob, float* __restrict__ bufs, int c, int slot) {
  const bool sp = SPLIT && MvsMlp::is_split(c);      // (the bf16 x 3 form of the chunk: 24 - 72 KB)
  const char* src = reinterpret_cast<const char*>(sp ? blob + MvsMlp::S_SPLIT + MvsMlp::split_offset(c)
                                                     : blob + MvsMlp::offset(c));
  char* dst = reinterpret_cast<char*>(bufs + slot * MvsMlp::CHUNK_MAX);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int npieces = chunk_pieces<SPLIT>(c);
  for (int p = wave; p < npieces; p += 4) mvs_dma_piece(src + p * 1024, mvs_lds_address(dst + p * 1024), lane * 16);
}
// pts_bias' A operands of this lane, [tile 4][k-step 10] (chunks 0, 1 of the fp32 blob): resident in registers for the whole
// launch -- 40 of the ~150 AGPRs the kernel leaves unused -- instead of two more chunks in the stream of every tile (two
// barriers, two DMA requests issued as bursts: there is no bf16 MFMA group in front of them to spread them over)
__device__ __forceinline__ void load_pts_bias_weights(const float* __restrict__ blob, int lane, float (&wb)[4][10]) {
#pragma unroll
  for (int tl = 0; tl < 4; ++tl)
#pragma unroll
    for (int t = 0; t < 10; ++t) wb[tl][t] = blob[MvsMlp::offset(tl >> 1) + t * 128 + (tl & 1) * 64 + lane];
}
// the first chunk(s) of a launch (every wave, before the first tile's gathers)
template <bool SPLIT = false>
__device__ __forceinline__ void start_chunks(const float* __restrict__ blob, float* __restrict__ bufs) {
  issue_chunk<SPLIT>(blob, bufs, MvsMlp::FIRST_STREAMED, 0);
}
// acc{0,1} += W_chunk[:, steps T0..T0+NT) * B, B given by `bval(t)` for the chunk-local step t
// (A operands software-pipelined by one group of MVS_G k-steps, as BMV_CHAIN2 of mlp.hpp)
#ifndef MVS_G
#define MVS_G 2
#endif
#define MVS_GEMM(buf, T0, NT, BEXPR, ACC0, ACC1)                                                        \
  {                                                                                                     \
    float a_[2][MVS_G][2];                                                                              \
    _Pragma("unroll") for (int u_ = 0; u_ < MVS_G; ++u_)                                                \
      if (u_ < (NT)) a_[0][u_][0] = (buf)[((T0) + u_) * 128 + lane], a_[0][u_][1] = (buf)[((T0) + u_) * 128 + 64 + lane]; \
    _Pragma("unroll") for (int g_ = 0; g_ < ((NT) + MVS_G - 1) / MVS_G; ++g_) {                         \
      _Pragma("unroll") for (int u_ = 0; u_ < MVS_G; ++u_) {                                            \
        const int tn_ = (g_ + 1) * MVS_G + u_;                                                          \
        if (tn_ < (NT))                                                                                 \
          a_[(g_ + 1) & 1][u_][0] = (buf)[((T0) + tn_) * 128 + lane], a_[(g_ + 1) & 1][u_][1] = (buf)[((T0) + tn_) * 128 + 64 + lane]; \
      }                                                                                                 \
      BMV_FENCE();                                                                                      \
      _Pragma("unroll") for (int u_ = 0; u_ < MVS_G; ++u_) {                                            \
        const int t = g_ * MVS_G + u_;                                                                  \
        if (t < (NT)) {                                                                                 \
          const float b_ = (BEXPR);                                                                     \
          ACC0 = BMV_MFMA(a_[g_ & 1][u_][0], b_, ACC0);                                                 \
          ACC1 = BMV_MFMA(a_[g_ & 1][u_][1], b_, ACC1);                                                 \
        }                                                                                               \
      }                                                                                                 \
      BMV_FENCE();                                                                                      \
    }                                                                                                   \
  }

// a K -> 64 product of a split chunk of KS bf16 k-steps on the B pieces of the layer's input (computed once per layer:
// both output halves use them): the bf16 x 3 form of BMV_SPLIT_CHAIN2 (mlp.hpp), small terms first
#ifndef BMV_MVS_ADIST
#define BMV_MVS_ADIST 2     // groups the A pieces are read ahead of their MFMAs
#endif
#ifndef BMV_MVS_ASPREAD
#define BMV_MVS_ASPREAD 1   // those reads between the group's first MFMAs instead of in front of them
#endif
// The A pieces are read by hand-written ds_read_b128 with COUNTED waits.  Left to the compiler, every wait in front of a
// group's MFMAs is s_waitcnt lgkmcnt(0) -- with the next chunk's LDS-DMA in flight it does not count LDS reads -- which
// also waits for the reads issued just before it for the NEXT group: the read-ahead was void in every second group and
// the LDS latency of the four waves queueing at the pipe exposed (18.8 k of the 122 k cycles of a tile:
// scripts/ablate_mvs_mlp.py, profiles/r6/mvs_pipeline.txt).  LDS operations complete in order, so "at most 3 n
// outstanding" says the oldest are done whatever else the compiler has in flight; scalar loads share the counter and
// return out of order -- none is issued between the drain at the top and the last group (the fences keep the
// compiler's code out of the group regions).
#define MVS_LDS_READ128(DST, ADDR, OFF)                                                                 \
  {                                                                                                     \
    mlp_u32x4& d_ = DST; /* (named outside the asm: operands alone do not capture in a generic lambda) */ \
    const unsigned a_ = ADDR;                                                                           \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d_) : "v"(a_), "n"(OFF) : "memory");            \
  }
template <int I, int N, class F>
__device__ __forceinline__ void mvs_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    mvs_static_for<I + 1, N>(f);
  }
}
#define MVS_GEMM_SPLIT(buf, KS, BH, BM, BL, ACC0, ACC1, DMA_)                                                 \
  {                                                                                                     \
    /* 2 KS groups g = (bf16 k-step T, tile tl) of 3 A pieces and 6 MFMAs; the pieces of group g + BMV_MVS_ADIST are   \
       read from LDS under the MFMAs of group g */                                                      \
    constexpr int AD_ = BMV_MVS_ADIST, NG_ = 2 * (KS);                                                  \
    unsigned ab_[3];   /* LDS byte address of this lane's 16 bytes of piece pc, group 0 */             \
    _Pragma("unroll") for (int pc_ = 0; pc_ < 3; ++pc_)                                                 \
      ab_[pc_] = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(buf) + (unsigned)lane * 16u + pc_ * NG_ * 1024; \
    mlp_u32x4 as_[AD_ + 1][3];                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
    mvs_static_for<0, (AD_ < NG_ ? AD_ : NG_) * 3>([&](auto I_) {                                       \
      constexpr int g0_ = decltype(I_)::value / 3, pc_ = decltype(I_)::value % 3;                       \
      if constexpr ((kMvsAblate & 16) != 0) as_[g0_][pc_] = mlp_u32x4{(unsigned)lane, 1u, 2u, 3u};      \
      else MVS_LDS_READ128(as_[g0_][pc_], ab_[pc_], g0_ * 1024);                                        \
    });                                                                                                 \
    mvs_static_for<0, NG_>([&](auto G_) {                                                               \
      constexpr int g_ = decltype(G_)::value, T_ = g_ >> 1, tl_ = g_ & 1, cur_ = g_ % (AD_ + 1);        \
      constexpr int nxt_ = (g_ + AD_) % (AD_ + 1);                                                      \
      constexpr bool more_ = g_ + AD_ < NG_;                                                            \
      /* the reads of group g + AD between the first MFMAs of group g (BMV_MVS_ASPREAD), not in front of them: when   \
         the four waves queue at the LDS pipe the issue of a read waits, and a wave issues in order -- behind an MFMA \
         that wait is covered by its 32 cycles, in front of the group it is not */                                 \
      auto read_ = [&](auto P_) {                                                                       \
        constexpr int pc_ = decltype(P_)::value;                                                        \
        if constexpr (more_) {                                                                          \
          if constexpr ((kMvsAblate & 16) != 0) as_[nxt_][pc_] = as_[cur_][pc_];                        \
          else MVS_LDS_READ128(as_[nxt_][pc_], ab_[pc_], (g_ + AD_) * 1024);                            \
        }                                                                                               \
      };                                                                                                \
      if constexpr ((kMvsAblate & 16) == 0) {                                                           \
        /* (with the reads spread, those of group g + AD are issued AFTER this wait: one group fewer is newer) */   \
        constexpr int ahead_ = BMV_MVS_ASPREAD ? AD_ - 1 : AD_;                                         \
        constexpr int newer_ = 3 * (NG_ - 1 - g_ < ahead_ ? NG_ - 1 - g_ : ahead_);                     \
        if constexpr (!BMV_MVS_ASPREAD) { read_(std::integral_constant<int, 0>{}); read_(std::integral_constant<int, 1>{}); read_(std::integral_constant<int, 2>{}); } \
        mlp_u32x4 &w0_ = as_[cur_][0], &w1_ = as_[cur_][1], &w2_ = as_[cur_][2];                        \
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(w0_), "+v"(w1_), "+v"(w2_) : "n"(newer_));          \
      }                                                                                                 \
      BMV_FENCE();                                                                                      \
      const mlp_bf16x8 Bh_ = __builtin_bit_cast(mlp_bf16x8, BH[T_]), Bm_ = __builtin_bit_cast(mlp_bf16x8, BM[T_]), \
                       Bl_ = __builtin_bit_cast(mlp_bf16x8, BL[T_]);                                    \
      const mlp_bf16x8 Ah_ = __builtin_bit_cast(mlp_bf16x8, as_[cur_][0]);                              \
      const mlp_bf16x8 Am_ = __builtin_bit_cast(mlp_bf16x8, as_[cur_][1]);                              \
      const mlp_bf16x8 Al_ = __builtin_bit_cast(mlp_bf16x8, as_[cur_][2]);                              \
      f32x16 c_ = tl_ == 0 ? ACC0 : ACC1;                                                               \
      if constexpr ((kMvsAblate & 8) != 0) {                                                            \
        asm volatile("" :: "v"(Al_), "v"(Am_), "v"(Ah_), "v"(Bh_), "v"(Bm_), "v"(Bl_));                 \
        if constexpr (BMV_MVS_ASPREAD) { read_(std::integral_constant<int, 0>{}); read_(std::integral_constant<int, 1>{}); read_(std::integral_constant<int, 2>{}); } \
        DMA_(); DMA_();                                                                                 \
      } else {                                                                                          \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al_, Bh_, c_, 0, 0, 0);                            \
        if constexpr (BMV_MVS_ASPREAD) { BMV_FENCE(); read_(std::integral_constant<int, 0>{}); BMV_FENCE(); } \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah_, Bl_, c_, 0, 0, 0);                            \
        if constexpr (BMV_MVS_ASPREAD) { BMV_FENCE(); read_(std::integral_constant<int, 1>{}); BMV_FENCE(); } \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am_, Bm_, c_, 0, 0, 0);                            \
        if constexpr (BMV_MVS_ASPREAD) { BMV_FENCE(); read_(std::integral_constant<int, 2>{}); BMV_FENCE(); } \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am_, Bh_, c_, 0, 0, 0);                            \
        BMV_FENCE(); DMA_(); BMV_FENCE();     /* two pieces of the next chunk's request per group */     \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah_, Bm_, c_, 0, 0, 0);                            \
        BMV_FENCE(); DMA_(); BMV_FENCE();                                                               \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah_, Bh_, c_, 0, 0, 0);                            \
        if constexpr (tl_ == 0) ACC0 = c_; else ACC1 = c_;                                              \
      }                                                                                                 \
      BMV_FENCE();                                                                                      \
    });                                                                                                 \
  }
typedef float mvs_f32x2 __attribute__((ext_vector_type(2)));
// Timing ablations of the split form (WRONG results; scripts/ablate_mvs_mlp.py builds them beside the product library):
// 1 no operand split (the pieces are the value's own bits), 2 no bias * relu epilogue, 4 no chunk barrier / wait,
// 8 no matrix instructions, 16 no LDS reads of the A pieces, 32 accumulators start at zero instead of the layer's bias vector
#ifndef BMV_MVS_ABLATE
#define BMV_MVS_ABLATE 0
#endif
constexpr int kMvsAblate = BMV_MVS_ABLATE;
// the three bf16 pieces of NK bf16 k-steps of a layer's input into B{H,M,L}[OFF ..): BEXPR is this lane's value for the
// fp32 k-step `t` (0 .. 8 NK - 1) of that part of the input
// The pieces are TRUNCATED, not rounded (8 + 8 + 8 significand bits: hi = the value's upper 16 bits, mid = the upper 16 bits
// of value - hi, low = value - hi - mid, which has at most 8 significant bits left -- an error-free split like the rounded
// one of mlp.hpp, what is dropped is nothing): v_and / v_perm_b32 / v_pk_add_f32 instead of v_cvt_pk_bf16_f32 and two expands,
// 9 instructions per pair as well but only 2 of them floating point and no inline asm -- 50 instead of 71 cycles per pair
// for a wave alone on its SIMD (scripts/ubench/split_under_mfma.hip, profiles/r6/split_under_mfma_ubench.txt).
typedef __bf16 mvs_bf16x2 __attribute__((ext_vector_type(2)));
// fp32 pair -> packed bf16 pair, round to nearest even: the compiler's own conversion (it selects v_cvt_pk_bf16_f32, the
// instruction mlp.hpp's inline asm names): no hazard s_nop on either side of an asm statement (1081 -> 587 per tile) and
// schedulable -- 1.3 % on config 4, bit-identical.  (While the layer kinds were selected at run time it made the kernel
// spill; BMV_MVS_CVT_ASM brings the asm form back.)
#ifdef BMV_MVS_CVT_ASM
#define MVS_CVT_PK(V) mlp_cvt_pk_bf16((V)[0], (V)[1])
#else
#define MVS_CVT_PK(V) __builtin_bit_cast(unsigned, __builtin_convertvector(V, mvs_bf16x2))
#endif
#ifndef BMV_MVS_SPLIT_TRUNC
#define BMV_MVS_SPLIT_TRUNC 0   // measured in the kernel: 111.1 k against 108.4 k cycles per tile for the rounded form and 84 bytes of scratch in the renderer
#endif
__device__ __forceinline__ unsigned mvs_upper_halves(mvs_f32x2 v) {   // bf16 pair (v[0] low | v[1] high) by truncation
  unsigned p = __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u);
  asm volatile("" : "+v"(p));   // a VGPR, here: left to itself the allocator spills (184 - 296 bytes of scratch)
  return p;
}
__device__ __forceinline__ mvs_f32x2 mvs_upper_floats(mvs_f32x2 v) {
  return mvs_f32x2{__uint_as_float(__float_as_uint(v[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(v[1]) & 0xffff0000u)};
}
#define MVS_SPLIT_INPUT(NK, OFF, BEXPR, BH, BM, BL)                                                     \
  _Pragma("unroll") for (int T_ = 0; T_ < (NK); ++T_)                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
      mvs_f32x2 v_;                                                                                     \
      _Pragma("unroll") for (int jj_ = 0; jj_ < 2; ++jj_) {                                             \
        const int t = 8 * T_ + 2 * q_ + jj_;                                                            \
        v_[jj_] = (BEXPR);                                                                              \
      }                                                                                                 \
      if constexpr ((kMvsAblate & 1) != 0) {                                                            \
        BH[(OFF) + T_][q_] = __float_as_uint(v_[0]), BM[(OFF) + T_][q_] = __float_as_uint(v_[1]);       \
        BL[(OFF) + T_][q_] = __float_as_uint(v_[0]);                                                    \
        continue;                                                                                       \
      }                                                                                                 \
      if constexpr (BMV_MVS_SPLIT_TRUNC) {                                                              \
        const mvs_f32x2 r1_ = v_ - mvs_upper_floats(v_);                                                \
        const mvs_f32x2 r2_ = r1_ - mvs_upper_floats(r1_);                                              \
        BH[(OFF) + T_][q_] = mvs_upper_halves(v_), BM[(OFF) + T_][q_] = mvs_upper_halves(r1_);          \
        BL[(OFF) + T_][q_] = mvs_upper_halves(r2_);                                                     \
        BMV_FENCE();   /* (pair by pair: left free, the scheduler interleaves all pairs of a layer -- spills) */ \
        continue;                                                                                       \
      }                                                                                                 \
      /* rounded pieces; the two residuals as packed fp32 subtractions (v_pk_add_f32): 9 instead of 11 per pair */        \
      const unsigned ph_ = MVS_CVT_PK(v_);                                               \
      const mvs_f32x2 h_ = {__uint_as_float(ph_ << 16), __uint_as_float(ph_ & 0xffff0000u)};            \
      const mvs_f32x2 r1_ = v_ - h_;                                                                    \
      const unsigned pm_ = MVS_CVT_PK(r1_);                                             \
      const mvs_f32x2 m_ = {__uint_as_float(pm_ << 16), __uint_as_float(pm_ & 0xffff0000u)};            \
      const mvs_f32x2 r2_ = r1_ - m_;                                                                   \
      BH[(OFF) + T_][q_] = ph_, BM[(OFF) + T_][q_] = pm_, BL[(OFF) + T_][q_] = MVS_CVT_PK(r2_); \
    }

#define MVS_SMALL(BASE, IDX) Sq[(BASE) + 2 * (IDX) - ((IDX) & 15)]
// -DBMV_MVS_STAMPS: shader-clock stamps at the phase boundaries of every wave's third tile (scripts/stamps_mvs_mlp.py reads
// them through bmv_debug_fetch_mvs_stamps); a tuning build, never the shipped library -- the stamps drain every counter,
// so they serialise what the shipped kernel overlaps
#ifdef BMV_MVS_STAMPS
constexpr int kMvsStamps = 52;
__device__ float g_mvs_stamps[256 * 4 * kMvsStamps];
#define MSTAMP(i)                                                       \
  {                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         \
    st[i] = __builtin_amdgcn_s_memtime();                               \
    __builtin_amdgcn_sched_barrier(0);                                  \
  }
#elif defined(BMV_MVS_PHASE_FENCE)
#if BMV_MVS_PHASE_FENCE == 2
#define MSTAMP(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#elif BMV_MVS_PHASE_FENCE == 3
#define MSTAMP(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define MSTAMP(i) __builtin_amdgcn_sched_barrier(0);
#endif
#else
#define MSTAMP(i)
#endif
// wb: pts_bias' weights (load_pts_bias_weights), e[32]: embedded point (slot t -> input 2t+h), f[10]: 20-ch feature,
// dv[2]: view direction.  Must be called by all 4 waves of the workgroup together (chunk staging uses workgroup barriers).
// The first streamed chunk of this tile must already be in flight (start_chunks at the top of the launch; afterwards
// the last chunk of the previous tile requests it).
template <bool SPLIT = false>
__device__ __forceinline__ void mvs_mlp_forward(const float* __restrict__ blob, const float* __restrict__ small,
                                                float* __restrict__ buf2, ChunkPipe& pipe, int lane,
                                                const float (&wb)[4][10], const float (&e)[32], const float (&f)[10],
                                                const float (&dv)[2], float (&out)[4]) {
  const int h = lane >> 5;
  const float* __restrict__ Sq = small + 16 * h;   // this lane half's 16 values of a tile are contiguous: MVS_SMALL
  f32x16 bias[4], hcur[4], hnew[4];
  int chunk = MvsMlp::FIRST_STREAMED;
  const float* buf = buf2;
  // The weight stream.  The request for the chunk behind chunk c is SET UP at the barrier in front of c and issued piece
  // by piece between the MFMAs of c (MVS_GEMM_SPLIT calls dma_step twice per group), not as a burst behind the barrier:
  // a wave issues in order, and its 6 - 18 global_load_lds of 1 KB queue behind the other three waves' at the CU's one
  // address path -- 1 - 2 k cycles per chunk in which no MFMA of the wave was issued, 37 k of the 114 k cycles of a tile
  // (scripts/stamps_mvs_mlp.py, profiles/r6/mvs_pipeline.txt).  In front of an fp32 chunk (BMV_MVS_SPLIT=0: no groups to
  // spread it over) the request is flushed at once as before.  All counts below fold to constants: the chunk counter does.
  static_assert(kMvsBuffers == 2, "prefetch distance 1");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int dma_np = 0, dma_i = 0;            // 1 KB pieces of the request; pieces of THIS wave (p = wave + 4 i) issued so far
  const char* dma_src = nullptr;        // this wave's first piece (+ 4 KB per piece), wave-uniform: mvs_dma_piece
  unsigned dma_dst = 0;                 // its LDS byte address
  const unsigned lane16 = lane * 16;
  auto dma_step = [&]() {
    if (dma_i < (dma_np + 3) / 4) {
      if (dma_i < dma_np / 4 || 4 * dma_i + wave < dma_np)
        mvs_dma_piece(dma_src + dma_i * 4096, dma_dst + dma_i * 4096, lane16);
      ++dma_i;
    }
  };
  auto dma_flush = [&]() {
#pragma unroll
    for (int k = 0; k < (MvsMlp::CHUNK_MAX / 256 + 3) / 4; ++k) dma_step();
  };
  auto next_chunk = [&]() {
    dma_flush();                                       // (nothing left behind a split chunk)
    if constexpr ((kMvsAblate & 4) == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my pieces of this chunk have landed ...
      asm volatile("s_barrier" ::: "memory");          // ... everyone's have; everyone is done with the previous chunk
    }
    buf = buf2 + pipe.slot * MvsMlp::CHUNK_MAX;
    const int free_slot = pipe.slot == 0 ? kMvsBuffers - 1 : pipe.slot - 1;   // the previous chunk's buffer
    const bool last = chunk + 1 == MvsMlp::N_CHUNKS;
    const int nc = last ? MvsMlp::FIRST_STREAMED : chunk + 1;   // (the next tile's first chunk under this tile's last one)
    const bool sp = SPLIT && MvsMlp::is_split(nc);
    dma_np = last && !pipe.more ? 0 : chunk_pieces<SPLIT>(nc);
    dma_i = 0;
    dma_src = reinterpret_cast<const char*>(sp ? blob + MvsMlp::S_SPLIT + MvsMlp::split_offset(nc) : blob + MvsMlp::offset(nc)) +
              wave * 1024;
    dma_dst = mvs_lds_address(buf2 + free_slot * MvsMlp::CHUNK_MAX) + wave * 1024;
    if constexpr (!SPLIT) dma_flush();
    pipe.slot = pipe.slot == kMvsBuffers - 1 ? 0 : pipe.slot + 1;
    ++chunk;
  };
#ifdef BMV_MVS_STAMPS
  unsigned long long st[kMvsStamps];
#endif
  MSTAMP(0)
  // pts_bias (network.py:210): bias = W_b feat + b_b
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[tl][r] = MVS_SMALL(MvsMlp::S_BBIAS, tl * 16 + r);
#pragma unroll
    for (int t = 0; t < 10; ++t) bias[tl] = BMV_MFMA(wb[tl][t], f[t], bias[tl]);   // (weights resident: load_pts_bias_weights)
  }
  MSTAMP(1)
  // pts_linears.0..5 (network.py:211-216): h = relu((W_i h + b_i) * bias), skip-concat of pts after i == 4.
  // One body for the three kinds of layer -- 0: pts_linears.0 (the embedded point), 1: pts_linears.1-4 (128 -> 128),
  // 2: pts_linears.5 (embedded point | 128) --, the kind a template argument: as ONE `#pragma unroll` loop over six layers
  // with `if (layer == ...)` inside, hipcc gave up on the unrolling half of the time ("unrolled size is too large": which
  // build did was a matter of what else changed) and then selected the kind at run time.  BMV_MVS_LAYER_LOOP=1 runs the four
  // layers of kind 1 as a run-time loop instead of four copies (measured 1.5 % slower on config 4; MvsMlp's chunk tables are
  // closed forms so that the chunk counter may be a run-time value there).
  auto pts_layer = [&](auto kind_, const int layer) {
    constexpr int kind = decltype(kind_)::value;
    mlp_u32x4 bh[12], bm[12], bl[12];   // [embedded point 4 |] hidden 8 bf16 k-steps
    if constexpr (SPLIT) {
      if constexpr (kind != 1) { MVS_SPLIT_INPUT(4, 0, e[t], bh, bm, bl) }
      if constexpr (kind == 2) { MVS_SPLIT_INPUT(8, 4, hcur[t >> 4][t & 15], bh, bm, bl) }
      if constexpr (kind == 1) { MVS_SPLIT_INPUT(8, 0, hcur[t >> 4][t & 15], bh, bm, bl) }
    }
    MSTAMP(2 + 6 * layer)
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hnew[2 * tp][r] = (kMvsAblate & 32) != 0 ? 0.f : MVS_SMALL(MvsMlp::S_BL, layer * 64 + (2 * tp) * 16 + r);
        hnew[2 * tp + 1][r] = (kMvsAblate & 32) != 0 ? 0.f : MVS_SMALL(MvsMlp::S_BL, layer * 64 + (2 * tp + 1) * 16 + r);
      }
      next_chunk();
      MSTAMP(3 + 6 * layer + 2 * tp)
      if constexpr (SPLIT) {
        MVS_GEMM_SPLIT(buf, (kind == 0 ? 4 : kind == 1 ? 8 : 12), bh, bm, bl, hnew[2 * tp], hnew[2 * tp + 1], dma_step);
      } else if constexpr (kind == 0) {
        MVS_GEMM(buf, 0, 32, e[t], hnew[2 * tp], hnew[2 * tp + 1]);
      } else if constexpr (kind == 2) {
        MVS_GEMM(buf, 0, 32, e[t], hnew[2 * tp], hnew[2 * tp + 1]);
        MVS_GEMM(buf, 32, 64, hcur[t >> 4][t & 15], hnew[2 * tp], hnew[2 * tp + 1]);
      } else {
        MVS_GEMM(buf, 0, 64, hcur[t >> 4][t & 15], hnew[2 * tp], hnew[2 * tp + 1]);
      }
      MSTAMP(4 + 6 * layer + 2 * tp)
    }
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {                // (pairs: v_pk_mul_f32)
        if constexpr ((kMvsAblate & 2) != 0) {
          hcur[tl][r] = hnew[tl][r], hcur[tl][r + 1] = hnew[tl][r + 1];
          continue;
        }
        const mvs_f32x2 m = mvs_f32x2{hnew[tl][r], hnew[tl][r + 1]} * mvs_f32x2{bias[tl][r], bias[tl][r + 1]};
        hcur[tl][r] = fmaxf(m[0], 0.f), hcur[tl][r + 1] = fmaxf(m[1], 0.f);
      }
    BMV_FENCE();
    MSTAMP(7 + 6 * layer)
  };
  pts_layer(std::integral_constant<int, 0>{}, 0);
#if !defined(BMV_MVS_LAYER_LOOP) || defined(BMV_MVS_STAMPS)
  pts_layer(std::integral_constant<int, 1>{}, 1);
  pts_layer(std::integral_constant<int, 1>{}, 2);
  pts_layer(std::integral_constant<int, 1>{}, 3);
  pts_layer(std::integral_constant<int, 1>{}, 4);
#else
#pragma unroll 1
  for (int layer = 1; layer <= 4; ++layer) pts_layer(std::integral_constant<int, 1>{}, layer);
#endif
  pts_layer(std::integral_constant<int, 2>{}, 5);
  // alpha head (network.py:220)
  {
    float s = 0.f;
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += MVS_SMALL(MvsMlp::S_WA, tl * 16 + r) * hcur[tl][r];
    out[3] = fmaxf(xhalf_sum(s) + small[MvsMlp::S_SC], 0.f);
  }
  MSTAMP(38)
  // feature_linear (network.py:221), no activation
  mlp_u32x4 fh[8], fm[8], fl[8];
  if constexpr (SPLIT) { MVS_SPLIT_INPUT(8, 0, hcur[t >> 4][t & 15], fh, fm, fl) }
  MSTAMP(39)
#pragma unroll
  for (int tp = 0; tp < 2; ++tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      hnew[2 * tp][r] = (kMvsAblate & 32) != 0 ? 0.f : MVS_SMALL(MvsMlp::S_BF, (2 * tp) * 16 + r);
      hnew[2 * tp + 1][r] = (kMvsAblate & 32) != 0 ? 0.f : MVS_SMALL(MvsMlp::S_BF, (2 * tp + 1) * 16 + r);
    }
    next_chunk();
    MSTAMP(40 + 2 * tp)
    if constexpr (SPLIT) {
      MVS_GEMM_SPLIT(buf, 8, fh, fm, fl, hnew[2 * tp], hnew[2 * tp + 1], dma_step);
    } else {
      MVS_GEMM(buf, 0, 64, hcur[t >> 4][t & 15], hnew[2 * tp], hnew[2 * tp + 1]);
    }
    MSTAMP(41 + 2 * tp)
  }
  // views_linears.0 (network.py:222-226): relu(W_v [feature, dir] + b_v), 131 -> 64
  f32x16 hv[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int r = 0; r < 16; ++r) hv[tl][r] = (kMvsAblate & 32) != 0 ? 0.f : MVS_SMALL(MvsMlp::S_BV, tl * 16 + r);
  next_chunk();
  MSTAMP(44)
  if constexpr (SPLIT) {
    mlp_u32x4 vh[9], vm[9], vl[9];      // feature 8 bf16 k-steps | direction (fp32 k-steps 64, 65; zeros behind)
    MVS_SPLIT_INPUT(8, 0, hnew[t >> 4][t & 15], vh, vm, vl)
    MVS_SPLIT_INPUT(1, 8, (t < 2 ? dv[t & 1] : 0.f), vh, vm, vl)
    MSTAMP(45)
    MVS_GEMM_SPLIT(buf, 9, vh, vm, vl, hv[0], hv[1], dma_step);
    MSTAMP(46)
  } else {
    MVS_GEMM(buf, 0, 64, hnew[t >> 4][t & 15], hv[0], hv[1]);
    MVS_GEMM(buf, 64, 2, dv[t], hv[0], hv[1]);
  }
  // rgb head (network.py:228): sigmoid(W_rgb relu(hv) + b)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = 0.f;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += MVS_SMALL(MvsMlp::S_WRGB, c * 32 + tl * 16 + r) * fmaxf(hv[tl][r], 0.f);
    float pre = xhalf_sum(s) + small[MvsMlp::S_SC + 1 + c];
    out[c] = 1.f / (1.f + __expf(-pre));
  }
  MSTAMP(47)
#ifdef BMV_MVS_STAMPS
  if constexpr (SPLIT) {
    if (lane == 0 && blockIdx.x < 256) {
      float* o = g_mvs_stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kMvsStamps;
      if (pipe.tile_no == 2) {
        for (int i = 1; i < 48; ++i) o[i] = (float)(unsigned)(st[i] - st[0]);
        o[0] = 1.f;
      }
      if (pipe.tile_no == 3) o[48] = (float)(unsigned)(st[0] - pipe.prev_start);
    }
    pipe.prev_start = st[0];
    ++pipe.tile_no;
  }
#endif
}

// ---------------------------------------------------------------------------
// a21-a24: per-sample inputs of the MLP
// ---------------------------------------------------------------------------
struct MvsCams {
  Cam cam[4];
  float near_v, far_v;
};

// value j (0..62) of Embedder.embed(ndc): [x | sin(x 2^k) k=0..9 | cos(x 2^k)], component fastest
__device__ __forceinline__ float embed_value(int j, const float (&ndc)[3], const float (&sn)[30],
                                             const float (&cs)[30]) {
  return j < 3 ? ndc[j < 3 ? j : 0] : j < 33 ? sn[j >= 3 && j < 33 ? j - 3 : 0] : j < 63 ? cs[j >= 33 && j < 63 ? j - 33 : 0] : 0.f;
}

template <int S>
__device__ __forceinline__ void mvs_point_inputs(const bmv_mvs_render_args& a, const MvsCams& mc, int ray, int k,
                                                 int h, float (&e)[32], float (&f)[10], float (&dv)[2], float& z,
                                                 float& vis) {
  // No multiply-add contraction in the geometry: which products the compiler fuses differs between the kernels this is
  // inlined into (BMV_MVS_SPLIT 0 / 1 are two instantiations), one ulp of ndc is 6e-5 of sin(512 ndc), and six gated
  // layers amplify that to 5e-6 of alpha -- the two forms must see the SAME MLP inputs (tests/test_gpu_mvs.py)
#pragma clang fp contract(off)
  const float* r = a.rays + (size_t)ray * 8;
  const float o[3] = {r[0], r[1], r[2]}, d[3] = {r[3], r[4], r[5]};
  const float t = linspace01(k, a.Ns);
  z = r[6] * (1.f - t) + r[7] * t;  // network.py:952
  const float xyz[3] = {o[0] + d[0] * z, o[1] + d[1] * z, o[2] + d[2] * z};
  const float inv_w = (float)(a.W - 1), inv_h = (float)(a.H - 1);
  float ndc[3];
  {  // a22: reference view (view 0), depth normalised by the sweep bounds, re-mapped into the padded volume
    const Cam& c = mc.cam[0];
    float cx = xyz[0] * c.E[0] + xyz[1] * c.E[1] + xyz[2] * c.E[2] + c.E[3];
    float cy = xyz[0] * c.E[4] + xyz[1] * c.E[5] + xyz[2] * c.E[6] + c.E[7];
    float cz = xyz[0] * c.E[8] + xyz[1] * c.E[9] + xyz[2] * c.E[10] + c.E[11];
    float qx = cx * c.Kf[0] + cy * c.Kf[1] + cz * c.Kf[2];
    float qy = cx * c.Kf[3] + cy * c.Kf[4] + cz * c.Kf[5];
    float qz = cx * c.Kf[6] + cy * c.Kf[7] + cz * c.Kf[8];
    float u = (qx / qz + 0.f) / inv_w, v = (qy / qz + 0.f) / inv_h;
    float wf = (inv_w + 1.f) / 4.f, hf = (inv_h + 1.f) / 4.f;
    float p2 = (float)(a.pad * 2);
    ndc[1] = v * hf / (hf + p2) + (float)a.pad / (hf + p2);
    ndc[0] = u * wf / (wf + p2) + (float)a.pad / (wf + p2);
    ndc[2] = (qz - mc.near_v) / (mc.far_v - mc.near_v);
  }
  {  // a23: regularised volume, trilinear, zeros padding; this half's 4 channels
    Taps3 t3 = taps3_zeros(ndc[0], ndc[1], ndc[2], a.wp, a.hp, a.D);
    size_t cs = (size_t)a.D * a.hp * a.wp;
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = tap3_fetch(a.volume + (size_t)(2 * j + h) * cs, t3);
  }
  vis = 0.f;
  const size_t plane = (size_t)a.H * a.W;
#pragma unroll
  for (int i = 0; i < S; ++i) {  // build_color_volume: rgb (bilinear, border) + inside flag per view
    const Cam& c = mc.cam[i];
    float cx = xyz[0] * c.E[0] + xyz[1] * c.E[1] + xyz[2] * c.E[2] + c.E[3];
    float cy = xyz[0] * c.E[4] + xyz[1] * c.E[5] + xyz[2] * c.E[6] + c.E[7];
    float cz = xyz[0] * c.E[8] + xyz[1] * c.E[9] + xyz[2] * c.E[10] + c.E[11];
    float qx = cx * c.Kf[0] + cy * c.Kf[1] + cz * c.Kf[2];
    float qy = cx * c.Kf[3] + cy * c.Kf[4] + cz * c.Kf[5];
    float qz = cx * c.Kf[6] + cy * c.Kf[7] + cz * c.Kf[8];
    float gx = ((qx / qz + 0.f) / inv_w) * 2.f - 1.f, gy = ((qy / qz + 0.f) / inv_h) * 2.f - 1.f;
    Taps2 t2 = taps_border(unnorm(gx, a.W), unnorm(gy, a.H), a.W, a.H);
    const float* img = a.src_inps + (size_t)i * 3 * plane;
    // feature slots 4+2i, 5+2i hold (r | g) and (b | inside) for (half 0 | half 1)
    float c0 = tap_fetch(img + (size_t)h * plane, t2) * 0.5f + 0.5f;          // r or g
    float c1 = tap_fetch(img + (size_t)2 * plane, t2) * 0.5f + 0.5f;          // b
    float ins = (gx > -1.f && gx < 1.f && gy > -1.f && gy < 1.f) ? 1.f : 0.f;
    f[4 + 2 * i] = c0;
    f[5 + 2 * i] = h ? ins : c1;
    if (a.mask) vis += visible(c, xyz, inv_w, inv_h);
  }
  {  // gen_dir_feature: unit ray direction in the reference camera frame
    float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float dn[3] = {d[0] / n, d[1] / n, d[2] / n};
    const Cam& c = mc.cam[0];
    float a0 = dn[0] * c.E[0] + dn[1] * c.E[1] + dn[2] * c.E[2];
    float a1 = dn[0] * c.E[4] + dn[1] * c.E[5] + dn[2] * c.E[6];
    float a2 = dn[0] * c.E[8] + dn[1] * c.E[9] + dn[2] * c.E[10];
    dv[0] = h ? a1 : a0;
    dv[1] = h ? 0.f : a2;
  }
  {  // a24: positional encoding of the NDC point
    float sn[30], cs[30];
#pragma unroll
    for (int q = 0; q < 10; ++q)
#pragma unroll
      for (int c = 0; c < 3; ++c) sincosf(ndc[c] * (float)(1 << q), &sn[q * 3 + c], &cs[q * 3 + c]);
#pragma unroll
    for (int tt = 0; tt < 32; ++tt) {
      float v0 = embed_value(2 * tt, ndc, sn, cs), v1 = embed_value(2 * tt + 1, ndc, sn, cs);
      e[tt] = h ? v1 : v0;
    }
  }
}

template <int S, bool SPLIT = false>
__global__ void __launch_bounds__(256, 1) mvs_render_kernel(bmv_mvs_render_args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* small = lds;                                   // MvsMlp::S_TOTAL floats (padded to 1 KB for the DMA buffers)
  float* buf = lds + kMvsSmall;                         // kMvsBuffers weight-chunk buffers
  MvsCams* mc = reinterpret_cast<MvsCams*>(buf + kMvsBuffers * MvsMlp::CHUNK_MAX);
  if (a.blob)
    for (int i = threadIdx.x; i < MvsMlp::S_TOTAL; i += blockDim.x) small[i] = a.blob[MvsMlp::A_TOTAL + i];
  if ((int)threadIdx.x < S) load_cam(a.src_exts + threadIdx.x * 16, a.src_ixts + threadIdx.x * 9, 1.f, mc->cam[threadIdx.x]);
  if (threadIdx.x == 32) {
    float n0 = a.near_far[0], n1 = a.near_far[1];
    mc->near_v = fminf(n0, n1), mc->far_v = fmaxf(n0, n1);  // batch['near_far'].min() / .max()
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 31, h = lane >> 5;
  const long npts = (long)(a.ray_end - a.ray_begin) * a.Ns;
  const long ntiles = (npts + 31) / 32;
  const long per_round = (long)gridDim.x * 4;
  ChunkPipe pipe{0, false};
  float wb[4][10] = {};
  if (a.blob) {
    start_chunks<SPLIT>(a.blob, buf);               // under the gathers / sincos of the first tile
    load_pts_bias_weights(a.blob, lane, wb);
  }
  for (long round = 0; round * per_round < ntiles; ++round) {   // uniform trip count across the workgroup
    pipe.more = (round + 1) * per_round < ntiles;
    long tile = round * per_round + (long)blockIdx.x * 4 + wave;
    long pt = tile * 32 + s;
    bool valid = pt < npts;
    long pc = valid ? pt : npts - 1;
    int ray = a.ray_begin + (int)(pc / a.Ns), k = (int)(pc % a.Ns);
    float e[32], f[10], dv[2], z, vis, res[4];
    mvs_point_inputs<S>(a, *mc, ray, k, h, e, f, dv, z, vis);
    size_t gi = (size_t)ray * a.Ns + k;
    if (a.inputs86 && valid) {  // test / API-parity hook: the reference's 86-wide MLP input row
      float* row = a.inputs86 + gi * 86;
#pragma unroll
      for (int t = 0; t < 32; ++t)
        if (2 * t + h < 63) row[2 * t + h] = e[t];
#pragma unroll
      for (int t = 0; t < 10; ++t) row[63 + 2 * t + h] = f[t];
      if (h == 0) row[83] = dv[0], row[85] = dv[1];
      else row[84] = dv[0];
    }
    if (a.blob) {
      mvs_mlp_forward<SPLIT>(a.blob, small, buf, pipe, lane, wb, e, f, dv, res);
      if (valid && h == 0) {
        float4 o4 = {res[0], res[1], res[2], res[3]};
        reinterpret_cast<float4*>(a.raw)[gi] = o4;
      }
    }
    if (valid && h == 0) {
      if (a.z_vals) a.z_vals[gi] = z;
      if (a.mask) a.mask[gi] = vis / (float)S;
    }
  }
}

template <bool SPLIT>
__global__ void __launch_bounds__(256, 1) mvs_mlp_kernel(const float* __restrict__ x, const float* __restrict__ blob,
                                                          long npts, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* small = lds;
  float* buf = lds + kMvsSmall;
  for (int i = threadIdx.x; i < MvsMlp::S_TOTAL; i += blockDim.x) small[i] = blob[MvsMlp::A_TOTAL + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 31, h = lane >> 5;
  const long ntiles = (npts + 31) / 32;
  const long per_round = (long)gridDim.x * 4;
  ChunkPipe pipe{0, false};
  start_chunks<SPLIT>(blob, buf);
  float wb[4][10];
  load_pts_bias_weights(blob, lane, wb);
  for (long round = 0; round * per_round < ntiles; ++round) {
    pipe.more = (round + 1) * per_round < ntiles;
    long pt = (round * per_round + (long)blockIdx.x * 4 + wave) * 32 + s;
    bool valid = pt < npts;
    const float* row = x + (valid ? pt : npts - 1) * 86;
    float e[32], f[10], dv[2], res[4];
#pragma unroll
    for (int t = 0; t < 32; ++t) e[t] = (2 * t + h < 63) ? row[2 * t + h] : 0.f;
#pragma unroll
    for (int t = 0; t < 10; ++t) f[t] = row[63 + 2 * t + h];
    dv[0] = row[83 + h];
    dv[1] = h ? 0.f : row[85];
    // the row is in registers BEFORE the next weight chunk is requested: a wait for these loads behind that request
    // would have to be vmcnt(0), i.e. would drain the prefetch
#pragma unroll
    for (int t = 0; t < 32; ++t) asm volatile("" ::"v"(e[t]));
#pragma unroll
    for (int t = 0; t < 10; ++t) asm volatile("" ::"v"(f[t]));
    asm volatile("" ::"v"(dv[0]), "v"(dv[1]));
    mvs_mlp_forward<SPLIT>(blob, small, buf, pipe, lane, wb, e, f, dv, res);
    if (valid && h == 0) {
      float4 o4 = {res[0], res[1], res[2], res[3]};
      reinterpret_cast<float4*>(out)[pt] = o4;
    }
  }
}

// boost_mvsnerf calc_mask (boost_mvsnerf/network.py:23-45): march Ns samples, viewport visibility fraction
__global__ void mvs_march_mask_kernel(const float* __restrict__ rays, const float* __restrict__ exts,
                                      const float* __restrict__ ixts, long total, int Ns, int V, float inv_w,
                                      float inv_h, float* __restrict__ zv, float* __restrict__ mask) {
  extern __shared__ float smem[];
  Cam* cams = reinterpret_cast<Cam*>(smem);
  for (int v = threadIdx.x; v < V; v += blockDim.x) load_cam(exts + (size_t)v * 16, ixts + (size_t)v * 9, 1.f, cams[v]);
  __syncthreads();
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  long ray = i / Ns;
  int k = (int)(i - ray * Ns);
  const float* r = rays + ray * 8;
  float t = linspace01(k, Ns);
  float z = r[6] * (1.f - t) + r[7] * t;
  float xyz[3] = {r[0] + r[3] * z, r[1] + r[4] * z, r[2] + r[5] * z};
  float acc = 0.f;
  for (int v = 0; v < V; ++v) acc += visible(cams[v], xyz, inv_w, inv_h);
  zv[i] = z;
  mask[i] = acc / (float)V;
}

static constexpr size_t kMvsLds = (kMvsSmall + kMvsBuffers * MvsMlp::CHUNK_MAX) * sizeof(float) + sizeof(MvsCams);
static_assert(kMvsLds <= 160 * 1024, "LDS of a gfx950 CU");

// ---------------------------------------------------------------------------
// Backward of the MVSNeRF path (training).  Gradient reaches the network only through the masked variance channels of
// the padded sweep (-> source features) and through the trilinear lookup of the regularised volume (-> volume); the
// colour channels, the positional encoding and the view directions are data.
// ---------------------------------------------------------------------------
// a19 + a20 backward: d_out (B, 3S+C, D, hp, wp) -> d_feats (B,S,C,h,w) (atomics); one thread per padded voxel and CB
// channels.  var = sum(v^2) inv - (sum(v) inv)^2 with inv = 1 / count (a constant of the mask): d var / d v_s =
// 2 inv (v_s - m) for every view that contributes (the reference view inside the un-padded window, a source view
// through its 4 zero-padded taps).
template <int CB, int S, int MODE>
__global__ void __launch_bounds__(256) mvs_sweep_bwd_kernel(const float* __restrict__ feats, const float* __restrict__ proj,
                                                             const float* __restrict__ depth_values,
                                                             const float* __restrict__ d_out, int C, int h, int w, int D,
                                                             int pad, float* __restrict__ d_feats, FixedWs fixed) {
  ScatterAcc<MODE> sacc = make_acc<MODE>(d_feats, fixed, 0);
  const int b = blockIdx.z, c0 = blockIdx.y * CB;
  const int hp = h + 2 * pad, wp = w + 2 * pad;
  const size_t nvox = (size_t)D * hp * wp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox) return;
  const int xp = (int)(i % wp), yp = (int)((i / wp) % hp), d = (int)(i / ((size_t)hp * wp));
  const int x = xp - pad, y = yp - pad;
  const bool inside = x >= 0 && x < w && y >= 0 && y < h;
  const size_t plane = (size_t)h * w;
  const float depth = depth_values[b * D + d];
  Taps2 tp[S];
  float count = 1.f;
#pragma unroll
  for (int s = 1; s < S; ++s) {
    const float* P = proj + ((size_t)b * S + s) * 12;
    float px = P[0] * x + P[1] * y + P[2] + P[3] / depth;
    float py = P[4] * x + P[5] * y + P[6] + P[7] / depth;
    float pz = P[8] * x + P[9] * y + P[10] + P[11] / depth;
    float gx = (px / pz) / ((float)(w - 1) * 0.5f) - 1.f;
    float gy = (py / pz) / ((float)(h - 1) * 0.5f) - 1.f;
    count += (gx > -1.f && gx < 1.f && gy > -1.f && gy < 1.f) ? 1.f : 0.f;
    tp[s] = taps_zeros(unnorm(gx, w), unnorm(gy, h), w, h);
  }
  const float inv = 1.f / count;
  const float* g_var = d_out + ((size_t)b * (3 * S + C) + 3 * S + c0) * nvox + i;
  const size_t ref_off = inside ? (size_t)y * w + x : 0;
  for (int c = 0; c < CB; ++c) {
    float v[S], sum = 0.f;
    v[0] = inside ? feats[(((size_t)b * S) * C + c0 + c) * plane + ref_off] : 0.f;
    sum = v[0];
#pragma unroll
    for (int s = 1; s < S; ++s) {
      v[s] = tap_fetch(feats + (((size_t)b * S + s) * C + c0 + c) * plane, tp[s]);
      sum += v[s];
    }
    const float m = sum * inv, g = g_var[(size_t)c * nvox];
    if (g == 0.f) continue;
    if (inside) sacc.add(d_feats + (((size_t)b * S) * C + c0 + c) * plane + ref_off, 2.f * inv * g * (v[0] - m));
#pragma unroll
    for (int s = 1; s < S; ++s) {
      const float gw = 2.f * inv * g * (v[s] - m);
      float* df = d_feats + (((size_t)b * S + s) * C + c0 + c) * plane;
      if (tp[s].w00 != 0.f) sacc.add(df + tp[s].o00, tp[s].w00 * gw);
      if (tp[s].w01 != 0.f) sacc.add(df + tp[s].o01, tp[s].w01 * gw);
      if (tp[s].w10 != 0.f) sacc.add(df + tp[s].o10, tp[s].w10 * gw);
      if (tp[s].w11 != 0.f) sacc.add(df + tp[s].o11, tp[s].w11 * gw);
    }
  }
}

// a22 + a23 backward: d_feat (N, Ns, 8) = gradient of the 8 volume channels of the MLP input -> d_volume (8,D,hp,wp)
// (atomics, trilinear, zeros padding); the NDC point of every sample is recomputed from its ray exactly as the forward
// does (mvs_point_inputs).
template <int MODE>
__global__ void __launch_bounds__(256) mvs_vol_feat_bwd_kernel(const float* __restrict__ rays, const float* __restrict__ ext0,
                                                                const float* __restrict__ ixt0,
                                                                const float* __restrict__ near_far,
                                                                const float* __restrict__ d_feat, long npts, int Ns, int H,
                                                                int W, int D, int hp, int wp, int pad,
                                                                float* __restrict__ d_volume, FixedWs fixed) {
  ScatterAcc<MODE> sacc = make_acc<MODE>(d_volume, fixed, 0);
  __shared__ Cam cam;
  __shared__ float nf[2];
  if (threadIdx.x == 0) load_cam(ext0, ixt0, 1.f, cam);
  if (threadIdx.x == 32) {
    float n0 = near_far[0], n1 = near_far[1];
    nf[0] = fminf(n0, n1), nf[1] = fmaxf(n0, n1);
  }
  __syncthreads();
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npts) return;
  const long ray = idx / Ns;
  const int k = (int)(idx - ray * Ns);
  const float* r = rays + ray * 8;
  const float t = linspace01(k, Ns);
  const float z = r[6] * (1.f - t) + r[7] * t;
  const float xyz[3] = {r[0] + r[3] * z, r[1] + r[4] * z, r[2] + r[5] * z};
  const float inv_w = (float)(W - 1), inv_h = (float)(H - 1);
  const Cam& c = cam;
  float cx = xyz[0] * c.E[0] + xyz[1] * c.E[1] + xyz[2] * c.E[2] + c.E[3];
  float cy = xyz[0] * c.E[4] + xyz[1] * c.E[5] + xyz[2] * c.E[6] + c.E[7];
  float cz = xyz[0] * c.E[8] + xyz[1] * c.E[9] + xyz[2] * c.E[10] + c.E[11];
  float qx = cx * c.Kf[0] + cy * c.Kf[1] + cz * c.Kf[2];
  float qy = cx * c.Kf[3] + cy * c.Kf[4] + cz * c.Kf[5];
  float qz = cx * c.Kf[6] + cy * c.Kf[7] + cz * c.Kf[8];
  float u = (qx / qz + 0.f) / inv_w, v = (qy / qz + 0.f) / inv_h;
  float wf = (inv_w + 1.f) / 4.f, hf = (inv_h + 1.f) / 4.f;
  float p2 = (float)(pad * 2);
  const float n1 = v * hf / (hf + p2) + (float)pad / (hf + p2);
  const float n0 = u * wf / (wf + p2) + (float)pad / (wf + p2);
  const float n2 = (qz - nf[0]) / (nf[1] - nf[0]);
  const Taps3 t3 = taps3_zeros(n0, n1, n2, wp, hp, D);
  const size_t cs = (size_t)D * hp * wp;
  const float* g = d_feat + idx * 8;
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    const float gc = g[ch];
    if (gc == 0.f) continue;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (t3.w[q] != 0.f) sacc.add(d_volume + (size_t)ch * cs + t3.o[q], t3.w[q] * gc);
  }
}

}  // namespace bmv

using namespace bmv;

extern "C" {

int bmv_mvs_proj_mats(const float* src_exts, const float* src_ixts, int B, int S, float* proj, bmv_stream_t stream) {
  BMV_REQUIRE(src_exts && src_ixts && proj, "bmv_mvs_proj_mats: null pointer");
  BMV_REQUIRE(B > 0 && S > 0, "bmv_mvs_proj_mats: bad shape");
  hipLaunchKernelGGL(mvs_proj_mats_kernel, dim3(cdiv(B * S, 64)), dim3(64), 0, as_stream(stream), src_exts, src_ixts, B,
                     S, proj);
  BMV_LAUNCH_END("bmv_mvs_proj_mats");
}

int bmv_resize_bilinear(const float* src, int n, int C, int H, int W, int h, int w, float* dst, bmv_stream_t stream) {
  BMV_REQUIRE(src && dst, "bmv_resize_bilinear: null pointer");
  BMV_REQUIRE(n > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0, "bmv_resize_bilinear: bad shape");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(cdiv(h * w, 256), n * C), dim3(256), 0, as_stream(stream), src, H, W,
                     h, w, dst);
  BMV_LAUNCH_END("bmv_resize_bilinear");
}

int bmv_mvs_sweep_fwd(const float* imgs, const float* feats, const float* proj, const float* depth_values, int B,
                      int S, int C, int h, int w, int D, int pad, float* volume, bmv_stream_t stream) {
  BMV_REQUIRE(imgs && feats && proj && depth_values && volume, "bmv_mvs_sweep_fwd: null pointer");
  BMV_REQUIRE(B > 0 && h > 1 && w > 1 && D > 0 && pad >= 0, "bmv_mvs_sweep_fwd: bad shape");
  if (C != 32 || S != 3) {
    set_error("bmv_mvs_sweep_fwd: built for C=32 feature channels and S=3 views (got C=%d S=%d)", C, S);
    return BMV_ERR_UNSUPPORTED;
  }
  size_t nvox = (size_t)D * (h + 2 * pad) * (w + 2 * pad);
  hipLaunchKernelGGL((mvs_sweep_kernel<32, 3>), dim3(cdiv(nvox, 256), B), dim3(256), 0, as_stream(stream), imgs, feats,
                     proj, depth_values, h, w, D, pad, volume);
  BMV_LAUNCH_END("bmv_mvs_sweep_fwd");
}

int bmv_mvs_sweep_cl_fwd(const float* imgs, const float* feats_cl, const float* proj, const float* depth_values, int B,
                         int S, int C, int h, int w, int D, int pad, float* volume, bmv_stream_t stream) {
  BMV_REQUIRE(imgs && feats_cl && proj && depth_values && volume, "bmv_mvs_sweep_cl_fwd: null pointer");
  BMV_REQUIRE(B > 0 && h > 1 && w > 1 && D > 0 && pad >= 0, "bmv_mvs_sweep_cl_fwd: bad shape");
  if (C != 32 || S != 3) {
    set_error("bmv_mvs_sweep_cl_fwd: built for C=32 feature channels and S=3 views (got C=%d S=%d)", C, S);
    return BMV_ERR_UNSUPPORTED;
  }
  const size_t nvox = (size_t)D * (h + 2 * pad) * (w + 2 * pad);
  if ((3 * S + C) * nvox * 4 >= ((size_t)1 << 31) || (size_t)S * h * w * C * 4 >= ((size_t)1 << 31)) {
    set_error("bmv_mvs_sweep_cl_fwd: one batch item of the volume / the features must stay below 2 GiB (32-bit offsets)");
    return BMV_ERR_UNSUPPORTED;
  }
  // BMV_MVS_SWEEP_AUX: cache-policy bits of the stores (2 = nt); + 256 = stores staged through LDS as whole channel rows
  const int aux = bmv::tuning("BMV_MVS_SWEEP_AUX", 0x102);
  const dim3 grid(cdiv(4 * nvox, 256), B);
#define BMV_MVS_CL(A)                                                                                            \
  hipLaunchKernelGGL((mvs_sweep_cl_kernel<3, A>), grid, dim3(256), 0, as_stream(stream), imgs, feats_cl, proj, \
                     depth_values, h, w, D, pad, volume)
  if (aux == 2) BMV_MVS_CL(2);
  else if (aux == 0x102) BMV_MVS_CL(0x102);
  else if (aux == 0x100) BMV_MVS_CL(0x100);
  else BMV_MVS_CL(0);
#undef BMV_MVS_CL
  BMV_LAUNCH_END("bmv_mvs_sweep_cl_fwd");
}

int bmv_mvs_march_mask(const float* rays, const float* src_exts, const float* src_ixts, int N, int Ns, int V,
                       float inv_w, float inv_h, float* z_vals, float* mask, bmv_stream_t stream) {
  BMV_REQUIRE(rays && src_exts && src_ixts && z_vals && mask, "bmv_mvs_march_mask: null pointer");
  BMV_REQUIRE(N >= 0 && Ns > 0 && V > 0 && V <= 64, "bmv_mvs_march_mask: bad shape");
  long total = (long)N * Ns;
  if (total == 0) return BMV_OK;
  hipLaunchKernelGGL(mvs_march_mask_kernel, dim3(cdiv(total, 256)), dim3(256), sizeof(Cam) * V, as_stream(stream), rays,
                     src_exts, src_ixts, total, Ns, V, inv_w, inv_h, z_vals, mask);
  BMV_LAUNCH_END("bmv_mvs_march_mask");
}

#ifdef BMV_MVS_STAMPS
int bmv_debug_fetch_mvs_stamps(float* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_mvs_stamps), sizeof(float) * 256 * 4 * kMvsStamps);
}
#endif
int bmv_mvs_mlp_blob_size(void) { return MvsMlp::TOTAL_S; }   // (fp32 chunks + small tables + the fifteen bf16 x 3 chunks)

int bmv_mvs_mlp_pack_weights(const bmv_mvs_mlp_params* p, float* blob, bmv_stream_t stream) {
  BMV_REQUIRE(p && blob, "bmv_mvs_mlp_pack_weights: null pointer");
  const float* const* pp = reinterpret_cast<const float* const*>(p);
  for (int i = 0; i < 22; ++i) BMV_REQUIRE(pp[i], "bmv_mvs_mlp_pack_weights: parameter %d is null", i);
  hipLaunchKernelGGL(mvs_mlp_pack_kernel, dim3(cdiv(MvsMlp::TOTAL_S, 256)), dim3(256), 0, as_stream(stream), *p, blob);
  BMV_LAUNCH_END("bmv_mvs_mlp_pack_weights");
}

static int mvs_lds_ok(const void* k) {
  return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMvsLds) == hipSuccess;
}

int bmv_mvs_mlp_fwd(const float* x, const float* blob, long npts, float* out, bmv_stream_t stream) {
  BMV_REQUIRE(x && blob && out, "bmv_mvs_mlp_fwd: null pointer");
  BMV_REQUIRE(npts >= 0, "bmv_mvs_mlp_fwd: npts=%ld", npts);
  if (npts == 0) return BMV_OK;
  // BMV_MVS_SPLIT (default 1): all matrix chunks but pts_bias as bf16 MFMAs on three-piece fp32 operands (fp32 accuracy); 0: all fp32
  const bool split = bmv::tuning("BMV_MVS_SPLIT", 1) != 0;
  BMV_REQUIRE(mvs_lds_ok(split ? reinterpret_cast<const void*>(mvs_mlp_kernel<true>) : reinterpret_cast<const void*>(mvs_mlp_kernel<false>)),
              "bmv_mvs_mlp_fwd: cannot reserve LDS");
  long ntiles = (npts + 31) / 32;
  unsigned grid = (unsigned)((ntiles + 3) / 4 < 256 ? (ntiles + 3) / 4 : 256);
  if (split) hipLaunchKernelGGL(mvs_mlp_kernel<true>, dim3(grid), dim3(256), kMvsLds, as_stream(stream), x, blob, npts, out);
  else hipLaunchKernelGGL(mvs_mlp_kernel<false>, dim3(grid), dim3(256), kMvsLds, as_stream(stream), x, blob, npts, out);
  BMV_LAUNCH_END("bmv_mvs_mlp_fwd");
}

int bmv_mvs_render_fwd(const bmv_mvs_render_args* a, bmv_stream_t stream) {
  BMV_REQUIRE(a, "bmv_mvs_render_fwd: null args");
  BMV_REQUIRE(a->rays && a->volume && a->src_inps && a->src_exts && a->src_ixts && a->near_far,
              "bmv_mvs_render_fwd: null pointer");
  BMV_REQUIRE(a->blob ? (a->raw != nullptr) : (a->inputs86 != nullptr),
              "bmv_mvs_render_fwd: need raw output (with weights) or inputs86 (without)");
  BMV_REQUIRE(a->S == 3, "bmv_mvs_render_fwd: S=%d, built for 3 source views per cost volume", a->S);
  BMV_REQUIRE(a->N > 0 && a->Ns > 0 && a->D > 0 && a->hp > 0 && a->wp > 0 && a->H > 1 && a->W > 1,
              "bmv_mvs_render_fwd: bad shape");
  BMV_REQUIRE(a->ray_begin >= 0 && a->ray_end <= a->N && a->ray_begin <= a->ray_end, "bmv_mvs_render_fwd: ray range");
  if (a->ray_begin == a->ray_end) return BMV_OK;
  const bool split = bmv::tuning("BMV_MVS_SPLIT", 1) != 0;
  BMV_REQUIRE(mvs_lds_ok(split ? reinterpret_cast<const void*>(mvs_render_kernel<3, true>) : reinterpret_cast<const void*>(mvs_render_kernel<3, false>)),
              "bmv_mvs_render_fwd: cannot reserve LDS");
  long ntiles = ((long)(a->ray_end - a->ray_begin) * a->Ns + 31) / 32;
  unsigned grid = (unsigned)((ntiles + 3) / 4 < 256 ? (ntiles + 3) / 4 : 256);
  if (split) hipLaunchKernelGGL((mvs_render_kernel<3, true>), dim3(grid), dim3(256), kMvsLds, as_stream(stream), *a);
  else hipLaunchKernelGGL((mvs_render_kernel<3, false>), dim3(grid), dim3(256), kMvsLds, as_stream(stream), *a);
  BMV_LAUNCH_END("bmv_mvs_render_fwd");
}

static int mvs_sweep_bwd_impl(const float* feats, const float* proj, const float* depth_values, const float* d_volume,
                              int B, int S, int C, int h, int w, int D, int pad, float* d_feats, FixedWs fixed,
                              bmv_stream_t stream, const char* name) {
  BMV_REQUIRE(feats && proj && depth_values && d_volume && d_feats, "%s: null pointer", name);
  BMV_REQUIRE(B > 0 && h > 1 && w > 1 && D > 0 && pad >= 0, "%s: bad shape", name);
  if (S != 3 || C % 8 != 0) {
    set_error("%s: built for S=3 views and C %% 8 == 0 (got S=%d C=%d)", name, S, C);
    return BMV_ERR_UNSUPPORTED;
  }
  const size_t nvox = (size_t)D * (h + 2 * pad) * (w + 2 * pad);
  launch_modes(fixed, as_stream(stream), [&](auto mode) {
    hipLaunchKernelGGL((mvs_sweep_bwd_kernel<8, 3, decltype(mode)::value>), dim3(cdiv(nvox, 256), C / 8, B), dim3(256), 0,
                       as_stream(stream), feats, proj, depth_values, d_volume, C, h, w, D, pad, d_feats, fixed);
  });
  if (fixed.ws) fixed_finish(fixed, 0, (size_t)B * S * C * h * w, d_feats, as_stream(stream));
  BMV_LAUNCH_END(name);
}
int bmv_mvs_sweep_bwd(const float* feats, const float* proj, const float* depth_values, const float* d_volume, int B,
                      int S, int C, int h, int w, int D, int pad, float* d_feats, bmv_stream_t stream) {
  return mvs_sweep_bwd_impl(feats, proj, depth_values, d_volume, B, S, C, h, w, D, pad, d_feats, FixedWs{}, stream,
                            "bmv_mvs_sweep_bwd");
}
int bmv_mvs_sweep_bwd_fixed(const float* feats, const float* proj, const float* depth_values, const float* d_volume, int B,
                            int S, int C, int h, int w, int D, int pad, float* d_feats, long long* workspace,
                            bmv_stream_t stream) {
  BMV_REQUIRE(workspace, "bmv_mvs_sweep_bwd_fixed: null workspace");
  return mvs_sweep_bwd_impl(feats, proj, depth_values, d_volume, B, S, C, h, w, D, pad, d_feats, FixedWs{workspace}, stream,
                            "bmv_mvs_sweep_bwd_fixed");
}

static int mvs_vol_feat_bwd_impl(const float* rays, const float* src_ext0, const float* src_ixt0, const float* near_far,
                                 const float* d_feat, long N, int Ns, int H, int W, int D, int hp, int wp, int pad,
                                 float* d_volume, FixedWs fixed, bmv_stream_t stream, const char* name) {
  BMV_REQUIRE(rays && src_ext0 && src_ixt0 && near_far && d_feat && d_volume, "%s: null pointer", name);
  BMV_REQUIRE(N >= 0 && Ns > 0 && H > 1 && W > 1 && D > 0 && hp > 0 && wp > 0, "%s: bad shape", name);
  if (N == 0) {      // the float form adds into a caller-zeroed buffer; the fixed-point form WRITES its output
    if (fixed.ws) {
      fixed_zero(d_volume, (size_t)8 * D * hp * wp, as_stream(stream));
      BMV_LAUNCH_END(name);
    }
    return BMV_OK;
  }
  const long npts = N * Ns;
  launch_modes(fixed, as_stream(stream), [&](auto mode) {
    hipLaunchKernelGGL(mvs_vol_feat_bwd_kernel<decltype(mode)::value>, dim3(cdiv(npts, 256)), dim3(256), 0,
                       as_stream(stream), rays, src_ext0, src_ixt0, near_far, d_feat, npts, Ns, H, W, D, hp, wp, pad,
                       d_volume, fixed);
  });
  if (fixed.ws) fixed_finish(fixed, 0, (size_t)8 * D * hp * wp, d_volume, as_stream(stream));
  BMV_LAUNCH_END(name);
}
int bmv_mvs_vol_feat_bwd(const float* rays, const float* src_ext0, const float* src_ixt0, const float* near_far,
                         const float* d_feat, long N, int Ns, int H, int W, int D, int hp, int wp, int pad,
                         float* d_volume, bmv_stream_t stream) {
  return mvs_vol_feat_bwd_impl(rays, src_ext0, src_ixt0, near_far, d_feat, N, Ns, H, W, D, hp, wp, pad, d_volume, FixedWs{},
                               stream, "bmv_mvs_vol_feat_bwd");
}
int bmv_mvs_vol_feat_bwd_fixed(const float* rays, const float* src_ext0, const float* src_ixt0, const float* near_far,
                               const float* d_feat, long N, int Ns, int H, int W, int D, int hp, int wp, int pad,
                               float* d_volume, long long* workspace, bmv_stream_t stream) {
  BMV_REQUIRE(workspace, "bmv_mvs_vol_feat_bwd_fixed: null workspace");
  return mvs_vol_feat_bwd_impl(rays, src_ext0, src_ixt0, near_far, d_feat, N, Ns, H, W, D, hp, wp, pad, d_volume,
                               FixedWs{workspace}, stream, "bmv_mvs_vol_feat_bwd_fixed");
}

}  // extern "C"
