// Backward of ENeRF's tiny MLP (a11: Agg + NeRF, lib/networks/enerf/nerf.py:29-43, 74-89) on
// the fp32 matrix cores.
//
// Split.  One kernel (this file) recomputes the forward of a 32-sample tile in registers
// and back-propagates the DATA path: every product W^T d_pre is again a transposed MFMA
// product with the sample on the lane, using weight tables transposed once by a pack
// kernel, laid out so that the gradient of every layer input comes out in exactly the
// register slots the forward consumed it from (output register r == forward k-step r).
// The pre-activation gradients (and the three hidden activations that cannot be rebuilt
// from the inputs) are written to HBM as [row][sample] matrices; the WEIGHT gradients
// are then plain library GEMMs over the sample dimension, dW = D_pre @ ACT^T (host side,
// rocBLAS via torch.matmul) -- the reduction over 10^5..10^6 samples is the K dimension of
// a tall-skinny GEMM, not something to do with atomics.  Only the three 1-wide heads
// (agg_w_fc, sigma, color.2 weights) are reduced in-kernel (per-lane running sums, one
// wave reduction and 160 atomics per wave at the very end).
#include "mlp.hpp"

namespace bmv {

__host__ __device__ constexpr int inv_reg(int rho) { return (rho & 3) + 4 * (rho >> 3); }
__host__ __device__ constexpr int inv_half(int rho) { return (rho >> 2) & 1; }

template <int FEAT_CH>
struct MlpBwdLayout {
  using F = MlpLayout<FEAT_CH>;
  static constexpr int FC = F::FC, KFC = F::KFC, KF = F::KF;
  static constexpr int GSH_TILES = (2 * KFC + 15) / 16;  // 16 input slots (32 inputs) per tile
  // transposed tables: [step over output-neuron pairs][input-slot tile][64 lanes]
  static constexpr int T_CSH = 0;                              // color.0 shared:   32 steps x 3 tiles
  static constexpr int T_CV = T_CSH + 32 * 3 * 64;             // color.0 per view: 32 steps x 1..2 tiles
  static constexpr int CV_TILES = (KF + 15) / 16;
  static constexpr int T_L0 = T_CV + 32 * CV_TILES * 64;       // lr0:              32 steps x 1 tile
  static constexpr int T_FC = T_L0 + 32 * 64;                  // agg.fc:            8 steps x 1 tile
  static constexpr int T_GSH = T_FC + 8 * 64;                  // global_fc shared: 16 steps x GSH_TILES
  static constexpr int T_GV = T_GSH + 16 * GSH_TILES * 64;     // global_fc / view: 16 steps x GV_TILES
  static constexpr int GV_TILES = (KFC + 15) / 16;
  static constexpr int V_WV = T_GV + 16 * GV_TILES * 64;       // view_fc^T for d_dir: [KFC][4][2]
  static constexpr int TOTAL = (V_WV + KFC * 4 * 2 + 3) / 4 * 4;
  // rows of the per-sample matrices written for the weight-gradient GEMMs
  static constexpr int FCP = 2 * KFC;                          // padded channel rows per view
  static constexpr int R_DH = 0;                               // d_pre color.0        3 x 64
  static constexpr int R_DX = R_DH + 192;                      // d_pre lr0            64
  static constexpr int R_DFC = R_DX + 64;                      // d_pre agg.fc         16
  static constexpr int R_DG = R_DFC + 16;                      // d_pre global_fc      3 x 32
  static constexpr int R_DV = R_DG + 96;                       // d_pre view_fc        3 x FCP
  static constexpr int R_DS = R_DV + 3 * FCP;                  // d_pre agg_w x3, sigma, color.2 x3  (7)
  static constexpr int R_AX = R_DS + 8;                        // x                    64
  static constexpr int R_AIM16 = R_AX + 64;                    // relu(agg.fc)         16
  static constexpr int R_AIM = R_AIM16 + 16;                   // sum_i w_i g_i        32
  static constexpr int R_TOTAL = R_AIM + 32;
  static constexpr int IN_ROWS = FCP + 4;                      // d_img rows per view: channels (padded) + 4 dir
};

template <int FEAT_CH>
__global__ void nerf_pack_bwd_kernel(bmv_nerf_params p, float* __restrict__ blob) {
  using L = MlpBwdLayout<FEAT_CH>;
  constexpr int FC = L::FC, KFC = L::KFC, KF = L::KF, CW = 88 + FC + 4;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L::TOTAL) return;
  float v = 0.f;
  if (idx < L::V_WV) {
    int lane = idx & 63, e = idx >> 6;
    int rho = lane & 31, hh = lane >> 5;              // lane = (output row rho, half hh of the d_pre pair)
    int reg = inv_reg(rho), half = inv_half(rho);     // the forward slot this output row stands for
    if (idx < L::T_CV) {
      int q = e - L::T_CSH / 64, u = q / 3, t = 16 * (q % 3) + reg;
      int n = 32 * (u >> 4) + n16(u & 15, hh);
      int k = t < 32 ? 32 * (t >> 4) + n16(t & 15, half) : t < 36 ? 64 + 2 * (t - 32) + half : t < 44 ? 72 + n16(t - 36, half) : -1;
      if (k >= 0) v = p.color0_w[n * CW + k];
    } else if (idx < L::T_L0) {
      int q = e - L::T_CV / 64, u = q / L::CV_TILES, t = 16 * (q % L::CV_TILES) + reg;
      int n = 32 * (u >> 4) + n16(u & 15, hh);
      int k = -1;
      if (t < KFC) {
        int c = 2 * t + half;
        if (c < FC) k = 88 + c;
      } else if (t < KF) {
        k = 88 + FC + 2 * (t - KFC) + half;
      }
      if (k >= 0) v = p.color0_w[n * CW + k];
    } else if (idx < L::T_FC) {
      int u = e - L::T_L0 / 64, t = reg;
      int n = 32 * (u >> 4) + n16(u & 15, hh);
      int k = t < 4 ? 2 * t + half : t < 12 ? 8 + n16(t - 4, half) : -1;
      if (k >= 0) v = p.lr0_w[n * 24 + k];
    } else if (idx < L::T_GSH) {
      int u = e - L::T_FC / 64, t = reg;              // 16 neurons: pairs n16(u, hh), u < 8
      v = p.fc_w[n16(u, hh) * 32 + n16(t, half)];
    } else if (idx < L::T_GV) {
      int q = e - L::T_GSH / 64, u = q / L::GSH_TILES, t = 16 * (q % L::GSH_TILES) + reg;
      int n = n16(u, hh);
      if (t < 2 * KFC) {
        int which = t / KFC, c = 2 * (t % KFC) + half;
        if (c < FC) v = p.global_fc_w[n * 3 * FC + FC * (1 + which) + c];
      }
    } else {
      int q = e - L::T_GV / 64, u = q / L::GV_TILES, t = 16 * (q % L::GV_TILES) + reg;
      int n = n16(u, hh);
      if (t < KFC) {
        int c = 2 * t + half;
        if (c < FC) v = p.global_fc_w[n * 3 * FC + c];
      }
    }
  } else {
    int rel = idx - L::V_WV;
    if (rel < KFC * 4 * 2) {
      int h = rel & 1, q = (rel >> 1) & 3, j = rel >> 3, c = 2 * j + h;
      if (c < FC) v = p.view_fc_w[c * 4 + q];
    }
  }
  blob[idx] = v;
}

template <int FEAT_CH>
struct BwdOut {
  float* rows;   // (R_TOTAL, P)
  float* d_vox;  // (8, P)
  float* d_img;  // (3, IN_ROWS, P)
  float* vecs;   // 64 color.2 weight grad | 64 sigma weight grad | 32 agg_w_fc weight grad  (atomics)
};

template <int FEAT_CH>
__global__ void __launch_bounds__(256, 1) nerf_mlp_bwd_kernel(const float* __restrict__ vox_feat,
                                                               const float* __restrict__ img,
                                                               const float* __restrict__ d_out,
                                                               const float* __restrict__ blob_fwd,
                                                               const float* __restrict__ blob_bwd, long npts,
                                                               BwdOut<FEAT_CH> o) {
  using L = MlpLayout<FEAT_CH>;
  using LB = MlpBwdLayout<FEAT_CH>;
  constexpr int KFC = L::KFC, KF = L::KF;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* W = lds;                 // forward blob
  float* WB = lds + L::TOTAL;     // transposed tables
  for (int i = threadIdx.x; i < L::TOTAL / 4; i += blockDim.x)
    reinterpret_cast<float4*>(W)[i] = reinterpret_cast<const float4*>(blob_fwd)[i];
  for (int i = threadIdx.x; i < LB::TOTAL / 4; i += blockDim.x)
    reinterpret_cast<float4*>(WB)[i] = reinterpret_cast<const float4*>(blob_bwd)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 31, h = lane >> 5;
  const float* __restrict__ Wa = W + lane;
  const float* __restrict__ Wv = W + h;
  const float* __restrict__ Ta = WB + lane;
  const float ba = W[L::V_SC + 0], bs = W[L::V_SC + 1], bc2 = W[L::V_SC + 2];
  const long P = npts;

  // running sums of the three 1-wide weight gradients (this lane's neurons, over its samples)
  f32x16 acc_wc2[2], acc_ws[2], acc_wa;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_wc2[0][r] = acc_wc2[1][r] = acc_ws[0][r] = acc_ws[1][r] = acc_wa[r] = 0.f;

  const long ntiles = (npts + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long pt = tile * 32 + s;
    const bool valid = pt < npts;
    const long pc = valid ? pt : npts - 1;
    // ------------------------------------------------------------------ inputs
    float fin[3][KF], dir[3][4], vox[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vox[j] = vox_feat[pc * 8 + 2 * j + h];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float* q = img + (pc * 3 + i) * L::IN;
#pragma unroll
      for (int j = 0; j < KFC; ++j) fin[i][j] = (2 * j + h < L::FC) ? q[2 * j + h] : 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) fin[i][KFC + k] = q[L::FC + 2 * k + h];
#pragma unroll
      for (int k = 0; k < 4; ++k) dir[i][k] = q[L::FC + k];
    }
    float go[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) go[k] = valid ? d_out[pc * 4 + k] : 0.f;

    // ------------------------------------------------------------------ forward recompute (mlp.hpp, intermediates kept)
    auto pre_v = [&](int i, int j) -> float {
      return Wv[L::V_VF + (j * 5 + 4) * 2] + Wv[L::V_VF + (j * 5 + 0) * 2] * dir[i][0] +
             Wv[L::V_VF + (j * 5 + 1) * 2] * dir[i][1] + Wv[L::V_VF + (j * 5 + 2) * 2] * dir[i][2] +
             Wv[L::V_VF + (j * 5 + 3) * 2] * dir[i][3];
    };
    auto fval = [&](int i, int j) -> float { return fin[i][j] + fmaxf(pre_v(i, j), 0.f); };
    float var[KFC], mean[KFC];
#pragma unroll
    for (int j = 0; j < KFC; ++j) {
      float f0 = fval(0, j), f1 = fval(1, j), f2 = fval(2, j);
      float m = (f0 + f1 + f2) / 3.f;
      mean[j] = m;
      var[j] = ((f0 - m) * (f0 - m) + (f1 - m) * (f1 - m) + (f2 - m) * (f2 - m)) * 0.5f;
      BMV_FENCE_EVERY(j, 6);
    }
    BMV_FENCE();
    f32x16 gsh;
#pragma unroll
    for (int r = 0; r < 16; ++r) gsh[r] = Wv[L::V_BG + r * 2];
#pragma unroll
    for (int t = 0; t < KFC; ++t) {
      gsh = BMV_MFMA(Wa[L::A_GSH + t * 64], var[t], gsh);
      BMV_FENCE_EVERY(t, 6);
    }
#pragma unroll
    for (int t = 0; t < KFC; ++t) {
      gsh = BMV_MFMA(Wa[L::A_GSH + (KFC + t) * 64], mean[t], gsh);
      BMV_FENCE_EVERY(t, 6);
    }
    BMV_FENCE();
    f32x16 g[3];
    float aw[3], apre[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      g[i] = gsh;
#pragma unroll
      for (int t = 0; t < KFC; ++t) {
        g[i] = BMV_MFMA(Wa[L::A_GV + t * 64], fval(i, t), g[i]);
        BMV_FENCE_EVERY(t, 6);
      }
      float sdot = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        g[i][r] = fmaxf(g[i][r], 0.f);
        sdot += Wv[L::V_WA + r * 2] * g[i][r];
      }
      apre[i] = xhalf_sum(sdot) + ba;
      aw[i] = fmaxf(apre[i], 0.f);
      BMV_FENCE();
    }
    {
      float m = fmaxf(aw[0], fmaxf(aw[1], aw[2]));
      float e0 = __expf(aw[0] - m), e1 = __expf(aw[1] - m), e2 = __expf(aw[2] - m);
      float inv = 1.f / (e0 + e1 + e2);
      aw[0] = e0 * inv, aw[1] = e1 * inv, aw[2] = e2 * inv;
    }
    f32x16 im, q16;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      im[r] = aw[0] * g[0][r] + aw[1] * g[1][r] + aw[2] * g[2][r];
      q16[r] = Wv[L::V_BFC + r * 2];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      q16 = BMV_MFMA(Wa[L::A_FC + t * 64], im[t], q16);
      BMV_FENCE_EVERY(t, 8);
    }
    BMV_FENCE();
    float im16[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) im16[r] = fmaxf(q16[r], 0.f);
    f32x16 x[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[tl][r] = Wv[L::V_B0 + (tl * 16 + r) * 2];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      float b = t < 4 ? vox[t < 4 ? t : 0] : im16[t >= 4 ? t - 4 : 0];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) x[tl] = BMV_MFMA(Wa[L::A_L0 + (t * 2 + tl) * 64], b, x[tl]);
      BMV_FENCE_EVERY(t, 4);
    }
    float spre;
    {
      float sdot = 0.f;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          x[tl][r] = fmaxf(x[tl][r], 0.f);
          sdot += Wv[L::V_WS + (tl * 16 + r) * 2] * x[tl][r];
        }
      spre = xhalf_sum(sdot) + bs;
    }
    f32x16 csh[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) csh[tl][r] = Wv[L::V_BC + (tl * 16 + r) * 2];
#pragma unroll
    for (int t = 0; t < 44; ++t) {
      float b;
      if (t < 32)
        b = x[t < 32 ? (t >> 4) : 0][t & 15];
      else if (t < 36)
        b = vox[t >= 32 && t < 36 ? t - 32 : 0];
      else
        b = im16[t >= 36 ? t - 36 : 0];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) csh[tl] = BMV_MFMA(Wa[L::A_CSH + (t * 2 + tl) * 64], b, csh[tl]);
      BMV_FENCE_EVERY(t, 4);
    }
    BMV_FENCE();
    // colour logits need all three views before the softmax backward: first pass keeps only c_i
    float cl[3], cpre[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f32x16 hc[2] = {csh[0], csh[1]};
#pragma unroll
      for (int t = 0; t < KF; ++t) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) hc[tl] = BMV_MFMA(Wa[L::A_CV + (t * 2 + tl) * 64], fin[i][t], hc[tl]);
        BMV_FENCE_EVERY(t, 4);
      }
      float sdot = 0.f;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) sdot += Wv[L::V_WC2 + (tl * 16 + r) * 2] * fmaxf(hc[tl][r], 0.f);
      cpre[i] = xhalf_sum(sdot) + bc2;
      cl[i] = fmaxf(cpre[i], 0.f);
      BMV_FENCE();
    }
    {
      float m = fmaxf(cl[0], fmaxf(cl[1], cl[2]));
      float e0 = __expf(cl[0] - m), e1 = __expf(cl[1] - m), e2 = __expf(cl[2] - m);
      float inv = 1.f / (e0 + e1 + e2);
      cl[0] = e0 * inv, cl[1] = e1 * inv, cl[2] = e2 * inv;
    }

    // ------------------------------------------------------------------ backward
    float* rows = o.rows + pt;  // + row * P
    auto put = [&](int row, float v) {
      if (valid) rows[(long)row * P] = v;
    };
    // rgb = sum_i cw_i rgb_i : the colour channels FEAT_CH+{0,1,2} sit at (half0, slot J), (half1, J), (half0, J+1)
    constexpr int J = FEAT_CH / 2;
    const float g_mine0 = h == 0 ? go[0] : go[1];  // gradient of the channel in slot J of this half
    const float g_mine1 = h == 0 ? go[2] : 0.f;    // slot J+1: blue (half 0) or padding (half 1)
    float d_cw[3], dotc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      d_cw[i] = xhalf_sum(g_mine0 * fin[i][J] + g_mine1 * fin[i][J + 1]);
      dotc += cl[i] * d_cw[i];
    }
    float d_cpre[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      d_cpre[i] = cpre[i] > 0.f ? cl[i] * (d_cw[i] - dotc) : 0.f;
      if (h == 0) put(LB::R_DS + 4 + i, d_cpre[i]);
    }
    // per view: recompute hc_i, d_pre_h_i, its weight-gradient rows, input gradient, and the running sum over views
    f32x16 dhs[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dhs[0][r] = dhs[1][r] = 0.f;
    float d_in[3][KF];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f32x16 hc[2] = {csh[0], csh[1]};
#pragma unroll
      for (int t = 0; t < KF; ++t) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) hc[tl] = BMV_MFMA(Wa[L::A_CV + (t * 2 + tl) * 64], fin[i][t], hc[tl]);
        BMV_FENCE_EVERY(t, 4);
      }
      f32x16 dh[2];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float act = fmaxf(hc[tl][r], 0.f);
          acc_wc2[tl][r] += d_cpre[i] * act;
          float d = act > 0.f ? d_cpre[i] * Wv[L::V_WC2 + (tl * 16 + r) * 2] : 0.f;
          dh[tl][r] = d;
          dhs[tl][r] += d;
          put(LB::R_DH + i * 64 + 32 * tl + n16(r, h), d);
        }
      BMV_FENCE();
      // d_in_i = Wc[:, 88:]^T d_pre_h_i  (output register r == forward slot r)
      f32x16 di[LB::CV_TILES];
#pragma unroll
      for (int tt = 0; tt < LB::CV_TILES; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) di[tt][r] = 0.f;
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        float b = dh[u >> 4][u & 15];
#pragma unroll
        for (int tt = 0; tt < LB::CV_TILES; ++tt)
          di[tt] = BMV_MFMA(Ta[LB::T_CV + (u * LB::CV_TILES + tt) * 64], b, di[tt]);
        BMV_FENCE_EVERY(u, 4);
      }
#pragma unroll
      for (int t = 0; t < KF; ++t) d_in[i][t] = di[t >> 4][t & 15];
      // the blended colour reads the sampled source colour directly
      d_in[i][J] += cl[i] * g_mine0;
      d_in[i][J + 1] += cl[i] * g_mine1;
      BMV_FENCE();
    }
    // d_[x, vox, im16] = Wc[:, :88]^T sum_i d_pre_h_i
    f32x16 dxv[3];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dxv[tt][r] = 0.f;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      float b = dhs[u >> 4][u & 15];
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) dxv[tt] = BMV_MFMA(Ta[LB::T_CSH + (u * 3 + tt) * 64], b, dxv[tt]);
      BMV_FENCE_EVERY(u, 4);
    }
    BMV_FENCE();
    // sigma head and lr0
    const float d_spre = go[3] * (1.f / (1.f + __expf(-spre)));  // softplus' = sigmoid (threshold branch: 1)
    if (h == 0) put(LB::R_DS + 3, spre > 20.f ? go[3] : d_spre);
    const float d_sp = spre > 20.f ? go[3] : d_spre;
    f32x16 dx[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_ws[tl][r] += d_sp * x[tl][r];
        float d = dxv[tl][r] + d_sp * Wv[L::V_WS + (tl * 16 + r) * 2];
        d = x[tl][r] > 0.f ? d : 0.f;
        dx[tl][r] = d;
        put(LB::R_DX + 32 * tl + n16(r, h), d);
        put(LB::R_AX + 32 * tl + n16(r, h), x[tl][r]);
      }
    f32x16 dv24;  // slots 0..3 vox, 4..11 im16
#pragma unroll
    for (int r = 0; r < 16; ++r) dv24[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      dv24 = BMV_MFMA(Ta[LB::T_L0 + u * 64], dx[u >> 4][u & 15], dv24);
      BMV_FENCE_EVERY(u, 8);
    }
    BMV_FENCE();
    // vox gradient: slots 32..35 of the colour input + slots 0..3 of lr0
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (valid) o.d_vox[(long)(2 * j + h) * P + pt] = dxv[2][j] + dv24[j];
    // agg.fc
    float dfc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float d = dxv[2][4 + r] + dv24[4 + r];
      d = im16[r] > 0.f ? d : 0.f;
      dfc[r] = d;
      put(LB::R_DFC + n16(r, h), d);
      put(LB::R_AIM16 + n16(r, h), im16[r]);
    }
    f32x16 dim;
#pragma unroll
    for (int r = 0; r < 16; ++r) dim[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) dim = BMV_MFMA(Ta[LB::T_FC + u * 64], dfc[u], dim);
    BMV_FENCE();
#pragma unroll
    for (int r = 0; r < 16; ++r) put(LB::R_AIM + n16(r, h), im[r]);
    // softmax over views of the aggregation weights, agg_w_fc, global_fc
    float d_w[3], dotw = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float sdot = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sdot += dim[r] * g[i][r];
      d_w[i] = xhalf_sum(sdot);
      dotw += aw[i] * d_w[i];
    }
    f32x16 dgs;
#pragma unroll
    for (int r = 0; r < 16; ++r) dgs[r] = 0.f;
    f32x16 dg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float d_apre = apre[i] > 0.f ? aw[i] * (d_w[i] - dotw) : 0.f;
      if (h == 0) put(LB::R_DS + i, d_apre);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_wa[r] += d_apre * g[i][r];
        float d = aw[i] * dim[r] + d_apre * Wv[L::V_WA + r * 2];
        d = g[i][r] > 0.f ? d : 0.f;
        dg[i][r] = d;
        dgs[r] += d;
        put(LB::R_DG + i * 32 + n16(r, h), d);
      }
    }
    BMV_FENCE();
    // d_[var, mean] = Wg[:, F:]^T sum_i d_pre_g_i ; d_f_i = Wg[:, :F]^T d_pre_g_i
    f32x16 dvm[LB::GSH_TILES];
#pragma unroll
    for (int tt = 0; tt < LB::GSH_TILES; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dvm[tt][r] = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int tt = 0; tt < LB::GSH_TILES; ++tt)
        dvm[tt] = BMV_MFMA(Ta[LB::T_GSH + (u * LB::GSH_TILES + tt) * 64], dgs[u], dvm[tt]);
      BMV_FENCE_EVERY(u, 8);
    }
    BMV_FENCE();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f32x16 dfv[LB::GV_TILES];
#pragma unroll
      for (int tt = 0; tt < LB::GV_TILES; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dfv[tt][r] = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int tt = 0; tt < LB::GV_TILES; ++tt)
          dfv[tt] = BMV_MFMA(Ta[LB::T_GV + (u * LB::GV_TILES + tt) * 64], dg[i][u], dfv[tt]);
        BMV_FENCE_EVERY(u, 8);
      }
      float d_dir[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KFC; ++j) {
        float fj = fval(i, j);
        // var = sum (f - m)^2 / 2, mean = sum f / 3
        float d_f = dfv[j >> 4][j & 15] + dvm[j >> 4][j & 15] * (fj - mean[j]) +
                    dvm[(KFC + j) >> 4][(KFC + j) & 15] * (1.f / 3.f);
        d_in[i][j] += d_f;
        float d_pv = pre_v(i, j) > 0.f ? d_f : 0.f;
        if (valid) o.rows[(long)(LB::R_DV + i * LB::FCP + 2 * j + h) * P + pt] = d_pv;
#pragma unroll
        for (int q = 0; q < 4; ++q) d_dir[q] += d_pv * WB[LB::V_WV + ((j * 4 + q) << 1) + h];
        BMV_FENCE_EVERY(j, 6);
      }
      // the 4 direction inputs: this half's slots from color.0, plus view_fc^T summed over BOTH halves' channels
#pragma unroll
      for (int q = 0; q < 4; ++q) d_dir[q] = xhalf_sum(d_dir[q]);
      d_in[i][KFC] += h ? d_dir[1] : d_dir[0];
      d_in[i][KFC + 1] += h ? d_dir[3] : d_dir[2];
      // store the per-view input gradient rows: channel 2j+h, then dir 2k+h
      float* dimg = o.d_img + (long)i * LB::IN_ROWS * P + pt;
      if (valid) {
#pragma unroll
        for (int j = 0; j < KFC; ++j) dimg[(long)(2 * j + h) * P] = d_in[i][j];
        dimg[(long)(LB::FCP + h) * P] = d_in[i][KFC];
        dimg[(long)(LB::FCP + 2 + h) * P] = d_in[i][KFC + 1];
      }
      BMV_FENCE();
    }
  }
  // ---- flush the 1-wide weight gradients: reduce over the 32 samples of each half, then one atomic per neuron
  auto reduce32 = [&](float v) {
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
  };
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = reduce32(acc_wc2[tl][r]), b = reduce32(acc_ws[tl][r]);
      if (s == 0) {
        atomicAdd(o.vecs + 32 * tl + n16(r, h), a);
        atomicAdd(o.vecs + 64 + 32 * tl + n16(r, h), b);
      }
    }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float a = reduce32(acc_wa[r]);
    if (s == 0) atomicAdd(o.vecs + 128 + n16(r, h), a);
  }
}

}  // namespace bmv

using namespace bmv;

extern "C" {

int bmv_nerf_bwd_blob_size(int feat_ch) {
  if (feat_ch == 8) return MlpBwdLayout<8>::TOTAL;
  if (feat_ch == 32) return MlpBwdLayout<32>::TOTAL;
  set_error("bmv_nerf_bwd_blob_size: feat_ch=%d unsupported (8 or 32)", feat_ch);
  return BMV_ERR_UNSUPPORTED;
}

int bmv_nerf_bwd_rows(int feat_ch, int* d_img_rows) {
  if (feat_ch == 8) {
    if (d_img_rows) *d_img_rows = MlpBwdLayout<8>::IN_ROWS;
    return MlpBwdLayout<8>::R_TOTAL;
  }
  if (feat_ch == 32) {
    if (d_img_rows) *d_img_rows = MlpBwdLayout<32>::IN_ROWS;
    return MlpBwdLayout<32>::R_TOTAL;
  }
  set_error("bmv_nerf_bwd_rows: feat_ch=%d unsupported (8 or 32)", feat_ch);
  return BMV_ERR_UNSUPPORTED;
}

int bmv_nerf_pack_bwd_weights(const bmv_nerf_params* p, int feat_ch, float* blob, bmv_stream_t stream) {
  BMV_REQUIRE(p && blob, "bmv_nerf_pack_bwd_weights: null pointer");
  if (feat_ch == 8)
    hipLaunchKernelGGL(nerf_pack_bwd_kernel<8>, dim3(cdiv(MlpBwdLayout<8>::TOTAL, 256)), dim3(256), 0, as_stream(stream),
                       *p, blob);
  else if (feat_ch == 32)
    hipLaunchKernelGGL(nerf_pack_bwd_kernel<32>, dim3(cdiv(MlpBwdLayout<32>::TOTAL, 256)), dim3(256), 0,
                       as_stream(stream), *p, blob);
  else {
    set_error("bmv_nerf_pack_bwd_weights: feat_ch=%d unsupported (8 or 32)", feat_ch);
    return BMV_ERR_UNSUPPORTED;
  }
  BMV_LAUNCH_END("bmv_nerf_pack_bwd_weights");
}

int bmv_nerf_mlp_bwd(const float* vox_feat, const float* img, const float* d_out, const float* blob_fwd,
                     const float* blob_bwd, int feat_ch, long npts, float* rows, float* d_vox, float* d_img, float* vecs,
                     bmv_stream_t stream) {
  BMV_REQUIRE(vox_feat && img && d_out && blob_fwd && blob_bwd && rows && d_vox && d_img && vecs,
              "bmv_nerf_mlp_bwd: null pointer");
  BMV_REQUIRE(npts >= 0, "bmv_nerf_mlp_bwd: npts=%ld", npts);
  if (npts == 0) return BMV_OK;
  long ntiles = (npts + 31) / 32;
  unsigned grid = (unsigned)((ntiles + 3) / 4 < 256 ? (ntiles + 3) / 4 : 256);
#define BWD_CASE(FC)                                                                                               \
  if (feat_ch == FC) {                                                                                             \
    size_t lds = (size_t)(MlpLayout<FC>::TOTAL + MlpBwdLayout<FC>::TOTAL) * 4;                                     \
    BMV_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_mlp_bwd_kernel<FC>),                        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,           \
                "bmv_nerf_mlp_bwd: cannot reserve %zu B of LDS", lds);                                             \
    BwdOut<FC> o{rows, d_vox, d_img, vecs};                                                                        \
    hipLaunchKernelGGL(nerf_mlp_bwd_kernel<FC>, dim3(grid), dim3(256), lds, as_stream(stream), vox_feat, img, d_out, \
                       blob_fwd, blob_bwd, npts, o);                                                               \
    BMV_LAUNCH_END("bmv_nerf_mlp_bwd");                                                                            \
  }
  BWD_CASE(8)
  BWD_CASE(32)
#undef BWD_CASE
  set_error("bmv_nerf_mlp_bwd: feat_ch=%d unsupported (8 or 32)", feat_ch);
  return BMV_ERR_UNSUPPORTED;
}

}  // extern "C"
