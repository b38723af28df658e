// Backward of ENeRF's tiny MLP (a11: Agg + NeRF, lib/networks/enerf/nerf.py:29-43, 74-89) on
// the fp32 matrix cores.
//
// Three kernels.
//  1. nerf_mlp_bwd_kernel recomputes the forward of a 32-sample tile in registers and
//     back-propagates the DATA path: every product W^T d_pre is again a transposed MFMA
//     product with the sample on the lane, using weight tables transposed once by a pack
//     kernel, laid out so that the gradient of every layer input comes out in exactly the
//     register slots the forward consumed it from (output register r == forward k-step r).
//     The pre-activation gradients and the layer inputs are written to HBM as per-tile
//     matrices rows[tile][row][32 samples].  Phases are ordered so that little stays live
//     across them (the aggregation forward and lr0 are recomputed where their backward
//     needs them; 46 + 24 of ~700 MFMAs per tile).
//  2. nerf_wgrad_kernel computes every WEIGHT gradient dW = D_pre ACT^T on the matrix cores
//     with the SAMPLE index as the MFMA k dimension: lane (m, kk) holds row m of a 32-row
//     block for the 32 samples of tile 2p + kk (one contiguous 128-byte read), so 32
//     v_mfma_f32_32x32x2 steps reduce 64 samples into a 32x32 block of dW.  The four waves of
//     a workgroup own disjoint blocks (no LDS, no barrier); each workgroup writes its partial
//     blocks once.  Bias gradients are row sums of the same operands (VALU, free).
//  3. nerf_wgrad_finish_kernel sums the partials in a fixed order (deterministic, no
//     atomics) straight into tensors of the reference's parameter shapes.
// Only the two 1-wide heads whose inputs are not stored (agg_w_fc, color.2 weights) are
// reduced in kernel 1 (per-lane running sums, one wave reduction, one 96-float partial per
// wave; kernel 3 sums the waves' partials in a fixed order: no atomics anywhere, the whole
// backward is bit-reproducible).
#include "mlp.hpp"

namespace bmv {

__host__ __device__ constexpr int inv_reg(int rho) { return (rho & 3) + 4 * (rho >> 3); }
__host__ __device__ constexpr int inv_half(int rho) { return (rho >> 2) & 1; }

template <int FEAT_CH, int NV = 3>
struct MlpBwdLayout {
  static_assert(NV >= 2 && NV <= 4, "source views per cost volume");
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
  // rows of the per-tile matrices rows[tile][row][32] read by the weight-gradient kernel
  static constexpr int FCP = 2 * KFC;                          // padded channel rows per view
  static constexpr int INR = FCP + 4;                          // per-view input rows: channels (padded) + 4 dir
  static constexpr int R_DH = 0;                               // d_pre color.0        NV x 64
  static constexpr int R_DX = R_DH + NV * 64;                  // d_pre lr0            64
  static constexpr int R_DFC = R_DX + 64;                      // d_pre agg.fc         16
  static constexpr int R_DG = R_DFC + 16;                      // d_pre global_fc      NV x 32
  static constexpr int R_DV = R_DG + NV * 32;                  // d_pre view_fc        NV x FCP
  static constexpr int R_DS = R_DV + NV * FCP;                 // d_pre agg_w x NV, sigma, color.2 x NV, one zero row
  static constexpr int DS_ROWS = 2 * NV + 2;
  static constexpr int DS_SIGMA = NV, DS_C2 = NV + 1;          // rows of R_DS: agg_w 0.., sigma, color.2 NV + 1..
  static constexpr int R_AX = R_DS + DS_ROWS;                  // x = relu(lr0)        64
  static constexpr int R_AV24 = R_AX + 64;                     // vox 8 | relu(agg.fc) 16
  static constexpr int R_AIM = R_AV24 + 24;                    // sum_i w_i g_i        32
  static constexpr int R_IN = R_AIM + 32;                      // per-view inputs      NV x INR
  static constexpr int R_F = R_IN + NV * INR;                  // f_i (Agg residual)   NV x FCP
  static constexpr int R_VAR = R_F + NV * FCP;                 // var | mean           2 x FCP
  static constexpr int R_TOTAL = R_VAR + 2 * FCP;
  static constexpr int IN_ROWS = INR;                          // d_img rows per view
  // 32x32 blocks of the weight-gradient partials (one workgroup writes NBLK blocks + NBB bias vectors of 64)
  static constexpr int NB_IN = (INR + 31) / 32, NB_F = (FCP + 31) / 32, NB_VM = (2 * FCP + 31) / 32;
  static constexpr int BLK_WC_SH = 0;                          // [tl 2][tb 3]  sum_i D_h x [x, vox|im16]
  static constexpr int BLK_WC_V = BLK_WC_SH + 6;               // [tl 2][NB_IN] sum_i D_h_i x in_i
  static constexpr int BLK_W0 = BLK_WC_V + 2 * NB_IN;          // [tl 2]        D_x x [vox|im16]
  static constexpr int BLK_WS = BLK_W0 + 2;                    // [tb 2]        D_s x x      (row DS_SIGMA = sigma weight)
  static constexpr int BLK_WFC = BLK_WS + 2;                   //               D_fc x im
  static constexpr int BLK_WV = BLK_WFC + 1;                   // [NB_F]        sum_i D_v_i x dir_i
  static constexpr int BLK_WG_V = BLK_WV + NB_F;               // [NB_F]        sum_i D_g_i x f_i
  static constexpr int BLK_WG_SH = BLK_WG_V + NB_F;            // [NB_VM]       sum_i D_g_i x [var|mean]
  static constexpr int NBLK = BLK_WG_SH + NB_VM;
  static constexpr int BB_DH = 0, BB_DX = 2, BB_DS = 4, BB_DFC = 5, BB_DG = 6, BB_DV = 7, NBB = 7 + NB_F;
  static constexpr int PART = NBLK * 1024 + NBB * 64;          // floats per workgroup partial
};

template <int FEAT_CH>
__global__ void nerf_pack_bwd_kernel(bmv_nerf_params p, float* __restrict__ blob) {
  using L = MlpBwdLayout<FEAT_CH>;      // (the transposed tables do not depend on the number of views)
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

// MFMA chain over NT k-steps into TILES accumulator tiles; A of (step t, tile tt) at A[(t * TILES + tt) * 64] (LDS,
// lane-offset pointer).  The A reads of group g + 1 are issued before the MFMAs of group g (see BMV_CHAIN1).
template <int NT, int TILES, int G, class BF>
__device__ __forceinline__ void mfma_chain(const float* __restrict__ A, BF&& b_of, f32x16* acc) {
  float a[2][G][TILES];
#pragma unroll
  for (int u = 0; u < G; ++u)
    if (u < NT)
#pragma unroll
      for (int tt = 0; tt < TILES; ++tt) a[0][u][tt] = A[(u * TILES + tt) * 64];
#pragma unroll
  for (int g = 0; g < (NT + G - 1) / G; ++g) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int tn = (g + 1) * G + u;
      if (tn < NT)
#pragma unroll
        for (int tt = 0; tt < TILES; ++tt) a[(g + 1) & 1][u][tt] = A[(tn * TILES + tt) * 64];
    }
    BMV_FENCE();
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int t = g * G + u;
      if (t < NT) {
        const float b = b_of(t);
#pragma unroll
        for (int tt = 0; tt < TILES; ++tt) acc[tt] = BMV_MFMA(a[g & 1][u][tt], b, acc[tt]);
      }
    }
    BMV_FENCE();
  }
}

template <int FEAT_CH>
struct BwdOut {
  float* rows;   // (ntiles, R_TOTAL, 32)
  float* d_vox;  // (8, P)
  float* d_img;  // (NV, IN_ROWS, P)
  float* vecs;   // per wave of the grid [128]: 64 color.2 weight grad | 32 agg_w_fc weight grad | 32 unused
};

template <int FEAT_CH, int NV>
__global__ void __launch_bounds__(256, 1) nerf_mlp_bwd_kernel(const float* __restrict__ vox_feat,
                                                               const float* __restrict__ img,
                                                               const float* __restrict__ d_out,
                                                               const float* __restrict__ blob_fwd,
                                                               const float* __restrict__ blob_bwd, long npts,
                                                               BwdOut<FEAT_CH> o) {
  using L = MlpLayout<FEAT_CH>;
  using LB = MlpBwdLayout<FEAT_CH, NV>;
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

  // running sums of the two 1-wide weight gradients whose inputs are not stored (this lane's neurons, its samples)
  f32x16 acc_wc2[2], acc_wa;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_wc2[0][r] = acc_wc2[1][r] = acc_wa[r] = 0.f;

  const long ntiles = (npts + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long pt = tile * 32 + s;
    const bool valid = pt < npts;
    const long pc = valid ? pt : npts - 1;
    float* __restrict__ rows = o.rows + tile * (long)(LB::R_TOTAL * 32) + s;   // + row * 32
    // rows past npts hold finite activations of the clamped sample and exactly-zero gradients (go = 0)
    auto put = [&](int row, float v) { rows[row * 32] = v; };
    // ------------------------------------------------------------------ inputs
    float fin[NV][KF], dir[NV][4], vox[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vox[j] = vox_feat[pc * 8 + 2 * j + h];
      put(LB::R_AV24 + 2 * j + h, vox[j]);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float* q = img + (pc * NV + i) * L::IN;
#pragma unroll
      for (int j = 0; j < KFC; ++j) fin[i][j] = (2 * j + h < L::FC) ? q[2 * j + h] : 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) fin[i][KFC + k] = q[L::FC + 2 * k + h];
#pragma unroll
      for (int k = 0; k < 4; ++k) dir[i][k] = q[L::FC + k];
#pragma unroll
      for (int j = 0; j < KF; ++j) put(LB::R_IN + i * LB::INR + 2 * j + h, fin[i][j]);
    }
    float go[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) go[k] = valid ? d_out[pc * 4 + k] : 0.f;
    BMV_FENCE();

    auto pre_v = [&](int i, int j) -> float {
      return Wv[L::V_VF + (j * 5 + 4) * 2] + Wv[L::V_VF + (j * 5 + 0) * 2] * dir[i][0] +
             Wv[L::V_VF + (j * 5 + 1) * 2] * dir[i][1] + Wv[L::V_VF + (j * 5 + 2) * 2] * dir[i][2] +
             Wv[L::V_VF + (j * 5 + 3) * 2] * dir[i][3];
    };
    auto fval = [&](int i, int j) -> float { return fin[i][j] + fmaxf(pre_v(i, j), 0.f); };

    // ------------------------------------------------------------------ aggregation forward (mlp.hpp); run twice:
    // first for im16 (and the stored activations), again right before its own backward
    float mean[KFC];
    f32x16 g[NV];
    float aw[NV], apre[NV];
    auto agg_forward = [&](bool store) {
      float var[KFC];
#pragma unroll
      for (int j = 0; j < KFC; ++j) {
        float f[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) f[i] = fval(i, j);
        float m = f[0];
#pragma unroll
        for (int i = 1; i < NV; ++i) m += f[i];
        m = m / (float)NV;
        float ss = (f[0] - m) * (f[0] - m);
#pragma unroll
        for (int i = 1; i < NV; ++i) ss += (f[i] - m) * (f[i] - m);
        mean[j] = m;
        var[j] = ss * (1.f / (float)(NV - 1));
        if (store) {
#pragma unroll
          for (int i = 0; i < NV; ++i) put(LB::R_F + i * LB::FCP + 2 * j + h, f[i]);
          put(LB::R_VAR + 2 * j + h, var[j]), put(LB::R_VAR + LB::FCP + 2 * j + h, m);
        }
        BMV_FENCE_EVERY(j, 6);
      }
      BMV_FENCE();
      f32x16 gsh;
#pragma unroll
      for (int r = 0; r < 16; ++r) gsh[r] = Wv[L::V_BG + r * 2];
      mfma_chain<2 * KFC, 1, BMV_MLP_G1>(Wa + L::A_GSH, [&](int t) { return t < KFC ? var[t < KFC ? t : 0] : mean[t >= KFC ? t - KFC : 0]; }, &gsh);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        g[i] = gsh;
        mfma_chain<KFC, 1, BMV_MLP_G1>(Wa + L::A_GV, [&](int t) { return fval(i, t); }, &g[i]);
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
      softmax_views<NV>(aw);
    };
    float im16[8];
    {
      agg_forward(true);
      f32x16 im, q16;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        im[r] = weighted_views<NV>(aw, g, r);
        q16[r] = Wv[L::V_BFC + r * 2];
        put(LB::R_AIM + n16(r, h), im[r]);
      }
      mfma_chain<16, 1, BMV_MLP_G1>(Wa + L::A_FC, [&](int t) { return im[t]; }, &q16);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        im16[r] = fmaxf(q16[r], 0.f);
        put(LB::R_AV24 + 8 + n16(r, h), im16[r]);
      }
    }
    BMV_FENCE();
    // ------------------------------------------------------------------ lr0 forward (again before its backward)
    f32x16 x[2];
    auto lr0_forward = [&]() {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[tl][r] = Wv[L::V_B0 + (tl * 16 + r) * 2];
      mfma_chain<12, 2, BMV_MLP_G2>(Wa + L::A_L0, [&](int t) { return t < 4 ? vox[t < 4 ? t : 0] : im16[t >= 4 ? t - 4 : 0]; }, x);
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[tl][r] = fmaxf(x[tl][r], 0.f);
    };
    lr0_forward();
    float spre;
    {
      float sdot = 0.f;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sdot += Wv[L::V_WS + (tl * 16 + r) * 2] * x[tl][r];
          put(LB::R_AX + 32 * tl + n16(r, h), x[tl][r]);
        }
      spre = xhalf_sum(sdot) + bs;
    }
    // ------------------------------------------------------------------ colour forward
    f32x16 csh[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) csh[tl][r] = Wv[L::V_BC + (tl * 16 + r) * 2];
    mfma_chain<44, 2, BMV_MLP_G2>(Wa + L::A_CSH, [&](int t) {
      return t < 32 ? x[t < 32 ? (t >> 4) : 0][t & 15] : t < 36 ? vox[t >= 32 && t < 36 ? t - 32 : 0] : im16[t >= 36 ? t - 36 : 0];
    }, csh);
    BMV_FENCE();
    // colour logits need all three views before the softmax backward: first pass keeps only c_i
    float cl[NV], cpre[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x16 hc[2] = {csh[0], csh[1]};
      mfma_chain<KF, 2, BMV_MLP_G2>(Wa + L::A_CV, [&](int t) { return fin[i][t]; }, hc);
      float sdot = 0.f;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) sdot += Wv[L::V_WC2 + (tl * 16 + r) * 2] * fmaxf(hc[tl][r], 0.f);
      cpre[i] = xhalf_sum(sdot) + bc2;
      cl[i] = fmaxf(cpre[i], 0.f);
      BMV_FENCE();
    }
    softmax_views<NV>(cl);

    // ------------------------------------------------------------------ colour backward
    // rgb = sum_i cw_i rgb_i : the colour channels FEAT_CH+{0,1,2} sit at (half0, slot J), (half1, J), (half0, J+1)
    constexpr int J = FEAT_CH / 2;
    const float g_mine0 = h == 0 ? go[0] : go[1];  // gradient of the channel in slot J of this half
    const float g_mine1 = h == 0 ? go[2] : 0.f;    // slot J+1: blue (half 0) or padding (half 1)
    float d_cpre[NV];
    {
      float d_cw[NV], dotc = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        d_cw[i] = xhalf_sum(g_mine0 * fin[i][J] + g_mine1 * fin[i][J + 1]);
        dotc += cl[i] * d_cw[i];
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        d_cpre[i] = cpre[i] > 0.f ? cl[i] * (d_cw[i] - dotc) : 0.f;
        if (h == 0) put(LB::R_DS + LB::DS_C2 + i, d_cpre[i]);
      }
      if (h == 1) put(LB::R_DS + LB::DS_ROWS - 1, 0.f);
    }
    // per view: recompute hc_i, d_pre_h_i, its rows, the input gradient, and the running sum over views
    f32x16 dhs[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dhs[0][r] = dhs[1][r] = 0.f;
    float d_in[NV][KF];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x16 hc[2] = {csh[0], csh[1]};
      mfma_chain<KF, 2, BMV_MLP_G2>(Wa + L::A_CV, [&](int t) { return fin[i][t]; }, hc);
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
      mfma_chain<32, LB::CV_TILES, BMV_MLP_G2>(Ta + LB::T_CV, [&](int u) { return dh[u >> 4][u & 15]; }, di);
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
    mfma_chain<32, 3, BMV_MLP_G2>(Ta + LB::T_CSH, [&](int u) { return dhs[u >> 4][u & 15]; }, dxv);
    BMV_FENCE();
    // ------------------------------------------------------------------ sigma head and lr0 backward (x recomputed)
    const float d_sp = spre > 20.f ? go[3] : go[3] * (1.f / (1.f + __expf(-spre)));  // softplus' = sigmoid
    if (h == 0) put(LB::R_DS + LB::DS_SIGMA, d_sp);
    f32x16 dv24;  // slots 0..3 vox, 4..11 im16
    {
      lr0_forward();
      f32x16 dx[2];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float d = dxv[tl][r] + d_sp * Wv[L::V_WS + (tl * 16 + r) * 2];
          d = x[tl][r] > 0.f ? d : 0.f;
          dx[tl][r] = d;
          put(LB::R_DX + 32 * tl + n16(r, h), d);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) dv24[r] = 0.f;
      mfma_chain<32, 1, BMV_MLP_G1>(Ta + LB::T_L0, [&](int u) { return dx[u >> 4][u & 15]; }, &dv24);
    }
    BMV_FENCE();
    // vox gradient: slots 32..35 of the colour input + slots 0..3 of lr0
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (valid) o.d_vox[(long)(2 * j + h) * P + pt] = dxv[2][j] + dv24[j];
    // agg.fc
    f32x16 dim;
    {
      float dfc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float d = dxv[2][4 + r] + dv24[4 + r];
        d = im16[r] > 0.f ? d : 0.f;
        dfc[r] = d;
        put(LB::R_DFC + n16(r, h), d);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dim[r] = 0.f;
      mfma_chain<8, 1, BMV_MLP_G1>(Ta + LB::T_FC, [&](int u) { return dfc[u]; }, &dim);
    }
    BMV_FENCE();
    // ------------------------------------------------------------------ aggregation backward (forward recomputed)
    agg_forward(false);
    // softmax over views of the aggregation weights, agg_w_fc, global_fc
    f32x16 dgs;
    f32x16 dg[NV];
    {
      float d_w[NV], dotw = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float sdot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sdot += dim[r] * g[i][r];
        d_w[i] = xhalf_sum(sdot);
        dotw += aw[i] * d_w[i];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dgs[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
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
    }
    BMV_FENCE();
    // d_[var, mean] = Wg[:, F:]^T sum_i d_pre_g_i ; d_f_i = Wg[:, :F]^T d_pre_g_i
    f32x16 dvm[LB::GSH_TILES];
#pragma unroll
    for (int tt = 0; tt < LB::GSH_TILES; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dvm[tt][r] = 0.f;
    mfma_chain<16, LB::GSH_TILES, BMV_MLP_G2>(Ta + LB::T_GSH, [&](int u) { return dgs[u]; }, dvm);
    BMV_FENCE();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x16 dfv[LB::GV_TILES];
#pragma unroll
      for (int tt = 0; tt < LB::GV_TILES; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dfv[tt][r] = 0.f;
      mfma_chain<16, LB::GV_TILES, BMV_MLP_G2>(Ta + LB::T_GV, [&](int u) { return dg[i][u]; }, dfv);
      float d_dir[4] = {0.f, 0.f, 0.f, 0.f};
      float* dimg = o.d_img + (long)i * LB::IN_ROWS * P + pt;
#pragma unroll
      for (int j = 0; j < KFC; ++j) {
        float pv = pre_v(i, j);
        float fj = fin[i][j] + fmaxf(pv, 0.f);
        // var = sum (f - m)^2 / (NV - 1), mean = sum f / NV
        float d_f = dfv[j >> 4][j & 15] + dvm[j >> 4][j & 15] * ((fj - mean[j]) * (2.f / (float)(NV - 1))) +
                    dvm[(KFC + j) >> 4][(KFC + j) & 15] * (1.f / (float)NV);
        if (valid) dimg[(long)(2 * j + h) * P] = d_in[i][j] + d_f;
        float d_pv = pv > 0.f ? d_f : 0.f;
        put(LB::R_DV + i * LB::FCP + 2 * j + h, d_pv);
#pragma unroll
        for (int q = 0; q < 4; ++q) d_dir[q] += d_pv * WB[LB::V_WV + ((j * 4 + q) << 1) + h];
        BMV_FENCE_EVERY(j, 6);
      }
      // the 4 direction inputs: this half's slots from color.0, plus view_fc^T summed over BOTH halves' channels
#pragma unroll
      for (int q = 0; q < 4; ++q) d_dir[q] = xhalf_sum(d_dir[q]);
      if (valid) {
        dimg[(long)(LB::FCP + h) * P] = d_in[i][KFC] + (h ? d_dir[1] : d_dir[0]);
        dimg[(long)(LB::FCP + 2 + h) * P] = d_in[i][KFC + 1] + (h ? d_dir[3] : d_dir[2]);
      }
      BMV_FENCE();
    }
  }
  // ---- flush the 1-wide weight gradients: reduce over the 32 samples of each half into this wave's partial
  float* __restrict__ vpart = o.vecs + ((long)blockIdx.x * 4 + wave) * 128;
  auto reduce32 = [&](float v) {
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
  };
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = reduce32(acc_wc2[tl][r]);
      if (s == 0) vpart[32 * tl + n16(r, h)] = a;
    }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float a = reduce32(acc_wa[r]);
    if (s == 0) vpart[64 + n16(r, h)] = a;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients: dW block (32 x 32) += A (32 rows x 64 samples) B^T (32 rows x 64 samples), samples = MFMA k.
// ---------------------------------------------------------------------------------------------------------------
template <int FEAT_CH, int NV>
__global__ void __launch_bounds__(256, 1) nerf_wgrad_kernel(const float* __restrict__ rows, long ntiles,
                                                             float* __restrict__ partials) {
  using LB = MlpBwdLayout<FEAT_CH, NV>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 31, kk = lane >> 5;
  const long npairs = (ntiles + 1) / 2;
  float* __restrict__ part = partials + (long)blockIdx.x * LB::PART;

  // operand: row (row_base + m) of tile, 32 consecutive samples (128 B per lane); rows past nrows / tiles past the end: 0
  auto load = [&](long tile, int row_base, int nrows, float (&v)[32]) {
    if (m < nrows && tile < ntiles) {
      const float4* p = reinterpret_cast<const float4*>(rows + (tile * LB::R_TOTAL + row_base + m) * 32);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float4 t = p[q];
        v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 32; ++q) v[q] = 0.f;
    }
  };
  auto mm = [&](const float (&a)[32], const float (&b)[32], f32x16& acc) {
#pragma unroll
    for (int j = 0; j < 32; ++j) acc = BMV_MFMA(a[j], b[j], acc);
  };
  auto rowsum = [&](const float (&a)[32]) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) t += a[j];
    return t;
  };
  auto zero = [&](f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
  };
  // accumulator register r, lane (n, hh) holds D[m = n16(r, hh)][n]
  auto store_block = [&](int blk, const f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[blk * 1024 + n16(r, kk) * 32 + m] = acc[r];
  };
  auto store_bias = [&](int bb, float v) { part[LB::NBLK * 1024 + bb * 64 + lane] = v; };

  if (wave < 2) {
    // colour layer, output neurons 32 * wave ..: sum_i D_h_i x in_i, then (sum_i D_h_i) x [x | vox, im16]
    const int tl = wave;
    f32x16 acc_sh[3], acc_v[LB::NB_IN];
    float bias = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) zero(acc_sh[q]);
#pragma unroll
    for (int q = 0; q < LB::NB_IN; ++q) zero(acc_v[q]);
    for (long pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
      const long tile = 2 * pair + kk;
      float asum[32], a[32], b[32];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        load(tile, LB::R_DH + i * 64 + tl * 32, 32, a);
#pragma unroll
        for (int bb = 0; bb < LB::NB_IN; ++bb) {
          load(tile, LB::R_IN + i * LB::INR + bb * 32, LB::INR - bb * 32, b);
          mm(a, b, acc_v[bb]);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) asum[j] = i == 0 ? a[j] : asum[j] + a[j];
      }
      bias += rowsum(asum);
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        load(tile, tb < 2 ? LB::R_AX + tb * 32 : LB::R_AV24, tb < 2 ? 32 : 24, b);
        mm(asum, b, acc_sh[tb]);
      }
    }
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) store_block(LB::BLK_WC_SH + tl * 3 + tb, acc_sh[tb]);
#pragma unroll
    for (int bb = 0; bb < LB::NB_IN; ++bb) store_block(LB::BLK_WC_V + tl * LB::NB_IN + bb, acc_v[bb]);
    store_bias(LB::BB_DH + tl, bias);
  } else if (wave == 2) {
    // lr0, sigma head, agg.fc, view_fc
    f32x16 acc_w0[2], acc_ws[2], acc_fc, acc_wv[LB::NB_F];
    float bias_dx[2] = {0.f, 0.f}, bias_ds = 0.f, bias_fc = 0.f, bias_dv[LB::NB_F];
    zero(acc_w0[0]), zero(acc_w0[1]), zero(acc_ws[0]), zero(acc_ws[1]), zero(acc_fc);
#pragma unroll
    for (int q = 0; q < LB::NB_F; ++q) zero(acc_wv[q]), bias_dv[q] = 0.f;
    for (long pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
      const long tile = 2 * pair + kk;
      float a[32], b[32];
      load(tile, LB::R_AV24, 24, b);
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        load(tile, LB::R_DX + tl * 32, 32, a);
        mm(a, b, acc_w0[tl]);
        bias_dx[tl] += rowsum(a);
      }
      load(tile, LB::R_DS, LB::DS_ROWS, a);
      bias_ds += rowsum(a);
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        load(tile, LB::R_AX + tb * 32, 32, b);
        mm(a, b, acc_ws[tb]);
      }
      load(tile, LB::R_DFC, 16, a);
      load(tile, LB::R_AIM, 32, b);
      mm(a, b, acc_fc);
      bias_fc += rowsum(a);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        load(tile, LB::R_IN + i * LB::INR + LB::FCP, 4, b);
#pragma unroll
        for (int ab = 0; ab < LB::NB_F; ++ab) {
          load(tile, LB::R_DV + i * LB::FCP + ab * 32, LB::FCP - ab * 32, a);
          mm(a, b, acc_wv[ab]);
          bias_dv[ab] += rowsum(a);
        }
      }
    }
    store_block(LB::BLK_W0, acc_w0[0]), store_block(LB::BLK_W0 + 1, acc_w0[1]);
    store_block(LB::BLK_WS, acc_ws[0]), store_block(LB::BLK_WS + 1, acc_ws[1]);
    store_block(LB::BLK_WFC, acc_fc);
    store_bias(LB::BB_DX, bias_dx[0]), store_bias(LB::BB_DX + 1, bias_dx[1]);
    store_bias(LB::BB_DS, bias_ds), store_bias(LB::BB_DFC, bias_fc);
#pragma unroll
    for (int ab = 0; ab < LB::NB_F; ++ab) store_block(LB::BLK_WV + ab, acc_wv[ab]), store_bias(LB::BB_DV + ab, bias_dv[ab]);
  } else {
    // global_fc: sum_i D_g_i x f_i, then (sum_i D_g_i) x [var | mean]
    f32x16 acc_gv[LB::NB_F], acc_gs[LB::NB_VM];
    float bias = 0.f;
#pragma unroll
    for (int q = 0; q < LB::NB_F; ++q) zero(acc_gv[q]);
#pragma unroll
    for (int q = 0; q < LB::NB_VM; ++q) zero(acc_gs[q]);
    for (long pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
      const long tile = 2 * pair + kk;
      float asum[32], a[32], b[32];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        load(tile, LB::R_DG + i * 32, 32, a);
#pragma unroll
        for (int bb = 0; bb < LB::NB_F; ++bb) {
          load(tile, LB::R_F + i * LB::FCP + bb * 32, LB::FCP - bb * 32, b);
          mm(a, b, acc_gv[bb]);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) asum[j] = i == 0 ? a[j] : asum[j] + a[j];
      }
      bias += rowsum(asum);
#pragma unroll
      for (int bb = 0; bb < LB::NB_VM; ++bb) {
        load(tile, LB::R_VAR + bb * 32, 2 * LB::FCP - bb * 32, b);
        mm(asum, b, acc_gs[bb]);
      }
    }
#pragma unroll
    for (int bb = 0; bb < LB::NB_F; ++bb) store_block(LB::BLK_WG_V + bb, acc_gv[bb]);
#pragma unroll
    for (int bb = 0; bb < LB::NB_VM; ++bb) store_block(LB::BLK_WG_SH + bb, acc_gs[bb]);
    store_bias(LB::BB_DG, bias);
  }
}

// Eight consecutive lanes per parameter-gradient element: where it sits in a partial (up to 3 places for the summed
// 1-wide biases), the workgroups' partials summed in a fixed order (lane q takes partials q, q + 8, ...; then a
// 3-step butterfly), so the result does not depend on scheduling.
template <int FEAT_CH, int NV>
__global__ void nerf_wgrad_finish_kernel(const float* __restrict__ partials, int nparts, const float* __restrict__ vecs,
                                         int nvecs, bmv_nerf_grads g) {
  using LB = MlpBwdLayout<FEAT_CH, NV>;
  constexpr int FC = LB::FC, FCP = LB::FCP, CW = 88 + FC + 4;
  constexpr int N_VW = FC * 4, N_VB = FC, N_GW = 32 * 3 * FC, N_GB = 32, N_AW = 32, N_AB = 1, N_FW = 16 * 32, N_FB = 16,
                N_0W = 64 * 24, N_0B = 64, N_SW = 64, N_SB = 1, N_CW = 64 * CW, N_CB = 64, N_2W = 64, N_2B = 1;
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, sub = threadIdx.x & 7;
  auto blk = [](int b, int mm, int nn) { return b * 1024 + mm * 32 + nn; };
  auto bias = [](int bb, int mm) { return LB::NBLK * 1024 + bb * 64 + mm; };   // + 32 for the other sample half
  float* dst = nullptr;
  int src[4] = {-1, -1, -1, -1};
  bool is_bias = false;
  int vec = -1;
  int e = idx;
  if (e < N_VW) {
    int c = e / 4, q = e % 4;
    dst = g.view_fc_w + e, src[0] = blk(LB::BLK_WV + c / 32, c % 32, q);
  } else if ((e -= N_VW) < N_VB) {
    dst = g.view_fc_b + e, src[0] = bias(LB::BB_DV + e / 32, e % 32), is_bias = true;
  } else if ((e -= N_VB) < N_GW) {
    int n = e / (3 * FC), k = e % (3 * FC);
    dst = g.global_fc_w + e;
    if (k < FC) {
      src[0] = blk(LB::BLK_WG_V + k / 32, n, k % 32);
    } else {
      int row = k < 2 * FC ? k - FC : FCP + (k - 2 * FC);
      src[0] = blk(LB::BLK_WG_SH + row / 32, n, row % 32);
    }
  } else if ((e -= N_GW) < N_GB) {
    dst = g.global_fc_b + e, src[0] = bias(LB::BB_DG, e), is_bias = true;
  } else if ((e -= N_GB) < N_AW) {
    dst = g.agg_w_w + e, vec = 64 + e;
  } else if ((e -= N_AW) < N_AB) {
    dst = g.agg_w_b, is_bias = true;
#pragma unroll
    for (int i = 0; i < NV; ++i) src[i] = bias(LB::BB_DS, i);
  } else if ((e -= N_AB) < N_FW) {
    dst = g.fc_w + e, src[0] = blk(LB::BLK_WFC, e / 32, e % 32);
  } else if ((e -= N_FW) < N_FB) {
    dst = g.fc_b + e, src[0] = bias(LB::BB_DFC, e), is_bias = true;
  } else if ((e -= N_FB) < N_0W) {
    int n = e / 24, k = e % 24;
    dst = g.lr0_w + e, src[0] = blk(LB::BLK_W0 + n / 32, n % 32, k);
  } else if ((e -= N_0W) < N_0B) {
    dst = g.lr0_b + e, src[0] = bias(LB::BB_DX + e / 32, e % 32), is_bias = true;
  } else if ((e -= N_0B) < N_SW) {
    dst = g.sigma_w + e, src[0] = blk(LB::BLK_WS + e / 32, LB::DS_SIGMA, e % 32);
  } else if ((e -= N_SW) < N_SB) {
    dst = g.sigma_b, src[0] = bias(LB::BB_DS, LB::DS_SIGMA), is_bias = true;
  } else if ((e -= N_SB) < N_CW) {
    int n = e / CW, k = e % CW, tl = n / 32;
    dst = g.color0_w + e;
    if (k < 88) {
      src[0] = blk(LB::BLK_WC_SH + tl * 3 + (k < 64 ? k / 32 : 2), n % 32, k < 64 ? k % 32 : k - 64);
    } else {
      int c = k - 88, row = c < FC ? c : FCP + (c - FC);
      src[0] = blk(LB::BLK_WC_V + tl * LB::NB_IN + row / 32, n % 32, row % 32);
    }
  } else if ((e -= N_CW) < N_CB) {
    dst = g.color0_b + e, src[0] = bias(LB::BB_DH + e / 32, e % 32), is_bias = true;
  } else if ((e -= N_CB) < N_2W) {
    dst = g.color2_w + e, vec = e;
  } else if ((e -= N_2W) < N_2B) {
    dst = g.color2_b, is_bias = true;
#pragma unroll
    for (int i = 0; i < NV; ++i) src[i] = bias(LB::BB_DS, LB::DS_C2 + i);
  } else {
    return;
  }
  float acc = 0.f;
  if (vec >= 0) {
    for (int w = sub; w < nvecs; w += 8) acc += vecs[(long)w * 128 + vec];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
  } else {
#pragma unroll 4
    for (int w = sub; w < nparts; w += 8) {
      const float* p = partials + (long)w * LB::PART;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (src[q] >= 0) acc += p[src[q]] + (is_bias ? p[src[q] + 32] : 0.f);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
  }
  if (sub == 0)   *dst = acc;
}

}  // namespace bmv

using namespace bmv;

namespace {
constexpr int kWgradGrid = 256;   // one workgroup per CU (the kernel takes > 256 registers)
constexpr int kBwdGrid = 256;     // workgroups of the data-path kernel (4 waves each write a 128-float head partial)

template <int FC, int NV>
long workspace_floats(long npts) {
  using LB = MlpBwdLayout<FC, NV>;
  long ntiles = (npts + 31) / 32;
  return ntiles * (long)LB::R_TOTAL * 32 + (long)kWgradGrid * LB::PART + (long)kBwdGrid * 4 * 128;
}

template <int FC, int NV>
int run_bwd(const float* vox_feat, const float* img, const float* d_out, const float* blob_fwd, const float* blob_bwd,
            long npts, float* ws, float* d_vox, float* d_img, const bmv_nerf_grads* grads, hipStream_t st) {
  using LB = MlpBwdLayout<FC, NV>;
  const long ntiles = (npts + 31) / 32;
  float* rows = ws;
  float* partials = rows + ntiles * (long)LB::R_TOTAL * 32;
  float* vecs = partials + (long)kWgradGrid * LB::PART;
  // (round 3 cleared a 128-float atomics buffer here with hipMemsetAsync: inside a captured training step the memset NODE
  // did not reliably clear it on later replays; since round 5 every wave writes its own partial, nothing to clear)
  const unsigned grid = (unsigned)((ntiles + 3) / 4 < kBwdGrid ? (ntiles + 3) / 4 : kBwdGrid);
  const size_t lds = (size_t)(MlpLayout<FC>::TOTAL + LB::TOTAL) * 4;
  BMV_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_mlp_bwd_kernel<FC, NV>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
              "bmv_nerf_mlp_bwd: cannot reserve %zu B of LDS", lds);
  BwdOut<FC> o{rows, d_vox, d_img, vecs};
  hipLaunchKernelGGL((nerf_mlp_bwd_kernel<FC, NV>), dim3(grid), dim3(256), lds, st, vox_feat, img, d_out, blob_fwd, blob_bwd,
                     npts, o);
  const long npairs = (ntiles + 1) / 2;
  const int nparts = (int)(npairs < kWgradGrid ? npairs : kWgradGrid);
  hipLaunchKernelGGL((nerf_wgrad_kernel<FC, NV>), dim3(nparts), dim3(256), 0, st, rows, ntiles, partials);
  constexpr int F = FC + 3;
  constexpr int total = F * 4 + F + 32 * 3 * F + 32 + 32 + 1 + 16 * 32 + 16 + 64 * 24 + 64 + 64 + 1 + 64 * (88 + F + 4) + 64 + 64 + 1;
  hipLaunchKernelGGL((nerf_wgrad_finish_kernel<FC, NV>), dim3(cdiv((long)total * 8, 256)), dim3(256), 0, st, partials, nparts,
                     vecs, (int)grid * 4, *grads);
  BMV_LAUNCH_END("bmv_nerf_mlp_bwd");
}

// (feat_ch, S) -> the instantiation: feat_ch in {8, 32} x S in {2, 3, 4} source views
#define BMV_MLP_BWD_DISPATCH(EXPR)                  \
  if (feat_ch == 8 && S == 3) return EXPR(8, 3);    \
  if (feat_ch == 32 && S == 3) return EXPR(32, 3);  \
  if (feat_ch == 8 && S == 2) return EXPR(8, 2);    \
  if (feat_ch == 32 && S == 2) return EXPR(32, 2);  \
  if (feat_ch == 8 && S == 4) return EXPR(8, 4);    \
  if (feat_ch == 32 && S == 4) return EXPR(32, 4);
}  // namespace

extern "C" {

int bmv_nerf_bwd_blob_size(int feat_ch) {
  if (feat_ch == 8) return MlpBwdLayout<8>::TOTAL;
  if (feat_ch == 32) return MlpBwdLayout<32>::TOTAL;
  set_error("bmv_nerf_bwd_blob_size: feat_ch=%d unsupported (8 or 32)", feat_ch);
  return BMV_ERR_UNSUPPORTED;
}

int bmv_nerf_bwd_rows(int feat_ch, int S, int* d_img_rows) {
#define ROWS(FC, NVV) ((d_img_rows ? (void)(*d_img_rows = MlpBwdLayout<FC, NVV>::IN_ROWS) : (void)0), MlpBwdLayout<FC, NVV>::R_TOTAL)
  BMV_MLP_BWD_DISPATCH(ROWS)
#undef ROWS
  set_error("bmv_nerf_bwd_rows: feat_ch=%d (8 or 32) with S=%d source views (2, 3 or 4) unsupported", feat_ch, S);
  return BMV_ERR_UNSUPPORTED;
}

long bmv_nerf_bwd_workspace(int feat_ch, int S, long npts) {
  if (npts < 0) {
    set_error("bmv_nerf_bwd_workspace: npts=%ld", npts);
    return BMV_ERR_INVALID;
  }
#define WSF(FC, NVV) workspace_floats<FC, NVV>(npts)
  BMV_MLP_BWD_DISPATCH(WSF)
#undef WSF
  set_error("bmv_nerf_bwd_workspace: feat_ch=%d (8 or 32) with S=%d source views (2, 3 or 4) unsupported", feat_ch, S);
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
                     const float* blob_bwd, int feat_ch, int S, long npts, float* workspace, float* d_vox, float* d_img,
                     const bmv_nerf_grads* grads, bmv_stream_t stream) {
  BMV_REQUIRE(vox_feat && img && d_out && blob_fwd && blob_bwd && workspace && d_vox && d_img && grads,
              "bmv_nerf_mlp_bwd: null pointer");
  BMV_REQUIRE(grads->view_fc_w && grads->view_fc_b && grads->global_fc_w && grads->global_fc_b && grads->agg_w_w &&
                  grads->agg_w_b && grads->fc_w && grads->fc_b && grads->lr0_w && grads->lr0_b && grads->sigma_w &&
                  grads->sigma_b && grads->color0_w && grads->color0_b && grads->color2_w && grads->color2_b,
              "bmv_nerf_mlp_bwd: null gradient pointer");
  BMV_REQUIRE(npts > 0, "bmv_nerf_mlp_bwd: npts=%ld", npts);
#define RUN(FC, NVV) \
  run_bwd<FC, NVV>(vox_feat, img, d_out, blob_fwd, blob_bwd, npts, workspace, d_vox, d_img, grads, as_stream(stream))
  BMV_MLP_BWD_DISPATCH(RUN)
#undef RUN
  set_error("bmv_nerf_mlp_bwd: feat_ch=%d (8 or 32) with S=%d source views (2, 3 or 4) unsupported", feat_ch, S);
  return BMV_ERR_UNSUPPORTED;
}

}  // extern "C"
