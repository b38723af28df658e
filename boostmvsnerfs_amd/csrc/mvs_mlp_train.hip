// Training forward + backward of MVSNeRF's Renderer_ours (a25: 6 x 128 MLP with the pts_bias gate, skip concat after
// layer 4, alpha / feature / views / rgb heads; lib/networks/mvsnerf/network.py:153-229) on the fp32 matrix cores.
//
// Inference runs the fused kernel of mvs.hip (weights streamed through LDS, activations never leave registers).  Under
// autograd every layer's input and pre-activation are needed again, so the training path keeps them: activations live
// in HBM as per-tile ROW MATRICES act[tile][row][32 samples] (a row = one neuron / input channel of 32 consecutive
// points = 128 contiguous bytes), and three small generic kernels do all the work:
//
//   rows_gemm_kernel   Out[m][n] = sum_k Table[k][col0 + m] * In[k][n] over up to two input row segments, m in blocks of
//                      32 rows, n = the 32 samples of a tile, on v_mfma_f32_32x32x2: the B operand of k-step u is rows
//                      2u, 2u+1 of the tile (ONE 256-byte coalesced load per wave), the A operand 64 consecutive floats
//                      of the k-major table.  The forward uses the transposed weights (packed once per step), the
//                      data gradient dIn = W^T dOut uses the parameter tensor AS STORED (its major index is the
//                      reduction index).  Epilogues work in the accumulator layout (a register = two 128-byte row
//                      pieces): bias, the pts_bias gate + ReLU (stores z and h), and for the backward the ReLU mask,
//                      the gate, and the running d(pts_bias) sum.
//   mvs_heads_*        the 1- and 3-wide heads (alpha, rgb): one lane per point, dot products over rows.
//   rows_wgrad_kernel  dW[m][k] = sum_n dOut[m][n] In[k][n] with the SAMPLE index as the MFMA k dimension (the scheme of
//                      mlp_bwd.hip): a wave owns a strip of 32 output rows x up to 6 blocks of 32 input rows of one
//                      layer for a slice of the tiles; partial strips are summed in a fixed order by
//                      rows_wgrad_finish_kernel straight into tensors of the parameters' shapes (deterministic, no
//                      atomics).  Bias gradients are row sums of the same operands.
#include <initializer_list>

#include "mlp.hpp"

namespace bmv {

// ---- row map of one tile (rows of 32 floats) -------------------------------------------------------------------------
struct TR {
  static constexpr int PTS = 0;                  // embedded point          63 (+1 zero)
  static constexpr int FEAT = 64;                // 20-ch feature           20 (+12 zero)
  static constexpr int VIEW = 96;                // view direction           3 (+1 zero)
  static constexpr int BIAS = 100;               // pts_bias(feat)         128
  static constexpr int Z0 = 228;                 // layer l: z_l at Z0 + 256 l (dz_l after the backward), h_l 128 behind
  __host__ __device__ static constexpr int Z(int l) { return Z0 + 256 * l; }
  __host__ __device__ static constexpr int H(int l) { return Z0 + 256 * l + 128; }
  static constexpr int FEATL = 1764;             // feature_linear(h_5)    128
  static constexpr int HV = 1892;                // relu(views layer)       64
  static constexpr int DHV = 1956;               // d pre(views layer)      64
  static constexpr int DFEATL = 2020;            // d feature_linear out   128
  static constexpr int TMP = 2148;               // alpha head's share of d h_5
  static constexpr int DBIAS = 2276;             // d pts_bias output      128
  static constexpr int GOUT = 2404;              // d pre of the heads: alpha, r, g, b
  static constexpr int DPTS = 2408;              // d embedded point        64
  static constexpr int DFEAT = 2472;             // d feature               32
  static constexpr int DVIEW = 2504;             // d view direction         4
  static constexpr int TOTAL = 2508;
};

// k-major transposed weights for the forward: [k][M] per layer, k padded to the row segments
struct TW {
  static constexpr int BIAS = 0;                           // [20][128]
  static constexpr int L0 = BIAS + 20 * 128;               // [64][128]   (row 63 = 0)
  __host__ __device__ static constexpr int L(int l) { return L0 + 64 * 128 + (l - 1) * 128 * 128; }   // l = 1..4: [128][128]
  static constexpr int L5 = L0 + 64 * 128 + 4 * 128 * 128; // [64 + 128][128] (row 63 = 0)
  static constexpr int FEATL = L5 + 192 * 128;             // [128][128]
  static constexpr int VIEWS = FEATL + 128 * 128;          // [128 + 4][64]  (row 131 = 0)
  static constexpr int TOTAL = VIEWS + 132 * 64;
};

__global__ void mvs_train_pack_kernel(bmv_mvs_mlp_params p, float* __restrict__ wt) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= TW::TOTAL) return;
  float v = 0.f;
  if (idx < TW::L0) {
    const int k = idx / 128, m = idx % 128;
    v = p.bias_w[m * 20 + k];
  } else if (idx < TW::L(1)) {
    const int e = idx - TW::L0, k = e / 128, m = e % 128;
    if (k < 63) v = p.pts_w[0][m * 63 + k];
  } else if (idx < TW::L5) {
    const int e = idx - TW::L(1), l = 1 + e / (128 * 128), k = (e / 128) % 128, m = e % 128;
    v = p.pts_w[l][m * 128 + k];
  } else if (idx < TW::FEATL) {
    const int e = idx - TW::L5, k = e / 128, m = e % 128;     // torch.cat([input_pts, h]): columns 0..62 pts, 63..190 h
    if (k < 63) v = p.pts_w[5][m * 191 + k];
    else if (k >= 64) v = p.pts_w[5][m * 191 + 63 + (k - 64)];
  } else if (idx < TW::VIEWS) {
    const int e = idx - TW::FEATL, k = e / 128, m = e % 128;
    v = p.feature_w[m * 128 + k];
  } else {
    const int e = idx - TW::VIEWS, k = e / 64, m = e % 64;    // torch.cat([feature, input_views])
    if (k < 131) v = p.views_w[m * 131 + k];
  }
  wt[idx] = v;
}

// x (N, 86) -> rows PTS / FEAT / VIEW of every tile (padding rows and samples past N: 0); and back for the gradient
__global__ void mvs_train_scatter_kernel(const float* __restrict__ x, long N, float* __restrict__ act) {
  const long tile = blockIdx.x;
  for (int e = threadIdx.x; e < 100 * 32; e += blockDim.x) {
    const int row = e >> 5, n = e & 31;
    const long s = tile * 32 + n;
    int col = -1;
    if (row < 63) col = row;
    else if (row >= TR::FEAT && row < TR::FEAT + 20) col = 63 + row - TR::FEAT;
    else if (row >= TR::VIEW && row < TR::VIEW + 3) col = 83 + row - TR::VIEW;
    act[(tile * TR::TOTAL + row) * 32 + n] = (col >= 0 && s < N) ? x[s * 86 + col] : 0.f;
  }
}
__global__ void mvs_train_gather_dx_kernel(const float* __restrict__ act, long N, float* __restrict__ dx) {
  const long tile = blockIdx.x;
  for (int e = threadIdx.x; e < 86 * 32; e += blockDim.x) {
    const int col = e % 86, n = e / 86;
    const long s = tile * 32 + n;
    if (s >= N) continue;
    const int row = col < 63 ? TR::DPTS + col : col < 83 ? TR::DFEAT + col - 63 : TR::DVIEW + col - 83;
    dx[s * 86 + col] = act[(tile * TR::TOTAL + row) * 32 + n];
  }
}

// ---- the generic rows GEMM ---------------------------------------------------------------------------------------------
enum : int { EP_STORE = 0, EP_GATE_RELU = 1, EP_RELU = 2, EP_BWD_HID = 3 };
struct RowsGemm {
  float* act;
  const float* table;      // k-major: table[k * ld + col0 + m]
  const float* bias;       // [M] or null
  long ntiles;
  int ld, col0, M;         // rows m >= M are neither computed from the table (0) nor stored
  int seg_row[2], seg_k[2];
  int out_row;
  int aux0, aux1, aux2;    // EP_GATE_RELU: aux0 = BIAS rows, aux1 = H rows (out_row = Z rows)
                           // EP_BWD_HID:   aux0 = BIAS rows, aux1 = DBIAS rows, aux2 = rows added to the product or -1
                           //               (out_row = Z rows: z is read, dz written)
  int accumulate;          // EP_STORE: add to what is there; EP_BWD_HID: DBIAS += instead of =
};

template <int MB, int EP>
__global__ void __launch_bounds__(256, 2) rows_gemm_kernel(RowsGemm a) {
  // (wave index through readfirstlane: tile pointers are then scalar, loads take SGPR base + 32-bit lane offset)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, kh = lane >> 5;
  const long tile0 = ((long)blockIdx.x * 4 + wave) * 2;
  if (tile0 >= a.ntiles) return;
  const bool two = tile0 + 1 < a.ntiles;
  const int mbase = blockIdx.y * MB * 32;
  f32x16 acc[2][MB];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][mb][r] = 0.f;
  float* __restrict__ t0 = a.act + tile0 * TR::TOTAL * 32;
  float* __restrict__ t1 = two ? t0 + TR::TOTAL * 32 : t0;
  // table column of this lane per 32-row block: clamped (no branch in the loop), masked to 0 past M
  int colc[MB];
  bool live[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mbase + mb * 32 + n;
    live[mb] = m < a.M;
    colc[mb] = a.col0 + (live[mb] ? m : a.M - 1);
  }
  // k loop: operands of step u + PF are requested before the MFMAs of step u are issued (PF x 6 loads in flight)
  constexpr int PF = 4;
  int krow = 0;
  for (int seg = 0; seg < 2; ++seg) {
    const int steps = a.seg_k[seg] >> 1;
    if (steps == 0) continue;
    const float* __restrict__ in0 = t0 + a.seg_row[seg] * 32 + (kh * 32 + n);
    const float* __restrict__ in1 = t1 + a.seg_row[seg] * 32 + (kh * 32 + n);
    const float* __restrict__ tab = a.table + (long)krow * a.ld + kh * a.ld;
    const int ld2 = 2 * a.ld;
    float q0[PF], q1[PF], qw[PF][MB];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int uc = j < steps ? j : steps - 1;
      q0[j] = in0[uc * 64], q1[j] = in1[uc * 64];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) qw[j][mb] = tab[uc * ld2 + colc[mb]];
    }
    for (int u0 = 0; u0 < steps; u0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        if (u0 + j < steps) {
          const float b0 = q0[j], b1 = q1[j];
          float w[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) w[mb] = live[mb] ? qw[j][mb] : 0.f;
          const int un = u0 + j + PF, uc = un < steps ? un : steps - 1;
          q0[j] = in0[uc * 64], q1[j] = in1[uc * 64];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) qw[j][mb] = tab[uc * ld2 + colc[mb]];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            acc[0][mb] = BMV_MFMA(w[mb], b0, acc[0][mb]);
            acc[1][mb] = BMV_MFMA(w[mb], b1, acc[1][mb]);
          }
        }
      }
    }
    krow += a.seg_k[seg];
  }
  // epilogue: register r of lane (n, kh) is element (row n16(r, kh), sample n) of its 32-row block.  All loads of a
  // block are issued before its stores (the rows alias through `act` as far as the compiler knows).
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t == 1 && !two) break;
    float* __restrict__ T = (t ? t1 : t0) + n;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m0 = mbase + mb * 32 + 4 * kh;            // row of register r: m0 + (r & 3) + 8 (r >> 2)
      if (mbase + mb * 32 >= a.M) break;                  // (uniform)
      float v[16], x0[16], x1[16], x2[16];
      bool ok[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2);
        ok[r] = m < a.M;
        const int mc = ok[r] ? m : a.M - 1;
        v[r] = acc[t][mb][r];
        if constexpr (EP == EP_STORE) {
          x0[r] = a.bias ? a.bias[mc] : 0.f;
          x1[r] = a.accumulate ? T[(a.out_row + mc) * 32] : 0.f;
        } else if constexpr (EP == EP_GATE_RELU) {
          x0[r] = a.bias[mc];
          x1[r] = T[(a.aux0 + mc) * 32];
        } else if constexpr (EP == EP_RELU) {
          x0[r] = a.bias[mc];
        } else {
          x0[r] = T[(a.out_row + mc) * 32];                                   // z
          x1[r] = T[(a.aux0 + mc) * 32];                                      // gate
          x2[r] = a.aux2 >= 0 ? T[(a.aux2 + mc) * 32] : 0.f;
          v[r] += x2[r];
          x2[r] = a.accumulate ? T[(a.aux1 + mc) * 32] : 0.f;                 // running d gate
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!ok[r]) continue;
        const int m = m0 + (r & 3) + 8 * (r >> 2);
        if constexpr (EP == EP_STORE) {
          T[(a.out_row + m) * 32] = v[r] + x0[r] + x1[r];
        } else if constexpr (EP == EP_GATE_RELU) {
          const float z = v[r] + x0[r];
          T[(a.out_row + m) * 32] = z;                                        // z_l
          T[(a.aux1 + m) * 32] = fmaxf(z * x1[r], 0.f);                       // h_l = relu(z_l * gate)
        } else if constexpr (EP == EP_RELU) {
          T[(a.out_row + m) * 32] = fmaxf(v[r] + x0[r], 0.f);
        } else {
          const float dpre = x0[r] * x1[r] > 0.f ? v[r] : 0.f;                // d relu(z * gate)
          T[(a.aux1 + m) * 32] = x2[r] + dpre * x0[r];                        // d gate
          T[(a.out_row + m) * 32] = dpre * x1[r];                             // d z
        }
      }
      asm volatile("" ::: "memory");      // the next block's loads stay behind these stores (register pressure)
    }
  }
}

// ---- heads -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mvs_heads_fwd_kernel(const float* __restrict__ act, bmv_mvs_mlp_params p, long N,
                                                            float* __restrict__ out) {
  const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const float* __restrict__ T = act + (s >> 5) * TR::TOTAL * 32 + (s & 31);
  float al = p.alpha_b[0];
  for (int k = 0; k < 128; ++k) al += p.alpha_w[k] * T[(TR::H(5) + k) * 32];
  float c[3] = {p.rgb_b[0], p.rgb_b[1], p.rgb_b[2]};
  for (int k = 0; k < 64; ++k) {
    const float h = T[(TR::HV + k) * 32];
#pragma unroll
    for (int j = 0; j < 3; ++j) c[j] += p.rgb_w[j * 64 + k] * h;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) out[s * 4 + j] = 1.f / (1.f + __expf(-c[j]));
  out[s * 4 + 3] = fmaxf(al, 0.f);
}

__global__ void __launch_bounds__(256) mvs_heads_bwd_kernel(float* __restrict__ act, bmv_mvs_mlp_params p, long N,
                                                            const float* __restrict__ out, const float* __restrict__ d_out) {
  const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ((N + 31) / 32) * 32) return;
  float* __restrict__ T = act + (s >> 5) * TR::TOTAL * 32 + (s & 31);
  float g[4] = {0.f, 0.f, 0.f, 0.f};                       // d pre: alpha, r, g, b
  if (s < N) {
    g[0] = out[s * 4 + 3] > 0.f ? d_out[s * 4 + 3] : 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float y = out[s * 4 + j];
      g[1 + j] = d_out[s * 4 + j] * y * (1.f - y);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) T[(TR::GOUT + j) * 32] = g[j];
  for (int k = 0; k < 128; ++k) T[(TR::TMP + k) * 32] = p.alpha_w[k] * g[0];
  for (int k = 0; k < 64; ++k) {
    const float v = p.rgb_w[k] * g[1] + p.rgb_w[64 + k] * g[2] + p.rgb_w[128 + k] * g[3];
    T[(TR::DHV + k) * 32] = T[(TR::HV + k) * 32] > 0.f ? v : 0.f;
  }
}

// ---- weight gradients --------------------------------------------------------------------------------------------------
constexpr int kMaxKB = 6, kMaxStrips = 40;
struct Strip {
  short a_row, a_rows;            // 32-row block of the output-side gradient: first row, valid rows
  short nkb;                      // input-side blocks of 32 rows
  short b_row[kMaxKB], b_rows[kMaxKB];
  // where the strip goes (finish kernel): out[(i) * ld + col[kb] + j] = D[sel0 + i][j], i < out_rows, j < b_rows[kb]
  short sel0, out_rows, out_row0, ld;
  short col[kMaxKB];
  short param, bias_param;        // indices into the gradient pointer table (bias_param < 0: none)
};
constexpr int kSplitPerKb = 16;   // workgroups per 32 x 32 block column of a layer: ~2500 waves = 2.5 per SIMD
constexpr int kMaxGroups = 12;
struct WgradArgsMvs {
  const float* act;
  long ntiles;
  int nstrips, ngroups, nwgs;
  float* partials;                // [workgroup][wave][kMaxKB * 1024 + 64]
  // a GROUP = the (up to 4) strips of one layer: same input-side blocks, different 32-row blocks of the output side.
  // A workgroup is (group, split), its wave j is strip first_strip + j: the four waves read the same input-side rows
  // at about the same time (one trip to HBM, three L1 / L2 hits) -- the strips one by one moved 2.8 GB per call
  short first_strip[kMaxGroups + 1];
  short first_wg[kMaxGroups + 1];
  Strip strip[kMaxStrips];
};
constexpr int kPartFloats = kMaxKB * 1024 + 64;

__global__ void __launch_bounds__(256, 2) rows_wgrad_kernel(WgradArgsMvs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, kk = lane >> 5;
  int g = 0;
  while (g + 1 < a.ngroups && (int)blockIdx.x >= a.first_wg[g + 1]) ++g;
  const int si = a.first_strip[g] + wave;
  if (si >= a.first_strip[g + 1]) return;
  const int split = blockIdx.x - a.first_wg[g], nsplit = a.first_wg[g + 1] - a.first_wg[g];
  const Strip& S = a.strip[si];
  const long npairs = (a.ntiles + 1) / 2;
  f32x16 acc[kMaxKB];
#pragma unroll
  for (int kb = 0; kb < kMaxKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
  float bsum = 0.f;
  auto load = [&](long tile, int row0, int nrows, float (&v)[32]) {
    // (rows past the block / tiles past the end: row 0 of tile 0 is read instead and zeroed -- no divergent branch)
    const bool ok = m < nrows && tile < a.ntiles;
    const float4* p = reinterpret_cast<const float4*>(a.act + ((ok ? tile : 0) * TR::TOTAL + row0 + (ok ? m : 0)) * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 t = p[q];
      v[4 * q] = ok ? t.x : 0.f, v[4 * q + 1] = ok ? t.y : 0.f, v[4 * q + 2] = ok ? t.z : 0.f, v[4 * q + 3] = ok ? t.w : 0.f;
    }
  };
  for (long pair = split; pair < npairs; pair += nsplit) {
    const long tile = 2 * pair + kk;
    float av[32], bv[2][32];
    load(tile, S.a_row, S.a_rows, av);
    load(tile, S.b_row[0], S.b_rows[0], bv[0]);
#pragma unroll
    for (int j = 0; j < 32; ++j) bsum += av[j];
#pragma unroll
    for (int kb = 0; kb < kMaxKB; ++kb) {
      if (kb < S.nkb) {
        if (kb + 1 < S.nkb) load(tile, S.b_row[kb + 1 < kMaxKB ? kb + 1 : kb], S.b_rows[kb + 1 < kMaxKB ? kb + 1 : kb], bv[(kb + 1) & 1]);
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[kb] = BMV_MFMA(av[j], bv[kb & 1][j], acc[kb]);
      }
    }
  }
  float* __restrict__ part = a.partials + ((long)blockIdx.x * 4 + wave) * kPartFloats;
#pragma unroll
  for (int kb = 0; kb < kMaxKB; ++kb)
    if (kb < S.nkb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[kb * 1024 + n16(r, kk) * 32 + m] = acc[kb][r];
    }
  part[kMaxKB * 1024 + lane] = bsum;
}

struct GradPtrs {
  float* p[22];
};
// grid (strip, block): blocks 0 .. kMaxKB-1 sum one 32 x 32 weight block each over the strip's splits (fixed order),
// block kMaxKB the bias row sums
__global__ void __launch_bounds__(256) rows_wgrad_finish_kernel(WgradArgsMvs a, GradPtrs g) {
  const int si = blockIdx.x, kb = blockIdx.y;
  const Strip& S = a.strip[si];
  int gr = 0;
  while (gr + 1 < a.ngroups && si >= a.first_strip[gr + 1]) ++gr;
  const int nsplit = a.first_wg[gr + 1] - a.first_wg[gr];
  const long stride = 4L * kPartFloats;                    // between the splits of one strip
  const float* __restrict__ part = a.partials + ((long)a.first_wg[gr] * 4 + (si - a.first_strip[gr])) * kPartFloats;
  // fixed summation order (8 interleaved running sums, then a fixed tree): deterministic
  auto total = [&](int off) {
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8)
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] += part[(sp + q) * stride + off];
    for (; sp < nsplit; ++sp) t[sp & 7] += part[sp * stride + off];
    return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  };
  if (kb < kMaxKB) {
    if (kb >= S.nkb) return;
    float* __restrict__ out = g.p[S.param];
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) {
      const int row = e >> 5, j = e & 31;
      const int i = row - S.sel0;
      if (i < 0 || i >= S.out_rows || j >= S.b_rows[kb]) continue;
      out[(S.out_row0 + i) * S.ld + S.col[kb] + j] = total(kb * 1024 + e);
    }
  } else if (S.bias_param >= 0) {
    for (int e = threadIdx.x; e < S.out_rows; e += blockDim.x) {
      const int row = S.sel0 + e;
      g.p[S.bias_param][S.out_row0 + e] = total(kMaxKB * 1024 + row) + total(kMaxKB * 1024 + 32 + row);
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
template <int EP>
static void launch_gemm(const RowsGemm& a, hipStream_t st) {
  const int mpad = (a.M + 31) / 32 * 32;
  const unsigned gx = cdiv(a.ntiles, 8);
  if (mpad % 128 == 0) hipLaunchKernelGGL((rows_gemm_kernel<4, EP>), dim3(gx, mpad / 128), dim3(256), 0, st, a);
  else if (mpad % 64 == 0) hipLaunchKernelGGL((rows_gemm_kernel<2, EP>), dim3(gx, mpad / 64), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((rows_gemm_kernel<1, EP>), dim3(gx, mpad / 32), dim3(256), 0, st, a);
}
static RowsGemm gemm(float* act, long ntiles, const float* table, int ld, int col0, int M, const float* bias, int r0,
                     int k0, int r1, int k1, int out_row) {
  RowsGemm a;
  a.act = act, a.table = table, a.bias = bias, a.ntiles = ntiles, a.ld = ld, a.col0 = col0, a.M = M;
  a.seg_row[0] = r0, a.seg_k[0] = k0, a.seg_row[1] = r1, a.seg_k[1] = k1, a.out_row = out_row;
  a.aux0 = a.aux1 = 0, a.aux2 = -1, a.accumulate = 0;
  return a;
}

// parameter order of the gradient table = the order of bmv_mvs_mlp_params' members
enum : int { G_W0 = 0, G_B0 = 6, G_BIAS_W = 12, G_BIAS_B, G_VIEWS_W, G_VIEWS_B, G_FEAT_W, G_FEAT_B, G_ALPHA_W, G_ALPHA_B,
             G_RGB_W, G_RGB_B };

}  // namespace bmv

extern "C" {

long bmv_mvs_mlp_train_act_floats(long npts) { return ((npts + 31) / 32) * (long)bmv::TR::TOTAL * 32; }
long bmv_mvs_mlp_train_scratch_floats(void) {
  return (long)bmv::TW::TOTAL + (long)bmv::kMaxGroups * bmv::kMaxKB * bmv::kSplitPerKb * 4 * bmv::kPartFloats;
}

int bmv_mvs_mlp_train_fwd(const float* x, const bmv_mvs_mlp_params* params, long npts, float* act, float* scratch,
                          float* out, bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(x && params && act && scratch && out, "bmv_mvs_mlp_train_fwd: null pointer");
  BMV_REQUIRE(npts > 0, "bmv_mvs_mlp_train_fwd: npts=%ld", npts);
  const float* const* pp = reinterpret_cast<const float* const*>(params);
  for (int i = 0; i < 22; ++i) BMV_REQUIRE(pp[i], "bmv_mvs_mlp_train_fwd: parameter %d is null", i);
  hipStream_t st = as_stream(stream);
  const bmv_mvs_mlp_params& p = *params;
  const long T = (npts + 31) / 32;
  float* wt = scratch;
  hipLaunchKernelGGL(mvs_train_pack_kernel, dim3(cdiv(TW::TOTAL, 256)), dim3(256), 0, st, p, wt);
  hipLaunchKernelGGL(mvs_train_scatter_kernel, dim3((unsigned)T), dim3(256), 0, st, x, npts, act);
  // gate = pts_bias(feat)                                                      network.py:210
  launch_gemm<EP_STORE>(gemm(act, T, wt + TW::BIAS, 128, 0, 128, p.bias_b, TR::FEAT, 20, 0, 0, TR::BIAS), st);
  // h_l = relu(pts_linears[l](h) * gate), h = cat([pts, h]) after l == 4       network.py:211-216
  for (int l = 0; l < 6; ++l) {
    RowsGemm a = l == 0 ? gemm(act, T, wt + TW::L0, 128, 0, 128, p.pts_b[0], TR::PTS, 64, 0, 0, TR::Z(0))
                 : l == 5 ? gemm(act, T, wt + TW::L5, 128, 0, 128, p.pts_b[5], TR::PTS, 64, TR::H(4), 128, TR::Z(5))
                          : gemm(act, T, wt + TW::L(l), 128, 0, 128, p.pts_b[l], TR::H(l - 1), 128, 0, 0, TR::Z(l));
    a.aux0 = TR::BIAS, a.aux1 = TR::H(l);
    launch_gemm<EP_GATE_RELU>(a, st);
  }
  // feature = feature_linear(h); hv = relu(views_linears[0](cat([feature, views])))    network.py:221-226
  launch_gemm<EP_STORE>(gemm(act, T, wt + TW::FEATL, 128, 0, 128, p.feature_b, TR::H(5), 128, 0, 0, TR::FEATL), st);
  launch_gemm<EP_RELU>(gemm(act, T, wt + TW::VIEWS, 64, 0, 64, p.views_b, TR::FEATL, 128, TR::VIEW, 4, TR::HV), st);
  // alpha = relu(alpha_linear(h)), rgb = sigmoid(rgb_linear(hv))               network.py:220, 228
  hipLaunchKernelGGL(mvs_heads_fwd_kernel, dim3(cdiv(npts, 256)), dim3(256), 0, st, act, p, npts, out);
  BMV_LAUNCH_END("bmv_mvs_mlp_train_fwd");
}

int bmv_mvs_mlp_train_bwd(const bmv_mvs_mlp_params* params, float* act, float* scratch, const float* out,
                          const float* d_out, long npts, float* dx, const bmv_mvs_mlp_params* grads,
                          bmv_stream_t stream) {
  using namespace bmv;
  BMV_REQUIRE(params && act && scratch && out && d_out && dx && grads, "bmv_mvs_mlp_train_bwd: null pointer");
  BMV_REQUIRE(npts > 0, "bmv_mvs_mlp_train_bwd: npts=%ld", npts);
  const float* const* gp = reinterpret_cast<const float* const*>(grads);
  for (int i = 0; i < 22; ++i) BMV_REQUIRE(gp[i], "bmv_mvs_mlp_train_bwd: gradient %d is null", i);
  hipStream_t st = as_stream(stream);
  const bmv_mvs_mlp_params& p = *params;
  const long T = (npts + 31) / 32;
  hipLaunchKernelGGL(mvs_heads_bwd_kernel, dim3(cdiv(T * 32, 256)), dim3(256), 0, st, act, p, npts, out, d_out);
  // data path: dIn = W^T dOut with the parameter tensor as the k-major table (its rows are the reduction index)
  launch_gemm<EP_STORE>(gemm(act, T, p.views_w, 131, 0, 128, nullptr, TR::DHV, 64, 0, 0, TR::DFEATL), st);
  launch_gemm<EP_STORE>(gemm(act, T, p.views_w, 131, 128, 3, nullptr, TR::DHV, 64, 0, 0, TR::DVIEW), st);
  {
    RowsGemm a = gemm(act, T, p.feature_w, 128, 0, 128, nullptr, TR::DFEATL, 128, 0, 0, TR::Z(5));
    a.aux0 = TR::BIAS, a.aux1 = TR::DBIAS, a.aux2 = TR::TMP, a.accumulate = 0;
    launch_gemm<EP_BWD_HID>(a, st);
  }
  launch_gemm<EP_STORE>(gemm(act, T, p.pts_w[5], 191, 0, 63, nullptr, TR::Z(5), 128, 0, 0, TR::DPTS), st);
  for (int l = 5; l >= 1; --l) {
    RowsGemm a = gemm(act, T, p.pts_w[l], l == 5 ? 191 : 128, l == 5 ? 63 : 0, 128, nullptr, TR::Z(l), 128, 0, 0, TR::Z(l - 1));
    a.aux0 = TR::BIAS, a.aux1 = TR::DBIAS, a.aux2 = -1, a.accumulate = 1;
    launch_gemm<EP_BWD_HID>(a, st);
  }
  {
    RowsGemm a = gemm(act, T, p.pts_w[0], 63, 0, 63, nullptr, TR::Z(0), 128, 0, 0, TR::DPTS);
    a.accumulate = 1;
    launch_gemm<EP_STORE>(a, st);
  }
  launch_gemm<EP_STORE>(gemm(act, T, p.bias_w, 20, 0, 20, nullptr, TR::DBIAS, 128, 0, 0, TR::DFEAT), st);
  hipLaunchKernelGGL(mvs_train_gather_dx_kernel, dim3((unsigned)T), dim3(256), 0, st, act, npts, dx);

  // weight gradients
  WgradArgsMvs w;
  w.act = act, w.ntiles = T, w.partials = scratch + TW::TOTAL;
  int ns = 0;
  auto add = [&](int a_row, int a_rows, int out_row0, int sel0, int out_rows, int param, int bias_param, int ld,
                 std::initializer_list<int> brow, std::initializer_list<int> brows, std::initializer_list<int> col) {
    Strip& S = w.strip[ns++];
    S.a_row = (short)a_row, S.a_rows = (short)a_rows, S.nkb = (short)brow.size();
    S.sel0 = (short)sel0, S.out_rows = (short)out_rows, S.out_row0 = (short)out_row0, S.ld = (short)ld;
    S.param = (short)param, S.bias_param = (short)bias_param;
    int i = 0;
    for (int v : brow) S.b_row[i++] = (short)v;
    i = 0;
    for (int v : brows) S.b_rows[i++] = (short)v;
    i = 0;
    for (int v : col) S.col[i++] = (short)v;
  };
  int ng = 0;
  auto group = [&]() { w.first_strip[ng++] = (short)ns; };
  group();
  for (int o = 0; o < 128; o += 32) add(TR::DBIAS + o, 32, o, 0, 32, G_BIAS_W, G_BIAS_B, 20, {TR::FEAT}, {20}, {0});
  group();
  for (int o = 0; o < 128; o += 32)
    add(TR::Z(0) + o, 32, o, 0, 32, G_W0 + 0, G_B0 + 0, 63, {TR::PTS, TR::PTS + 32}, {32, 31}, {0, 32});
  for (int l = 1; l <= 4; ++l) {
    group();
    for (int o = 0; o < 128; o += 32)
      add(TR::Z(l) + o, 32, o, 0, 32, G_W0 + l, G_B0 + l, 128,
          {TR::H(l - 1), TR::H(l - 1) + 32, TR::H(l - 1) + 64, TR::H(l - 1) + 96}, {32, 32, 32, 32}, {0, 32, 64, 96});
  }
  group();
  for (int o = 0; o < 128; o += 32)
    add(TR::Z(5) + o, 32, o, 0, 32, G_W0 + 5, G_B0 + 5, 191,
        {TR::PTS, TR::PTS + 32, TR::H(4), TR::H(4) + 32, TR::H(4) + 64, TR::H(4) + 96}, {32, 31, 32, 32, 32, 32},
        {0, 32, 63, 95, 127, 159});
  group();
  for (int o = 0; o < 128; o += 32)
    add(TR::DFEATL + o, 32, o, 0, 32, G_FEAT_W, G_FEAT_B, 128, {TR::H(5), TR::H(5) + 32, TR::H(5) + 64, TR::H(5) + 96},
        {32, 32, 32, 32}, {0, 32, 64, 96});
  group();
  for (int o = 0; o < 64; o += 32)
    add(TR::DHV + o, 32, o, 0, 32, G_VIEWS_W, G_VIEWS_B, 131,
        {TR::FEATL, TR::FEATL + 32, TR::FEATL + 64, TR::FEATL + 96, TR::VIEW}, {32, 32, 32, 32, 3}, {0, 32, 64, 96, 128});
  group();
  add(TR::GOUT, 4, 0, 0, 1, G_ALPHA_W, G_ALPHA_B, 128, {TR::H(5), TR::H(5) + 32, TR::H(5) + 64, TR::H(5) + 96},
      {32, 32, 32, 32}, {0, 32, 64, 96});
  group();
  add(TR::GOUT, 4, 0, 1, 3, G_RGB_W, G_RGB_B, 64, {TR::HV, TR::HV + 32}, {32, 32}, {0, 32});
  w.first_strip[ng] = (short)ns;
  w.nstrips = ns, w.ngroups = ng;
  const long npairs = (T + 1) / 2;
  int nwg = 0;
  for (int i = 0; i < ng; ++i) {
    w.first_wg[i] = (short)nwg;
    const long sp = (long)kSplitPerKb * w.strip[w.first_strip[i]].nkb;
    nwg += (int)(sp < npairs ? sp : npairs);             // (never more splits than tile pairs)
  }
  w.first_wg[ng] = (short)nwg, w.nwgs = nwg;
  GradPtrs g;
  for (int i = 0; i < 22; ++i) g.p[i] = const_cast<float*>(gp[i]);
  hipLaunchKernelGGL(rows_wgrad_kernel, dim3(nwg), dim3(256), 0, st, w);
  hipLaunchKernelGGL(rows_wgrad_finish_kernel, dim3(ns, kMaxKB + 1), dim3(256), 0, st, w, g);
  BMV_LAUNCH_END("bmv_mvs_mlp_train_bwd");
}

}  // extern "C"
