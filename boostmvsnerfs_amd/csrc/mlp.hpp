// ENeRF's tiny MLP (a11: Agg + NeRF, lib/networks/enerf/nerf.py:29-43, 74-89) on
// the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Orientation.  Every layer is computed transposed, OUT^T[n, sample] = W[n, k] IN^T[k, sample],
// with the weights as the MFMA A operand and the activations as B, so that the
// SAMPLE index lives on the lane (lane & 31) and the neurons live in the 16
// accumulator registers: lane (s, h = lane >> 5) register r holds neuron
// n16(r, h) = (r & 3) + 8 (r >> 2) + 4 h of sample s.  A 32x32x2 step consumes
// k = 2 inputs, one from each lane half, so the accumulator registers of one
// layer are directly the B operands of the next (k-step r takes register r from
// both halves = inputs n16(r,0), n16(r,1)); the weights are pre-permuted to that
// k order once (pack kernel below).  No activation ever crosses lanes or LDS
// between layers; only the 1-wide heads (agg weight, sigma, colour logit) need a
// single cross-half add.
//
// Algebra.  global_fc and color.0 are split into a part shared by the NV source views
// (computed once, then used as the C-in of each view's chain) and a per-view part:
//   global_fc([f_i, var, mean]) = Wg[:, :F] f_i + (Wg[:, F:2F] var + Wg[:, 2F:] mean + b)
//   color.0([x, v, in_i])       = Wc[:, 88:] in_i + (Wc[:, :88] [x, v] + b)
// 205 MFMAs per 32 samples for feat_ch = 8 and 3 views (26.2 kFLOP/sample instead of 50.9).
//
// Views.  The reference's Agg / NeRF are written for ANY number of source views (nerf.py:29-43, 74-89: var / mean /
// softmax over dim -2; ENeRF pre-trains with train_input_views [2, 3, 4], configs/exps/pretrain/enerf/
// dtu_pretrain.yaml:22-23): NV is a template parameter of everything below (2, 3, 4 instantiated); the weight blob does
// not depend on it.
#pragma once
#include "bmv_common.hpp"

namespace bmv {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__host__ __device__ constexpr int n16(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int FEAT_CH>
struct MlpLayout {
  static constexpr int FC = FEAT_CH + 3;       // image feature + rgb channels per view
  static constexpr int KFC = (FC + 1) / 2;     // k-steps covering them (2 per step)
  static constexpr int KF = KFC + 2;           // + 4 direction components
  static constexpr int IN = FC + 4;            // per-view input width of the reference
  // MFMA A tables: [step][tile][64 lanes]
  static constexpr int A_GSH = 0;                          // global_fc, var|mean part: 2*KFC steps
  static constexpr int A_GV = A_GSH + 2 * KFC * 64;        // global_fc, per-view part:  KFC steps
  static constexpr int A_FC = A_GV + KFC * 64;             // agg.fc: 16 steps
  static constexpr int A_L0 = A_FC + 16 * 64;              // lr0: 12 steps x 2 tiles
  static constexpr int A_CSH = A_L0 + 12 * 2 * 64;         // color.0 shared part: 44 steps x 2 tiles
  static constexpr int A_CV = A_CSH + 44 * 2 * 64;         // color.0 per-view part: KF steps x 2 tiles
  // VALU tables: [idx][2 halves]
  static constexpr int V_VF = A_CV + KF * 2 * 64;          // view_fc: [KFC][4 w + bias]
  static constexpr int V_BG = V_VF + KFC * 5 * 2;          // global_fc bias [16]
  static constexpr int V_WA = V_BG + 32;                   // agg_w_fc weight [16]
  static constexpr int V_BFC = V_WA + 32;                  // agg.fc bias [16]
  static constexpr int V_B0 = V_BFC + 32;                  // lr0 bias [2*16]
  static constexpr int V_WS = V_B0 + 64;                   // sigma weight [2*16]
  static constexpr int V_BC = V_WS + 64;                   // color.0 bias [2*16]
  static constexpr int V_WC2 = V_BC + 64;                  // color.2 weight [2*16]
  static constexpr int V_SC = V_WC2 + 64;                  // scalars: agg_w bias, sigma bias, color.2 bias
  static constexpr int TOTAL = (V_SC + 4 + 3) / 4 * 4;
  // Experiment (round 5, bmv_tuning BMV_RENDER_SPLIT): the two-tile chains -- lr0, color.0's shared and per-view parts: 160
  // of the MLP's 206 fp32 MFMAs per tile -- on the bf16 matrix pipe with both operands split into THREE bf16 pieces (8 + 8
  // + 8 mantissa bits: the fp32 value exactly).  Their A tables a second time, behind the fp32 blob, pre-split:
  // [piece 3][bf16 k-step][tile 2][lane 64] x 8 bf16 (16 bytes); one bf16 k-step = 8 fp32 k-steps (zero padded).
  static constexpr int SK_L0 = 2;                          // lr0: 12 fp32 k-steps
  static constexpr int SK = 6;                             // color.0 shared: 44
  static constexpr int SK_CV = (KF + 7) / 8;               // color.0 per view: KF
  static constexpr int S_L0 = TOTAL;
  static constexpr int S_CSH = S_L0 + 3 * SK_L0 * 2 * 64 * 4;
  static constexpr int S_CV = S_CSH + 3 * SK * 2 * 64 * 4;
  static constexpr int TOTAL_S = S_CV + 3 * SK_CV * 2 * 64 * 4;
  // the split form keeps blob[0, A_L0) ++ blob[V_VF, TOTAL_S) in LDS: the fp32 tables of the three chains stay out
  static constexpr int LDS_HOLE = V_VF - A_L0;
  static constexpr int LDS_S = TOTAL_S - LDS_HOLE;
};

// --------------------------------------------------------------------------
// Weight packing: one thread per blob element, reading the reference's
// parameter tensors (row-major (out, in) weights) directly.
// --------------------------------------------------------------------------
template <int FEAT_CH>
__global__ void nerf_pack_kernel(bmv_nerf_params p, float* __restrict__ blob) {
  using L = MlpLayout<FEAT_CH>;
  constexpr int FC = L::FC, KFC = L::KFC;
  constexpr int CW = 88 + FC + 4;  // color.0 input width
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L::TOTAL_S) return;
  if (idx >= L::TOTAL) {
    // split tables: dword q of (piece, bf16 k-step T, tile, lane (i, h)) = that piece of the fp32 table's entries of fp32
    // k-steps t = 8 T + 2 q, 8 T + 2 q + 1 for this tile and lane (low | high half), zero beyond the chain's length
    const int which = idx >= L::S_CV ? 2 : idx >= L::S_CSH ? 1 : 0;
    int e = idx - (which == 2 ? L::S_CV : which == 1 ? L::S_CSH : L::S_L0);
    const int nk = which == 2 ? L::SK_CV : which == 1 ? L::SK : L::SK_L0;
    const int q = e & 3;
    e >>= 2;
    const int lane = e & 63;
    e >>= 6;
    const int tl = e & 1;
    e >>= 1;
    const int T = e % nk, pc = e / nk;
    const int i = lane & 31, h = lane >> 5, n = 32 * tl + i;
    unsigned packed = 0;
    for (int jj = 0; jj < 2; ++jj) {
      const int t = 8 * T + 2 * q + jj;
      float w = 0.f;
      if (which == 0) {
        if (t < 12) w = p.lr0_w[n * 24 + (t < 4 ? 2 * t + h : 8 + n16(t - 4, h))];
      } else if (which == 1) {
        if (t < 44) {
          const int k = t < 32 ? 32 * (t >> 4) + n16(t & 15, h) : t < 36 ? 64 + 2 * (t - 32) + h : 72 + n16(t - 36, h);
          w = p.color0_w[n * CW + k];
        }
      } else if (t < L::KF) {
        if (t < KFC) {
          const int c = 2 * t + h;
          if (c < FC) w = p.color0_w[n * CW + 88 + c];
        } else {
          w = p.color0_w[n * CW + 88 + FC + 2 * (t - KFC) + h];
        }
      }
      // round to nearest even, as v_cvt_pk_bf16_f32 does for the activations (finite weights)
      auto rn = [](float v) {
        const unsigned u = __float_as_uint(v);
        return __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
      };
      const float hi = rn(w);
      const float r1 = w - hi;
      const float mid = rn(r1);
      const float r2 = r1 - mid;                                    // (exact: at most 8 significant bits are left)
      const float piece = pc == 0 ? hi : pc == 1 ? mid : r2;
      packed |= (__float_as_uint(piece) >> 16) << (16 * jj);
    }
    blob[idx] = __uint_as_float(packed);
    return;
  }
  float v = 0.f;
  if (idx < L::V_VF) {
    int lane = idx & 63, e = idx >> 6;  // all A tables are 64-aligned
    int i = lane & 31, h = lane >> 5;
    if (idx < L::A_GV) {
      int t = e - L::A_GSH / 64;
      int which = t / KFC, c = 2 * (t % KFC) + h;
      if (c < FC) v = p.global_fc_w[i * 3 * FC + FC * (1 + which) + c];
    } else if (idx < L::A_FC) {
      int t = e - L::A_GV / 64, c = 2 * t + h;
      if (c < FC) v = p.global_fc_w[i * 3 * FC + c];
    } else if (idx < L::A_L0) {
      int t = e - L::A_FC / 64;
      if (i < 16) v = p.fc_w[i * 32 + n16(t, h)];
    } else if (idx < L::A_CSH) {
      int q = e - L::A_L0 / 64, t = q >> 1, n = 32 * (q & 1) + i;
      int k = t < 4 ? 2 * t + h : 8 + n16(t - 4, h);
      v = p.lr0_w[n * 24 + k];
    } else if (idx < L::A_CV) {
      int q = e - L::A_CSH / 64, t = q >> 1, n = 32 * (q & 1) + i;
      int k;
      if (t < 32)
        k = 32 * (t >> 4) + n16(t & 15, h);
      else if (t < 36)
        k = 64 + 2 * (t - 32) + h;
      else
        k = 72 + n16(t - 36, h);
      v = p.color0_w[n * CW + k];
    } else {
      int q = e - L::A_CV / 64, t = q >> 1, n = 32 * (q & 1) + i;
      if (t < KFC) {
        int c = 2 * t + h;
        if (c < FC) v = p.color0_w[n * CW + 88 + c];
      } else {
        v = p.color0_w[n * CW + 88 + FC + 2 * (t - KFC) + h];
      }
    }
  } else {
    int rel, h;
    if (idx < L::V_BG) {
      rel = idx - L::V_VF, h = rel & 1, rel >>= 1;
      int j = rel / 5, q = rel % 5, c = 2 * j + h;
      if (c < FC) v = q < 4 ? p.view_fc_w[c * 4 + q] : p.view_fc_b[c];
    } else if (idx < L::V_WA) {
      rel = idx - L::V_BG, h = rel & 1, rel >>= 1;
      v = p.global_fc_b[n16(rel, h)];
    } else if (idx < L::V_BFC) {
      rel = idx - L::V_WA, h = rel & 1, rel >>= 1;
      v = p.agg_w_w[n16(rel, h)];
    } else if (idx < L::V_B0) {
      rel = idx - L::V_BFC, h = rel & 1, rel >>= 1;
      int n = n16(rel, h);
      if (n < 16) v = p.fc_b[n];
    } else if (idx < L::V_WS) {
      rel = idx - L::V_B0, h = rel & 1, rel >>= 1;
      v = p.lr0_b[32 * (rel >> 4) + n16(rel & 15, h)];
    } else if (idx < L::V_BC) {
      rel = idx - L::V_WS, h = rel & 1, rel >>= 1;
      v = p.sigma_w[32 * (rel >> 4) + n16(rel & 15, h)];
    } else if (idx < L::V_WC2) {
      rel = idx - L::V_BC, h = rel & 1, rel >>= 1;
      v = p.color0_b[32 * (rel >> 4) + n16(rel & 15, h)];
    } else if (idx < L::V_SC) {
      rel = idx - L::V_WC2, h = rel & 1, rel >>= 1;
      v = p.color2_w[32 * (rel >> 4) + n16(rel & 15, h)];
    } else {
      rel = idx - L::V_SC;
      v = rel == 0 ? p.agg_w_b[0] : rel == 1 ? p.sigma_b[0] : rel == 2 ? p.color2_b[0] : 0.f;
    }
  }
  blob[idx] = v;
}

#define BMV_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#ifndef BMV_MLP_G1
#define BMV_MLP_G1 3   // k-steps per software-pipelined group, one-tile chains
#endif
#ifndef BMV_MLP_G2
#define BMV_MLP_G2 2   // two-tile chains (2 MFMAs per k-step)
#endif
// The weight reads are LDS loads with compile-time offsets; left alone, the
// scheduler hoists dozens of them ahead of the MFMA chain they feed and spills.
// A scheduling fence every few k-steps keeps the live set to one chunk.
#define BMV_FENCE() __builtin_amdgcn_sched_barrier(0)
#define BMV_FENCE_EVERY(t, n) \
  do {                         \
    if (((t) % (n)) == (n)-1) BMV_FENCE(); \
  } while (0)

// MFMA chains with the A operands (weights, LDS) software-pipelined by one group of G k-steps: the ds_reads of
// group g + 1 are issued BEFORE the MFMAs of group g, so no MFMA waits for an LDS round trip (with the reads issued
// right behind a scheduling fence, as before, the first MFMA of every group stalled ~100-200 cycles: one s_waitcnt
// per MFMA in the ISA, the matrix pipe 46-54 % busy).  `t` is the k-step visible to BEXPR.
//   BMV_CHAIN1: one accumulator tile,  A of step t at Wa[AOFF + t * 64]
//   BMV_CHAIN2: two accumulator tiles, A of (step t, tile tl) at Wa[AOFF + (t * 2 + tl) * 64]
#define BMV_CHAIN1(AOFF, NT, G, BEXPR, ACC)                                                        \
  {                                                                                                \
    float a_[2][G];                                                                                \
    _Pragma("unroll") for (int u_ = 0; u_ < (G); ++u_)                                             \
      if (u_ < (NT)) a_[0][u_] = Wa[(AOFF) + u_ * 64];                                             \
    _Pragma("unroll") for (int g_ = 0; g_ < ((NT) + (G)-1) / (G); ++g_) {                          \
      _Pragma("unroll") for (int u_ = 0; u_ < (G); ++u_) {                                         \
        const int tn_ = (g_ + 1) * (G) + u_;                                                       \
        if (tn_ < (NT)) a_[(g_ + 1) & 1][u_] = Wa[(AOFF) + tn_ * 64];                              \
      }                                                                                            \
      BMV_FENCE();                                                                                 \
      _Pragma("unroll") for (int u_ = 0; u_ < (G); ++u_) {                                         \
        const int t = g_ * (G) + u_;                                                               \
        if (t < (NT)) {                                                                            \
          const float b_ = (BEXPR);                                                                \
          ACC = BMV_MFMA(a_[g_ & 1][u_], b_, ACC);                                                 \
        }                                                                                          \
      }                                                                                            \
      BMV_FENCE();                                                                                 \
    }                                                                                              \
  }
#define BMV_CHAIN2(AOFF, NT, G, BEXPR, ACC0, ACC1)                                                 \
  {                                                                                                \
    float a_[2][G][2];                                                                             \
    _Pragma("unroll") for (int u_ = 0; u_ < (G); ++u_)                                             \
      if (u_ < (NT)) a_[0][u_][0] = Wa[(AOFF) + (u_ * 2) * 64], a_[0][u_][1] = Wa[(AOFF) + (u_ * 2 + 1) * 64]; \
    _Pragma("unroll") for (int g_ = 0; g_ < ((NT) + (G)-1) / (G); ++g_) {                          \
      _Pragma("unroll") for (int u_ = 0; u_ < (G); ++u_) {                                         \
        const int tn_ = (g_ + 1) * (G) + u_;                                                       \
        if (tn_ < (NT))                                                                            \
          a_[(g_ + 1) & 1][u_][0] = Wa[(AOFF) + (tn_ * 2) * 64], a_[(g_ + 1) & 1][u_][1] = Wa[(AOFF) + (tn_ * 2 + 1) * 64]; \
      }                                                                                            \
      BMV_FENCE();                                                                                 \
      _Pragma("unroll") for (int u_ = 0; u_ < (G); ++u_) {                                         \
        const int t = g_ * (G) + u_;                                                               \
        if (t < (NT)) {                                                                            \
          const float b_ = (BEXPR);                                                                \
          ACC0 = BMV_MFMA(a_[g_ & 1][u_][0], b_, ACC0);                                            \
          ACC1 = BMV_MFMA(a_[g_ & 1][u_][1], b_, ACC1);                                            \
        }                                                                                          \
      }                                                                                            \
      BMV_FENCE();                                                                                 \
    }                                                                                              \
  }

// A two-tile chain on v_mfma_f32_32x32x16_bf16 with three-piece operands (CSPLIT): 8 fp32 k-steps = one bf16 k-step -- a
// lane's 8 B values are its own registers of those steps, as the fp32 chain takes them one by one; operands = hi + mid +
// lo bf16 pieces by ROUND TO NEAREST (hi = rn(x), mid = rn(x - hi), lo = x - hi - mid: exact, |mid| <= 2^-8 |x|, |lo| <= 2^-16
// |x|), products hi hi + hi mid + mid hi + hi lo + lo hi + mid mid (what is dropped -- mid lo, lo mid, lo lo -- is at
// most 2^-23 of the product: one fp32 rounding), small terms first.  `t` is the fp32 k-step visible to BEXPR; steps >= NT are zeros.
// SPTR: the chain's split table in LDS, [piece][k-step][tile][lane] x 16 bytes, already offset by the lane.
__device__ __forceinline__ unsigned mlp_cvt_pk_bf16(float a, float b) {   // [rn_bf16(a) | rn_bf16(b) << 16]
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
using mlp_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using mlp_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
// BMV_SPLIT_AHEAD (a constexpr bool in scope, mlp_forward's AHEAD): the three A pieces of the NEXT group (k-step, tile) are read from
// LDS under the six MFMAs and the operand split of this one (+ 12 VGPRs: 157 -> 168 of the 170 three waves per SIMD leave
// the fused renderer -- only where that does not spill; renderer 143.5 -> 142.0 us, bit-identical, round 6)
#define BMV_SPLIT_CHAIN2(SPTR, NK, NT, BEXPR, ACC0, ACC1)                                          \
  {                                                                                                \
    mlp_u32x4 ax_[3];                                                                              \
    _Pragma("unroll") for (int T_ = 0; T_ < (NK); ++T_) {                                          \
      mlp_u32x4 bh_, bm_, bl_;                                                                     \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                           \
        float bv_[2];                                                                              \
        _Pragma("unroll") for (int jj_ = 0; jj_ < 2; ++jj_) {                                      \
          const int t = 8 * T_ + 2 * q_ + jj_;                                                     \
          bv_[jj_] = t < (NT) ? (BEXPR) : 0.f;                                                     \
        }                                                                                          \
        const unsigned ph_ = mlp_cvt_pk_bf16(bv_[0], bv_[1]);                                      \
        const float r10_ = bv_[0] - __uint_as_float(ph_ << 16), r11_ = bv_[1] - __uint_as_float(ph_ & 0xffff0000u); \
        const unsigned pm_ = mlp_cvt_pk_bf16(r10_, r11_);                                          \
        const float r20_ = r10_ - __uint_as_float(pm_ << 16), r21_ = r11_ - __uint_as_float(pm_ & 0xffff0000u);     \
        bh_[q_] = ph_, bm_[q_] = pm_, bl_[q_] = mlp_cvt_pk_bf16(r20_, r21_);                       \
      }                                                                                            \
      const mlp_bf16x8 Bh_ = __builtin_bit_cast(mlp_bf16x8, bh_), Bm_ = __builtin_bit_cast(mlp_bf16x8, bm_),  \
                       Bl_ = __builtin_bit_cast(mlp_bf16x8, bl_);                                  \
      _Pragma("unroll") for (int tl_ = 0; tl_ < 2; ++tl_) {                                        \
        mlp_u32x4 an_[3];                                                                          \
        if (BMV_SPLIT_AHEAD && !(T_ == 0 && tl_ == 0)) { an_[0] = ax_[0], an_[1] = ax_[1], an_[2] = ax_[2]; }  \
        else {                                                                                     \
          _Pragma("unroll") for (int pc_ = 0; pc_ < 3; ++pc_) an_[pc_] = (SPTR)[((pc_ * (NK) + T_) * 2 + tl_) * 64];  \
        }                                                                                          \
        if (BMV_SPLIT_AHEAD && !(T_ == (NK) - 1 && tl_ == 1)) {   /* the next group's pieces under this group's MFMAs */ \
          const int Tn_ = tl_ == 1 ? T_ + 1 : T_, tn_ = tl_ ^ 1;                                   \
          _Pragma("unroll") for (int pc_ = 0; pc_ < 3; ++pc_) ax_[pc_] = (SPTR)[((pc_ * (NK) + Tn_) * 2 + tn_) * 64]; \
        }                                                                                          \
        const mlp_bf16x8 Ah_ = __builtin_bit_cast(mlp_bf16x8, an_[0]);                             \
        const mlp_bf16x8 Am_ = __builtin_bit_cast(mlp_bf16x8, an_[1]);                             \
        const mlp_bf16x8 Al_ = __builtin_bit_cast(mlp_bf16x8, an_[2]);                             \
        f32x16 c_ = tl_ == 0 ? ACC0 : ACC1;                                                        \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al_, Bh_, c_, 0, 0, 0);                       \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah_, Bl_, c_, 0, 0, 0);                       \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am_, Bm_, c_, 0, 0, 0);                       \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am_, Bh_, c_, 0, 0, 0);                       \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah_, Bm_, c_, 0, 0, 0);                       \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah_, Bh_, c_, 0, 0, 0);                       \
        if (tl_ == 0) ACC0 = c_; else ACC1 = c_;                                                   \
        BMV_FENCE();   /* (one tile's 12 A registers at a time: the kernel sits at its register cap) */ \
      }                                                                                            \
    }                                                                                              \
  }

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// ReLU as ONE instruction.  fmaxf(x, 0.f) on an MFMA result compiles to TWO v_max_f32: llvm.maxnum wants a
// canonicalised operand and the compiler cannot prove an accumulator register is one, so it emits v_max x, x, x first
// (round 3: 412 v_max in the MLP of a tile, ~190 of them such canonicalisations; fp32 MFMA and the vector ALU share the
// SIMD's FMA lanes, so every vector instruction of the MLP adds to the tile's time).  v_max_f32 with a NaN operand
// returns the other one: NaN -> 0, as fmaxf.
__device__ __forceinline__ float relu1(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

// x / NV, correctly rounded like the IEEE division it replaces (torch's mean divides by the count) in 3 instructions
// instead of the ~10 of the division expansion: NV = 2, 4 are exact multiplications; NV = 3 is Markstein's sequence
// q = RN(x y), r = x - 3 q (exact in an FMA), RN(q + r y) with y = RN(1/3) (correctly rounded for normal x)
template <int NV>
__device__ __forceinline__ float div_views(float x) {
  if constexpr (NV == 3) {
    const float y = 1.f / 3.f;
    const float q = x * y;
    const float r = __builtin_fmaf(-3.f, q, x);
    return __builtin_fmaf(r, y, q);
  } else {
    return x * (1.f / (float)NV);
  }
}

// softmax over the views of a sample (nerf.py:41, 88), in place; max first, sums left to right
template <int NV>
__device__ __forceinline__ void softmax_views(float (&a)[NV]) {
  float m = a[NV - 1];
#pragma unroll
  for (int i = NV - 2; i >= 0; --i) m = fmaxf(a[i], m);
  float e[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) e[i] = __expf(a[i] - m);
  float den = e[0];
#pragma unroll
  for (int i = 1; i < NV; ++i) den += e[i];
  const float inv = BMV_DIV(1.f, den);
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = e[i] * inv;
}
template <int NV>
__device__ __forceinline__ float weighted_views(const float (&w)[NV], const f32x16 (&g)[NV], int t) {
  float v = w[0] * g[0][t];
#pragma unroll
  for (int i = 1; i < NV; ++i) v += w[i] * g[i][t];
  return v;
}

// --------------------------------------------------------------------------
// Per-lane inputs (lane = (sample s, half h)):
//   fin[i][j]  j <  KFC : channel 2j+h of view i's [feature, rgb] vector (0 beyond FC)
//              j >= KFC : direction component 2(j-KFC)+h
//   dir[i][q]  all four direction components of view i
//   vox[j]     feature-volume channel 2j+h
// W: the packed blob in LDS.  out = [r, g, b, sigma], identical in both halves.
// --------------------------------------------------------------------------
template <int FEAT_CH, int NV = 3, bool CSPLIT = false, bool AHEAD = false>
__device__ __forceinline__ void mlp_forward(const float* __restrict__ W, int lane,
                                            const float (&fin)[NV][MlpLayout<FEAT_CH>::KF], const float (&dir)[NV][4],
                                            const float (&vox)[4], float (&out)[4]) {
  static_assert(NV >= 2 && NV <= 4, "source views per cost volume");
  constexpr bool BMV_SPLIT_AHEAD = AHEAD;
  using L = MlpLayout<FEAT_CH>;
  constexpr int KFC = L::KFC, KF = L::KF;
  const int h = lane >> 5;
  // (CSPLIT: the LDS image is blob[0, A_L0) ++ blob[V_VF, TOTAL_S) -- everything from the vector tables on sits LDS_HOLE
  // floats lower than in the blob)
  constexpr int HOLE = CSPLIT ? L::LDS_HOLE : 0;
  const float* __restrict__ Wa = W + lane;
  const float* __restrict__ Wv = W + h - HOLE;

  // Agg.view_fc + residual (nerf.py:77-79): f_i = in_i[:F] + relu(W_v dir_i + b).  f is
  // cheap, so it is recomputed where the per-view chain needs it instead of kept live.
  auto fval = [&](int i, int j) -> float {
    float pre = Wv[L::V_VF + (j * 5 + 4) * 2] + Wv[L::V_VF + (j * 5 + 0) * 2] * dir[i][0] +
                Wv[L::V_VF + (j * 5 + 1) * 2] * dir[i][1] + Wv[L::V_VF + (j * 5 + 2) * 2] * dir[i][2] +
                Wv[L::V_VF + (j * 5 + 3) * 2] * dir[i][3];
    return fin[i][j] + relu1(pre);
  };
  // unbiased variance and mean over the NV views (nerf.py:83-84); the f are kept for the per-view chain (feat_ch 8:
  // 18 registers against 144 vector instructions to recompute them)
  constexpr bool KEEP_F = KFC <= 6;
  float fkeep[KEEP_F ? NV : 1][KEEP_F ? KFC : 1];
  float var[KFC], mean[KFC];
#pragma unroll
  for (int j = 0; j < KFC; ++j) {
    float f[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f[i] = fval(i, j);
      if constexpr (KEEP_F) fkeep[i][j] = f[i];
    }
    float m = f[0];
#pragma unroll
    for (int i = 1; i < NV; ++i) m += f[i];
    m = div_views<NV>(m);
    float ss = (f[0] - m) * (f[0] - m);
#pragma unroll
    for (int i = 1; i < NV; ++i) ss += (f[i] - m) * (f[i] - m);
    mean[j] = m;
    var[j] = ss * (1.f / (float)(NV - 1));
    BMV_FENCE_EVERY(j, 6);
  }
  BMV_FENCE();
  // global_fc (nerf.py:86-87)
  f32x16 gsh;
#pragma unroll
  for (int r = 0; r < 16; ++r) gsh[r] = Wv[L::V_BG + r * 2];
  BMV_CHAIN1(L::A_GSH, 2 * KFC, BMV_MLP_G1, (t < KFC ? var[t < KFC ? t : 0] : mean[t >= KFC ? t - KFC : 0]), gsh)
  f32x16 g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g[i] = gsh;
    BMV_CHAIN1(L::A_GV, KFC, BMV_MLP_G1, (KEEP_F ? fkeep[KEEP_F ? i : 0][KEEP_F ? t : 0] : fval(i, t)), g[i])
#pragma unroll
    for (int r = 0; r < 16; ++r) g[i][r] = relu1(g[i][r]);
    BMV_FENCE();
  }
  // agg_w_fc + softmax over views + weighted sum (nerf.py:88-89)
  float aw[NV];
  const float ba = W[L::V_SC - HOLE + 0], bs = W[L::V_SC - HOLE + 1], bc2 = W[L::V_SC - HOLE + 2];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += Wv[L::V_WA + r * 2] * g[i][r];
    aw[i] = fmaxf(xhalf_sum(s) + ba, 0.f);
  }
  softmax_views<NV>(aw);
  // agg.fc (nerf.py:90): 32 -> 16
  f32x16 q;
#pragma unroll
  for (int r = 0; r < 16; ++r) q[r] = Wv[L::V_BFC + r * 2];
  BMV_CHAIN1(L::A_FC, 16, BMV_MLP_G1, (weighted_views<NV>(aw, g, t)), q)
  float im16[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) im16[r] = relu1(q[r]);
  // lr0 (nerf.py:34-35): [vox(8), im(16)] -> 64
  f32x16 x[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[tl][r] = Wv[L::V_B0 + (tl * 16 + r) * 2];
  if constexpr (CSPLIT) {
    const mlp_u32x4* sp = reinterpret_cast<const mlp_u32x4*>(W + L::S_L0 - HOLE) + lane;
    BMV_SPLIT_CHAIN2(sp, L::SK_L0, 12, (t < 4 ? vox[t & 3] : im16[(t - 4) & 7]), x[0], x[1])
  } else {
    BMV_CHAIN2(L::A_L0, 12, BMV_MLP_G2, (t < 4 ? vox[t < 4 ? t : 0] : im16[t >= 4 ? t - 4 : 0]), x[0], x[1])
  }
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[tl][r] = relu1(x[tl][r]);
  // sigma head (nerf.py:38)
  {
    float s = 0.f;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += Wv[L::V_WS + (tl * 16 + r) * 2] * x[tl][r];
    float pre = xhalf_sum(s) + bs;
    out[3] = pre > 20.f ? pre : log1pf(__expf(pre));
  }
  // color.0 shared part: [x(64), vox(8), im(16)]
  f32x16 csh[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int r = 0; r < 16; ++r) csh[tl][r] = Wv[L::V_BC + (tl * 16 + r) * 2];
  if constexpr (CSPLIT) {
    const mlp_u32x4* sp = reinterpret_cast<const mlp_u32x4*>(W + L::S_CSH - HOLE) + lane;
    BMV_SPLIT_CHAIN2(sp, L::SK, 44, (t < 32 ? x[(t >> 4) & 1][t & 15] : t < 36 ? vox[t & 3] : im16[(t - 36) & 7]), csh[0], csh[1])
  } else {
  BMV_CHAIN2(L::A_CSH, 44, BMV_MLP_G2,
             (t < 32 ? x[t < 32 ? (t >> 4) : 0][t & 15] : t < 36 ? vox[t >= 32 && t < 36 ? t - 32 : 0] : im16[t >= 36 ? t - 36 : 0]),
             csh[0], csh[1])
  }
  // per-view part + color.2 + softmax over views (nerf.py:39-42)
  float cl[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    f32x16 hc[2] = {csh[0], csh[1]};
    if constexpr (CSPLIT) {
      const mlp_u32x4* sp = reinterpret_cast<const mlp_u32x4*>(W + L::S_CV - HOLE) + lane;
      BMV_SPLIT_CHAIN2(sp, L::SK_CV, KF, fin[i][t < KF ? t : 0], hc[0], hc[1])
    } else {
      BMV_CHAIN2(L::A_CV, KF, BMV_MLP_G2, fin[i][t], hc[0], hc[1])
    }
    float s = 0.f;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += Wv[L::V_WC2 + (tl * 16 + r) * 2] * relu1(hc[tl][r]);
    cl[i] = fmaxf(xhalf_sum(s) + bc2, 0.f);
    BMV_FENCE();
  }
  softmax_views<NV>(cl);
  // blend the sampled source colours: channels FEAT_CH + {0,1,2} sit at
  // (half 0, slot J), (half 1, slot J), (half 0, slot J+1) with J = FEAT_CH / 2
  constexpr int J = FEAT_CH / 2;
  static_assert(FEAT_CH % 2 == 0, "feat_ch must be even");
  float pa = cl[0] * fin[0][J], pb = cl[0] * fin[0][J + 1];
#pragma unroll
  for (int i = 1; i < NV; ++i) pa += cl[i] * fin[i][J], pb += cl[i] * fin[i][J + 1];
  float oa = __shfl_xor(pa, 32, 64), ob = __shfl_xor(pb, 32, 64);
  out[0] = h == 0 ? pa : oa;
  out[1] = h == 0 ? oa : pa;
  out[2] = h == 0 ? pb : ob;
}

}  // namespace bmv
