// FeatureNet's encoder convolutions (feature_net.py:11-19: conv1 = ConvBnReLU(8, 16, 5, 2) + ConvBnReLU(16, 16, 3),
// conv2 = ConvBnReLU(16, 32, 5, 2) + ConvBnReLU(32, 32, 3), batch norm folded) on the BF16 matrix cores with three-piece
// fp32 operands (round 6; the arithmetic of csrc/conv_c4s.hip / csrc/fpn_s.hip: six v_mfma_f32_16x16x32_bf16 per product
// group on hi + mid + lo pieces of both operands, fp32 accumulation, smallest terms first).  Replaces csrc/conv.hip's
// fp32 16x16x4 kernels for these layers at sizes that fill the chip (25 + 19 + 27 us of a 0.73 ms frame).
//
//  * matrix rows = 16 output channels (an "M tile"; 32-channel layers: two waves per strip, one per M tile), matrix
//    columns = 16 adjacent output pixels of a row, one k-step of 32 = FOUR (input octet, filter column) pairs x the 8
//    channels of the octet: the pairs of a filter row are enumerated octet-major and packed four to a step, so the
//    k-dimension is 5/8 used for 8 -> 16 5x5, 10/12 for 16 -> 32 5x5, 6/8 for 16 -> 16 3x3, 12/12 for 32 -> 32 3x3;
//  * WEIGHTS STATIONARY: a wave's A operands (filter rows x steps x 3 pieces, 72-180 registers) are loaded once and stay
//    in registers for its whole strip;
//  * ROW WALK, input-stationary: a wave owns a strip of 16 NT output columns x TY output rows and walks the
//    STR (TY - 1) + KS input rows once; a staged row serves every output row it touches (stride 2: filter rows of its own
//    parity) from the same B operands: one ds_read_b128 per piece feeds 6 x (2 or 3) matrix instructions;
//  * staging: lane l loads input columns (2 l, 2 l + 1) of the strip's window (origin STR X0 - 2: even, so a lane's two
//    columns are in or out of the image TOGETHER and the zero padding is the buffer's range check) for every channel as
//    one 8-byte load, two rows ahead of their use; splits the 8 channels of an octet into three bf16 pieces per column
//    (11 vector instructions per two values) and parks them as 16-byte records in its wave's own LDS, even and odd
//    columns apart (the B reads of 16 lanes are then consecutive records for both strides);
//  * WAVES ARE INDEPENDENT (as fpn_s.hip's): no workgroup barrier; the two M-tile waves of a 32-channel layer stage the
//    same rows twice (11 vector instructions per two values is cheap next to 90 matrix instructions per row) rather
//    than meet at a barrier per row.
#include <stdlib.h>

#include <type_traits>

#include "bmv_common.hpp"

namespace bmv {

using f32x4s = __attribute__((ext_vector_type(4))) float;
using i32x4s = __attribute__((ext_vector_type(4))) int;
using i32x2s = __attribute__((ext_vector_type(2))) int;
using bf16x8s = __attribute__((ext_vector_type(8))) __bf16;

struct Conv2dSArgs {
  const float* in;       // (B, Cin, H, W) planar fp32, or null when in_rec is given
  const int* in_rec;     // SPLIT RECORDS (B, Cin / 8, piece 3, H, W, 4): the 8 channels of an octet at a pixel as 8 bf16 = 16 bytes
  const int* wsplit;     // [M tile][filter row KS][step][piece 3][lane 64][4]
  const float* bias;     // (Cout)
  float* out;            // (B, Cout, Ho, Wo), or null
  int* out_rec;          // (B, Cout / 8, 3, Ho, Wo, 4) split records of the result, or null
  int B, H, W, Ho, Wo, Cout;
  float slope;
  int strips, tiles_y, nmt, ntiles;
};

__device__ __forceinline__ unsigned c2_pack_hi(float a, float b) {   // [bf16(a) | bf16(b) << 16] by truncation
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ unsigned c2_pack_rne(float a, float b) {  // round to nearest even (exact here: <= 8 bits left)
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float c2_trunc(float v) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u);
}

template <int I, int N, class F>
__device__ __forceinline__ void c2_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c2_static_for<I + 1, N>(f);
  }
}

// ablation builds (timing only, wrong results): 1 no matrix instructions, 2 no loads, 4 no split + LDS writes, 8 no stores
#ifndef BMV_C2S_ABLATE
#define BMV_C2S_ABLATE 0
#endif
constexpr int kC2Ablate = BMV_C2S_ABLATE;

constexpr int kC2_PH = 40;    // records per column parity of a staged row (80 columns; 40 x 16 bytes = 128 (mod 256):
                              // the even and the odd run of a stride-1 read fall on disjoint LDS banks)

// what the columns left and right of the image read in the split-record form (LDS-DMA has no range check to turn into zeros)
__device__ const i32x4s c2_zero_record = {0, 0, 0, 0};

// INREC: the input arrives as SPLIT RECORDS written by the producing layer's epilogue (once per value, instead of once per
// consuming wave: the M tiles and the halo rows of the strips repeat the split 1.4-2.8 x) and is staged by LDS-DMA
// (global_load_lds_dwordx4: 64 records per instruction, no registers, no vector instructions); the kernel is then the
// matrix instructions, their LDS reads and the stores (profiles/r6/conv2d_s_ablation.txt: 12.0 / 8.0 / 10.2 of the 20.8 /
// 18.0 / 21.6 us of the three layers).  Rows in flight: DIST ahead, DIST + 1 linear row buffers [octet][piece][POS].
template <int KS, int STR, int NOCT, int NT, int TY, int OCC, bool INREC>
__global__ void __launch_bounds__(256, OCC) conv2d_s_kernel(Conv2dSArgs a) {
  constexpr int PAD = KS / 2, XO = 2 - PAD;             // staged column 0 = input column STR X0 - 2
  constexpr int NP = STR * (TY - 1) + KS;               // input rows of a strip
  constexpr int NPAIR = NOCT * KS, NSTEP = (NPAIR + 3) / 4;
  constexpr int PH = kC2_PH;
  constexpr int TS = 8 * STR;                           // records between the reads of adjacent 16-pixel tiles
  constexpr int WIN = STR * (16 * NT - 1) + KS + XO;    // staged columns that are read
  static_assert(WIN <= 2 * PH, "strip window");
  // INREC: staged columns of a row (linear).  One DMA instruction when they fit 64 lanes (the lanes past POS are masked
  // off); else two FULL instructions and 128 columns: with the second one under `if (lane < 8)` hipcc 7.2 regrouped the
  // requests of consecutive planes across the divergent region and rows arrived incomplete
  constexpr int POS = WIN <= 64 ? (WIN + 7) / 8 * 8 : 128;
  constexpr int DIST = INREC ? (NOCT * STR >= 4 ? 1 : 2) : 1;   // INREC: rows requested ahead (the 16 -> 32 5x5 strips: 1, LDS)
  constexpr int NBUF = DIST + 1;
  constexpr int BUF = INREC ? NOCT * 3 * POS : NOCT * 3 * 2 * PH;   // records of one staged row: [octet][piece][parity][PH]
  extern __shared__ i32x4s c2_lds[];                    // [wave 4][buffer NBUF][BUF]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // consecutive workgroup ids go round-robin over the 8 XCDs: every XCD gets a contiguous run of strips = a band of rows
  // of the batch, the same band (as a fraction of the map) in every layer of the chain, so a layer reads what the
  // previous one left in ITS L2 (csrc/conv.hip's tile order; without it the chain was 18 us slower in the frame than
  // stand-alone timings promised)
  const int tile = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wave;
  if (tile >= a.ntiles) return;                         // (no barrier in this kernel: a wave may leave)
  const int mt = tile % a.nmt;
  int rest = tile / a.nmt;
  const int strip = rest % a.strips;
  rest /= a.strips;
  const int ty = rest % a.tiles_y, b = rest / a.tiles_y;
  const int X0 = strip * 16 * NT, Y0 = ty * TY;
  const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
  i32x4s* my = c2_lds + wave * (NBUF * BUF);
  const int n = lane & 15, kg = lane >> 4;

  // producer role: input columns gx, gx + 1
  const int gx = STR * X0 - 2 + 2 * lane;
  const bool xin = (gx >= 0) & (gx < W) & (2 * lane < WIN);
  const int voff = xin ? 4 * gx : (int)0x80000000u;
  __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (size_t)b * (NOCT * 8) * H * W), 0,
                                                                 (int)(4u * (unsigned)(NOCT * 8) * (unsigned)(H * W)), 0x00020000);
  // consumer role: record of (step, tile 0, piece 0) in a staged row
  int roff[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int pi = min(4 * s + kg, NPAIR - 1);          // (pairs past the end have zero weights: any finite data will do)
    const int oct = pi / KS, col = pi % KS + XO;
    const int c2 = STR == 2 ? col : n + col;
    if (INREC)
      roff[s] = oct * 3 * POS + STR * n + col;          // (linear rows: the stride-2 reads take two LDS passes)
    else
      roff[s] = (oct * 3 * 2 + (c2 & 1)) * PH + (STR == 2 ? n + (c2 >> 1) : (c2 >> 1));
  }
  constexpr int QS = INREC ? POS : 2 * PH;              // records between the pieces of an octet
  constexpr int TSR = INREC ? 16 * STR : TS;            // ... between adjacent 16-pixel tiles
  // A operands: [filter row][step][piece]
  i32x4s A[KS][NSTEP][3];
  {
    const i32x4s* __restrict__ wp = reinterpret_cast<const i32x4s*>(a.wsplit) + (size_t)mt * (KS * NSTEP * 3 * 64) + lane;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int st = 0; st < NSTEP; ++st)
#pragma unroll
        for (int q = 0; q < 3; ++q) A[ky][st][q] = wp[(size_t)((ky * NSTEP + st) * 3 + q) * 64];
  }
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[mt * 16 + 4 * kg + j];

  auto rowok = [&](int p) { return (STR * Y0 - PAD + p >= 0) & (STR * Y0 - PAD + p < H); };
  // input row p of the strip: the lane's two columns of every channel
  auto fetch = [&](int p, float (&t)[NOCT * 8][2]) {
    const int gy = STR * Y0 - PAD + p;
    const int dead = ((gy >= 0) & (gy < H)) ? 0 : (int)0x80000000u;
    const int gyc = min(max(gy, 0), H - 1);
    // (the elements go through named ints: `__builtin_bit_cast(float, v[e])` straight on an element of an ext-vector reads
    // element 0 whatever e is -- hipcc 7.2; the first build of this kernel staged every even column twice)
#pragma unroll
    for (int c = 0; c < NOCT * 8; ++c) {
      if (kC2Ablate & 2) {
        t[c][0] = (float)(c + lane + p), t[c][1] = (float)(c - lane);
        continue;
      }
      const i32x2s v = __builtin_bit_cast(i32x2s, __builtin_amdgcn_raw_buffer_load_b64(irs, voff | dead, 4 * ((c * H + gyc) * W), 0));
      const int v0 = v[0], v1 = v[1];
      t[c][0] = __builtin_bit_cast(float, v0), t[c][1] = __builtin_bit_cast(float, v1);
    }
  };
  // split the 8 channels of every octet at the lane's two columns into three bf16 pieces and park them in buffer `buf`
  auto park = [&](const float (&t)[NOCT * 8][2], int buf) {
    if (kC2Ablate & 4) {
      if (t[0][0] == 12345.f) my[lane] = i32x4s{1, 2, 3, 4};
      return;
    }
#pragma unroll
    for (int oct = 0; oct < NOCT; ++oct)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        i32x4s pc[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v0 = t[oct * 8 + 2 * i][e], v1 = t[oct * 8 + 2 * i + 1][e];
          pc[0][i] = (int)c2_pack_hi(v0, v1);
          const float r0 = v0 - c2_trunc(v0), r1 = v1 - c2_trunc(v1);
          pc[1][i] = (int)c2_pack_hi(r0, r1);
          pc[2][i] = (int)c2_pack_rne(r0 - c2_trunc(r0), r1 - c2_trunc(r1));
        }
        if (lane < PH) {
#pragma unroll
          for (int q = 0; q < 3; ++q) my[buf * BUF + ((oct * 3 + q) * 2 + e) * PH + lane] = pc[q];
        }
      }
  };

  f32x4s acc[TY][NT];
  // the matrix instructions of input row p (in buffer p & 1) for every output row it touches
  auto multiply = [&](auto pc) {
    constexpr int p = decltype(pc)::value;
    const i32x4s* bp = my + (p % NBUF) * BUF;
    c2_static_for<0, NSTEP * NT>([&](auto sc) {
      constexpr int s = decltype(sc)::value / NT, t = decltype(sc)::value % NT;
      bf16x8s bx[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) bx[q] = __builtin_bit_cast(bf16x8s, bp[roff[s] + q * QS + TSR * t]);
      c2_static_for<0, KS>([&](auto kc) {
        constexpr int ky = decltype(kc)::value;
        constexpr int zo = (p - ky) / STR;               // input row p = STR zo + ky
        if constexpr (p - ky >= 0 && (p - ky) % STR == 0 && zo < TY) {
          if (kC2Ablate & 1) {
            acc[zo][t] += __builtin_bit_cast(f32x4s, A[ky][s][0]) + __builtin_bit_cast(f32x4s, bx[ky % 3]);
            return;
          }
          // smallest terms first: (lo, hi), (mid, mid), (hi, lo), (mid, hi), (hi, mid), (hi, hi)
#pragma unroll
          for (int sum = 2; sum >= 0; --sum)
#pragma unroll
            for (int i = 0; i <= sum; ++i)
              acc[zo][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, A[ky][s][i]), bx[sum - i], acc[zo][t], 0, 0, 0);
        }
      });
    });
  };
  // lane (n, kg) holds output channels 16 mt + 4 kg + j of pixel (Y0 + zo, X0 + 16 t + n)
  auto store_row = [&](int zo) {
    const int y = Y0 + zo;
    if (y >= Ho) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int x = X0 + 16 * t + n;
      if (x >= Wo) continue;
      if ((kC2Ablate & 8) && acc[0][0][0] != 12345.f) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = acc[zo][t][j] + bs[j];
        v[j] = fmaxf(u, 0.f) + a.slope * fminf(u, 0.f);
      }
      if (a.out) {
        float* o = a.out + (((size_t)b * a.Cout + mt * 16 + 4 * kg) * Ho + y) * Wo + x;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[(size_t)j * Ho * Wo] = v[j];
      }
      if (a.out_rec) {
        // the lane's four channels are half h = kg & 1 of octet 2 mt + (kg >> 1): 8 bytes of each of the three records
        const int oct = 2 * mt + (kg >> 1);
        const float r0 = v[0] - c2_trunc(v[0]), r1 = v[1] - c2_trunc(v[1]), r2 = v[2] - c2_trunc(v[2]), r3 = v[3] - c2_trunc(v[3]);
        const i32x2s hi = {(int)c2_pack_hi(v[0], v[1]), (int)c2_pack_hi(v[2], v[3])};
        const i32x2s mid = {(int)c2_pack_hi(r0, r1), (int)c2_pack_hi(r2, r3)};
        const i32x2s lo = {(int)c2_pack_rne(r0 - c2_trunc(r0), r1 - c2_trunc(r1)), (int)c2_pack_rne(r2 - c2_trunc(r2), r3 - c2_trunc(r3))};
        const size_t plane = (size_t)Ho * Wo;
        i32x2s* o = reinterpret_cast<i32x2s*>(a.out_rec) + ((((size_t)b * (a.Cout >> 3) + oct) * 3) * plane + (size_t)y * Wo + x) * 2 + (kg & 1);
        o[0] = hi, o[2 * plane] = mid, o[4 * plane] = lo;
      }
    }
  };

  if constexpr (INREC) {
    // producer role: staged column 64 j + lane of the strip's window.  A column outside the image reads the zero record
    // (its address does not move with the row: multiplier 0)
    constexpr int NCH = (POS + 63) / 64;                // DMA instructions per (octet, piece, row)
    constexpr int DPR = NOCT * 3 * NCH;                 // ... per row
    const char* rbase = reinterpret_cast<const char*>(a.in_rec) + (size_t)b * (NOCT * 3) * H * W * 16;
    const char* cadr[NCH];
    unsigned cmul[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int col = 64 * j + lane, gxr = STR * X0 - 2 + col;
      const bool in = (gxr >= 0) & (gxr < W) & (col < WIN);
      cadr[j] = in ? rbase + (size_t)gxr * 16 : reinterpret_cast<const char*>(&c2_zero_record);
      cmul[j] = in ? 1u : 0u;
    }
    auto request = [&](int p) {
      const int gyc = min(max(STR * Y0 - PAD + p, 0), H - 1);      // (a row outside the image is requested too -- the counts of
      char* dst = reinterpret_cast<char*>(my + (p % NBUF) * BUF);  // the waits below are static -- and never multiplied)
#pragma unroll
      for (int o = 0; o < NOCT * 3; ++o) {
        const unsigned off = (unsigned)((o * H + gyc) * W) * 16u;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          if (POS >= 64 || lane < POS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cadr[j] + (size_t)cmul[j] * off),
                                             (__attribute__((address_space(3))) void*)(dst + (o * POS + 64 * j) * 16), 16, 0, 0);
        }
      }
    };
#pragma unroll
    for (int p = 0; p < DIST; ++p) request(p);
    c2_static_for<0, NP>([&](auto pc) {
      constexpr int p = decltype(pc)::value;
      if (p + DIST < NP) request(p + DIST);          // (into the buffer of row p - 1: its reads fed matrix instructions already issued)
      if constexpr (p % STR == 0 && p / STR < TY) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[p / STR][t] = f32x4s{0.f, 0.f, 0.f, 0.f};
      }
      // row p has landed: at most the requests of the rows behind it are outstanding (vector memory operations retire in
      // order; the stores issued meanwhile only make this wait for more than it needs)
      constexpr int younger = DPR * ((NP - 1 - p) < DIST ? (NP - 1 - p) : DIST);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger) : "memory");
      if (rowok(p)) multiply(pc);
      if (p >= KS - 1 && (p - (KS - 1)) % STR == 0) store_row((p - (KS - 1)) / STR);
      __builtin_amdgcn_sched_barrier(0);
    });
  } else {
  float nx[2][NOCT * 8][2];
  fetch(0, nx[0]);
  fetch(1, nx[1]);
  park(nx[0], 0);
  // (the row index is a compile-time constant of every iteration: `#pragma unroll` gave up on the 19 rows of the 5x5
  // strips and left the accumulators in scratch memory behind a run-time index)
  c2_static_for<0, NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    if (p + 2 < NP) fetch(p + 2, nx[p & 1]);       // (two rows ahead; row p's values were parked one iteration ago)
    if constexpr (p % STR == 0 && p / STR < TY) {  // the first row output row p / STR reads
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[p / STR][t] = f32x4s{0.f, 0.f, 0.f, 0.f};
    }
    if (rowok(p)) multiply(pc);
    if (p >= KS - 1 && (p - (KS - 1)) % STR == 0) store_row((p - (KS - 1)) / STR);   // its last contribution is in
    if (p + 1 < NP) park(nx[(p + 1) & 1], (p + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);             // (one row's loads, matrix instructions and vector work per region)
  });
  }
}

struct C2Shape {
  int ks, stride, noct, nmt, nstep;
};
inline bool c2_shape(int Cin, int Cout, int ks, int stride, C2Shape& s) {
  // the shapes of FeatureNet's encoder (what is instantiated below)
  if (ks == 5 && stride == 2) {
    if (Cin != 8 && Cin != 16) return false;
  } else if (ks == 3 && stride == 1) {
    if (Cin != 16 && Cin != 32) return false;
  } else {
    return false;
  }
  if (Cout != 16 && Cout != 32) return false;
  s.ks = ks, s.stride = stride, s.noct = Cin / 8, s.nmt = Cout / 16, s.nstep = (s.noct * ks + 3) / 4;
  return true;
}

template <int KS, int STR, int NOCT, int NT, int TY, int OCC, bool INREC>
void c2_launch(Conv2dSArgs& a, hipStream_t st) {
  a.strips = (a.Wo + 16 * NT - 1) / (16 * NT);
  a.tiles_y = (a.Ho + TY - 1) / TY;
  a.ntiles = a.B * a.tiles_y * a.strips * a.nmt;
  constexpr int WIN = STR * (16 * NT - 1) + KS + 2 - KS / 2, POS = WIN <= 64 ? (WIN + 7) / 8 * 8 : 128;
  constexpr int NBUF = INREC ? (NOCT * STR >= 4 ? 2 : 3) : 2;
  constexpr int BUF = INREC ? NOCT * 3 * POS : NOCT * 3 * 2 * kC2_PH;
  const size_t lds = (size_t)4 * NBUF * BUF * sizeof(i32x4s);
  auto kern = conv2d_s_kernel<KS, STR, NOCT, NT, TY, OCC, INREC>;
  static bool raised = false;
  if (lds > 64 * 1024 && !raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    raised = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.ntiles + 3) / 4), dim3(256), lds, st, a);
}

}  // namespace bmv

using namespace bmv;

extern "C" {

// int32 words of bmv_conv2d_s_fwd's split weights: [M tile Cout / 16][filter row ks][step][piece 3][lane 64][4]; 0: shape not covered
int bmv_conv2d_s_wsplit_ints(int Cin, int Cout, int ks, int stride) {
  C2Shape s;
  if (!c2_shape(Cin, Cout, ks, stride, s)) return 0;
  return s.nmt * ks * s.nstep * 3 * 64 * 4;
}

// act(conv2d(in; k = ks, stride, padding ks / 2) + bias) on the bf16 matrix cores with three-piece fp32 operands.
// Covered: (ks, stride, Cin) = (5, 2, 8 | 16), (3, 1, 16 | 32); Cout in {16, 32}; H and W even.  The input is `in`
// (B,Cin,H,W) planar fp32 or `in_records` (B,Cin/8,3,H,W,4) split records (the other one null); the result goes to `out`
// (B,Cout,H/stride,W/stride) planar fp32 and / or `out_records` (B,Cout/8,3,H/stride,W/stride,4) (either may be null, not
// both).  wsplit: boostmvsnerfs_amd/convnet.py pack_conv2d_s.
int bmv_conv2d_s_fwd(const float* in, const int* in_records, const int* wsplit, const float* bias, float* out, int* out_records,
                     int B, int Cin, int H, int W, int Cout, int ks, int stride, float act_slope, bmv_stream_t stream) {
  BMV_REQUIRE((in != nullptr) != (in_records != nullptr), "conv2d_s: exactly one of in / in_records");
  BMV_REQUIRE(wsplit && bias && (out || out_records), "conv2d_s: null pointer");
  C2Shape s;
  if (!c2_shape(Cin, Cout, ks, stride, s) || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) ||
      (size_t)Cin * H * W * 6 >= ((size_t)1 << 31)) {
    set_error("bmv_conv2d_s_fwd: shape not covered (Cin=%d Cout=%d ks=%d stride=%d H=%d W=%d)", Cin, Cout, ks, stride, H, W);
    return BMV_ERR_UNSUPPORTED;
  }
  Conv2dSArgs a;
  a.in = in, a.in_rec = in_records, a.wsplit = wsplit, a.bias = bias, a.out = out, a.out_rec = out_records;
  a.B = B, a.H = H, a.W = W, a.Ho = H / stride, a.Wo = W / stride, a.Cout = Cout, a.slope = act_slope, a.nmt = s.nmt;
  hipStream_t st = as_stream(stream);
  // rows per wave (BMV_CONV2D_S_ROWS: 4 or 8): 8 walks 19 % (5x5 stride 2) / 25 % (3x3) of halo rows, 4 twice the waves
  int rows = bmv::tuning("BMV_CONV2D_S_ROWS", 0);
  if (rows != 4 && rows != 8) {
    const long waves8 = (long)B * ((a.Ho + 7) / 8) * ((a.Wo + 31) / 32) * s.nmt;
    rows = waves8 >= 1536 ? 8 : 4;
  }
  const bool rec = in_records != nullptr;
#define C2(KS_, STR_, NOCT_, OCC_)                                                                  \
  if (ks == KS_ && stride == STR_ && s.noct == NOCT_) {                                             \
    if (rec) {                                                                                      \
      if (rows == 8) c2_launch<KS_, STR_, NOCT_, 2, 8, OCC_, true>(a, st);                          \
      else c2_launch<KS_, STR_, NOCT_, 2, 4, OCC_, true>(a, st);                                    \
    } else {                                                                                        \
      if (rows == 8) c2_launch<KS_, STR_, NOCT_, 2, 8, OCC_, false>(a, st);                         \
      else c2_launch<KS_, STR_, NOCT_, 2, 4, OCC_, false>(a, st);                                   \
    }                                                                                               \
    BMV_LAUNCH_END("bmv_conv2d_s_fwd");                                                             \
  }
  C2(5, 2, 1, 2)
  C2(5, 2, 2, 1)
  C2(3, 1, 2, 2)
  C2(3, 1, 4, 2)
#undef C2
  set_error("bmv_conv2d_s_fwd: no instantiation");
  return BMV_ERR_UNSUPPORTED;
}

}  // extern "C"
