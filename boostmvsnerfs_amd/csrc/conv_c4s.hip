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
//   * INPUT-STATIONARY over z (v2): a workgroup owns 16 x 16 outputs of TZ consecutive planes and walks the TZ + 2 input
//     planes once per octet; a staged plane serves its (up to) three output planes from the SAME B operands (a B operand
//     depends on the position and the in-plane tap slot only, the kz weights differ): one ds_read_b128 per piece feeds 18
//     matrix instructions, the tile's halo costs (TZ + 2) / TZ x 1.27 staged positions per output instead of 3.8;
//   * the A operands of an octet -- 3 kz x 3 steps x 3 pieces = 27 registers quads, split and laid out in lane order on the
//     host -- are loaded ONCE per octet and workgroup and stay in registers for all its planes (v1 re-read them from L1 for
//     every plane: 27 KB per wave and 108 matrix instructions = the L1's whole bandwidth; it gained only 50 -> 43 us);
//   * planes are staged through a double-buffered LDS plane (2 x 3 x 324 records): the loads of plane p + 2 are in flight
//     and plane p + 1 is split and written while plane p is multiplied; one barrier per plane; planes outside the volume
//     are skipped altogether (wave-uniform).
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
  float* out;           // mode 8: quad records (B, Cout/4, D, H, W, 4); mode 2: the renderer's volume records (B, D, H, W, 8)
                        // of channels 0..7 (16-byte stores either way; a planar result is a view of these)
  float* out2;          // mode 2: channel 8 (the depth logits), planar (B, D, H, W)
  int B, Cin, D, H, W, Cout;
  float slope;          // activation: v > 0 ? v : slope * v
  int mode;
};

// ablation builds (scripts/ablate_conv_c4s.py: BMV_C4S_DEFS=-DBMV_C4S_ABLATE=n; timing only, wrong results): 1 no matrix
// instructions, 2 no plane loads, 4 no split / LDS writes, 8 no A-operand loads, 16 no stores, 32 no per-plane barrier
#ifndef BMV_C4S_ABLATE
#define BMV_C4S_ABLATE 0
#endif
constexpr int kC4SAblate = BMV_C4S_ABLATE;

constexpr int kS_RS = 18;   // row pitch of a staged plane: 16 outputs + halo

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

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() carries a workgroup-scope fence = s_waitcnt
// vmcnt(0): it would drain the plane loads issued two iterations ahead and the early stores at EVERY plane -- the first
// build did, and its matrix time and memory time simply added up (profiles/r6/conv_c4s_ablation.txt)
__device__ __forceinline__ void c4s_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// PAIR: Cout == 8, matrix rows = (output row y + (m >> 3), channel m & 7), in-plane slots t = (j in 0..3, kx) = (t / 3, t % 3);
// else rows = channel m (Cout <= 16), slots t < 9 = (ky, kx), slots 9..11 carry zero weights.
// A workgroup = 4 waves = 16 x by 4 RW y outputs of TZ planes; a wave = RW rows (4 or 2) = RW / 2 row pairs (PAIR) / RW rows.
// RW = 2 halves the tile (twice the workgroups: the level-0 volume is only 4 x 5 tiles of 16 x 16 per plane) at 1.4
// instead of 1.27 staged positions per output.
template <bool PAIR, int TZ, int RW>
__global__ void __launch_bounds__(256, ((PAIR && RW == 2) ? 3 : 2)) conv_c4s_kernel(C4SArgs a) {   // (paired, RW 2: 168 registers = 3 workgroups per CU)
  static_assert(TZ % 2 == 0, "the plane buffers alternate: an octet's TZ + 2 planes must be an even count");
  static_assert(RW == 2 || RW == 4, "rows per wave");
  constexpr int NQ = PAIR ? RW / 2 : RW, NP = TZ + 2, TY = 4 * RW;
  constexpr int kS_PPOS = (TY + 2) * kS_RS, kS_NSLOT = (kS_PPOS + 255) / 256;     // one input plane of a 16 x TY tile
  extern __shared__ i32x4s s_part[];                                // [2 buffers][3 pieces][kS_PPOS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kk = lane >> 4;
  const int ntx = (a.W + 15) >> 4, nty = (a.H + TY - 1) / TY, ntz = (a.D + TZ - 1) / TZ;
  int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int tx = bid % ntx;
  bid /= ntx;
  const int ty = bid % nty;
  bid /= nty;
  const int tz = bid % ntz, b = bid / ntz;
  const int x0 = tx * 16, y0 = ty * TY, z0 = tz * TZ;
  const int hw = a.H * a.W, plane = a.D * hw;
  const int nocts = a.Cin >> 3;

  // plane slots of this thread (a slot = one position of the 18 x 18 input box of a plane): byte offset of its record
  // inside the plane, or out of range (zero padding)
  unsigned goff[kS_NSLOT];
#pragma unroll
  for (int j = 0; j < kS_NSLOT; ++j) {
    const int slot = tid + 256 * j;
    const int sx = slot % kS_RS, sy = slot / kS_RS;
    const int gx = x0 - 1 + sx, gy = y0 - 1 + sy;
    const bool ok = (slot < kS_PPOS) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H);
    goff[j] = ok ? 16u * (unsigned)(gy * a.W + gx) : 0x80000000u;
  }
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + (size_t)b * a.Cin * plane), 0, (int)(4u * (unsigned)(a.Cin * plane)), 0x00020000);
  // input plane p of the tile is volume plane z0 - 1 + p; planes outside the volume are neither staged nor multiplied
  auto zin = [&](int p) { return z0 - 1 + p; };
  auto zok = [&](int p) { return zin(p) >= 0 && zin(p) < a.D; };
  // plane k of the linearised (octet, plane) sequence is loaded into pre[k & 1] (NP is even: k & 1 == p & 1) TWO
  // iterations before it is split and written: its loads are in flight under two planes' matrix instructions
  // Loads and writes are UNCONDITIONAL (a plane outside the volume / past the last octet is requested out of range: zeros
  // come back, are split and written, and nobody reads them): with `if (plane exists)` around both, the compiler could
  // not prove that a load it had issued was consumed before its registers were reused and put s_waitcnt vmcnt(0) in
  // front of every iteration's loads -- which drained the plane in flight AND the early stores (ISA of the first build).
  f32x4s pre[2][kS_NSLOT][2];
  auto load_plane = [&](int oct, int p, bool exists) {
    if (kC4SAblate & 2) return;
    const unsigned dead = (exists && zok(p)) ? 0u : 0x80000000u;
    const unsigned cb = 16u * (unsigned)(2 * min(oct, nocts - 1) * plane + min(max(zin(p), 0), a.D - 1) * hw);
#pragma unroll
    for (int j = 0; j < kS_NSLOT; ++j) {      // (every lane issues every slot's loads -- slots past the box are out of range --
      const unsigned o = (goff[j] + cb) | dead;   //  so that the count of loads in flight is the same on every path)
      pre[p & 1][j][0] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0));
      pre[p & 1][j][1] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + 16u * (unsigned)plane, 0, 0));
    }
  };
  auto write_plane = [&](int p) {   // split pre[p & 1] (plane p) into three bf16 pieces, once per staged value -> buffer p & 1
    if (kC4SAblate & 4) return;
#pragma unroll
    for (int j = 0; j < kS_NSLOT; ++j) {
      if ((j + 1) * 256 > kS_PPOS && tid + 256 * j >= kS_PPOS) continue;
      i32x4s pc[3];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float v0 = pre[p & 1][j][h][2 * e], v1 = pre[p & 1][j][h][2 * e + 1];
          pc[0][2 * h + e] = (int)s_pack_hi(v0, v1);
          const float r0 = v0 - s_trunc(v0), r1 = v1 - s_trunc(v1);
          pc[1][2 * h + e] = (int)s_pack_hi(r0, r1);
          pc[2][2 * h + e] = (int)s_pack_rne(r0 - s_trunc(r0), r1 - s_trunc(r1));
        }
#pragma unroll
      for (int q = 0; q < 3; ++q) s_part[((p & 1) * 3 + q) * kS_PPOS + tid + 256 * j] = pc[q];
    }
  };

  // per-lane tap offsets (positions inside a plane) of the 3 steps: slot t = 4 g + kk
  int tapoff[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int t = 4 * g + kk;
    tapoff[g] = PAIR ? (t / 3) * kS_RS + t % 3 : (t < 9 ? (t / 3) * kS_RS + t % 3 : 0);
  }
  // origins of this wave's 16-wide output pieces: PAIR: row pairs (4 wave + 2 q, + 1); else rows 4 wave + q
  int pbase[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) pbase[q] = (RW * wave + (PAIR ? 2 * q : q)) * kS_RS + n;

  f32x4s acc[TZ][NQ];
#pragma unroll
  for (int z = 0; z < TZ; ++z)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[z][q] = f32x4s{0.f, 0.f, 0.f, 0.f};

  // epilogue of ONE output plane: accumulator j of lane (n, kk) = matrix row 4 kk + j at x = n of the piece.  Issued as
  // soon as the plane's last contribution is in (the last octet's iteration zo + 2): the stores drain under the
  // matrix instructions of the planes that follow instead of in one tail at the end of the workgroup.  Buffer stores:
  // one per-lane byte offset per piece (computed once; lanes outside the volume / without a channel: out of range,
  // dropped by the bounds check), the plane's offset rides in the scalar offset.
  const int cg = PAIR ? (kk & 1) : kk;                 // group of 4 output channels this lane holds
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] = a.bias[4 * cg + j];
  // bytes per voxel of the lane's store and the lane's base inside a voxel's record / the channel block
  const unsigned vb = a.mode == 2 ? 32u : 16u;
  __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out + (size_t)b * (a.mode == 2 ? 8 : a.Cout) * plane, 0,
      (int)(4u * (unsigned)((a.mode == 2 ? 8 : a.Cout) * plane)), 0x00020000);
  __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(
      (a.mode == 2 && a.out2) ? a.out2 + (size_t)b * plane : a.out, 0, (a.mode == 2 && a.out2) ? (int)(4u * (unsigned)plane) : 0, 0x00020000);
  unsigned voff[NQ], doff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int y = y0 + RW * wave + (PAIR ? 2 * q + (kk >> 1) : q), x = x0 + n;
    const bool in = (x < a.W) & (y < a.H) & !((kC4SAblate & 16) != 0);
    const unsigned pix = (unsigned)(y * a.W + x);
    const bool has = a.mode == 2 ? cg < 2 : 4 * cg < a.Cout;
    voff[q] = (in && has) ? pix * vb + (a.mode == 2 ? 16u * (unsigned)cg : 16u * (unsigned)(cg * plane)) : 0x80000000u;
    doff[q] = (in && a.mode == 2 && cg == 2) ? pix * 4u : 0x80000000u;
  }
  auto store_plane = [&](int zo) {
    const int z = z0 + zo;
    if (z >= a.D) return;
    const unsigned so = (unsigned)(z * hw) * vb;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4s v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = acc[zo][q][j] + bs[j];
        v[j] = fmaxf(t, 0.f) + a.slope * fminf(t, 0.f);
      }
      {                           // quad records / the renderer's volume records: one 16-byte store
        // s_nop behind the store, tied to its data registers: with the plane offset in an SGPR hipcc 7.2 issues the next
        // piece's first v_add_f32 right behind this store and overwrites element 1 of its data before the store has read
        // it -- LLVM's hazard recogniser assumes that a buffer store of more than 8 bytes WITH a register soffset has no
        // VALU-write-data hazard; on gfx950 it has (intermittently wrong channels 1 / 5 in groups of four x, found by the
        // tiling test; with an immediate soffset the recogniser inserts the wait state itself, at the price of a vector
        // add per store and three spilled registers in the 168-register build).  ISA: profiles/r6/conv_c4s_notes.txt.
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4s, v), orsrc, (int)voff[q], (int)so, 0);
        asm volatile("s_nop 1" : "+v"(v));
        if (!PAIR && a.mode == 2)   // ... + the planar depth logits (channel 8 = row 0 of group 2)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[0]), drsrc, (int)doff[q], (int)((unsigned)(z * hw) * 4u), 0);
      }
    }
  };

  // pipeline: plane (oct, p) lives in LDS buffer p & 1.  Iteration (oct, p): the loads of the plane two ahead go out, plane
  // p is multiplied, the plane one ahead (loaded during the previous iteration) is split and written, barrier.
  const i32x4s* __restrict__ wp = reinterpret_cast<const i32x4s*>(a.wsplit) + lane;
  i32x4s A[3][3][3];      // [step][kz][piece]: an octet's A operands, in registers for all planes of the tile
  auto load_A = [&](int oct) {
#pragma unroll
    for (int i = 0; i < 27; ++i)
      A[i / 9][(i / 3) % 3][i % 3] = (kC4SAblate & 8) ? i32x4s{i + oct, lane, i, 1} : wp[((size_t)oct * 27 + i) * 64];
  };
  load_plane(0, 0, true);
  load_A(0);              // (in use order; in flight under the first planes' staging)
  load_plane(0, 1, true);
  write_plane(0);
  c4s_barrier();
  for (int oct = 0; oct < nocts; ++oct) {
    if (oct > 0) load_A(oct);
    const bool more = oct + 1 < nocts;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p + 2 < NP) load_plane(oct, p + 2, true);
      else load_plane(oct + 1, p + 2 - NP, more);
      if (zok(p)) {
        const i32x4s* bp = s_part + (p & 1) * 3 * kS_PPOS;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int pos = pbase[q] + tapoff[g];
            bf16x8s bx[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) bx[r] = __builtin_bit_cast(bf16x8s, bp[r * kS_PPOS + pos]);
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) {
              const int zo = p - kz;             // input plane p = output plane zo + kz (known after unrolling)
              if (zo < 0 || zo >= TZ) continue;
              if (kC4SAblate & 1) {
                acc[zo][q] += __builtin_bit_cast(f32x4s, A[g][kz][0]) + __builtin_bit_cast(f32x4s, bx[kz]);
                continue;
              }
              // smallest terms first: (lo, hi), (mid, mid), (hi, lo), (mid, hi), (hi, mid), (hi, hi)
#pragma unroll
              for (int sum = 2; sum >= 0; --sum)
#pragma unroll
                for (int i = 0; i <= sum; ++i)
                  acc[zo][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, A[g][kz][i]), bx[sum - i],
                                                                       acc[zo][q], 0, 0, 0);
            }
          }
      }
      if (!more && p >= 2) store_plane(p - 2);      // output plane p - 2 has its last contribution
      write_plane(p + 1 < NP ? p + 1 : 0);
      if (!(kC4SAblate & 32)) c4s_barrier();
    }
  }
}

template <bool PAIR, int TZ, int RW>
static void c4s_launch(const C4SArgs& a, hipStream_t st) {
  constexpr int TY = 4 * RW;
  constexpr size_t lds = (size_t)2 * 3 * (TY + 2) * kS_RS * sizeof(i32x4s);
  static_assert(lds <= 64 * 1024, "dynamic LDS within the default limit: no attribute call");
  const unsigned grid = (unsigned)(((a.W + 15) / 16) * ((a.H + TY - 1) / TY) * ((a.D + TZ - 1) / TZ) * a.B);
  hipLaunchKernelGGL((conv_c4s_kernel<PAIR, TZ, RW>), dim3(grid), dim3(256), lds, st, a);
}

}  // namespace bmv

using namespace bmv;

extern "C" {

// int32 words of the split weights of bmv_conv_c4s_fwd: [octet][step 3][kz 3][piece 3][lane 64][4] (the same count for the
// row-paired and the plain form)
int bmv_conv_c4s_wsplit_ints(int Cin, int pair) {
  (void)pair;
  if (Cin < 8 || (Cin & 7)) {
    set_error("bmv_conv_c4s_wsplit_ints: Cin=%d must be a multiple of 8", Cin);
    return BMV_ERR_UNSUPPORTED;
  }
  return (Cin / 8) * 27 * 64 * 4;
}

// 3x3x3 convolution, stride 1, padding 1, on the bf16 matrix cores with three-piece fp32 operands (fp32 accuracy):
// in = quad records (B, Cin/4, D, H, W, 4); out by mode (8 quad records, 2 volume records + out2 = channel 8);
// pair != 0: Cout == 8, weights packed row-paired (boostmvsnerfs_amd/convnet.py pack_conv_c4s)
int bmv_conv_c4s_fwd(const float* in, const int* wsplit, const float* bias, float* out, float* out2, int B, int Cin, int D,
                     int H, int W, int Cout, int pair, float slope, int mode, bmv_stream_t stream) {
  pair = pair ? 1 : 0;
  BMV_REQUIRE(in && wsplit && bias && out, "bmv_conv_c4s_fwd: null pointer");
  BMV_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "bmv_conv_c4s_fwd: bad shape");
  BMV_REQUIRE((mode == 2 && (Cout == 8 || (Cout == 9 && out2))) || (mode == 8 && Cout % 4 == 0),
              "bmv_conv_c4s_fwd: mode=%d with Cout=%d (2: volume records, Cout 8 | 9; 8: quad records, Cout %% 4 == 0)", mode, Cout);
  if (Cin < 8 || (Cin & 7) || Cout < 1 || Cout > 16 || (pair && Cout != 8) ||
      (size_t)Cin * D * H * W * 4 >= ((size_t)1 << 31) || (size_t)Cout * D * H * W * 4 >= ((size_t)1 << 32)) {
    set_error("bmv_conv_c4s_fwd: shape not covered (Cin=%d (%%8), Cout=%d (<= 16; 8 when paired), %dx%dx%d)", Cin, Cout, D, H, W);
    return BMV_ERR_UNSUPPORTED;
  }
  C4SArgs a{in, wsplit, bias, out, out2, B, Cin, D, H, W, Cout, slope, mode};
  hipStream_t st = as_stream(stream);
  // tile: rows per wave RW (tile height 4 RW) and planes per workgroup TZ; BMV_CONV_C4S_TZ / BMV_CONV_C4S_RW force one.
  // Default 16 x 8 outputs x 4 planes: measured on the frame's layers (profiles/r6/conv_c4s_tilings.txt) it wins or ties
  // everywhere -- the paired kernel fits 168 registers with it (3 workgroups per CU instead of 2), and the level-0
  // volume (4 x 5 tiles of 16 x 16 per plane) needs the small tile to fill the chip at all.
  int tzv = bmv::tuning("BMV_CONV_C4S_TZ", 0), rwv = bmv::tuning("BMV_CONV_C4S_RW", 0);
  if (rwv != 2 && rwv != 4) rwv = 2;
  if (tzv != 2 && tzv != 4) tzv = D >= 3 ? 4 : 2;
  if (!pair && rwv == 4 && tzv == 4) tzv = 2;
#define C4S_CASE(P, TZV, RWV) \
  if (pair == P && tzv == TZV && rwv == RWV) c4s_launch<P != 0, TZV, RWV>(a, st)
  C4S_CASE(1, 4, 4);
  C4S_CASE(1, 2, 4);
  C4S_CASE(1, 4, 2);
  C4S_CASE(1, 2, 2);
  C4S_CASE(0, 2, 4);
  C4S_CASE(0, 4, 2);
  C4S_CASE(0, 2, 2);
#undef C4S_CASE
  BMV_LAUNCH_END("bmv_conv_c4s_fwd");
}

}  // extern "C"
