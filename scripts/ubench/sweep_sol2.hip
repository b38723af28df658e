// Speed-of-light microbenchmark, part 2: access-pattern floors of ALTERNATIVE decompositions of the level-1 plane sweep
// (volume (16, 8, 256, 320), source 3 x 256 x 320 x 16 channels), no arithmetic beyond adds.
//
//   hipcc --offload-arch=gfx950 -O3 -o sweep_sol2 sweep_sol2.hip && ./sweep_sol2
//
// sweep_sol.hip showed: the windowed kernel's own pattern (64-byte records, one window per (tile, plane, view)) costs
// 19.6 us with NO arithmetic and NO skeleton -- fills 7.5 us (197 MB through the per-CU vector memory path at 42 B/clk/CU),
// tap reads, stores and the 4.2 us of the dispatch bracket ADD UP, with or without double buffering.  The fill volume is
// the lever: 1.33-1.5 records per voxel and view.
//
// Here: source in QUAD-PLANAR layout (S, C/4, Hs, Ws, 4): a record is 16 bytes (4 channels), a window of 34 x 10 records
// is 5.4 KB, so that the windows of ALL THREE views of a channel quad are resident together (16 KB) and a workgroup can
// walk several PLANES on one union window (planes of a pixel tile see the source shifted by the parallax: ~2 texels per
// plane on the frame's hypotheses).  A workgroup = one 32 x 8 pixel tile x PLW planes x QPW channel quads; the planes are
// walked in groups of PG that share one union window per view.
//   fill bytes per voxel-plane-view = (34 + SX (PG-1)) (10 + SY (PG-1)) / (256 PG) records of 64 B (all quads)
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int W = 320, H = 256, D = 8, C = 16, S = 3, Hs = 256, Ws = 320, TXW = 32, TYH = 8;
constexpr int SX = 2, SY = 2;   // parallax per plane, texels (the frame: 13 x 18 over 8 planes)

template <int PG, int PLW, int QPW, int FLAGS>
__global__ void __launch_bounds__(256) qu_kernel(const float* __restrict__ feats, float* __restrict__ out) {
  constexpr bool FILL = FLAGS & 1, READ = FLAGS & 2, STORE = FLAGS & 4, FILL_FIRST = FLAGS & 8;
  constexpr int WC = TXW + 2 + SX * (PG - 1), WR = TYH + 2 + SY * (PG - 1), NTEX = WC * WR, NP = (NTEX + 63) / 64;
  constexpr int CAPB = NP * 1024;   // bytes of one view's window
  extern __shared__ __attribute__((aligned(64))) char win[];
  // grid.x = 8 bands x plane blocks x quad blocks, grid.y = tile columns x tile rows of a band
  constexpr int TILES_X = W / TXW, TYB = (H / TYH) / 8;
  const int band = blockIdx.x & 7, kx = blockIdx.x >> 3;
  const int qb = kx % (4 / QPW), pb = kx / (4 / QPW);
  const int j = blockIdx.y / TILES_X, tx = blockIdx.y - j * TILES_X, ty = band * TYB + j;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = tid % TXW, ly = tid / TXW;
  const int x = tx * TXW + lx, y = ty * TYH + ly;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feats), 0, (int)((size_t)S * Hs * Ws * C * 4), 0x00020000);
  const size_t hw = (size_t)H * W;
  const unsigned cstride = (unsigned)(D * hw) * 4u;
  __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)C * D * hw * 4), 0x00020000);

  auto fill = [&](int q, int p0) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int x0 = tx * TXW - 1 + (s - 1) * (2 + p0), y0 = ty * TYH - 1 + (s - 1);
      for (int p = wave; p < NP; p += 4) {
        const int L = p * 64 + lane;
        const int row = L / WC, col = L - row * WC;
        const int gy = y0 + row, gx = x0 + col;
        const bool ok = ((unsigned)gy < (unsigned)Hs) & ((unsigned)gx < (unsigned)Ws) & (L < NTEX);
        const unsigned off = ((unsigned)(s * 4 + q) * (unsigned)(Hs * Ws) + (unsigned)(gy * Ws + gx)) * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(win + s * CAPB + p * 1024), 16,
                                                 (int)(ok ? off : 0x80000000u), 0, 0, 0);
      }
    }
  };
  float4 acc = {0, 0, 0, 0};
  auto read_plane = [&](int pl) {
    acc = {0, 0, 0, 0};
    const unsigned rec = (unsigned)((ly + 1 + pl * SY) * WC + lx + 1 + pl * SX);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const char* base = win + s * CAPB + (rec << 4);
      const float4 t00 = *(const float4*)(base), t01 = *(const float4*)(base + 16), t10 = *(const float4*)(base + WC * 16),
                   t11 = *(const float4*)(base + WC * 16 + 16);
      acc.x += (t00.x + t01.x) + (t10.x + t11.x), acc.y += (t00.y + t01.y) + (t10.y + t11.y);
      acc.z += (t00.z + t01.z) + (t10.z + t11.z), acc.w += (t00.w + t01.w) + (t10.w + t11.w);
    }
  };
  auto store_plane = [&](int q, int d) {
    const unsigned voff = (unsigned)((size_t)d * hw + (size_t)y * W + x) * 4u;
    unsigned soff = (unsigned)(q * 4) * cstride;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc.x), orsrc, (int)voff, (int)soff, 2);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc.y), orsrc, (int)voff, (int)(soff + cstride), 2);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc.z), orsrc, (int)voff, (int)(soff + 2 * cstride), 2);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc.w), orsrc, (int)voff, (int)(soff + 3 * cstride), 2);
  };

  bool first = true;
  for (int qq = 0; qq < QPW; ++qq) {
    const int q = qb * QPW + qq;
    for (int g = 0; g < PLW / PG; ++g) {
      const int p0 = pb * PLW + g * PG;
      if (!first) __syncthreads();   // every wave is done with the previous windows
      first = false;
      if (FILL) fill(q, p0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int pl = 0; pl < PG; ++pl) {
        if (READ) read_plane(pl);
        if (STORE) store_plane(q, p0 + pl);
      }
    }
  }
  if (!STORE && READ && acc.x == 1234.5f) out[0] = acc.x;
}

template <int PG, int PLW, int QPW, int FLAGS>
float run(const float* feats, std::vector<float*>& outs, int iters) {
  constexpr int WC = TXW + 2 + SX * (PG - 1), WR = TYH + 2 + SY * (PG - 1), NTEX = WC * WR, NP = (NTEX + 63) / 64;
  auto kern = qu_kernel<PG, PLW, QPW, FLAGS>;
  const size_t lds = (size_t)3 * NP * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(8u * (D / PLW) * (4 / QPW), (W / TXW) * ((H / TYH) / 8), 1), block(256);
  std::vector<hipEvent_t> ev(2 * iters);
  for (auto& e : ev) hipEventCreate(&e);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, grid, block, lds, 0, feats, outs[i % outs.size()]);
  hipDeviceSynchronize();
  for (int i = 0; i < iters; ++i) hipExtLaunchKernelGGL(kern, grid, block, lds, 0, ev[2 * i], ev[2 * i + 1], 0, feats, outs[i % outs.size()]);
  hipDeviceSynchronize();
  double sum = 0;
  float mn = 1e9f;
  for (int i = 0; i < iters; ++i) {
    float ms;
    hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
    sum += ms, mn = ms < mn ? ms : mn;
  }
  for (auto& e : ev) hipEventDestroy(e);
  if (hipGetLastError() != hipSuccess) printf("  (launch error)\n");
  const double alg = (double)S * Hs * Ws * C * 4 + (double)C * D * H * W * 4;
  const float t = (float)(sum / iters * 1e3);
  printf("  PG %d  planes/WG %d  quads/WG %d  flags %2d  window %2d x %2d  fill %.2f rec/voxel-plane  lds %6zu B  WGs %5u  avg %6.2f us  min %6.2f us  -> %.3f of 8 TB/s\n",
         PG, PLW, QPW, FLAGS, WC, WR, (double)NTEX / (256.0 * PG), lds, grid.x * grid.y, t, mn * 1e3, alg / t / 1e6 / 8.0);
  return t;
}

template <int PG, int PLW, int QPW>
void combo(const float* feats, std::vector<float*>& outs) {
  const int it = 40;
  run<PG, PLW, QPW, 0>(feats, outs, it);
  run<PG, PLW, QPW, 1>(feats, outs, it);
  run<PG, PLW, QPW, 4>(feats, outs, it);
  run<PG, PLW, QPW, 3>(feats, outs, it);
  run<PG, PLW, QPW, 5>(feats, outs, it);
  run<PG, PLW, QPW, 7>(feats, outs, it);
}

int main(int argc, char** argv) {
  const int nout = argc > 1 ? atoi(argv[1]) : 8;
  float* feats;
  const size_t fbytes = (size_t)S * Hs * Ws * C * 4, obytes = (size_t)C * D * H * W * 4;
  hipMalloc(&feats, fbytes);
  hipMemset(feats, 0x3c, fbytes);
  std::vector<float*> outs(nout);
  for (auto& o : outs) hipMalloc(&o, obytes), hipMemset(o, 0, obytes);
  printf("flags: 1 fills, 2 tap reads, 4 stores\n");
  combo<1, 1, 4>(feats, outs);   // today's decomposition on 16-byte records: one plane, all four quads in turn
  combo<1, 1, 1>(feats, outs);   // ... one quad per workgroup (10240 workgroups)
  combo<1, 8, 1>(feats, outs);   // all 8 planes of a quad in one workgroup, a window per plane
  combo<2, 8, 1>(feats, outs);   // plane pairs on a union window
  combo<4, 8, 1>(feats, outs);
  combo<8, 8, 1>(feats, outs);
  combo<4, 8, 4>(feats, outs);   // 1280 workgroups: a tile's whole column of voxels
  combo<4, 4, 2>(feats, outs);
  combo<2, 2, 4>(feats, outs);
  return 0;
}
