// Speed-of-light microbenchmark for the plane sweep's ACCESS PATTERN: the windowed kernel's
// grid, tile mapping, LDS-DMA window fills, LDS tap reads and variance stores with (almost) no arithmetic, so that the
// distance between the real kernel and what this decomposition can reach at all is attributable.
//
//   hipcc --offload-arch=gfx950 -O3 -o sweep_sol sweep_sol.hip && ./sweep_sol
//
// Level 1 of BASELINE config 2: volume (C 16, D 8, 256 x 320) = 41.9 MB written, source (3, 256, 320, 16) channel-last
// = 15.7 MB read; tile 32 x 8 pixels x 1 plane x 16 channels (256 threads, lane = voxel), 2560 workgroups.
// Level 0: volume (C 32, D 64, 64 x 80), source (3, 128, 160, 32); tile 16 x 2 pixels x 8 planes x 16 channels.
// Variants (bit flags): 1 fills (3 views, one window each, barrier-separated like the real kernel), 2 LDS tap reads
// (16 ds_read_b128 per voxel and view + 12 adds per slice), 4 stores (16 dwords per voxel, scalar channel offsets),
// 8 stores as dwordx4 (a wave writes whole 8 x 128-byte tiles of one channel), 16 all three fills issued up front into
// three windows (no barrier between views; LDS x 3), 32 units of 8 channels (32-byte records) with the fill of unit u + 1 in
// flight under the reads of unit u (two buffers in the LDS of one 16-channel window, one barrier per unit), 64 the same
// with 16-channel units (LDS x 2).
// Times: dispatch-bound events (hipExtLaunchKernelGGL start/stop = the kernel's begin and end) averaged over launches
// that rotate over 8 output volumes (336 MB > the 256 MB Infinity Cache: the stores have to drain to HBM).
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct Geo {
  int C, D, h, w, Hs, Ws, TXW, TYH, DP, chalves;
  int wc, wr;       // window columns / rows
};

template <int TXW, int TYH, int DP, int FLAGS>
__global__ void __launch_bounds__(TXW* TYH* DP) sol_kernel(const float* __restrict__ feats, float* __restrict__ out, Geo g, int tiles_x, int tyb,
                                                          int cap) {
  constexpr int NT = TXW * TYH * DP, NW = NT / 64;
  constexpr bool FILL = FLAGS & 1, READ = FLAGS & 2, STORE = FLAGS & 4, STORE4 = FLAGS & 8, UPFRONT = FLAGS & 16;
  constexpr bool DBUF_HALF = FLAGS & 32, DBUF_FULL = FLAGS & 64;
  extern __shared__ __attribute__((aligned(64))) char win[];
  const int band = blockIdx.x & 7, kx = blockIdx.x >> 3;
  const int chh = kx % g.chalves, pg = kx / g.chalves;
  const int j = blockIdx.y / tiles_x, tx = blockIdx.y - j * tiles_x, ty = band * tyb + j;
  if (ty * TYH >= g.h) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = tid % TXW, ly = (tid / TXW) % TYH, ld = tid / (TXW * TYH);
  const int x = tx * TXW + lx, y = ty * TYH + ly, d = pg * DP + ld;
  const unsigned REC = (unsigned)g.C * 4u;
  const int scale = g.Ws / g.w;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feats), 0, (int)((size_t)3 * g.Hs * g.Ws * REC), 0x00020000);
  float4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const int wc = g.wc, wr = g.wr, ntex = wc * wr, npieces = (ntex + 15) >> 4;
  auto fill = [&](int s, char* base) {
    // window origin: the tile's footprint in the source, shifted by a view / plane dependent parallax
    const int x0 = tx * TXW * scale - 1 + (s - 1) * (2 + pg % 5), y0 = ty * TYH * scale - 1 + (s - 1);
    for (int p = wave; p < npieces; p += NW) {
      const int L = p * 16 + (lane >> 2);
      const int row = L / wc, col = L - row * wc;
      const int gy = y0 + row, gx = x0 + col;
      const bool ok = ((unsigned)gy < (unsigned)g.Hs) & ((unsigned)gx < (unsigned)g.Ws) & (L < ntex);
      const unsigned off = (unsigned)s * (unsigned)(g.Hs * g.Ws) * REC + (unsigned)(gy * g.Ws + gx) * REC + (unsigned)chh * 64u + (unsigned)(lane & 3) * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + p * 1024), 16,
                                               (int)(ok ? off : 0x80000000u), 0, 0, 0);
    }
  };
  auto read = [&](const char* base) {
    const unsigned key = (unsigned)(lane >> 2) & 3u;
    const unsigned rec = (unsigned)((ly * scale + 1 + (ld & 1)) * wc + lx * scale + 1 + (ld >> 1));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned a0 = (rec << 6) + ((q ^ key) << 4), a1 = a0 + (unsigned)wc * 64u;
      const float4 t00 = *(const float4*)(base + a0), t01 = *(const float4*)(base + a0 + 64), t10 = *(const float4*)(base + a1),
                   t11 = *(const float4*)(base + a1 + 64);
      acc[q].x += (t00.x + t01.x) + (t10.x + t11.x), acc[q].y += (t00.y + t01.y) + (t10.y + t11.y);
      acc[q].z += (t00.z + t01.z) + (t10.z + t11.z), acc[q].w += (t00.w + t01.w) + (t10.w + t11.w);
    }
  };
  // units of 8 channels (32-byte records): lane = (record lane >> 1, 16-byte half lane & 1), 32 records per piece
  auto fill_half = [&](int u, char* base) {
    const int s = u >> 1, ch8 = u & 1;
    const int x0 = tx * TXW * scale - 1 + (s - 1) * (2 + pg % 5), y0 = ty * TYH * scale - 1 + (s - 1);
    const int np = (ntex + 31) >> 5;
    for (int p = wave; p < np; p += NW) {
      const int L = p * 32 + (lane >> 1);
      const int row = L / wc, col = L - row * wc;
      const int gy = y0 + row, gx = x0 + col;
      const bool ok = ((unsigned)gy < (unsigned)g.Hs) & ((unsigned)gx < (unsigned)g.Ws) & (L < ntex);
      const unsigned off = (unsigned)s * (unsigned)(g.Hs * g.Ws) * REC + (unsigned)(gy * g.Ws + gx) * REC + (unsigned)chh * 64u + (unsigned)ch8 * 32u +
                           (unsigned)(lane & 1) * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + p * 1024), 16,
                                               (int)(ok ? off : 0x80000000u), 0, 0, 0);
    }
  };
  auto read_half = [&](int u, const char* base) {
    const unsigned key = (unsigned)(lane >> 3) & 1u;
    const unsigned rec = (unsigned)((ly * scale + 1 + (ld & 1)) * wc + lx * scale + 1 + (ld >> 1));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned a0 = (rec << 5) + ((q ^ key) << 4), a1 = a0 + (unsigned)wc * 32u;
      const float4 t00 = *(const float4*)(base + a0), t01 = *(const float4*)(base + a0 + 32), t10 = *(const float4*)(base + a1),
                   t11 = *(const float4*)(base + a1 + 32);
      float4& A = acc[(u & 1) * 2 + q];
      A.x += (t00.x + t01.x) + (t10.x + t11.x), A.y += (t00.y + t01.y) + (t10.y + t11.y);
      A.z += (t00.z + t01.z) + (t10.z + t11.z), A.w += (t00.w + t01.w) + (t10.w + t11.w);
    }
  };
  if (DBUF_HALF) {
    char* buf[2] = {win, win + (size_t)cap * 32};
    if (FILL) fill_half(0, buf[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      if (FILL && u + 1 < 6) fill_half(u + 1, buf[(u + 1) & 1]);
      if (READ) read_half(u, buf[u & 1]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else if (DBUF_FULL) {
    char* buf[2] = {win, win + (size_t)cap * 64};
    if (FILL) fill(0, buf[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (FILL && u + 1 < 3) fill(u + 1, buf[(u + 1) & 1]);
      if (READ) read(buf[u & 1]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else if (UPFRONT) {
    if (FILL)
      for (int s = 0; s < 3; ++s) fill(s, win + (size_t)s * cap * 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (READ)
      for (int s = 0; s < 3; ++s) read(win + (size_t)s * cap * 64);
  } else {
    for (int s = 0; s < 3; ++s) {
      if (FILL) fill(s, win);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (READ) read(win);
      if (s < 2) __syncthreads();
    }
  }
  const size_t hw = (size_t)g.h * g.w;
  const unsigned cstride = (unsigned)(g.D * hw) * 4u;
  __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)g.C * g.D * hw * 4), 0x00020000);
  if (STORE) {
    const unsigned voff = (unsigned)((size_t)d * hw + (size_t)y * g.w + x) * 4u;
    unsigned soff = (unsigned)chh * 16u * cstride;
    const float* a = (const float*)acc;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a[c]), orsrc, (int)voff, (int)soff, TXW >= 32 ? 2 : 0);
      soff += cstride;
    }
  }
  if (STORE4) {   // wave w writes channels 4w .. 4w+3 (NW = 4) of the tile as 16-byte pieces: (TXW / 4) lanes per row
    constexpr int LPR = TXW / 4, ROWS = 64 / LPR;   // rows (of one plane) per wave-instruction
    constexpr int NI = (TYH * DP + ROWS - 1) / ROWS;   // instructions per channel
    typedef __attribute__((ext_vector_type(4))) int i4;
#pragma unroll
    for (int cc = 0; cc < 16 / NW; ++cc) {
      const int c = wave * (16 / NW) + cc;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r = i * ROWS + lane / LPR, ry = r % TYH, rd = r / TYH;
        const unsigned voff = (unsigned)((size_t)(pg * DP + rd) * hw + (size_t)(ty * TYH + ry) * g.w + tx * TXW + (lane % LPR) * 4) * 4u;
        i4 v = {__builtin_bit_cast(int, acc[cc & 3].x), __builtin_bit_cast(int, acc[cc & 3].y), __builtin_bit_cast(int, acc[cc & 3].z), i};
        if (r < TYH * DP) __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, (int)voff, (int)((unsigned)(chh * 16 + c) * cstride), 2);
      }
    }
  }
  if (!STORE && !STORE4 && READ) {
    float s = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
    if (s == 1234.5f) out[0] = s;
  }
}

template <int TXW, int TYH, int DP, int FLAGS>
float run(const Geo& g, const float* feats, std::vector<float*>& outs, int iters, int cap_records) {
  const int tiles_x = (g.w + TXW - 1) / TXW, tiles_y = (g.h + TYH - 1) / TYH, tyb = (tiles_y + 7) / 8, pgroups = (g.D + DP - 1) / DP;
  auto kern = sol_kernel<TXW, TYH, DP, FLAGS>;
  const size_t lds = (size_t)cap_records * 64 * ((FLAGS & 16) ? 3 : (FLAGS & 64) ? 2 : 1) + 64;   // (flag 32: two buffers of cap x 32 B)
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(8u * g.chalves * pgroups, tiles_x * tyb, 1), block(TXW * TYH * DP);
  std::vector<hipEvent_t> ev(2 * iters);
  for (auto& e : ev) hipEventCreate(&e);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, grid, block, lds, 0, feats, outs[i % outs.size()], g, tiles_x, tyb, cap_records);
  hipDeviceSynchronize();
  for (int i = 0; i < iters; ++i)
    hipExtLaunchKernelGGL(kern, grid, block, lds, 0, ev[2 * i], ev[2 * i + 1], 0, feats, outs[i % outs.size()], g, tiles_x, tyb, cap_records);
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
  printf("    flags %2d  lds %6zu B  grid %u x %u x %u thr  avg %7.2f us  min %7.2f us\n", FLAGS, lds, grid.x, grid.y, block.x, sum / iters * 1e3, mn * 1e3);
  return (float)(sum / iters * 1e3);
}

template <int TXW, int TYH, int DP>
void level(const char* name, Geo g, int cap, int nout) {
  float* feats;
  const size_t fbytes = (size_t)3 * g.Hs * g.Ws * g.C * 4, obytes = (size_t)g.C * g.D * g.h * g.w * 4;
  hipMalloc(&feats, fbytes);
  hipMemset(feats, 0x3c, fbytes);
  std::vector<float*> outs(nout);
  for (auto& o : outs) hipMalloc(&o, obytes), hipMemset(o, 0, obytes);
  const double alg = (double)(fbytes + obytes);
  printf("%s: tile %d x %d x %d planes, window %d x %d records (cap %d), algorithmic %.1f MB, %d output volume(s) in rotation\n", name, TXW, TYH, DP,
         g.wc, g.wr, cap, alg / 1e6, nout);
  const int it = 40;
  run<TXW, TYH, DP, 0>(g, feats, outs, it, cap);
  float t;
  t = run<TXW, TYH, DP, 4>(g, feats, outs, it, cap);
  printf("      stores only (dword, lane = voxel)          : %.2f TB/s of the volume\n", obytes / t / 1e6);
  t = run<TXW, TYH, DP, 8>(g, feats, outs, it, cap);
  printf("      stores only (dwordx4, wave = channel tile) : %.2f TB/s of the volume\n", obytes / t / 1e6);
  run<TXW, TYH, DP, 1>(g, feats, outs, it, cap);
  run<TXW, TYH, DP, 3>(g, feats, outs, it, cap);
  t = run<TXW, TYH, DP, 5>(g, feats, outs, it, cap);
  printf("      fills + dword stores                        : %.3f of 8 TB/s on the algorithmic bytes\n", alg / t / 1e6 / 8.0);
  t = run<TXW, TYH, DP, 7>(g, feats, outs, it, cap);
  printf("      fills + tap reads + dword stores            : %.3f of 8 TB/s on the algorithmic bytes\n", alg / t / 1e6 / 8.0);
  t = run<TXW, TYH, DP, 11>(g, feats, outs, it, cap);
  printf("      fills + tap reads + dwordx4 stores          : %.3f of 8 TB/s on the algorithmic bytes\n", alg / t / 1e6 / 8.0);
  t = run<TXW, TYH, DP, 32 + 7>(g, feats, outs, it, cap);
  printf("      8-channel units, double-buffered fills + reads + stores: %.3f of 8 TB/s on the algorithmic bytes\n", alg / t / 1e6 / 8.0);
  run<TXW, TYH, DP, 32 + 3>(g, feats, outs, it, cap);
  run<TXW, TYH, DP, 32 + 1>(g, feats, outs, it, cap);
  t = run<TXW, TYH, DP, 64 + 7>(g, feats, outs, it, cap);
  printf("      16-channel units, double-buffered (LDS x 2) fills + reads + stores: %.3f of 8 TB/s on the algorithmic bytes\n", alg / t / 1e6 / 8.0);
  if ((size_t)cap * 64 * 3 + 64 <= 160 * 1024) {
    t = run<TXW, TYH, DP, 16 + 7>(g, feats, outs, it, cap);
    printf("      three windows up front + reads + dword stores: %.3f of 8 TB/s on the algorithmic bytes\n", alg / t / 1e6 / 8.0);
  }
  hipFree(feats);
  for (auto& o : outs) hipFree(o);
}

int main(int argc, char** argv) {
  const int nout = argc > 1 ? atoi(argv[1]) : 8;
  // level 1: source at the volume's resolution; window = tile + halo + parallax slack
  level<32, 8, 1>("level 1", Geo{16, 8, 256, 320, 256, 320, 32, 8, 1, 1, 35, 11}, 448, nout);
  level<32, 8, 1>("level 1 (exact windows 34 x 10)", Geo{16, 8, 256, 320, 256, 320, 32, 8, 1, 1, 34, 10}, 352, nout);
  level<32, 4, 1>("level 1, 128-thread tiles", Geo{16, 8, 256, 320, 256, 320, 32, 4, 1, 1, 35, 7}, 256, nout);
  // level 0: source at twice the volume's resolution, 8 planes share a window along the epipolar line
  level<16, 2, 8>("level 0", Geo{32, 64, 64, 80, 128, 160, 16, 2, 8, 2, 48, 8}, 448, nout);
  level<16, 4, 8>("level 0, 512-thread tiles", Geo{32, 64, 64, 80, 128, 160, 16, 4, 8, 2, 48, 12}, 576, nout);
  return 0;
}
