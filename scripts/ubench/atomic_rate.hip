// fp32 atomic-add throughput on gfx950 for the scatter patterns of the backward kernels:
//   0  global, planar gradient: every lane its own texel (64 lines per wave instruction)
//   1  global, channel-last gradient: the 64 lanes = 2 taps x 32 consecutive channels (2 lines per instruction)
//   2  global, channel-last, 4 taps x 16 channels
//   3  LDS ds_add_f32, every lane its own word of a 48 KB window (the LDS-window scatter)
//   4  global, channel-last, 8 taps x 8 channels
// hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) k(float* __restrict__ img, int texels, int iters) {
  __shared__ float win[12288];
  const int lane = threadIdx.x & 63;
  const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (MODE == 3) {
    for (int i = threadIdx.x; i < 12288; i += 256) win[i] = 0.f;
    __syncthreads();
  }
  for (int i = 0; i < iters; ++i) {
    const unsigned r = hash(wid * 7919u + i);
    if (MODE == 0) {
      // neighbouring voxels -> neighbouring (not equal) texels of one channel plane: base + 2 * lane, rows 4 apart
      const int t = (r % (texels - 1024)) + 2 * (lane & 15) + 320 * (lane >> 4);
      atomicAdd(img + t, 1.f);
    } else if (MODE == 1) {
      const int tap = (hash(r + (lane >> 5)) % texels);
      atomicAdd(img + (size_t)tap * 32 + (lane & 31), 1.f);
    } else if (MODE == 2) {
      const int tap = (hash(r + (lane >> 4)) % texels);
      atomicAdd(img + (size_t)tap * 16 + (lane & 15), 1.f);
    } else if (MODE == 4) {
      const int tap = (hash(r + (lane >> 3)) % texels);
      atomicAdd(img + (size_t)tap * 8 + (lane & 7), 1.f);
    } else {
      const int t = (r % (12288 - 1024)) + 2 * (lane & 15) + 80 * (lane >> 4);
      atomicAdd(win + t, 1.f);
    }
  }
  if (MODE == 3) {
    __syncthreads();
    if (win[threadIdx.x] == 12345.f) img[0] = 1.f;
  }
}

template <int MODE>
void run(const char* name, float* img, int texels) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 512, grid = 2048;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, img, texels, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, img, texels, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double lanes = (double)grid * 256 * iters;
  printf("%-64s %8.1f us  %7.1f G lane-atomics/s  %6.2f G wave-instr/s\n", name, ms * 1e3, lanes / ms / 1e6, lanes / 64 / ms / 1e6);
}

int main() {
  const int texels = 128 * 160 * 3;     // level-0 source maps of three views
  float* img;
  hipMalloc(&img, (size_t)texels * 32 * 4);
  hipMemset(img, 0, (size_t)texels * 32 * 4);
  run<0>("global planar (own texel per lane, 64 lines / instr)", img, texels * 32);
  run<1>("global channel-last 2 taps x 32 ch", img, texels);
  run<2>("global channel-last 4 taps x 16 ch", img, texels * 2);
  run<4>("global channel-last 8 taps x 8 ch", img, texels * 4);
  run<3>("LDS ds_add_f32 (own word per lane)", img, texels);
  // the same channel-last pattern on ever smaller footprints: many waves on the same lines at the same time
  run<1>("global channel-last 2 x 32 ch, 8192 texels (1 MB)", img, 8192);
  run<1>("global channel-last 2 x 32 ch, 1024 texels (128 KB)", img, 1024);
  run<1>("global channel-last 2 x 32 ch, 128 texels (16 KB)", img, 128);
  return 0;
}
