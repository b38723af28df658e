// What does a launch of the plane sweep's GRID cost with nothing in it?  (the "4.2 us dispatch bracket" of
// DESIGN 4.1 was graph-timed; here the same empty grids run under rocprofv3 --kernel-trace --stats, whose kernel duration
// is the dispatch's own begin -> end.)  Grids: level 0 = 768 workgroups x 256 threads (8 x 8 x 12, 30 KB LDS), level 1 =
// 1280 x 256 (8 x 4 x 40), and 256 / 2560 for scale; each with and without the sweep's dynamic LDS (occupancy-limiting).
// hipcc --offload-arch=gfx950 -O3 -o empty_grid empty_grid.hip
// cd /tmp && rocprofv3 --kernel-trace --stats -d out -- ./empty_grid
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void __launch_bounds__(256) empty_l0(float* p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }
__global__ void __launch_bounds__(256) empty_l1(float* p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }
__global__ void __launch_bounds__(256) empty_256(float* p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }
__global__ void __launch_bounds__(256) empty_2560(float* p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }
__global__ void __launch_bounds__(256) empty_l0_lds(float* p) { extern __shared__ float s[]; if (p && threadIdx.x == 1024) p[0] = s[0]; }
__global__ void __launch_bounds__(256) empty_l1_lds(float* p) { extern __shared__ float s[]; if (p && threadIdx.x == 1024) p[0] = s[0]; }

int main() {
  float* p;
  hipMalloc(&p, 4);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int rep = 0; rep < 50; ++rep) {
    hipLaunchKernelGGL(empty_l0, dim3(8 * 8, 12, 1), dim3(256), 0, st, p);
    hipLaunchKernelGGL(empty_l1, dim3(8 * 4, 40, 1), dim3(256), 0, st, p);
    hipLaunchKernelGGL(empty_256, dim3(256), dim3(256), 0, st, p);
    hipLaunchKernelGGL(empty_2560, dim3(2560), dim3(256), 0, st, p);
    hipLaunchKernelGGL(empty_l0_lds, dim3(8 * 8, 12, 1), dim3(256), 30 * 1024, st, p);
    hipLaunchKernelGGL(empty_l1_lds, dim3(8 * 4, 40, 1), dim3(256), 30 * 1024, st, p);
  }
  hipStreamSynchronize(st);
  // back-to-back launches of ONE grid, event-timed: the per-launch cost when the queue never drains
  const char* names[2] = {"level-0 grid (768 x 256)", "level-1 grid (1280 x 256)"};
  for (int g = 0; g < 2; ++g) {
    hipEventRecord(e0, st);
    for (int i = 0; i < 200; ++i) {
      if (g == 0) hipLaunchKernelGGL(empty_l0_lds, dim3(8 * 8, 12, 1), dim3(256), 30 * 1024, st, p);
      else hipLaunchKernelGGL(empty_l1_lds, dim3(8 * 4, 40, 1), dim3(256), 30 * 1024, st, p);
    }
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.2f us per back-to-back launch (200 launches, event-timed)\n", names[g], ms * 1e3 / 200);
  }
  return 0;
}
