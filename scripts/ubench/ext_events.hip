// What do HIP events read around a kernel of KNOWN length?  (a) hipEventRecord before / after the launch, (b) the
// start / stop events of hipExtLaunchKernelGGL, which are bound to the kernel's own dispatch.  The kernel spins for a
// given number of 100 MHz ticks (s_memrealtime), so its true duration is known to ~0.1 us.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/ext_events.hip -o scripts/ubench/ext_events && scripts/ubench/ext_events
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void spin(unsigned long long ticks, int* sink) {
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
  }
  if (threadIdx.x == 9999) *sink = 1;
}
__global__ void tiny(int* sink) {
  if (threadIdx.x == 9999) *sink = 1;
}

int main() {
  int* sink;
  hipMalloc(&sink, 4);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t a, b, c, d;
  hipEventCreate(&a), hipEventCreate(&b), hipEventCreate(&c), hipEventCreate(&d);
  for (int us : {5, 22, 100}) {
    unsigned long long ticks = (unsigned long long)us * 100;
    float sum_rec = 0, sum_ext = 0;
    const int N = 20;
    for (int i = 0; i < N + 2; ++i) {
      // a tiny kernel in front, as in the frame (the sweep follows a one-wave launch)
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, sink);
      hipEventRecord(a, st);
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, ticks, sink);
      hipEventRecord(b, st);
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, sink);
      hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, c, d, 0, ticks, sink);
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, sink);
      hipStreamSynchronize(st);
      float m1 = 0, m2 = 0;
      hipEventElapsedTime(&m1, a, b);
      hipError_t e = hipEventElapsedTime(&m2, c, d);
      if (i >= 2) sum_rec += m1, sum_ext += m2;
      if (i == 2) printf("  (ext elapsed rc=%d)\n", (int)e);
    }
    printf("kernel %3d us: hipEventRecord bracket %.2f us, hipExtLaunchKernelGGL start/stop %.2f us\n", us,
           sum_rec / N * 1e3, sum_ext / N * 1e3);
  }
  return 0;
}
