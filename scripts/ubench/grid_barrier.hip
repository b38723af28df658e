// Cost of a grid-wide barrier inside one kernel on MI355X (8 XCDs, non-coherent L2s): what a fused "U-Net interior"
// kernel would pay between two dependent layers instead of a kernel boundary (~10 us per tiny layer in the frame's
// graph, of which ~4-5 us are the dispatch gap + ramp).
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
// Variants: fence scope (agent = write back / invalidate the XCD's L2, as cross-XCD visibility of a layer's output
// needs), with and without a small payload written before and read after the barrier (checks visibility).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);   // (HIP: agent scope by default for __threadfence-like fences)
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

template <bool PAYLOAD>
__global__ void __launch_bounds__(256) k(unsigned* counter, float* buf, int nbar, int n, unsigned base, int* bad) {
  const int gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
  for (int b = 0; b < nbar; ++b) {
    if (PAYLOAD)
      for (int i = gid; i < n; i += gsz) buf[(size_t)(b & 1) * n + i] = (float)(b * 7 + i % 13);
    grid_barrier(counter, base + gridDim.x * (unsigned)(b + 1));
    if (PAYLOAD) {
      // read what OTHER workgroups (other XCDs) wrote
      for (int i = gid; i < n; i += gsz) {
        const int j = (i + n / 2 + 12345) % n;
        if (buf[(size_t)(b & 1) * n + j] != (float)(b * 7 + j % 13)) atomicAdd(bad, 1);
      }
    }
  }
}

int main() {
  unsigned* counter;
  float* buf;
  int* bad;
  const int n = 1 << 20;   // 4 MB payload per phase (a deep U-Net layer's output is 0.3-2.6 MB)
  hipMalloc(&counter, 4), hipMalloc(&buf, 2 * (size_t)n * 4), hipMalloc(&bad, 4);
  hipMemset(counter, 0, 4), hipMemset(bad, 0, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  unsigned base = 0;
  for (int grid : {256, 512, 128}) {
    for (int nbar : {0, 1, 8, 32}) {
      for (int payload = 0; payload < 2; ++payload) {
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
          hipEventRecord(e0, 0);
          if (payload)
            hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, counter, buf, nbar, n, base, bad);
          else
            hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, counter, buf, nbar, n, base, bad);
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
          base += (unsigned)grid * nbar;
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (it && ms < best) best = ms;
        }
        printf("grid %4d  barriers %2d  payload %d : %8.2f us\n", grid, nbar, payload, best * 1e3);
      }
    }
  }
  int hb;
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("visibility errors: %d\n", hb);
  return 0;
}
