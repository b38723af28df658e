// Accumulation target of the scatter (gradient) kernels: float atomics, or ORDER-INDEPENDENT fixed-point accumulation.
//
// A float atomic add rounds after every addition, so the sum a texel ends up with depends on the order in which the
// hardware serves the adds: the gradients of the scatter kernels differ in their last bits from run to run, and
// training-mode batch norm over the few voxels of a deep U-Net level amplifies that (measured: up to 1.2e-2 relative L2
// between two runs of the SAME eager step on the tiny fixture, profiles/r5/graphed_step_flake.txt).  Integer addition is
// associative: with every contribution rounded ONCE to a multiple of 2^-k and added as a 64-bit integer, the result is
// the same whatever the order -- bit-reproducible training (BMV_DETERMINISTIC, include/bmv.h: the *_fixed entry points).
//
// A *_fixed launch is three passes of the same kernel code through ScatterAcc<MODE>:
//   MODE 1  the largest |contribution| of the launch (atomicMax on the float's bits: max is order-independent too)
//   scale   2^(38 - e) with max < 2^e, PER OUTPUT tensor of the launch: a contribution becomes an integer below 2^38 -- 38 bits
//           under the largest one of its tensor; 2^25 maximal contributions fit a texel's int64 without overflow (include/bmv.h)
//   MODE 2  accumulate  llrint(v * scale)  with 64-bit integer atomics into the workspace
//   finish  float(q * 2^-(38 - e)) into the caller's float gradient buffer
// MODE 0 is the plain float-atomic form the default path runs.  Non-finite contributions make the scale NaN and with it
// every output of the launch: loud, as the float path's NaN would be.
#pragma once
#include <type_traits>

#include "bmv_common.hpp"

namespace bmv {

constexpr int kFixedSlots = 256;                    // spread of the magnitude pass over addresses
constexpr int kFixedOuts = 2;                       // outputs of a launch with their OWN scale (d_feats + d_depth_values, d_depth + d_std:
                                                    // tensors of different units -- with one scale for both the smaller one kept only
                                                    // 38 - log2(ratio) bits: ADVICE r5)
constexpr int kFixedHeader = kFixedOuts * (1 + kFixedSlots / 2);   // int64 words in front of the accumulators: per output [scale, 1 / scale], then per output the slots
constexpr int kFixedBits = 38;

template <int MODE>
struct ScatterAcc {
  const float* base;      // the float destination the kernel's addresses are relative to
  long long* q;           // MODE 2: its fixed-point twin (same indexing)
  unsigned* slots;        // MODE 1: kFixedSlots running maxima of |v| (float bits)
  float scale;            // MODE 2
  unsigned seen;          // MODE 1: the largest magnitude this thread has already published
  __device__ __forceinline__ void add(float* p, float v) {
    if constexpr (MODE == 0) {
      atomicAdd(p, v);
    } else if constexpr (MODE == 1) {
      // (a thread publishes only what beats everything it has seen -- its own contributions, the slot's value when it
      // started and what the slot held at its last publication: skipping a value <= a former slot value cannot change
      // the final maximum, so the result stays order-independent while almost no thread ever issues an atomic)
      const unsigned b = __float_as_uint(fabsf(v));
      if (b > seen) seen = max(b, atomicMax(slots + ((blockIdx.x * blockDim.x + threadIdx.x) & (kFixedSlots - 1)), b));
    } else {
      atomicAdd(reinterpret_cast<unsigned long long*>(q + (p - base)), (unsigned long long)__float2ll_rn(v * scale));
    }
  }
};

// what a launcher hands its kernel: the workspace of a *_fixed call (null = float atomics)
struct FixedWs {
  long long* ws = nullptr;
  __host__ __device__ float* scale2(int which = 0) const { return reinterpret_cast<float*>(ws + which); }              // [scale, 1 / scale] of output `which`
  __host__ __device__ unsigned* slots(int which = 0) const { return reinterpret_cast<unsigned*>(ws + kFixedOuts + which * (kFixedSlots / 2)); }
  __host__ __device__ long long* acc(size_t offset) const { return ws + kFixedHeader + offset; }
};
// `which`: the launch's output this accumulator belongs to (0 or 1): its own magnitude slots and scale
template <int MODE>
__device__ __forceinline__ ScatterAcc<MODE> make_acc(const float* base, FixedWs f, size_t offset, int which = 0) {
  ScatterAcc<MODE> a;
  a.base = base, a.q = nullptr, a.slots = nullptr, a.scale = 0.f, a.seen = 0u;
  if constexpr (MODE == 1) {
    a.slots = f.slots(which);
    a.seen = __atomic_load_n(a.slots + ((blockIdx.x * blockDim.x + threadIdx.x) & (kFixedSlots - 1)), __ATOMIC_RELAXED);
  }
  if constexpr (MODE == 2) a.q = f.acc(offset), a.scale = f.scale2(which)[0];
  return a;
}

static __global__ void fixed_scale_kernel(FixedWs f) {
  bool bad = false;
  unsigned mm[kFixedOuts];
#pragma unroll
  for (int w = 0; w < kFixedOuts; ++w) {
    unsigned m = 0;
    for (int i = threadIdx.x; i < kFixedSlots; i += 64) m = max(m, f.slots(w)[i]);
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, s, 64));
    mm[w] = m;
    bad |= m > 0x7f7fffffu;         // inf / NaN somewhere: EVERY output of the launch becomes NaN
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kFixedOuts; ++w) {
      float scale = 1.f, inv = 1.f;
      if (bad) {
        scale = inv = __uint_as_float(0x7fc00000u);
      } else if (mm[w] != 0u) {
        int e;
        (void)frexpf(__uint_as_float(mm[w]), &e);      // max = mant * 2^e, mant in [0.5, 1)
        scale = ldexpf(1.f, kFixedBits - e), inv = ldexpf(1.f, e - kFixedBits);
      }
      f.scale2(w)[0] = scale, f.scale2(w)[1] = inv;
    }
  }
}

static __global__ void fixed_finish_kernel(const long long* __restrict__ q, const float* __restrict__ scale2, size_t n,
                                           float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)((double)q[i] * (double)scale2[1]);
}

static inline void fixed_finish(FixedWs f, size_t offset, size_t n, float* dst, hipStream_t st, int which = 0) {
  if (n) hipLaunchKernelGGL(fixed_finish_kernel, dim3(cdiv((long)n, 256)), dim3(256), 0, st, f.acc(offset), f.scale2(which), n, dst);
}

// An EMPTY input (no rays / samples): the float forms add nothing into their caller-zeroed outputs and return; the
// fixed-point forms WRITE their outputs (callers hand them uninitialised memory), so they write zeros.  A kernel, not
// hipMemsetAsync: a memset node did not reliably clear its buffer on later replays of a captured step.
static __global__ void fixed_zero_kernel(float* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = 0.f;
}
static inline void fixed_zero(float* dst, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(fixed_zero_kernel, dim3(cdiv((long)n, 256)), dim3(256), 0, st, dst, n);
}

// the passes of a launch: float atomics, or (fixed.ws set) magnitude pass -> scale -> fixed-point pass (scatter.hpp);
// `launch(mode)` issues the kernel with MODE = decltype(mode)::value
template <class L>
static inline void launch_modes(FixedWs fixed, hipStream_t st, L&& launch) {
  if (!fixed.ws) {
    launch(std::integral_constant<int, 0>{});
    return;
  }
  launch(std::integral_constant<int, 1>{});
  hipLaunchKernelGGL(fixed_scale_kernel, dim3(1), dim3(64), 0, st, fixed);
  launch(std::integral_constant<int, 2>{});
}

}  // namespace bmv
