// Shared device helpers for the BoostMVSNeRFs hot-path kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/bmv.h"

// fp32 division of the per-sample geometry.  Exact by default (the standalone ops and their tests); render.hip
// defines it as a * v_rcp_f32(b) (1 ulp): the fused renderer is co-limited by VALU issue and had 67 IEEE division
// expansions (~10 instructions each) per 32 samples.
#ifndef BMV_DIV
#define BMV_DIV(a, b) ((a) / (b))
#endif

namespace bmv {

void set_error(const char* fmt, ...);
// explicit tuning switch of the launchers (csrc/tuning.hip; include/bmv.h bmv_tuning_set): its value, or dflt when unset
int tuning(const char* name, int dflt);

// Events the NEXT sweep launch of this thread binds to its own dispatch (bmv_bind_next_launch, csrc/timing.hip):
// hipExtLaunchKernelGGL start / stop events read the kernel's begin and end, a hipEventRecord pair around a launch reads
// ~2.5 us more (scripts/ubench/ext_events.hip).  take_launch_events() hands them over once.
struct LaunchEvents {
  hipEvent_t start = nullptr, stop = nullptr;
};
LaunchEvents take_launch_events();
void set_launch_events(hipEvent_t start, hipEvent_t stop);

// Deferred pointers (bmv_defer_pointer, csrc/timing.hip): "argument `ptr` of the NEXT launch is to be read from
// table[slot] when the kernel RUNS".  A launcher that supports it for an argument asks deferred_for(); BMV_LAUNCH_END
// fails the call if a registered deferral was not taken (the entry point does not read that argument through a table).
struct DeferredPtr {
  const void* const* table = nullptr;
  int slot = -1;
};
DeferredPtr deferred_for(const void* ptr);
int deferred_finish();   // number of registered deferrals no launcher took; clears the list
template <typename T>
__device__ __forceinline__ T* deferred_load(const void* const* table, int slot, T* direct) {
  return (table && slot >= 0) ? static_cast<T*>(const_cast<void*>(table[slot])) : direct;
}

// (a failed check also drops the deferrals registered for the launch that is not going to happen: they must not attach
// themselves to the next, unrelated one)
#define BMV_REQUIRE(cond, ...)                  \
  do {                                          \
    if (!(cond)) {                              \
      ::bmv::set_error(__VA_ARGS__);            \
      (void)::bmv::deferred_finish();           \
      return BMV_ERR_INVALID;                   \
    }                                           \
  } while (0)

#define BMV_LAUNCH_END(name)                                                        \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      ::bmv::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return BMV_ERR_LAUNCH;                                                        \
    }                                                                               \
    if (::bmv::deferred_finish()) {                                                 \
      ::bmv::set_error("%s: a pointer registered with bmv_defer_pointer is not an argument this entry point reads through a table", name); \
      return BMV_ERR_UNSUPPORTED;                                                   \
    }                                                                               \
    return BMV_OK;                                                                  \
  } while (0)

// consecutive workgroup ids go round-robin over the 8 XCDs: XCD x runs workgroups x, x + 8, ...  Gives each XCD a
// CONTIGUOUS run of the n work items (per + (x < rem) of them)
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
  const int x = bid & 7, per = n >> 3, rem = n & 7;
  return x * per + min(x, rem) + (bid >> 3);
}

static inline hipStream_t as_stream(bmv_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

// ---------------------------------------------------------------------------
// Bilinear taps with torch.grid_sample(align_corners=True) semantics.
// Offsets are in elements of one channel plane (y * W + x); taps that fall
// outside get weight 0 and offset 0 so loads stay in bounds.
// ---------------------------------------------------------------------------
struct Taps2 {
  int o00, o01, o10, o11;
  float w00, w01, w10, w11;
};

// grid -> pixel:  ix = ((g + 1) / 2) * (W - 1)        (aten grid_sampler_unnormalize)
__device__ __forceinline__ float unnorm(float g, int size) { return ((g + 1.f) * 0.5f) * (float)(size - 1); }

// padding_mode='zeros'.  Finite but huge coordinates (the reference divides by
// clamp_min(z, 1e-6)) and NaN give all-zero weights; the float->int conversion
// only happens for coordinates known to be in range.
__device__ __forceinline__ Taps2 taps_zeros(float ix, float iy, int W, int H) {
  Taps2 t;
  float fx = floorf(ix), fy = floorf(iy);
  // branch-free: clamp to [-2, size] before the int conversion (fmaxf/fminf also map NaN
  // there), so anything outside comes out with both taps invalid and weight exactly 0
  int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W), y0 = (int)fminf(fmaxf(fy, -2.f), (float)H);
  int x1 = x0 + 1, y1 = y0 + 1;
  float ex = (fx + 1.f) - ix, ey = (fy + 1.f) - iy;  // (ix_se - ix), (iy_se - iy)
  float ax = ix - fx, ay = iy - fy;
  bool vx0 = (x0 >= 0) & (x0 <= W - 1), vx1 = (x1 >= 0) & (x1 <= W - 1);
  bool vy0 = (y0 >= 0) & (y0 <= H - 1), vy1 = (y1 >= 0) & (y1 <= H - 1);
  t.w00 = (vx0 && vy0) ? ex * ey : 0.f;
  t.w01 = (vx1 && vy0) ? ax * ey : 0.f;
  t.w10 = (vx0 && vy1) ? ex * ay : 0.f;
  t.w11 = (vx1 && vy1) ? ax * ay : 0.f;
  int cx0 = vx0 ? x0 : 0, cx1 = vx1 ? x1 : W - 1, cy0 = vy0 ? y0 : 0, cy1 = vy1 ? y1 : H - 1;
  t.o00 = cy0 * W + cx0;
  t.o01 = cy0 * W + cx1;
  t.o10 = cy1 * W + cx0;
  t.o11 = cy1 * W + cx1;
  return t;
}

// padding_mode='border': coordinates are clipped to [0, size-1] first.
__device__ __forceinline__ Taps2 taps_border(float ix, float iy, int W, int H) {
  ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
  if (!(ix == ix)) ix = 0.f;
  if (!(iy == iy)) iy = 0.f;
  float fx = floorf(ix), fy = floorf(iy);
  int x0 = (int)fx, y0 = (int)fy;
  int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  bool vx1 = (x0 + 1) <= W - 1, vy1 = (y0 + 1) <= H - 1;
  float ex = (fx + 1.f) - ix, ey = (fy + 1.f) - iy;
  float ax = ix - fx, ay = iy - fy;
  Taps2 t;
  t.w00 = ex * ey;
  t.w01 = vx1 ? ax * ey : 0.f;
  t.w10 = vy1 ? ex * ay : 0.f;
  t.w11 = (vx1 && vy1) ? ax * ay : 0.f;
  t.o00 = y0 * W + x0;
  t.o01 = y0 * W + x1;
  t.o10 = y1 * W + x0;
  t.o11 = y1 * W + x1;
  return t;
}

// p[i] for a 32-bit BYTE offset: `wave-uniform pointer + zero-extended 32-bit lane offset` is one addressing mode of
// global_load (scalar base + vector offset); an element index would need a 64-bit shift-add per load.
__device__ __forceinline__ float ld_byte_off(const float* __restrict__ p, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + byte_off);
}
// The same through a buffer resource: scalar descriptor + 32-bit lane byte offset + scalar byte offset (the channel
// plane): no vector ALU work per load at all.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, size_t bytes) {
  // the pointer is wave-uniform by construction; say so (readfirstlane), or every load through the descriptor is
  // wrapped in a waterfall loop
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  float* q = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float tap_fetch_buf(__amdgpu_buffer_rsrc_t r, const Taps2& t, unsigned soff) {
  float v = ld_buf(r, (unsigned)t.o00, soff) * t.w00;
  v += ld_buf(r, (unsigned)t.o01, soff) * t.w01;
  v += ld_buf(r, (unsigned)t.o10, soff) * t.w10;
  v += ld_buf(r, (unsigned)t.o11, soff) * t.w11;
  return v;
}
// taps whose offsets were pre-multiplied by 4 (tap_bytes)
__device__ __forceinline__ float tap_fetch_bytes(const float* __restrict__ p, const Taps2& t) {
  float v = ld_byte_off(p, (unsigned)t.o00) * t.w00;
  v += ld_byte_off(p, (unsigned)t.o01) * t.w01;
  v += ld_byte_off(p, (unsigned)t.o10) * t.w10;
  v += ld_byte_off(p, (unsigned)t.o11) * t.w11;
  return v;
}
__device__ __forceinline__ void tap_bytes(Taps2& t, int extra_elems) {
  t.o00 = (t.o00 + extra_elems) * 4, t.o01 = (t.o01 + extra_elems) * 4;
  t.o10 = (t.o10 + extra_elems) * 4, t.o11 = (t.o11 + extra_elems) * 4;
}

__device__ __forceinline__ float tap_fetch(const float* __restrict__ p, const Taps2& t) {
  // same accumulation order as aten's grid_sampler_2d: nw, ne, sw, se
  // offsets are >= 0 (taps_* clamp them into the plane): unsigned indices let the loads take the uniform plane
  // pointer as a scalar base + 32-bit lane offset instead of a 64-bit address computed per load
  float v = p[(unsigned)t.o00] * t.w00;
  v += p[(unsigned)t.o01] * t.w01;
  v += p[(unsigned)t.o10] * t.w10;
  v += p[(unsigned)t.o11] * t.w11;
  return v;
}

// F.interpolate(mode='bilinear', align_corners=True) source coordinate of one axis
// (aten area_pixel_compute_scale / compute_source_index_and_lambda).
struct Lerp1 {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp1 upsample_axis(int dst, int in_size, int out_size) {
  Lerp1 r;
  float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  float src = scale * (float)dst;
  r.i0 = min((int)src, in_size - 1);
  r.i1 = r.i0 + ((r.i0 < in_size - 1) ? 1 : 0);
  r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
  r.l0 = 1.f - r.l1;
  return r;
}
__device__ __forceinline__ float upsample_fetch(const float* __restrict__ p, int W, const Lerp1& ly, const Lerp1& lx) {
  return ly.l0 * (lx.l0 * p[ly.i0 * W + lx.i0] + lx.l1 * p[ly.i0 * W + lx.i1]) +
         ly.l1 * (lx.l0 * p[ly.i1 * W + lx.i0] + lx.l1 * p[ly.i1 * W + lx.i1]);
}

}  // namespace bmv
