// Fused renderer: rays -> per-ray bounds -> samples -> volume / image lookups ->
// MFMA MLP -> alpha compositing, one kernel (a6-a12, a14).
// Reference: lib/networks/enerf/network.py:24-43 (render_rays),
//            lib/networks/boost_enerf/network.py:123-149 (MLP-only variant + masks).
//
// Work decomposition: a wave owns tiles of 32 samples (= 32/Ns consecutive rays);
// lane = (sample s = lane & 31, half h = lane >> 5).  Both halves compute the
// sample's geometry; the gathers are split between them by channel parity, which
// is exactly the k-slot order the MLP wants (mlp.hpp).  The MLP weights sit in
// LDS once per workgroup (4 waves) and the workgroup walks tiles grid-stride.
#define BMV_DIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))   // see bmv_common.hpp
#include <stdlib.h>

#include "mlp.hpp"
#include "render_geom.hpp"

namespace bmv {

constexpr int kMaxViews = 4;   // source views per cost volume: 2, 3, 4 (dtu_pretrain.yaml:22-23)
struct RenderCams {
  Cam cam[kMaxViews];
  float tar_c[4];
};

// the weight blob into LDS; the split form (CS) keeps blob[0, A_L0) ++ blob[V_VF, TOTAL_S): the fp32 tables of the chains
// that run on the bf16 pipe stay out (mlp.hpp)
template <class L, bool CS>
__device__ __forceinline__ void blob_to_lds(float* lds, const float* __restrict__ blob) {
  if constexpr (CS) {
    static_assert(L::A_L0 % 4 == 0 && L::V_VF % 4 == 0 && L::TOTAL_S % 4 == 0, "16-byte pieces");
    for (int i = threadIdx.x; i < L::A_L0 / 4; i += blockDim.x)
      reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(blob)[i];
    for (int i = threadIdx.x; i < (L::TOTAL_S - L::V_VF) / 4; i += blockDim.x)
      reinterpret_cast<float4*>(lds)[L::A_L0 / 4 + i] = reinterpret_cast<const float4*>(blob)[L::V_VF / 4 + i];
  } else {
    for (int i = threadIdx.x; i < L::TOTAL / 4; i += blockDim.x)
      reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(blob)[i];
  }
}

template <int FEAT_CH, int NV, bool CS = false>
__global__ void __launch_bounds__(256, 2) nerf_mlp_kernel(const float* __restrict__ vox_feat,
                                                           const float* __restrict__ img,
                                                           const float* __restrict__ blob, long npts,
                                                           float* __restrict__ out) {
  using L = MlpLayout<FEAT_CH>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  blob_to_lds<L, CS>(lds, blob);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 31, h = lane >> 5;
  long ntiles = (npts + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    long pt = tile * 32 + s;
    bool valid = pt < npts;
    long p = valid ? pt : npts - 1;
    float fin[NV][L::KF], dir[NV][4], vox[4], res[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vox[j] = vox_feat[p * 8 + 2 * j + h];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float* q = img + (p * NV + i) * L::IN;
#pragma unroll
      for (int j = 0; j < L::KFC; ++j) fin[i][j] = (2 * j + h < L::FC) ? q[2 * j + h] : 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) fin[i][L::KFC + k] = q[L::FC + 2 * k + h];
#pragma unroll
      for (int k = 0; k < 4; ++k) dir[i][k] = q[L::FC + k];
    }
    mlp_forward<FEAT_CH, NV, CS>(lds, lane, fin, dir, vox, res);
    if (valid && h == 0) {
      float4 o = {res[0], res[1], res[2], res[3]};
      reinterpret_cast<float4*>(out)[pt] = o;
    }
  }
}

#ifdef BMV_RENDER_STAMPS
__device__ float g_stamps[512 * 4 * 8];
#endif

// PK: 0 = planar lookups, 1 = image lookups from 48-byte records, 3 = image and volume lookups from records
// the public argument block + the deferred pointers of this launch (bmv_defer_pointer: rays and the three outputs may be
// read from a device table when the kernel runs -- a captured frame then renders the caller's rays into the caller's
// tensors without copies)
struct RenderArgsDev : bmv_render_args {
  const void* const* table;
  int s_rays, s_out0, s_out1, s_out2;
};
__device__ __forceinline__ void resolve_deferred(RenderArgsDev& a) {
  if (a.table) {
    a.rays = deferred_load(a.table, a.s_rays, a.rays);
    a.out0 = deferred_load(a.table, a.s_out0, a.out0);
    a.out1 = deferred_load(a.table, a.s_out1, a.out1);
    a.out2 = deferred_load(a.table, a.s_out2, a.out2);
  }
}

template <int FEAT_CH, int NS, bool INV, int PK = 0, int NV = 3>
#ifndef BMV_RENDER_WPS
#define BMV_RENDER_WPS 2   // workgroups (= waves per SIMD) resident per CU
#endif
__global__ void __launch_bounds__(256, BMV_RENDER_WPS) render_rays_kernel(RenderArgsDev a) {
  resolve_deferred(a);
  using L = MlpLayout<FEAT_CH>;
  static_assert(32 % NS == 0, "samples per ray must divide 32");
  constexpr int RAYS_PER_TILE = 32 / NS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  RenderCams* rc = reinterpret_cast<RenderCams*>(lds + L::TOTAL);
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < L::TOTAL / 4; i += blockDim.x)
    reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(a.blob)[i];
  if (threadIdx.x < NV)
    load_cam(a.src_exts + ((size_t)b * NV + threadIdx.x) * 16, a.src_ixts + ((size_t)b * NV + threadIdx.x) * 9,
             a.render_scale, rc->cam[threadIdx.x]);
  if (threadIdx.x == NV) camera_centre(a.tar_ext + (size_t)b * 16, rc->tar_c);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 31, h = lane >> 5;
  const int k = s % NS;  // sample index along the ray
  const size_t hwv = (size_t)a.hv * a.wv;
  const size_t plane = (size_t)a.Hr * a.Wr;
  const float* depth = a.depth + b * hwv;
  const float* std_ = a.std + b * hwv;
  const float* nf = a.near_far + b * 2 * hwv;
  const float* vol = a.volume + (size_t)b * 8 * a.Dv * hwv;
  const float inv_w = (float)(a.Wr - 1), inv_h = (float)(a.Hr - 1);

  // buffer descriptors of the gathered tensors (wave-uniform, built once): loads are descriptor + 32-bit offsets
  const __amdgpu_buffer_rsrc_t rs_vol = make_rsrc(vol, (size_t)8 * a.Dv * hwv * 4);
  __amdgpu_buffer_rsrc_t rs_f[NV], rs_c[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    // source view i of this cost volume: slot i, or view_ids[b*NV + i] of tensors that hold all n_all views
    const size_t vslot = a.view_ids ? (size_t)b * a.n_all + a.view_ids[b * NV + i] : (size_t)b * NV + i;
    if constexpr (PK != 0) {
      rs_f[i] = make_rsrc(a.im_packed + vslot * 12 * plane, (size_t)12 * plane * 4);   // 48-byte records
      rs_c[i] = rs_f[i];
    } else {
      rs_f[i] = make_rsrc(a.im_feat + vslot * FEAT_CH * plane, (size_t)FEAT_CH * plane * 4);
      rs_c[i] = make_rsrc(a.rgb_src + vslot * 3 * plane, (size_t)3 * plane * 4);
    }
  }
  const int nrays = a.ray_end - a.ray_begin;
  const int ntiles = (nrays + RAYS_PER_TILE - 1) / RAYS_PER_TILE;
  // -DBMV_RENDER_STAMPS: shader-clock stamps of the phases of every wave's third tile (scripts/stamps_render.py reads
  // them through bmv_debug_fetch_stamps); a tuning build, never the shipped library
#ifdef BMV_RENDER_STAMPS
  unsigned long long st[8];
  int tile_no = 0;
#define RSTAMP(i)                                                       \
  {                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         \
    st[i] = __builtin_amdgcn_s_memtime();                               \
    __builtin_amdgcn_sched_barrier(0);                                  \
  }
#else
#define RSTAMP(i)
#endif
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    RSTAMP(0)
    int ray = a.ray_begin + tile * RAYS_PER_TILE + s / NS;
    bool valid = ray < a.ray_end;
    int rr = valid ? ray : a.ray_end - 1;
    const float* r = a.rays + ((size_t)b * a.N + rr) * 8;
    float o[3] = {r[0], r[1], r[2]}, d[3] = {r[3], r[4], r[5]};
    float px = r[6], py = r[7];
    float rn, rf, vn, vf;
    ray_bounds(depth, std_, nf, a.hv, a.wv, a.Hr, a.Wr, (int)px, (int)py, INV, rn, rf, vn, vf);
    float z, xyz[3], dn;
    sample_point(o, d, rn, rf, vn, vf, k, NS, INV, z, xyz, dn);

    float fin[NV][L::KF], dir[NV][4], vox[4], res[4];
    RSTAMP(1)
    {  // a9: trilinear lookup, this half's 4 channels
      Taps3 t3 = taps3_zeros(BMV_DIV(px, inv_w), BMV_DIV(py, inv_h), dn, a.wv, a.hv, a.Dv);
      const size_t cs = (size_t)a.Dv * hwv;
      // the half's channel offset (h * cs) goes into the 32-bit tap offsets once: every load is then
      // `wave-uniform plane pointer + 32-bit lane offset` instead of a 64-bit per-lane address
      const int hoff = h ? (int)cs : 0;
      if constexpr ((PK & 2) != 0) {
        // (Dv,hv,wv,8) records [ch 0 2 4 6 | ch 1 3 5 7]: this half's four channels of a tap are one 16-byte load
        float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 q = __builtin_bit_cast(
              float4, __builtin_amdgcn_raw_buffer_load_b128(rs_vol, (int)((unsigned)t3.o[k] * 32u + (h ? 16u : 0u)), 0, 0));
          acc.x += q.x * t3.w[k], acc.y += q.y * t3.w[k], acc.z += q.z * t3.w[k], acc.w += q.w * t3.w[k];
        }
        vox[0] = acc.x, vox[1] = acc.y, vox[2] = acc.z, vox[3] = acc.w;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) t3.o[k] = (t3.o[k] + hoff) * 4;   // byte offsets (a volume stays below 2 GiB)
#pragma unroll
        for (int j = 0; j < 4; ++j) vox[j] = tap3_fetch_buf(rs_vol, t3, (unsigned)(2 * j) * (unsigned)cs * 4u);
      }
    }
    BMV_FENCE();
    RSTAMP(2)
    float vis = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {  // a10 (+ a14)
      const Cam& cam = rc->cam[i];
      Taps2 t2 = project_taps(cam, xyz, a.Wr, a.Hr);
      if constexpr (PK != 0) {
        // lookup records: this half's four feature channels are one 16-byte load per tap, its two colour slots one
        // 8-byte load (taps outside the image carry weight 0 and offset 0, as in the planar path)
        static_assert(PK == 0 || FEAT_CH == 8, "packed records hold 8 feature channels");
        const unsigned of = h ? 16u : 0u, oc = 32u + (h ? 8u : 0u);
        const unsigned o[4] = {(unsigned)t2.o00 * 48u, (unsigned)t2.o01 * 48u, (unsigned)t2.o10 * 48u, (unsigned)t2.o11 * 48u};
        const float w[4] = {t2.w00, t2.w01, t2.w10, t2.w11};
        float4 fq[4];
        float2 cq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          fq[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_f[i], (int)(o[k] + of), 0, 0));
          cq[k] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_f[i], (int)(o[k] + oc), 0, 0));
        }
        float v0 = fq[0].x * w[0], v1 = fq[0].y * w[0], v2 = fq[0].z * w[0], v3 = fq[0].w * w[0];
        float c0 = cq[0].x * w[0], c1 = cq[0].y * w[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          v0 += fq[k].x * w[k], v1 += fq[k].y * w[k], v2 += fq[k].z * w[k], v3 += fq[k].w * w[k];
          c0 += cq[k].x * w[k], c1 += cq[k].y * w[k];
        }
        if (a.rgb_affine) c0 = c0 * 0.5f + 0.5f, c1 = c1 * 0.5f + 0.5f;
        if (h) c1 = 0.f;   // the last slot of the odd half is padding
        fin[i][0] = v0, fin[i][1] = v1, fin[i][2] = v2, fin[i][3] = v3, fin[i][4] = c0, fin[i][5] = c1;
      }
      Taps2 t2h = t2;  // same taps as byte offsets, one channel plane further for the odd half
      tap_bytes(t2h, h ? (int)plane : 0);
#pragma unroll
      for (int j = 0; j < (PK != 0 ? 0 : L::KFC); ++j) {
        int c = 2 * j + h;  // channel of [feature, rgb]
        float v = 0.f;
        if (2 * j + 1 < FEAT_CH) {  // both halves read a feature channel
          v = tap_fetch_buf(rs_f[i], t2h, (unsigned)(2 * j) * (unsigned)plane * 4u);
        } else if (2 * j >= FEAT_CH) {  // colour channels (the last slot of half 1 is padding)
          int cc = c - FEAT_CH;
          if (cc < 3) {
            v = tap_fetch_buf(rs_c[i], t2h, (unsigned)(2 * j - FEAT_CH) * (unsigned)plane * 4u);
            if (a.rgb_affine) v = v * 0.5f + 0.5f;
          }
        }
        fin[i][j] = v;
        BMV_FENCE_EVERY(j, 6);
      }
      dir_feature(xyz, rc->tar_c, cam.c, dir[i]);
      fin[i][L::KFC] = h ? dir[i][1] : dir[i][0];
      fin[i][L::KFC + 1] = h ? dir[i][3] : dir[i][2];
      if (a.mode == 1) vis += visible(cam, xyz, inv_w, inv_h);
      BMV_FENCE();
    }

    RSTAMP(3)
    mlp_forward<FEAT_CH, NV>(lds, lane, fin, dir, vox, res);
    RSTAMP(4)
#ifdef BMV_RENDER_STAMPS
    if (tile_no == 2 && lane == 0 && blockIdx.x < 512) {
      float* o = g_stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
      for (int i = 1; i < 5; ++i) o[i] = (float)(unsigned)(st[i] - st[0]);
      o[0] = 1.f;
    }
    if (tile_no == 3 && lane == 0 && blockIdx.x < 512)
      g_stamps[((size_t)blockIdx.x * 4 + wave) * 8 + 5] = (float)(unsigned)(st[0] - st[7]);
    st[7] = st[0];
    ++tile_no;
#endif

    if (a.mode == 1) {  // boost path: raw network output, depths and visibility, no compositing
      if (valid && h == 0) {
        size_t pt = ((size_t)b * a.N + ray) * NS + k;
        float4 o4 = {res[0], res[1], res[2], res[3]};
        reinterpret_cast<float4*>(a.out0)[pt] = o4;
        a.out1[pt] = z;
        a.out2[pt] = vis / (float)NV;
      }
      continue;
    }
    // a12: composite the NS samples of a ray; they sit on NS consecutive lanes.
    float alpha = 1.f - __expf(-res[3]);
    float om = 1.f - alpha + 1e-10f;
    float T = 1.f;  // exclusive product of om over samples before k
#pragma unroll
    for (int j = 1; j < NS; ++j) {
      float prev = __shfl_up(om, j, NS);
      if (k >= j) T *= prev;
    }
    float w = alpha * T;
    float c0 = w * res[0], c1 = w * res[1], c2 = w * res[2];
    float wmax = w;
#pragma unroll
    for (int m = 1; m < NS; m <<= 1) {
      c0 += __shfl_xor(c0, m, NS);
      c1 += __shfl_xor(c1, m, NS);
      c2 += __shfl_xor(c2, m, NS);
      wmax = fmaxf(wmax, __shfl_xor(wmax, m, NS));
    }
    float e = __expf(w - wmax), den = e;
#pragma unroll
    for (int m = 1; m < NS; m <<= 1) den += __shfl_xor(den, m, NS);
    float sm = BMV_DIV(e, den);
    float dz = sm * z, acc = sm;
#pragma unroll
    for (int m = 1; m < NS; m <<= 1) {
      dz += __shfl_xor(dz, m, NS);
      acc += __shfl_xor(acc, m, NS);
    }
    if (a.white_bkgd) c0 += 1.f - acc, c1 += 1.f - acc, c2 += 1.f - acc;
    if (valid && h == 0) {
      size_t ro = (size_t)b * a.N + ray;
      a.out2[ro * NS + k] = sm;
      if (k == 0) {
        a.out0[ro * 3] = c0, a.out0[ro * 3 + 1] = c1, a.out0[ro * 3 + 2] = c2;
        a.out1[ro] = dz;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Producer / consumer form of the fused renderer (round 3; lookup records for image AND volume, feat_ch 8).
//
// In the kernel above the two waves of a SIMD each run gather (geometry, 56 record loads, tap arithmetic: ~22 k cycles)
// and then the MLP (205 MFMAs = 13.2 k cycles of the matrix pipe) on their own tiles: tile period 48.6 k per wave, the
// matrix pipe 0.49-0.52 busy.  A first split (1 MLP wave + 2 gather waves per SIMD) showed that an MLP wave ALONE needs
// ~24 k cycles per tile -- between its MFMA chains sit ~1000 vector instructions and ~380 LDS reads (bias rows, ReLUs,
// the 1-wide heads, softmaxes, compositing) that depend on the chains' results -- so one MLP wave per SIMD cannot fill
// the pipe whatever feeds it.  Hence this shape: a workgroup is 8 MLP waves (waves 0-7: TWO per SIMD, one's vector
// sections under the other's MFMAs) and 4 gather waves (waves 8-11: one per SIMD).  A gather wave works on 64 samples =
// TWO tiles with lane = sample (the fused kernel computes every sample's geometry twice, once per lane half): 52
// 16-byte loads per 64 samples (whole 32-byte voxel records, whole 48-byte pixel records), then writes the two tiles'
// MLP inputs -- 36 dwords per MLP lane: fin[3][6], dir[3][4], vox[4], z, visibility, in the (sample, half) layout the
// MLP's B operands want -- into the mailboxes of MLP waves 2g, 2g + 1 and raises their flags.  An MLP wave copies its
// mailbox to registers, releases it, runs the MFMA chains and composites the ray.  No workgroup barrier after the
// prologue: flags in LDS (in-order per wave), bounded spins (a protocol error ends the kernel with NaNs, not a hang).
// Same arithmetic per value as render_rays_kernel (the same device functions, the same tap order): bit-identical.
// ---------------------------------------------------------------------------------------------------------------
#ifndef BMV_RENDER_PC_SPIN
#define BMV_RENDER_PC_SPIN (1 << 22)   // polls before a wave gives up
#endif
#ifndef BMV_RENDER_PC_PRIO
#define BMV_RENDER_PC_PRIO 0
#endif
#ifndef BMV_RENDER_PC_GATHER
#define BMV_RENDER_PC_GATHER 4   // gather waves per workgroup: 4 (one per SIMD, 168 registers) or 8 (two, 128 registers)
#endif
constexpr int kPcMlp = 8, kPcGather = BMV_RENDER_PC_GATHER;   // waves per role
// dwords per MLP lane of a mailbox: fin[NV][6], dir[NV][4], vox[4], z, visibility
constexpr int pc_box(int nv) { return 10 * nv + 6; }

#ifdef BMV_RENDER_PC_COUNT
__device__ unsigned long long g_pc_spins[4];   // tuning: polls that found the flag not ready / waits, per role
#endif
// A wave that gives up on a mailbox flag (a lost wake-up: a protocol error, or a wedged partner) ends its tile with
// unwritten / NaN pixels; it also COUNTS here, and bmv_render_pc_check (include/bmv.h) turns a non-zero count into an
// error with a message.  g_pc_inject (bmv_debug_render_pc_inject, tests): the gather role of workgroup 0 withholds one
// wake-up.
__device__ unsigned g_pc_faults;
__device__ int g_pc_inject;
__device__ __forceinline__ bool pc_wait(volatile int* flag, int want) {
  for (int it = 0; it < BMV_RENDER_PC_SPIN; ++it) {
    if (*flag == want) {
#ifdef BMV_RENDER_PC_COUNT
      if ((threadIdx.x & 31) == 0) {
        atomicAdd(&g_pc_spins[want & 1], (unsigned long long)it);
        atomicAdd(&g_pc_spins[2 + (want & 1)], 1ull);
      }
#endif
      // compiler barrier: the plain LDS loads / stores of the mailbox that follow a successful poll must not be
      // hoisted above it (the flag is volatile, the boxes are not)
      asm volatile("" ::: "memory");
      return true;
    }
    __builtin_amdgcn_s_sleep(4);
  }
  atomicAdd(&g_pc_faults, 1u);
  return false;
}

// CS: the MLP's two-tile chains on the bf16 matrix pipe with three-piece fp32 operands (mlp.hpp, CSPLIT): the default since
// the end of round 5; bmv_tuning BMV_RENDER_SPLIT=0 puts every chain back on fp32 MFMAs
template <int NS, bool INV, int NV, bool CS = false>
__global__ void __launch_bounds__(64 * (kPcMlp + kPcGather), 1) render_pc_kernel(RenderArgsDev a) {
  resolve_deferred(a);
  constexpr int FEAT_CH = 8;
  using L = MlpLayout<FEAT_CH>;
  static_assert(32 % NS == 0, "samples per ray must divide 32");
  static_assert(L::KFC == 6 && L::KF == 8, "mailbox layout: 6 feature / colour slots + 2 direction slots per view");
  static_assert(kPcGather == 4 || kPcGather == 8, "pair p goes to MLP waves 2 (p % 4), 2 (p % 4) + 1");
  // mailbox rows: [0, 6 NV) fin, [6 NV, 10 NV) dir, then vox[4], z, visibility
  constexpr int kPcBox = pc_box(NV), B_DIR = 6 * NV, B_VOX = 10 * NV, B_Z = 10 * NV + 4, B_VIS = 10 * NV + 5;
  constexpr int RAYS_PER_TILE = 32 / NS;
  constexpr int BLOB = CS ? L::LDS_S : L::TOTAL;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  RenderCams* rc = reinterpret_cast<RenderCams*>(lds + BLOB);
  // [kPcMlp] sequence words: 2 n = empty, waiting for this MLP wave's n-th job; 2 n + 1 = holds it (with more gather
  // waves than pairs of MLP waves a mailbox has two producers taking turns: the count keeps them in order)
  int* flags = reinterpret_cast<int*>(rc + 1);
  float* boxes = reinterpret_cast<float*>(flags + 16);              // [kPcMlp][kPcBox][64 MLP lanes]
  const int b = blockIdx.y;
  blob_to_lds<L, CS>(lds, a.blob);
  if (threadIdx.x < NV)
    load_cam(a.src_exts + ((size_t)b * NV + threadIdx.x) * 16, a.src_ixts + ((size_t)b * NV + threadIdx.x) * 9,
             a.render_scale, rc->cam[threadIdx.x]);
  if (threadIdx.x == NV) camera_centre(a.tar_ext + (size_t)b * 16, rc->tar_c);
  if (threadIdx.x >= 64 && threadIdx.x < 64 + 16) flags[threadIdx.x - 64] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = lane & 31;
  const int k = s % NS;  // sample index along the ray
  const int nrays = a.ray_end - a.ray_begin;
  const int ntiles = (nrays + RAYS_PER_TILE - 1) / RAYS_PER_TILE;
  // tiles of this workgroup: blockIdx.x + gridDim.x * j, j = 0 .. njobs - 1; job j belongs to MLP wave j % kPcMlp and
  // is produced, together with job j ^ 1, by gather wave (j >> 1) % kPcGather
  const int njobs = ntiles > (int)blockIdx.x ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const float inv_w = (float)(a.Wr - 1), inv_h = (float)(a.Hr - 1);

  if (wave >= kPcMlp) {
    // =============================================================================================== gather waves
#ifdef BMV_RENDER_PC_GPRIO
    __builtin_amdgcn_s_setprio(BMV_RENDER_PC_GPRIO);
#endif
    const int g = wave - kPcMlp;
    const int half_tile = lane >> 5;                      // which of the pair's two tiles this lane's sample is in
    const size_t hwv = (size_t)a.hv * a.wv;
    const size_t plane = (size_t)a.Hr * a.Wr;
    const float* depth = a.depth + b * hwv;
    const float* std_ = a.std + b * hwv;
    const float* nf = a.near_far + b * 2 * hwv;
    const float* vol = a.volume + (size_t)b * 8 * a.Dv * hwv;
    const __amdgpu_buffer_rsrc_t rs_vol = make_rsrc(vol, (size_t)8 * a.Dv * hwv * 4);
    __amdgpu_buffer_rsrc_t rs_f[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const size_t vslot = a.view_ids ? (size_t)b * a.n_all + a.view_ids[b * NV + i] : (size_t)b * NV + i;
      rs_f[i] = make_rsrc(a.im_packed + vslot * 12 * plane, (size_t)12 * plane * 4);   // 48-byte records
    }
    for (int p = g; 2 * p < njobs; p += kPcGather) {
#ifdef BMV_RENDER_PC_FAKE_GATHER   // tuning: the MLP side alone (mailboxes filled with a constant, no lookups)
      {
        const int j_ = 2 * p + half_tile;
        const int mw_ = 2 * (p & 3) + half_tile;
        volatile int* fl_ = flags + mw_;
        float* bx_ = boxes + (size_t)mw_ * kPcBox * 64 + s;
        if (j_ < njobs) {
          if (!pc_wait(fl_, 2 * (p >> 2))) return;
          for (int q = 0; q < kPcBox; ++q) bx_[q * 64] = 0.25f, bx_[q * 64 + 32] = 0.25f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j_ < njobs && s == 0) *fl_ = 2 * (p >> 2) + 1;
        continue;
      }
#endif
      const int j = 2 * p + half_tile;                    // this lane's job; the odd one of the last pair may not exist
      const bool job = j < njobs;
      const int mw = 2 * (p & 3) + half_tile;             // the MLP wave that takes it, as its job number p >> 2
      const int seq = 2 * (p >> 2);
      volatile int* flag = flags + mw;                    // (lanes 0 / 32 raise the two flags)
      float* box = boxes + (size_t)mw * kPcBox * 64 + s;  // MLP lane (s, h): + 32 h
      const int tile = (int)blockIdx.x + (int)gridDim.x * (job ? j : 2 * p);
      int ray = a.ray_begin + tile * RAYS_PER_TILE + s / NS;
      int rr = ray < a.ray_end ? ray : a.ray_end - 1;
      const float* r = a.rays + ((size_t)b * a.N + rr) * 8;
      float o[3] = {r[0], r[1], r[2]}, d[3] = {r[3], r[4], r[5]};
      float px = r[6], py = r[7];
      float rn, rf, vn, vf;
      ray_bounds(depth, std_, nf, a.hv, a.wv, a.Hr, a.Wr, (int)px, (int)py, INV, rn, rf, vn, vf);
      float z, xyz[3], dn;
      sample_point(o, d, rn, rf, vn, vf, k, NS, INV, z, xyz, dn);
      // The mailbox is written as the values are produced (holding both halves of all three views costs ~100
      // registers): it must be empty NOW.  The consumer empties it right after copying it to registers, ~300 cycles into
      // its 20 k-cycle tile, so this wait is almost always over before it starts.  (The poll is per lane; every lane of
      // a half polls the same word.)
      if (job && !pc_wait(flag, seq)) return;
      {  // a9: trilinear lookup from (Dv,hv,wv,8) records [ch 0 2 4 6 | ch 1 3 5 7]: a tap is two 16-byte loads
        Taps3 t3 = taps3_zeros(BMV_DIV(px, inv_w), BMV_DIV(py, inv_h), dn, a.wv, a.hv, a.Dv);
        float4 e = {0.f, 0.f, 0.f, 0.f}, od = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float4 q0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_vol, (int)((unsigned)t3.o[kk] * 32u), 0, 0));
          const float4 q1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_vol, (int)((unsigned)t3.o[kk] * 32u + 16u), 0, 0));
          e.x += q0.x * t3.w[kk], e.y += q0.y * t3.w[kk], e.z += q0.z * t3.w[kk], e.w += q0.w * t3.w[kk];
          od.x += q1.x * t3.w[kk], od.y += q1.y * t3.w[kk], od.z += q1.z * t3.w[kk], od.w += q1.w * t3.w[kk];
          if (kk == 3) BMV_FENCE();
        }
        if (job) {
          float* bv = box + B_VOX * 64;
          bv[0 * 64] = e.x, bv[1 * 64] = e.y, bv[2 * 64] = e.z, bv[3 * 64] = e.w;
          bv[0 * 64 + 32] = od.x, bv[1 * 64 + 32] = od.y, bv[2 * 64 + 32] = od.z, bv[3 * 64 + 32] = od.w;
        }
      }
      BMV_FENCE();
      float vis = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {  // a10 (+ a14)
        const Cam& cam = rc->cam[i];
        Taps2 t2 = project_taps(cam, xyz, a.Wr, a.Hr);
        const unsigned ob[4] = {(unsigned)t2.o00 * 48u, (unsigned)t2.o01 * 48u, (unsigned)t2.o10 * 48u, (unsigned)t2.o11 * 48u};
        const float w[4] = {t2.w00, t2.w01, t2.w10, t2.w11};
        float4 fe[4], fo[4], cc[4];   // even channels | odd channels | r b g 0 of the four taps
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          fe[kk] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_f[i], (int)ob[kk], 0, 0));
          fo[kk] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_f[i], (int)(ob[kk] + 16u), 0, 0));
          cc[kk] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_f[i], (int)(ob[kk] + 32u), 0, 0));
        }
        float4 ve = {fe[0].x * w[0], fe[0].y * w[0], fe[0].z * w[0], fe[0].w * w[0]};
        float4 vo = {fo[0].x * w[0], fo[0].y * w[0], fo[0].z * w[0], fo[0].w * w[0]};
        float cr = cc[0].x * w[0], cb = cc[0].y * w[0], cg = cc[0].z * w[0];
#pragma unroll
        for (int kk = 1; kk < 4; ++kk) {
          ve.x += fe[kk].x * w[kk], ve.y += fe[kk].y * w[kk], ve.z += fe[kk].z * w[kk], ve.w += fe[kk].w * w[kk];
          vo.x += fo[kk].x * w[kk], vo.y += fo[kk].y * w[kk], vo.z += fo[kk].z * w[kk], vo.w += fo[kk].w * w[kk];
          cr += cc[kk].x * w[kk], cb += cc[kk].y * w[kk], cg += cc[kk].z * w[kk];
        }
        if (a.rgb_affine) cr = cr * 0.5f + 0.5f, cb = cb * 0.5f + 0.5f, cg = cg * 0.5f + 0.5f;
        float dirv[4];
        dir_feature(xyz, rc->tar_c, cam.c, dirv);
        if (a.mode == 1) vis += visible(cam, xyz, inv_w, inv_h);
        if (job) {   // MLP lane half 0: even channels, (r, b); half 1: odd channels, (g, 0)
          float* bx = box + (i * 6) * 64;
          bx[0 * 64] = ve.x, bx[1 * 64] = ve.y, bx[2 * 64] = ve.z, bx[3 * 64] = ve.w, bx[4 * 64] = cr, bx[5 * 64] = cb;
          bx[0 * 64 + 32] = vo.x, bx[1 * 64 + 32] = vo.y, bx[2 * 64 + 32] = vo.z, bx[3 * 64 + 32] = vo.w;
          bx[4 * 64 + 32] = cg, bx[5 * 64 + 32] = 0.f;
          float* bd = box + (B_DIR + i * 4) * 64;
#pragma unroll
          for (int q = 0; q < 4; ++q) bd[q * 64] = dirv[q], bd[q * 64 + 32] = dirv[q];
        }
        BMV_FENCE();
      }
      if (job) {
        box[B_Z * 64] = z, box[B_Z * 64 + 32] = z;
        box[B_VIS * 64] = vis, box[B_VIS * 64 + 32] = vis;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (job && s == 0 && !(g_pc_inject && blockIdx.x == 0 && blockIdx.y == 0 && p == 0)) *flag = seq + 1;
    }
    return;
  }

  // ================================================================================================= MLP waves
  __builtin_amdgcn_s_setprio(BMV_RENDER_PC_PRIO);
  const int h = lane >> 5;
  const int m = wave;
  volatile int* flag = flags + m;
  const float* box = boxes + (size_t)m * kPcBox * 64 + lane;
  int seq = 0;
  for (int j = m; j < njobs; j += kPcMlp, seq += 2) {
    const int tile = (int)blockIdx.x + (int)gridDim.x * j;
    int ray = a.ray_begin + tile * RAYS_PER_TILE + s / NS;
    bool valid = ray < a.ray_end;
    float fin[NV][L::KF], dir[NV][4], vox[4], res[4];
    bool ok = pc_wait(flag, seq + 1);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int q = 0; q < 6; ++q) fin[i][q] = box[(i * 6 + q) * 64];
#pragma unroll
      for (int q = 0; q < 4; ++q) dir[i][q] = box[(B_DIR + i * 4 + q) * 64];
      fin[i][L::KFC] = h ? dir[i][1] : dir[i][0];
      fin[i][L::KFC + 1] = h ? dir[i][3] : dir[i][2];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) vox[q] = box[(B_VOX + q) * 64];
    const float z = box[B_Z * 64];
    const float vis = box[B_VIS * 64];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *flag = seq + 2;     // the next producer may refill while this tile runs through the MLP
    // (A pieces read one group ahead where the 12 registers that takes do not spill: mlp.hpp BMV_SPLIT_AHEAD)
    mlp_forward<FEAT_CH, NV, CS, (NV == 2 || (NV == 3 && NS <= 2))>(lds, lane, fin, dir, vox, res);
    if (!ok) res[0] = res[1] = res[2] = res[3] = __builtin_nanf("");   // protocol error: loud, not silent

    if (a.mode == 1) {  // boost path: raw network output, depths and visibility, no compositing
      if (valid && h == 0) {
        size_t pt = ((size_t)b * a.N + ray) * NS + k;
        float4 o4 = {res[0], res[1], res[2], res[3]};
        reinterpret_cast<float4*>(a.out0)[pt] = o4;
        a.out1[pt] = z;
        a.out2[pt] = vis / (float)NV;
      }
      continue;
    }
    // a12: composite the NS samples of a ray; they sit on NS consecutive lanes.
    float alpha = 1.f - __expf(-res[3]);
    float om = 1.f - alpha + 1e-10f;
    float T = 1.f;  // exclusive product of om over samples before k
#pragma unroll
    for (int jj = 1; jj < NS; ++jj) {
      float prev = __shfl_up(om, jj, NS);
      if (k >= jj) T *= prev;
    }
    float w = alpha * T;
    float c0 = w * res[0], c1 = w * res[1], c2 = w * res[2];
    float wmax = w;
#pragma unroll
    for (int mm = 1; mm < NS; mm <<= 1) {
      c0 += __shfl_xor(c0, mm, NS);
      c1 += __shfl_xor(c1, mm, NS);
      c2 += __shfl_xor(c2, mm, NS);
      wmax = fmaxf(wmax, __shfl_xor(wmax, mm, NS));
    }
    float ee = __expf(w - wmax), den = ee;
#pragma unroll
    for (int mm = 1; mm < NS; mm <<= 1) den += __shfl_xor(den, mm, NS);
    float sm = BMV_DIV(ee, den);
    float dz = sm * z, acc = sm;
#pragma unroll
    for (int mm = 1; mm < NS; mm <<= 1) {
      dz += __shfl_xor(dz, mm, NS);
      acc += __shfl_xor(acc, mm, NS);
    }
    if (a.white_bkgd) c0 += 1.f - acc, c1 += 1.f - acc, c2 += 1.f - acc;
    if (valid && h == 0) {
      size_t ro = (size_t)b * a.N + ray;
      a.out2[ro * NS + k] = sm;
      if (k == 0) {
        a.out0[ro * 3] = c0, a.out0[ro * 3 + 1] = c1, a.out0[ro * 3 + 2] = c2;
        a.out1[ro] = dz;
      }
    }
  }
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? 0 : -1;
}

}  // namespace bmv

using namespace bmv;

extern "C" {

// lost wake-ups of the producer / consumer renderer since the last reset: synchronises the device.  0 -> BMV_OK;
// otherwise BMV_ERR_LAUNCH with the count in bmv_last_error (the frames rendered since then hold unwritten pixels)
int bmv_render_pc_check(int reset) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_pc_faults), sizeof(n)) != hipSuccess) {
    set_error("bmv_render_pc_check: cannot read the fault counter: %s", hipGetErrorString(hipGetLastError()));
    return BMV_ERR_LAUNCH;
  }
  if (reset && n) {
    const unsigned zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pc_faults), &zero, sizeof(zero));
  }
  if (n) {
    set_error("render_pc_kernel: %u lane(s) gave up waiting for a mailbox flag (lost wake-up): the frames rendered since "
              "the last check hold unwritten pixels", n);
    return BMV_ERR_LAUNCH;
  }
  return BMV_OK;
}
int bmv_debug_render_pc_inject(int on) {   // tests: withhold one wake-up in workgroup 0
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pc_inject), &on, sizeof(on)) == hipSuccess ? BMV_OK : BMV_ERR_LAUNCH;
}

#ifdef BMV_RENDER_PC_COUNT
int bmv_debug_fetch_pc_spins(unsigned long long* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pc_spins), sizeof(unsigned long long) * 4);
}
#endif
#ifdef BMV_RENDER_STAMPS
int bmv_debug_fetch_stamps(float* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(float) * 512 * 4 * 8);
}
#endif

int bmv_nerf_blob_size(int feat_ch) {
  if (feat_ch == 8) return MlpLayout<8>::TOTAL_S;       // (the fp32 tables + the pre-split tables of the bf16 x 3 chains)
  if (feat_ch == 32) return MlpLayout<32>::TOTAL_S;
  set_error("bmv_nerf_blob_size: feat_ch=%d unsupported (8 or 32)", feat_ch);
  return BMV_ERR_UNSUPPORTED;
}

int bmv_nerf_pack_weights(const bmv_nerf_params* p, int feat_ch, float* blob, bmv_stream_t stream) {
  BMV_REQUIRE(p && blob, "bmv_nerf_pack_weights: null pointer");
  const float* const* pp = reinterpret_cast<const float* const*>(p);
  for (int i = 0; i < 16; ++i) BMV_REQUIRE(pp[i], "bmv_nerf_pack_weights: parameter %d is null", i);
  if (feat_ch == 8)
    hipLaunchKernelGGL(nerf_pack_kernel<8>, dim3(cdiv(MlpLayout<8>::TOTAL_S, 256)), dim3(256), 0, as_stream(stream), *p,
                       blob);
  else if (feat_ch == 32)
    hipLaunchKernelGGL(nerf_pack_kernel<32>, dim3(cdiv(MlpLayout<32>::TOTAL_S, 256)), dim3(256), 0, as_stream(stream),
                       *p, blob);
  else {
    set_error("bmv_nerf_pack_weights: feat_ch=%d unsupported (8 or 32)", feat_ch);
    return BMV_ERR_UNSUPPORTED;
  }
  BMV_LAUNCH_END("bmv_nerf_pack_weights");
}

int bmv_nerf_mlp_fwd(const float* vox_feat, const float* img, const float* blob, int feat_ch, int S, long npts,
                     float* out, bmv_stream_t stream) {
  BMV_REQUIRE(vox_feat && img && blob && out, "bmv_nerf_mlp_fwd: null pointer");
  BMV_REQUIRE(npts >= 0, "bmv_nerf_mlp_fwd: npts=%ld", npts);
  if (npts == 0) return BMV_OK;
  long ntiles = (npts + 31) / 32;
  unsigned grid = (unsigned)((ntiles + 3) / 4 < 1024 ? (ntiles + 3) / 4 : 1024);
#define MLP_CASE(FC, NVV)                                                                                            \
  if (feat_ch == FC && S == NVV) {                                                                                   \
    size_t lds = MlpLayout<FC>::TOTAL * 4;                                                                           \
    BMV_REQUIRE(set_lds(nerf_mlp_kernel<FC, NVV>, lds) == 0, "bmv_nerf_mlp_fwd: cannot reserve %zu B of LDS", lds);  \
    hipLaunchKernelGGL((nerf_mlp_kernel<FC, NVV>), dim3(grid), dim3(256), lds, as_stream(stream), vox_feat, img, blob, \
                       npts, out);                                                                                   \
    BMV_LAUNCH_END("bmv_nerf_mlp_fwd");                                                                              \
  }
  if (feat_ch == 8 && bmv::tuning("BMV_RENDER_SPLIT", 1)) {   // the two-tile chains as bf16 x 3, as in the fused renderer
    size_t lds = MlpLayout<8>::LDS_S * 4;
#define MLP_SPLIT_CASE(NVV)                                                                                          \
  if (S == NVV) {                                                                                                    \
    BMV_REQUIRE(set_lds(nerf_mlp_kernel<8, NVV, true>, lds) == 0, "bmv_nerf_mlp_fwd: cannot reserve %zu B of LDS", lds); \
    hipLaunchKernelGGL((nerf_mlp_kernel<8, NVV, true>), dim3(grid), dim3(256), lds, as_stream(stream), vox_feat, img, blob, \
                       npts, out);                                                                                   \
    BMV_LAUNCH_END("bmv_nerf_mlp_fwd");                                                                              \
  }
    MLP_SPLIT_CASE(3)
    MLP_SPLIT_CASE(2)
    MLP_SPLIT_CASE(4)
#undef MLP_SPLIT_CASE
  }
  MLP_CASE(8, 3)
  MLP_CASE(32, 3)
  MLP_CASE(8, 2)
  MLP_CASE(32, 2)
  MLP_CASE(8, 4)
  MLP_CASE(32, 4)
#undef MLP_CASE
  set_error("bmv_nerf_mlp_fwd: feat_ch=%d (8 or 32) with S=%d source views (2, 3 or 4) unsupported", feat_ch, S);
  return BMV_ERR_UNSUPPORTED;
}

// workgroups of the fused renderer: 2 are resident per CU (launch bounds), tiles are walked grid-stride
static unsigned render_grid() {
  const unsigned v = (unsigned)bmv::tuning("BMV_RENDER_GRID", (int)(256u * BMV_RENDER_WPS));   // every workgroup is resident from the start
  return v;
}

int bmv_render_rays_fwd(const bmv_render_args* a, bmv_stream_t stream) {
  const unsigned kRenderGrid = render_grid();
  BMV_REQUIRE(a, "bmv_render_rays_fwd: null args");
  BMV_REQUIRE(a->rays && a->depth && a->std && a->near_far && a->volume && (a->im_packed || (a->im_feat && a->rgb_src)) &&
                  a->src_exts && a->src_ixts && a->tar_ext && a->blob && a->out0 && a->out1 && a->out2,
              "bmv_render_rays_fwd: null pointer");
  BMV_REQUIRE(a->S >= 2 && a->S <= kMaxViews, "bmv_render_rays_fwd: S=%d source views (2, 3 or 4)", a->S);
  BMV_REQUIRE(a->B > 0 && a->N > 0 && a->hv > 0 && a->wv > 0 && a->Dv > 0 && a->Hr > 1 && a->Wr > 1,
              "bmv_render_rays_fwd: bad shape");
  BMV_REQUIRE(a->ray_begin >= 0 && a->ray_end <= a->N && a->ray_begin <= a->ray_end,
              "bmv_render_rays_fwd: ray range [%d,%d) outside [0,%d)", a->ray_begin, a->ray_end, a->N);
  BMV_REQUIRE(a->mode == 0 || a->mode == 1, "bmv_render_rays_fwd: mode=%d", a->mode);
  BMV_REQUIRE(a->view_ids == nullptr || a->n_all >= a->S, "bmv_render_rays_fwd: view_ids with n_all=%d < S", a->n_all);
  if (a->ray_begin == a->ray_end) {   // an empty ray shard / chunk: nothing is launched, so nothing is baked into a capture --
    (void)deferred_finish();          // the deferrals registered for this launch are simply dropped
    return BMV_OK;
  }
  int nrays = a->ray_end - a->ray_begin;
  RenderArgsDev dev;
  static_cast<bmv_render_args&>(dev) = *a;
  {
    const DeferredPtr dr = deferred_for(a->rays), d0 = deferred_for(a->out0), d1 = deferred_for(a->out1), d2 = deferred_for(a->out2);
    dev.table = dr.table ? dr.table : d0.table ? d0.table : d1.table ? d1.table : d2.table;
    BMV_REQUIRE((!dr.table || dr.table == dev.table) && (!d0.table || d0.table == dev.table) &&
                    (!d1.table || d1.table == dev.table) && (!d2.table || d2.table == dev.table),
                "bmv_render_rays_fwd: the deferred pointers of one launch must share one table");
    dev.s_rays = dr.slot, dev.s_out0 = d0.slot, dev.s_out1 = d1.slot, dev.s_out2 = d2.slot;
  }
  // producer / consumer form (lookup records for image and volume, feat_ch 8): BMV_RENDER_PC=0 keeps the kernel above
  const bool use_pc = bmv::tuning("BMV_RENDER_PC", 1) != 0;
  // the MLP's two-tile chains on the bf16 pipe with three-piece fp32 operands (mlp.hpp CSPLIT; fp32 accuracy: tests,
  // profiles/r5/mlp_split_accuracy.txt); BMV_RENDER_SPLIT=0: every chain on fp32 MFMAs
  const bool split = bmv::tuning("BMV_RENDER_SPLIT", 1) != 0;
#define RENDER_CASE_PC(NSV, NVV)                                                                                     \
  if (use_pc && a->im_packed && a->vol_packed && a->feat_ch == 8 && a->Ns == NSV && a->depth_inv == 0 && a->S == NVV) { \
    size_t lds = (split ? MlpLayout<8>::LDS_S : MlpLayout<8>::TOTAL) * 4 + sizeof(RenderCams) + 64 +                \
                 (size_t)kPcMlp * pc_box(NVV) * 64 * 4;                                                             \
    BMV_REQUIRE((split ? set_lds(render_pc_kernel<NSV, false, NVV, true>, lds) : set_lds(render_pc_kernel<NSV, false, NVV>, lds)) == 0, \
                "bmv_render_rays_fwd: cannot reserve LDS");                                                         \
    int ntiles = (nrays + (32 / NSV) - 1) / (32 / NSV);                                                             \
    const unsigned pc_grid = (unsigned)bmv::tuning("BMV_RENDER_PC_GRID", (int)(256u)); \
    unsigned grid = (unsigned)ntiles < pc_grid ? (unsigned)ntiles : pc_grid;                                        \
    if (split)                                                                                                       \
      hipLaunchKernelGGL((render_pc_kernel<NSV, false, NVV, true>), dim3(grid, a->B), dim3(64 * (kPcMlp + kPcGather)), lds, \
                         as_stream(stream), dev);                                                                    \
    else                                                                                                             \
      hipLaunchKernelGGL((render_pc_kernel<NSV, false, NVV>), dim3(grid, a->B), dim3(64 * (kPcMlp + kPcGather)), lds, \
                         as_stream(stream), dev);                                                                    \
    BMV_LAUNCH_END("bmv_render_rays_fwd");                                                                          \
  }
  RENDER_CASE_PC(2, 3)
  RENDER_CASE_PC(1, 3)
  RENDER_CASE_PC(4, 3)
  RENDER_CASE_PC(8, 3)
  RENDER_CASE_PC(2, 2)      // 2 / 4 source views (ENeRF pre-training, dtu_pretrain.yaml:22-23; test_input_views)
  RENDER_CASE_PC(2, 4)
  RENDER_CASE_PC(1, 2)      // ... at every sample count the 3-view form has (S = 2 / 4 x Ns = 1 / 4 / 8 failed)
  RENDER_CASE_PC(4, 2)
  RENDER_CASE_PC(8, 2)
  RENDER_CASE_PC(1, 4)
  RENDER_CASE_PC(4, 4)
  RENDER_CASE_PC(8, 4)
#undef RENDER_CASE_PC
#define RENDER_CASE_PK(FC, NSV, INVV, PKV, NVV)                                                                     \
  if (a->im_packed && (a->vol_packed ? 3 : 1) == PKV && a->feat_ch == FC && a->Ns == NSV && (a->depth_inv != 0) == INVV && \
      a->S == NVV) {                                                                                                \
    size_t lds = MlpLayout<FC>::TOTAL * 4 + sizeof(RenderCams);                                                     \
    BMV_REQUIRE(set_lds(render_rays_kernel<FC, NSV, INVV, PKV, NVV>, lds) == 0, "bmv_render_rays_fwd: cannot reserve LDS"); \
    int ntiles = (nrays + (32 / NSV) - 1) / (32 / NSV);                                                             \
    unsigned grid = (unsigned)((ntiles + 3) / 4 < kRenderGrid ? (ntiles + 3) / 4 : kRenderGrid);                    \
    hipLaunchKernelGGL((render_rays_kernel<FC, NSV, INVV, PKV, NVV>), dim3(grid, a->B), dim3(256), lds, as_stream(stream), dev); \
    BMV_LAUNCH_END("bmv_render_rays_fwd");                                                                          \
  }
  RENDER_CASE_PK(8, 2, false, 1, 3)
  RENDER_CASE_PK(8, 2, false, 3, 3)
  RENDER_CASE_PK(8, 1, false, 1, 3)
  RENDER_CASE_PK(8, 1, false, 3, 3)
  RENDER_CASE_PK(8, 4, false, 1, 3)
  RENDER_CASE_PK(8, 4, false, 3, 3)
  RENDER_CASE_PK(8, 8, false, 1, 3)
  RENDER_CASE_PK(8, 8, false, 3, 3)
  RENDER_CASE_PK(8, 2, false, 1, 2)
  RENDER_CASE_PK(8, 2, false, 3, 2)
  RENDER_CASE_PK(8, 2, false, 1, 4)
  RENDER_CASE_PK(8, 2, false, 3, 4)
  RENDER_CASE_PK(8, 1, false, 1, 2)
  RENDER_CASE_PK(8, 1, false, 3, 2)
  RENDER_CASE_PK(8, 4, false, 1, 2)
  RENDER_CASE_PK(8, 4, false, 3, 2)
  RENDER_CASE_PK(8, 8, false, 1, 2)
  RENDER_CASE_PK(8, 8, false, 3, 2)
  RENDER_CASE_PK(8, 1, false, 1, 4)
  RENDER_CASE_PK(8, 1, false, 3, 4)
  RENDER_CASE_PK(8, 4, false, 1, 4)
  RENDER_CASE_PK(8, 4, false, 3, 4)
  RENDER_CASE_PK(8, 8, false, 1, 4)
  RENDER_CASE_PK(8, 8, false, 3, 4)
#undef RENDER_CASE_PK
  BMV_REQUIRE(!a->im_packed && !a->vol_packed,
              "bmv_render_rays_fwd: no lookup-record kernel for feat_ch=%d Ns=%d depth_inv=%d S=%d (volume records: %d)",
              a->feat_ch, a->Ns, a->depth_inv, a->S, a->vol_packed);
#define RENDER_CASE(FC, NSV, INVV, NVV)                                                                             \
  if (a->feat_ch == FC && a->Ns == NSV && (a->depth_inv != 0) == INVV && a->S == NVV) {                             \
    size_t lds = MlpLayout<FC>::TOTAL * 4 + sizeof(RenderCams);                                                     \
    BMV_REQUIRE(set_lds(render_rays_kernel<FC, NSV, INVV, 0, NVV>, lds) == 0, "bmv_render_rays_fwd: cannot reserve LDS"); \
    int ntiles = (nrays + (32 / NSV) - 1) / (32 / NSV);                                                             \
    unsigned grid = (unsigned)((ntiles + 3) / 4 < kRenderGrid ? (ntiles + 3) / 4 : kRenderGrid);                    \
    hipLaunchKernelGGL((render_rays_kernel<FC, NSV, INVV, 0, NVV>), dim3(grid, a->B), dim3(256), lds, as_stream(stream), dev); \
    BMV_LAUNCH_END("bmv_render_rays_fwd");                                                                          \
  }
  RENDER_CASE(8, 2, false, 3)
  RENDER_CASE(32, 8, true, 3)
  RENDER_CASE(8, 1, false, 3)
  RENDER_CASE(8, 4, false, 3)
  RENDER_CASE(8, 8, false, 3)
  RENDER_CASE(32, 2, true, 3)
  RENDER_CASE(32, 4, true, 3)
  RENDER_CASE(8, 2, false, 2)
  RENDER_CASE(32, 8, true, 2)
  RENDER_CASE(8, 2, false, 4)
  RENDER_CASE(32, 8, true, 4)
  RENDER_CASE(8, 1, false, 2)
  RENDER_CASE(8, 4, false, 2)
  RENDER_CASE(8, 8, false, 2)
  RENDER_CASE(32, 2, true, 2)
  RENDER_CASE(32, 4, true, 2)
  RENDER_CASE(8, 1, false, 4)
  RENDER_CASE(8, 4, false, 4)
  RENDER_CASE(8, 8, false, 4)
  RENDER_CASE(32, 2, true, 4)
  RENDER_CASE(32, 4, true, 4)
#undef RENDER_CASE
  set_error("bmv_render_rays_fwd: no kernel for feat_ch=%d Ns=%d depth_inv=%d S=%d (built: feat_ch 8 / linear depth with Ns 1, 2, 4, 8; "
            "feat_ch 32 / inverse depth with Ns 2, 4, 8; S 2, 3, 4 each)", a->feat_ch, a->Ns, a->depth_inv, a->S);
  return BMV_ERR_UNSUPPORTED;
}

}  // extern "C"
