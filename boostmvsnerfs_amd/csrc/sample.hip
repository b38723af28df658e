// Stand-alone entry points for the sampler / lookup / compositing functions
// (a6-a10, a12, a14, a16).  The fused renderer (render.hip) reuses the same
// device functions; these kernels keep the reference's functional API callable
// piece by piece.  Reference: lib/networks/enerf/utils.py:392-520, 605-676, 753-786.
#include "render_geom.hpp"

namespace bmv {

__global__ void build_rays_kernel(const float* __restrict__ rays, const float* __restrict__ depth,
                                  const float* __restrict__ std_, const float* __restrict__ near_far, int N, int hv,
                                  int wv, int Hr, int Wr, int depth_inv, float* __restrict__ out) {
  int b = blockIdx.y;
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* r = rays + ((size_t)b * N + n) * 8;
  float* o = out + ((size_t)b * N + n) * 12;
  size_t hw = (size_t)hv * wv;
  float rn, rf, vn, vf;
  ray_bounds(depth + b * hw, std_ + b * hw, near_far + b * 2 * hw, hv, wv, Hr, Wr, (int)r[6], (int)r[7],
             depth_inv != 0, rn, rf, vn, vf);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = r[k];
  o[8] = rn, o[9] = rf, o[10] = vn, o[11] = vf;
}

__global__ void sample_along_depth_kernel(const float* __restrict__ rays, long total, int Ns, int depth_inv,
                                          float* __restrict__ xyz, float* __restrict__ uvd, float* __restrict__ zv) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  long ray = i / Ns;
  int k = (int)(i - ray * Ns);
  const float* r = rays + ray * 12;
  float z, p[3], dn;
  sample_point(r, r + 3, r[8], r[9], r[10], r[11], k, Ns, depth_inv != 0, z, p, dn);
  xyz[i * 3] = p[0], xyz[i * 3 + 1] = p[1], xyz[i * 3 + 2] = p[2];
  uvd[i * 3] = r[6], uvd[i * 3 + 1] = r[7], uvd[i * 3 + 2] = dn;
  zv[i] = z;
}

__global__ void unpreprocess_kernel(const float* __restrict__ src, int H, int W, int Ho, int Wo,
                                    float* __restrict__ out) {
  int plane = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ho * Wo) return;
  int y = i / Wo, x = i - y * Wo;
  const float* p = src + (size_t)plane * H * W;
  float v;
  if (Ho == H && Wo == W) {
    v = p[i] * 0.5f + 0.5f;
  } else {
    Lerp1 ly = upsample_axis(y, H, Ho), lx = upsample_axis(x, W, Wo);
    // interpolate(0.5*x + 0.5): the affine map is applied to the taps first, as the reference does
    float a = p[ly.i0 * W + lx.i0] * 0.5f + 0.5f, b = p[ly.i0 * W + lx.i1] * 0.5f + 0.5f;
    float c = p[ly.i1 * W + lx.i0] * 0.5f + 0.5f, d = p[ly.i1 * W + lx.i1] * 0.5f + 0.5f;
    v = ly.l0 * (lx.l0 * a + lx.l1 * b) + ly.l1 * (lx.l0 * c + lx.l1 * d);
  }
  out[(size_t)plane * Ho * Wo + i] = v;
}

__global__ void vox_feat_kernel(const float* __restrict__ uvd01, const float* __restrict__ vol, int P, int C, int D,
                                int h, int w, float* __restrict__ out) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float* q = uvd01 + ((size_t)b * P + i) * 3;
  Taps3 t = taps3_zeros(q[0], q[1], q[2], w, h, D);
  size_t cs = (size_t)D * h * w;
  const float* v = vol + (size_t)b * C * cs;
  for (int c = 0; c < C; ++c) out[((size_t)b * P + i) * C + c] = tap3_fetch(v + c * cs, t);
}

__global__ void img_feat_kernel(const float* __restrict__ xyz, const float* __restrict__ img,
                                const float* __restrict__ src_exts, const float* __restrict__ src_ixts,
                                const float* __restrict__ tar_ext, float render_scale, int P, int S, int C, int H,
                                int W, float* __restrict__ out) {
  extern __shared__ float smem[];
  Cam* cams = reinterpret_cast<Cam*>(smem);
  float* tar_c = smem + (sizeof(Cam) / 4) * S;
  int b = blockIdx.y;
  if ((int)threadIdx.x < S)
    load_cam(src_exts + ((size_t)b * S + threadIdx.x) * 16, src_ixts + ((size_t)b * S + threadIdx.x) * 9, render_scale,
             cams[threadIdx.x]);
  if ((int)threadIdx.x == S) camera_centre(tar_ext + (size_t)b * 16, tar_c);
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  float p[3] = {xyz[((size_t)b * P + i) * 3], xyz[((size_t)b * P + i) * 3 + 1], xyz[((size_t)b * P + i) * 3 + 2]};
  size_t plane = (size_t)H * W;
  for (int s = 0; s < S; ++s) {
    Taps2 t = project_taps(cams[s], p, W, H);
    const float* f = img + ((size_t)b * S + s) * C * plane;
    float* o = out + (((size_t)b * P + i) * S + s) * (C + 4);
    for (int c = 0; c < C; ++c) o[c] = tap_fetch(f + c * plane, t);
    dir_feature(p, tar_c, cams[s].c, o + C);
  }
}

__global__ void mask_viewport_kernel(const float* __restrict__ xyz, const float* __restrict__ src_exts,
                                     const float* __restrict__ src_ixts, float inv_w, float inv_h, int P, int V,
                                     float* __restrict__ mask) {
  extern __shared__ float smem[];
  Cam* cams = reinterpret_cast<Cam*>(smem);
  int b = blockIdx.y;
  for (int v = threadIdx.x; v < V; v += blockDim.x)
    load_cam(src_exts + ((size_t)b * V + v) * 16, src_ixts + ((size_t)b * V + v) * 9, 1.f, cams[v]);
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  float p[3] = {xyz[((size_t)b * P + i) * 3], xyz[((size_t)b * P + i) * 3 + 1], xyz[((size_t)b * P + i) * 3 + 2]};
  float acc = 0.f;
  for (int v = 0; v < V; ++v) acc += visible(cams[v], p, inv_w, inv_h);
  mask[(size_t)b * P + i] = acc / (float)V;
}

// a14 get_ndc_coords for one source view per batch item
__global__ void ndc_coords_kernel(const float* __restrict__ xyz, const float* __restrict__ src_ext,
                                  const float* __restrict__ src_ixt, float inv_w, float inv_h, int P,
                                  float* __restrict__ ndc) {
  __shared__ Cam cam;
  int b = blockIdx.y;
  if (threadIdx.x == 0) load_cam(src_ext + (size_t)b * 16, src_ixt + (size_t)b * 9, 1.f, cam);
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  float p[3] = {xyz[((size_t)b * P + i) * 3], xyz[((size_t)b * P + i) * 3 + 1], xyz[((size_t)b * P + i) * 3 + 2]};
  float u, v, pz;
  ndc_coords(cam, p, inv_w, inv_h, u, v, pz);
  float* o = ndc + ((size_t)b * P + i) * 3;
  o[0] = u, o[1] = v, o[2] = pz;
}

// a12: one thread per ray (Ns is 2..128; the fused renderer uses wave shuffles instead).
__global__ void composite_kernel(const float* __restrict__ raw, const float* __restrict__ zv, long nrays, int Ns,
                                 int white_bkgd, float* __restrict__ rgb, float* __restrict__ depth,
                                 float* __restrict__ weights) {
  long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrays) return;
  const float* q = raw + r * Ns * 4;
  float T = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, wmax = -INFINITY;
  for (int k = 0; k < Ns; ++k) {
    float alpha = 1.f - expf(-q[k * 4 + 3]);
    float wk = alpha * T;
    c0 += wk * q[k * 4], c1 += wk * q[k * 4 + 1], c2 += wk * q[k * 4 + 2];
    weights[r * Ns + k] = wk;
    wmax = fmaxf(wmax, wk);
    T *= (1.f - alpha + 1e-10f);
  }
  float den = 0.f;
  for (int k = 0; k < Ns; ++k) den += expf(weights[r * Ns + k] - wmax);
  float dsum = 0.f, acc = 0.f;
  for (int k = 0; k < Ns; ++k) {
    float s = expf(weights[r * Ns + k] - wmax) / den;
    weights[r * Ns + k] = s;
    dsum += s * zv[r * Ns + k];
    acc += s;
  }
  if (white_bkgd) {
    c0 += 1.f - acc, c1 += 1.f - acc, c2 += 1.f - acc;
  }
  rgb[r * 3] = c0, rgb[r * 3 + 1] = c1, rgb[r * 3 + 2] = c2;
  depth[r] = dsum;
}

// a16 (+ merge_mlp_outputs mask normalisation): one thread per ray, K volumes fused
// in registers so the (B,K,N,Ns,*) intermediates are read exactly once.
__global__ void blend_kernel(const float* __restrict__ raws, const float* __restrict__ masks,
                             const float* __restrict__ zv, int K, int N, int Ns, int normalise,
                             float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ weights) {
  int b = blockIdx.y;
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  size_t kstride = (size_t)N * Ns;
  const float* R = raws + ((size_t)b * K * N + n) * Ns * 4;
  const float* M = masks + ((size_t)b * K * N + n) * Ns;
  const float* Z = zv + ((size_t)b * K * N + n) * Ns;
  float* W = weights + ((size_t)b * N + n) * Ns;
  float T = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, wmax = -INFINITY;
  for (int s = 0; s < Ns; ++s) {
    float msum = 0.f;
    if (normalise)
      for (int k = 0; k < K; ++k) msum += M[k * kstride + s];
    float alpha = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int k = 0; k < K; ++k) {
      float m = M[k * kstride + s];
      if (normalise) m = msum > 0.f ? m / msum : 1.f / (float)K;
      const float* q = R + (k * kstride + s) * 4;
      float ak = 1.f - expf(-q[3]);
      alpha += ak * m;
      float tam = T * ak * m;  // (T * alpha_k * m_k) * rgb, utils.py:651
      a0 += tam * q[0], a1 += tam * q[1], a2 += tam * q[2];
    }
    c0 += a0, c1 += a1, c2 += a2;
    float wk = alpha * T;
    W[s] = wk;
    wmax = fmaxf(wmax, wk);
    T *= (1.f - alpha);
  }
  float den = 0.f;
  for (int s = 0; s < Ns; ++s) den += expf(W[s] - wmax);
  float dsum = 0.f;
  for (int s = 0; s < Ns; ++s) {
    float sm = expf(W[s] - wmax) / den;
    W[s] = sm;
    float zm = 0.f;
    for (int k = 0; k < K; ++k) zm += Z[k * kstride + s];
    dsum += sm * (zm / (float)K);
  }
  rgb[((size_t)b * N + n) * 3] = c0, rgb[((size_t)b * N + n) * 3 + 1] = c1, rgb[((size_t)b * N + n) * 3 + 2] = c2;
  depth[(size_t)b * N + n] = dsum;
}

// a16 with lane = SAMPLE (round 4; the kernel above is kept for sample counts that are not a power of two).  A ray's Ns
// samples sit on Ns consecutive lanes (Ns <= 64: 64 / Ns rays per wave) or on the 64 lanes in NC = Ns / 64 register
// chunks: every load is a coalesced run (16 bytes per lane for the raw outputs), each mask / depth is read ONCE, the
// transmittance is a segmented prefix product across lanes (log2 steps of shuffles), the per-ray sums and the softmax
// are xor butterflies inside the ray's aligned lane group, and the weights are written once.  The per-ray arithmetic
// does not depend on which rays share the wave (ray shards stay bit-identical to the full frame).
// Measured (config 4: K = 4, N = 78 848, Ns = 128, 970 MB): 2 825 us (0.04 of HBM) for the thread-per-ray loop above.
template <int NC, int KMAX>
__global__ void __launch_bounds__(256) blend_wave_kernel(const float* __restrict__ raws, const float* __restrict__ masks,
                                                         const float* __restrict__ zv, int K, int N, int Ns, int normalise,
                                                         float* __restrict__ rgb, float* __restrict__ depth,
                                                         float* __restrict__ weights) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * 256 + threadIdx.x) >> 6);
  const int G = NC > 1 ? 64 : Ns;             // lanes of a ray (a power of two)
  const int s0 = lane & (G - 1);
  const int n = wave * (64 / G) + lane / G;
  const bool live = n < N;
  const int nc = live ? n : N - 1;            // (idle lanes mirror the last ray: loads in bounds, no stores)
  const size_t kstride = (size_t)N * Ns;
  const size_t ray = ((size_t)b * K * N + nc) * Ns;
  const float4* R = reinterpret_cast<const float4*>(raws) + ray;
  const float* M = masks + ray;
  const float* Z = zv + ray;
  const float invK = 1.f / (float)K;
  float alpha[NC], c0[NC], c1[NC], c2[NC], zm[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int s = s0 + 64 * c;
    float mk[KMAX];
    float msum = 0.f, zs = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        mk[k] = M[k * kstride + s];
        msum += mk[k];
        zs += Z[k * kstride + s];
      }
    float a = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        float m = mk[k];
        if (normalise) m = msum > 0.f ? m / msum : invK;
        const float4 q = R[k * kstride + s];
        const float am = (1.f - expf(-q.w)) * m;   // alpha_k m_k
        a += am;
        q0 += am * q.x, q1 += am * q.y, q2 += am * q.z;
      }
    alpha[c] = a, c0[c] = q0, c1[c] = q1, c2[c] = q2, zm[c] = zs * invK;
  }
  // transmittance T_s = prod_{j < s} (1 - alpha_j) (no epsilon: enerf/utils.py:648): inclusive product scan in the lane
  // group, shifted by one, carried across the register chunks
  float T[NC], carry = 1.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float p = 1.f - alpha[c];
    for (int off = 1; off < G; off <<= 1) {
      const float t = __shfl_up(p, off);
      if (s0 >= off) p *= t;
    }
    float ex = __shfl_up(p, 1);
    if (s0 == 0) ex = 1.f;
    T[c] = carry * ex;
    carry *= __shfl(p, lane | (G - 1));
  }
  float w[NC], r0 = 0.f, r1 = 0.f, r2 = 0.f, wmax = -INFINITY;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    w[c] = alpha[c] * T[c];
    r0 += T[c] * c0[c], r1 += T[c] * c1[c], r2 += T[c] * c2[c];
    wmax = fmaxf(wmax, w[c]);
  }
  for (int off = G >> 1; off > 0; off >>= 1) {
    r0 += __shfl_xor(r0, off), r1 += __shfl_xor(r1, off), r2 += __shfl_xor(r2, off);
    wmax = fmaxf(wmax, __shfl_xor(wmax, off));
  }
  float e[NC], den = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) e[c] = expf(w[c] - wmax), den += e[c];
  for (int off = G >> 1; off > 0; off >>= 1) den += __shfl_xor(den, off);
  float dsum = 0.f;
  float* W = weights + ((size_t)b * N + nc) * Ns;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float sm = e[c] / den;
    dsum += sm * zm[c];
    if (live) W[s0 + 64 * c] = sm;
  }
  for (int off = G >> 1; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
  if (live && s0 == 0) {
    float* o = rgb + ((size_t)b * N + n) * 3;
    o[0] = r0, o[1] = r1, o[2] = r2;
    depth[(size_t)b * N + n] = dsum;
  }
}

}  // namespace bmv

using namespace bmv;

extern "C" {

int bmv_build_rays(const float* rays, const float* depth, const float* std_, const float* near_far, int B, int N,
                   int hv, int wv, int Hr, int Wr, int depth_inv, float* rays_out, bmv_stream_t stream) {
  BMV_REQUIRE(rays && depth && std_ && near_far && rays_out, "bmv_build_rays: null pointer");
  BMV_REQUIRE(B > 0 && N >= 0 && hv > 0 && wv > 0 && Hr > 0 && Wr > 0, "bmv_build_rays: bad shape");
  if (N == 0) return BMV_OK;
  hipLaunchKernelGGL(build_rays_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, as_stream(stream), rays, depth, std_,
                     near_far, N, hv, wv, Hr, Wr, depth_inv, rays_out);
  BMV_LAUNCH_END("bmv_build_rays");
}

int bmv_sample_along_depth(const float* rays, int B, int N, int Ns, int depth_inv, float* world_xyz, float* uvd,
                           float* z_vals, bmv_stream_t stream) {
  BMV_REQUIRE(rays && world_xyz && uvd && z_vals, "bmv_sample_along_depth: null pointer");
  BMV_REQUIRE(B > 0 && N >= 0 && Ns > 0, "bmv_sample_along_depth: bad shape");
  long total = (long)B * N * Ns;
  if (total == 0) return BMV_OK;
  hipLaunchKernelGGL(sample_along_depth_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), rays, total,
                     Ns, depth_inv, world_xyz, uvd, z_vals);
  BMV_LAUNCH_END("bmv_sample_along_depth");
}

int bmv_unpreprocess(const float* src, int n, int C, int H, int W, int Ho, int Wo, float* out, bmv_stream_t stream) {
  BMV_REQUIRE(src && out, "bmv_unpreprocess: null pointer");
  BMV_REQUIRE(n > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "bmv_unpreprocess: bad shape");
  hipLaunchKernelGGL(unpreprocess_kernel, dim3(cdiv(Ho * Wo, 256), n * C), dim3(256), 0, as_stream(stream), src, H, W,
                     Ho, Wo, out);
  BMV_LAUNCH_END("bmv_unpreprocess");
}

int bmv_vox_feat(const float* uvd01, const float* volume, int B, int P, int C, int D, int h, int w, float* out,
                 bmv_stream_t stream) {
  BMV_REQUIRE(uvd01 && volume && out, "bmv_vox_feat: null pointer");
  BMV_REQUIRE(B > 0 && P >= 0 && C > 0 && D > 0 && h > 0 && w > 0, "bmv_vox_feat: bad shape");
  if (P == 0) return BMV_OK;
  hipLaunchKernelGGL(vox_feat_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, as_stream(stream), uvd01, volume, P, C, D,
                     h, w, out);
  BMV_LAUNCH_END("bmv_vox_feat");
}

int bmv_img_feat(const float* xyz, const float* img_feat_rgb, const float* src_exts, const float* src_ixts,
                 const float* tar_ext, float render_scale, int B, int P, int S, int C, int H, int W, float* out,
                 bmv_stream_t stream) {
  BMV_REQUIRE(xyz && img_feat_rgb && src_exts && src_ixts && tar_ext && out, "bmv_img_feat: null pointer");
  BMV_REQUIRE(B > 0 && P >= 0 && S > 0 && S <= 16 && C > 0 && H > 1 && W > 1, "bmv_img_feat: bad shape");
  if (P == 0) return BMV_OK;
  size_t smem = sizeof(Cam) * S + 16;
  hipLaunchKernelGGL(img_feat_kernel, dim3(cdiv(P, 256), B), dim3(256), smem, as_stream(stream), xyz, img_feat_rgb,
                     src_exts, src_ixts, tar_ext, render_scale, P, S, C, H, W, out);
  BMV_LAUNCH_END("bmv_img_feat");
}

int bmv_mask_viewport(const float* xyz, const float* src_exts, const float* src_ixts, float inv_w, float inv_h, int B,
                      int P, int V, float* mask, bmv_stream_t stream) {
  BMV_REQUIRE(xyz && src_exts && src_ixts && mask, "bmv_mask_viewport: null pointer");
  BMV_REQUIRE(B > 0 && P >= 0 && V > 0 && V <= 64, "bmv_mask_viewport: bad shape");
  if (P == 0) return BMV_OK;
  hipLaunchKernelGGL(mask_viewport_kernel, dim3(cdiv(P, 256), B), dim3(256), sizeof(Cam) * V, as_stream(stream), xyz,
                     src_exts, src_ixts, inv_w, inv_h, P, V, mask);
  BMV_LAUNCH_END("bmv_mask_viewport");
}

int bmv_ndc_coords(const float* xyz, const float* src_ext, const float* src_ixt, float inv_w, float inv_h, int B, int P,
                   float* ndc, bmv_stream_t stream) {
  BMV_REQUIRE(xyz && src_ext && src_ixt && ndc, "bmv_ndc_coords: null pointer");
  BMV_REQUIRE(B > 0 && P >= 0, "bmv_ndc_coords: bad shape");
  if (P == 0) return BMV_OK;
  hipLaunchKernelGGL(ndc_coords_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, as_stream(stream), xyz, src_ext, src_ixt,
                     inv_w, inv_h, P, ndc);
  BMV_LAUNCH_END("bmv_ndc_coords");
}

int bmv_composite_fwd(const float* raw, const float* z_vals, long nrays, int Ns, int white_bkgd, float* rgb,
                      float* depth, float* weights, bmv_stream_t stream) {
  BMV_REQUIRE(raw && z_vals && rgb && depth && weights, "bmv_composite_fwd: null pointer");
  BMV_REQUIRE(nrays >= 0 && Ns > 0, "bmv_composite_fwd: bad shape");
  if (nrays == 0) return BMV_OK;
  hipLaunchKernelGGL(composite_kernel, dim3(cdiv(nrays, 256)), dim3(256), 0, as_stream(stream), raw, z_vals, nrays, Ns,
                     white_bkgd, rgb, depth, weights);
  BMV_LAUNCH_END("bmv_composite_fwd");
}

int bmv_blend_fwd(const float* raws, const float* masks, const float* z_vals, int B, int K, int N, int Ns,
                  int normalise, float* rgb, float* depth, float* weights, bmv_stream_t stream) {
  BMV_REQUIRE(raws && masks && z_vals && rgb && depth && weights, "bmv_blend_fwd: null pointer");
  BMV_REQUIRE(B > 0 && K > 0 && N >= 0 && Ns > 0, "bmv_blend_fwd: bad shape");
  if (N == 0) return BMV_OK;
  if ((Ns & (Ns - 1)) == 0 && Ns <= 256 && K <= 8) {   // lane = sample (power-of-two sample counts: every shipped config)
    const int G = Ns < 64 ? Ns : 64;
    const long waves = ((long)N * G + 63) / 64;
    dim3 grid(cdiv(waves, 4), B), block(256);
#define BW(NC, KM) hipLaunchKernelGGL((blend_wave_kernel<NC, KM>), grid, block, 0, as_stream(stream), raws, masks, z_vals, K, N, Ns, normalise, rgb, depth, weights)
    if (Ns <= 64) {
      if (K <= 4) BW(1, 4); else BW(1, 8);
    } else if (Ns == 128) {
      if (K <= 4) BW(2, 4); else BW(2, 8);
    } else {
      if (K <= 4) BW(4, 4); else BW(4, 8);
    }
#undef BW
    BMV_LAUNCH_END("bmv_blend_fwd");
  }
  hipLaunchKernelGGL(blend_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, as_stream(stream), raws, masks, z_vals, K, N,
                     Ns, normalise, rgb, depth, weights);
  BMV_LAUNCH_END("bmv_blend_fwd");
}

}  // extern "C"
