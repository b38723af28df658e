// Backward kernels of the ENeRF hot path (fine-tuning, BASELINE config 5; SURVEY.md section 8
// "Backward contract").  Each kernel is the hand-derived adjoint of the forward kernel of the same
// name; tests compare them with torch.autograd on the CPU oracle.  Scatter-adds are float atomics
// (global_atomic_add_f32); gradient buffers are zero-initialised by the caller.  The *_fixed entry points run the same
// kernels with order-independent fixed-point accumulation (scatter.hpp): bit-reproducible gradients.
#include "render_geom.hpp"
#include "scatter.hpp"

namespace bmv {

// ---------------------------------------------------------------------------
// a12 raw2outputs backward.  d_rgb (N,3), d_depth (N) or null -> d_raw (N,Ns,4).
// weights = softmax(alpha*T) is an output of the forward but never receives gradient in the
// reference's losses (lib/train/losses/enerf.py:22-24); z_vals are detached (utils.py:629).
// ---------------------------------------------------------------------------
__global__ void composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ zv,
                                     const float* __restrict__ d_rgb, const float* __restrict__ d_depth, long nrays,
                                     int Ns, float* __restrict__ d_raw) {
  long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrays) return;
  const float* q = raw + r * Ns * 4;
  float* dq = d_raw + r * Ns * 4;
  const float g0 = d_rgb[r * 3], g1 = d_rgb[r * 3 + 1], g2 = d_rgb[r * 3 + 2];
  // depth path: depth = sum_k softmax(w)_k z_k
  float wmax = -INFINITY, den = 0.f, dotz = 0.f;
  const float gd = d_depth ? d_depth[r] : 0.f;
  if (d_depth) {
    float T = 1.f;
    for (int k = 0; k < Ns; ++k) {
      float a = 1.f - expf(-q[k * 4 + 3]);
      wmax = fmaxf(wmax, a * T);
      T *= (1.f - a + 1e-10f);
    }
    T = 1.f;
    for (int k = 0; k < Ns; ++k) {
      float a = 1.f - expf(-q[k * 4 + 3]);
      float e = expf(a * T - wmax);
      den += e;
      dotz += e * zv[r * Ns + k];
      T *= (1.f - a + 1e-10f);
    }
    dotz /= den;  // = depth
  }
  // forward pass: park the exclusive transmittances T_k in the output buffer, then walk back
  {
    float Tf = 1.f;
    for (int k = 0; k < Ns; ++k) {
      dq[k * 4 + 3] = Tf;
      Tf *= (1.f - (1.f - expf(-q[k * 4 + 3])) + 1e-10f);
    }
  }
  float suffix = 0.f;  // suffix = sum_{j>k} d_T_j * T_j
  for (int k = Ns - 1; k >= 0; --k) {
    float a = 1.f - expf(-q[k * 4 + 3]);
    float om = 1.f - a + 1e-10f;
    float T = dq[k * 4 + 3];  // T_k (exclusive product)
    float w = a * T;
    float d_w = g0 * q[k * 4] + g1 * q[k * 4 + 1] + g2 * q[k * 4 + 2];
    if (d_depth) {
      float sm = expf(w - wmax) / den;
      d_w += gd * sm * (zv[r * Ns + k] - dotz);
    }
    dq[k * 4] = w * g0, dq[k * 4 + 1] = w * g1, dq[k * 4 + 2] = w * g2;
    float d_a = d_w * T - suffix / om;  // through w_k and through every later T_j
    dq[k * 4 + 3] = d_a * (1.f - a);    // alpha = 1 - exp(-sigma)
    suffix += d_w * a * T;              // d_T_k * T_k
  }
}

// ---------------------------------------------------------------------------
// a16 raw2outputs_blend backward (masks already normalised, no gradient to masks or z).
// raws (B,K,N,Ns,4), masks (B,K,N,Ns) -> d_raws.
// ---------------------------------------------------------------------------
__global__ void blend_bwd_kernel(const float* __restrict__ raws, const float* __restrict__ masks,
                                 const float* __restrict__ d_rgb, int K, int N, int Ns, float* __restrict__ d_raws) {
  int b = blockIdx.y;
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  size_t ks = (size_t)N * Ns;
  const float* R = raws + ((size_t)b * K * N + n) * Ns * 4;
  const float* M = masks + ((size_t)b * K * N + n) * Ns;
  float* DR = d_raws + ((size_t)b * K * N + n) * Ns * 4;
  const float g0 = d_rgb[((size_t)b * N + n) * 3], g1 = d_rgb[((size_t)b * N + n) * 3 + 1],
              g2 = d_rgb[((size_t)b * N + n) * 3 + 2];
  {  // park the exclusive transmittances T_s in the (k = 0) sigma slot of the output buffer
    float Tf = 1.f;
    for (int s = 0; s < Ns; ++s) {
      float alpha = 0.f;
      for (int k = 0; k < K; ++k) alpha += (1.f - expf(-R[(k * ks + s) * 4 + 3])) * M[k * ks + s];
      DR[s * 4 + 3] = Tf;
      Tf *= (1.f - alpha);
    }
  }
  float suffix = 0.f;
  for (int s = Ns - 1; s >= 0; --s) {
    const float T = DR[s * 4 + 3];
    float alpha = 0.f, A = 0.f;  // A = sum_k alpha_k m_k (c_k . g)
    for (int k = 0; k < K; ++k) {
      const float* q = R + (k * ks + s) * 4;
      float ak = 1.f - expf(-q[3]), m = M[k * ks + s];
      alpha += ak * m;
      A += ak * m * (q[0] * g0 + q[1] * g1 + q[2] * g2);
    }
    float om = 1.f - alpha;
    float om_safe = fabsf(om) > 1e-20f ? om : 1e-20f;
    float d_alpha_shared = -suffix / om_safe;  // alpha_s only enters later transmittances
    for (int k = 0; k < K; ++k) {
      const float* q = R + (k * ks + s) * 4;
      float* dq = DR + (k * ks + s) * 4;
      float ak = 1.f - expf(-q[3]), m = M[k * ks + s];
      float tam = T * ak * m;
      dq[0] = tam * g0, dq[1] = tam * g1, dq[2] = tam * g2;
      float d_ak = T * m * (q[0] * g0 + q[1] * g1 + q[2] * g2) + d_alpha_shared * m;
      dq[3] = d_ak * (1.f - ak);
    }
    suffix += A * T;  // d_T_s * T_s with d_T_s = A
  }
}

// ---------------------------------------------------------------------------
// Scatter-add of bilinear taps through an LDS window.  The threads of a workgroup handle a compact 2-D tile of
// voxels / rays, so the taps of one source view land in a small box of its feature map: the box is found with an
// LDS min/max, the taps are accumulated there with ds_add_f32, and the box is added to the gradient image once, row
// by row (coalesced, ~1 global atomic per touched texel instead of 4 per sample: neighbouring samples share texels).
// A box that does not fit (steep parallax) falls back to one global atomic per tap.
// ---------------------------------------------------------------------------
constexpr int kWinCap = 1536;   // texels per channel of the LDS window
constexpr int kWinCh = 8;       // channels per pass

struct WinBox {
  int minx, miny, maxx, maxy;
};

struct TapSet {                 // the 4 taps of one sample in one view; weight 0 = not written
  int x0, y0, x1, y1;
  float w00, w01, w10, w11;
};

// returns true if the window path is taken (uniform over the workgroup); box = inclusive texel bounds
__device__ __forceinline__ bool win_begin(WinBox* box, const TapSet& t, bool active) {
  if (threadIdx.x == 0) *box = WinBox{0x7fffffff, 0x7fffffff, -1, -1};
  __syncthreads();
  if (active) {
    const bool l = t.w00 != 0.f || t.w10 != 0.f, r = t.w01 != 0.f || t.w11 != 0.f;
    const bool u = t.w00 != 0.f || t.w01 != 0.f, d = t.w10 != 0.f || t.w11 != 0.f;
    if (l || r) {
      atomicMin(&box->minx, l ? t.x0 : t.x1), atomicMax(&box->maxx, r ? t.x1 : t.x0);
      atomicMin(&box->miny, u ? t.y0 : t.y1), atomicMax(&box->maxy, d ? t.y1 : t.y0);
    }
  }
  __syncthreads();
  const int wx = box->maxx - box->minx + 1, wy = box->maxy - box->miny + 1;
  return box->maxx >= 0 && wx * wy <= kWinCap;
}

// one pass of `nc` (<= kWinCh) channels: g[c] = gradient of this thread's sample for channel c of the pass;
// dst = gradient image plane of the pass's first channel, planes `plane` apart, rows `W` apart
template <class Acc>
__device__ __forceinline__ void win_scatter(Acc& acc, float* win, const WinBox* box, const TapSet& t, bool active,
                                            const float* g, int nc, float* __restrict__ dst, size_t plane, int W) {
  const int minx = box->minx, miny = box->miny, wx = box->maxx - minx + 1, wy = box->maxy - miny + 1;
  const int n = wx * wy;
  for (int i = threadIdx.x; i < nc * n; i += blockDim.x) win[i] = 0.f;
  __syncthreads();
  if (active) {
    const int o00 = (t.y0 - miny) * wx + (t.x0 - minx), o01 = (t.y0 - miny) * wx + (t.x1 - minx);
    const int o10 = (t.y1 - miny) * wx + (t.x0 - minx), o11 = (t.y1 - miny) * wx + (t.x1 - minx);
    for (int c = 0; c < nc; ++c) {
      float* wc = win + c * n;
#ifdef BMV_BWD_NOSCATTER
      if (g[c] == 123.456f) wc[o00] = g[c];
      continue;
#endif
      if (t.w00 != 0.f) atomicAdd(wc + o00, t.w00 * g[c]);
      if (t.w01 != 0.f) atomicAdd(wc + o01, t.w01 * g[c]);
      if (t.w10 != 0.f) atomicAdd(wc + o10, t.w10 * g[c]);
      if (t.w11 != 0.f) atomicAdd(wc + o11, t.w11 * g[c]);
    }
  }
  __syncthreads();
  // flush: 32 lanes along x, 8 (channel, row) pairs per sweep of the workgroup
  const int lx = threadIdx.x & 31, lr = threadIdx.x >> 5, nrows = nc * wy, rstep = blockDim.x >> 5;
  for (int r0 = lr; r0 < nrows; r0 += rstep) {
    const int c = r0 / wy, yy = r0 - c * wy;
    for (int xx = lx; xx < wx; xx += 32) {
      const float v = win[c * n + yy * wx + xx];
#ifndef BMV_BWD_NOFLUSH
      if (v != 0.f) acc.add(dst + c * plane + (size_t)(miny + yy) * W + (minx + xx), v);
#else
      if (v == 123.456f) dst[0] = v;
#endif
    }
  }
  __syncthreads();
}

template <class Acc>
__device__ __forceinline__ void direct_scatter(Acc& acc, const TapSet& t, const float* g, int nc, float* __restrict__ dst,
                                               size_t plane, int W) {
  for (int c = 0; c < nc; ++c) {
    float* d = dst + c * plane;
    if (t.w00 != 0.f) acc.add(d + (size_t)t.y0 * W + t.x0, t.w00 * g[c]);
    if (t.w01 != 0.f) acc.add(d + (size_t)t.y0 * W + t.x1, t.w01 * g[c]);
    if (t.w10 != 0.f) acc.add(d + (size_t)t.y1 * W + t.x0, t.w10 * g[c]);
    if (t.w11 != 0.f) acc.add(d + (size_t)t.y1 * W + t.x1, t.w11 * g[c]);
  }
}

// ---------------------------------------------------------------------------
// a9 get_vox_feat backward: d_out (B,P,C) -> d_volume (atomics), d_d01 (B,P) (only the depth
// coordinate of uvd carries gradient: u, v are pixel constants).
// ---------------------------------------------------------------------------
// Thread -> sample as in img_feat_bwd_kernel (2-D tiles of rays with the layout hint).  The 8 taps of the workgroup's
// samples land in a small box of the volume: found with an LDS min/max, accumulated with ds_add_f32 (C <= 8 channels x
// kWinCap voxels), added to d_volume once per touched voxel, x-contiguous; a box that does not fit falls back to one
// global atomic per tap.
template <int MODE>
__global__ void __launch_bounds__(256) vox_feat_bwd_kernel(const float* __restrict__ uvd01, const float* __restrict__ vol,
                                                           const float* __restrict__ d_out, int P, int C, int D, int h,
                                                           int w, int ray_w, int Ns, int tw, int th, int tiles_x,
                                                           float* __restrict__ d_vol, float* __restrict__ d_d01,
                                                           FixedWs fixed) {
  __shared__ float win[kWinCh * kWinCap];
  ScatterAcc<MODE> sacc = make_acc<MODE>(d_vol, fixed, 0);
  __shared__ int box[6];   // min x, y, z, max x, y, z
  const int b = blockIdx.y;
  int i;
  bool valid;
  if (ray_w > 0) {
    const int smp = threadIdx.x % Ns, r = threadIdx.x / Ns;
    const int rx = (blockIdx.x % tiles_x) * tw + r % tw, ry = (blockIdx.x / tiles_x) * th + r / tw;
    const int ray_h = P / (ray_w * Ns);
    valid = rx < ray_w && ry < ray_h;
    i = valid ? (ry * ray_w + rx) * Ns + smp : 0;
  } else {
    i = blockIdx.x * blockDim.x + threadIdx.x;
    valid = i < P;
    i = valid ? i : 0;
  }
  const float* q = uvd01 + ((size_t)b * P + i) * 3;
  float ix = unnorm(q[0] * 2.f - 1.f, w), iy = unnorm(q[1] * 2.f - 1.f, h), iz = unnorm(q[2] * 2.f - 1.f, D);
  float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const bool ok = valid && (fx >= -1.f) && (fx <= (float)(w - 1)) && (fy >= -1.f) && (fy <= (float)(h - 1)) &&
                  (fz >= -1.f) && (fz <= (float)(D - 1));
  const int x0 = ok ? (int)fx : 0, y0 = ok ? (int)fy : 0, z0 = ok ? (int)fz : 0;
  const float ax = ix - fx, ay = iy - fy, az = iz - fz, ex = (fx + 1.f) - ix, ey = (fy + 1.f) - iy, ez = (fz + 1.f) - iz;
  // per axis: which of the two taps exist (zeros padding)
  const bool vx0 = ok && x0 >= 0, vx1 = ok && x0 + 1 <= w - 1, vy0 = ok && y0 >= 0, vy1 = ok && y0 + 1 <= h - 1;
  const bool vz0 = ok && z0 >= 0, vz1 = ok && z0 + 1 <= D - 1;
  const bool any = (vx0 || vx1) && (vy0 || vy1) && (vz0 || vz1);
  if (threadIdx.x == 0) box[0] = box[1] = box[2] = 0x7fffffff, box[3] = box[4] = box[5] = -1;
  __syncthreads();
  if (any) {
    atomicMin(&box[0], vx0 ? x0 : x0 + 1), atomicMax(&box[3], vx1 ? x0 + 1 : x0);
    atomicMin(&box[1], vy0 ? y0 : y0 + 1), atomicMax(&box[4], vy1 ? y0 + 1 : y0);
    atomicMin(&box[2], vz0 ? z0 : z0 + 1), atomicMax(&box[5], vz1 ? z0 + 1 : z0);
  }
  __syncthreads();
  const int bx = box[0], by = box[1], bz = box[2];
  const int wx = box[3] - bx + 1, wy = box[4] - by + 1, wz = box[5] - bz + 1;
  // (the LDS pre-reduction adds floats in scheduling order: the fixed-point passes go straight to their accumulators)
  const bool use_win = MODE == 0 && box[3] >= 0 && wx * wy * wz <= kWinCap && C <= kWinCh;
  const int n = wx * wy * wz;
  if (use_win) {
    for (int k = threadIdx.x; k < C * n; k += blockDim.x) win[k] = 0.f;
    __syncthreads();
  }
  const size_t cs = (size_t)D * h * w;
  const float* v = vol + (size_t)b * C * cs;
  float* dv = d_vol + (size_t)b * C * cs;
  const float* go = d_out + ((size_t)b * P + i) * C;
  float gz = 0.f;
  if (any) {
    for (int k = 0; k < 8; ++k) {
      const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      if (!((dx ? vx1 : vx0) && (dy ? vy1 : vy0) && (dz ? vz1 : vz0))) continue;
      const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
      const float wxy = (dx ? ax : ex) * (dy ? ay : ey), wgt = wxy * (dz ? az : ez);
      const size_t o = ((size_t)z * h + y) * w + x;
      const int lo = ((z - bz) * wy + (y - by)) * wx + (x - bx);
      float acc = 0.f;
      for (int c = 0; c < C; ++c) {
        const float g = go[c];
        if (use_win)
          atomicAdd(win + c * n + lo, wgt * g);
        else
          sacc.add(dv + c * cs + o, wgt * g);
        acc += g * v[c * cs + o];
      }
      gz += (dz ? 1.f : -1.f) * wxy * acc;
    }
  }
  // iz = ((2 d - 1) + 1) / 2 * (D - 1)  ->  d iz / d d01 = D - 1
  if (valid) d_d01[(size_t)b * P + i] = gz * (float)(D - 1);
  if (use_win) {
    __syncthreads();
    // flush: 32 lanes along x, 8 (channel, z, y) rows per sweep of the workgroup
    const int lx = threadIdx.x & 31, lr = threadIdx.x >> 5, nrows = C * wz * wy, rstep = blockDim.x >> 5;
    for (int r0 = lr; r0 < nrows; r0 += rstep) {
      const int c = r0 / (wz * wy), rem = r0 - c * (wz * wy), zz = rem / wy, yy = rem - zz * wy;
      for (int xx = lx; xx < wx; xx += 32) {
        const float val = win[c * n + (zz * wy + yy) * wx + xx];
        if (val != 0.f) sacc.add(dv + c * cs + ((size_t)(bz + zz) * h + (by + yy)) * w + (bx + xx), val);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// a10 get_img_feat backward: d_out (B,P,S,C+4) -> d_img (B,S,C,H,W) (atomics), d_xyz (B,P,3).
// ---------------------------------------------------------------------------
// y = x / (|x| + eps): given dy -> dx
__device__ __forceinline__ void unit_eps_bwd(const float* x, float eps, const float* dy, float* dx) {
  float n = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  float s = n + eps;
  float dot = x[0] * dy[0] + x[1] * dy[1] + x[2] * dy[2];
  float k = n > 0.f ? dot / (s * s * n) : 0.f;
  for (int j = 0; j < 3; ++j) dx[j] = dy[j] / s - x[j] * k;
}

// Thread -> sample.  With the layout hint (rays row-major over an image `ray_w` wide, Ns samples per ray, Ns a power
// of two <= 64) a workgroup takes a compact tw x th tile of rays and all their samples, so that the taps of a view
// land in a small box (see win_begin); without it, 256 consecutive samples.
template <int MODE>
__global__ void __launch_bounds__(256) img_feat_bwd_kernel(
    const float* __restrict__ xyz, const float* __restrict__ img, const float* __restrict__ src_exts,
    const float* __restrict__ src_ixts, const float* __restrict__ tar_ext, float render_scale,
    const float* __restrict__ d_out, int P, int S, int C, int c_grad, int H, int W, int ray_w, int Ns, int tw, int th,
    int tiles_x, float* __restrict__ d_img, float* __restrict__ d_xyz, FixedWs fixed) {
  __shared__ float win[kWinCh * kWinCap];
  ScatterAcc<MODE> acc = make_acc<MODE>(d_img, fixed, 0);
  __shared__ WinBox box;
  __shared__ Cam cams[16];
  __shared__ float tar_c[4];
  const int b = blockIdx.y;
  if ((int)threadIdx.x < S)
    load_cam(src_exts + ((size_t)b * S + threadIdx.x) * 16, src_ixts + ((size_t)b * S + threadIdx.x) * 9, render_scale,
             cams[threadIdx.x]);
  if ((int)threadIdx.x == S) camera_centre(tar_ext + (size_t)b * 16, tar_c);
  __syncthreads();
  int i;
  bool valid;
  if (ray_w > 0) {
    const int smp = threadIdx.x % Ns, r = threadIdx.x / Ns;
    const int rx = (blockIdx.x % tiles_x) * tw + r % tw, ry = (blockIdx.x / tiles_x) * th + r / tw;
    const int ray_h = P / (ray_w * Ns);
    valid = rx < ray_w && ry < ray_h;
    i = valid ? (ry * ray_w + rx) * Ns + smp : 0;
  } else {
    i = blockIdx.x * blockDim.x + threadIdx.x;
    valid = i < P;
    i = valid ? i : 0;
  }
  const float p[3] = {xyz[((size_t)b * P + i) * 3], xyz[((size_t)b * P + i) * 3 + 1], xyz[((size_t)b * P + i) * 3 + 2]};
  float gp[3] = {0.f, 0.f, 0.f};
  const size_t plane = (size_t)H * W;
  for (int s = 0; s < S; ++s) {
    const Cam& cam = cams[s];
    const float* go = d_out + (((size_t)b * P + i) * S + s) * (C + 4);
    // ---- bilinear (border) part
    float cx = p[0] * cam.E[0] + p[1] * cam.E[1] + p[2] * cam.E[2] + cam.E[3];
    float cy = p[0] * cam.E[4] + p[1] * cam.E[5] + p[2] * cam.E[6] + cam.E[7];
    float cz = p[0] * cam.E[8] + p[1] * cam.E[9] + p[2] * cam.E[10] + cam.E[11];
    float qx = cx * cam.K[0] + cy * cam.K[1] + cz * cam.K[2];
    float qy = cx * cam.K[3] + cy * cam.K[4] + cz * cam.K[5];
    float qz = cx * cam.K[6] + cy * cam.K[7] + cz * cam.K[8];
    float z = fmaxf(qz, 1e-6f);
    float ix = unnorm((qx / z) / (float)(W - 1) * 2.f - 1.f, W), iy = unnorm((qy / z) / (float)(H - 1) * 2.f - 1.f, H);
    // border clip: gradient passes only strictly inside [0, size-1] (aten clip_coordinates_set_grad)
    float mx = (ix > 0.f && ix < (float)(W - 1)) ? 1.f : 0.f, my = (iy > 0.f && iy < (float)(H - 1)) ? 1.f : 0.f;
    float cix = fminf(fmaxf(ix, 0.f), (float)(W - 1)), ciy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    if (!(cix == cix)) cix = 0.f;      // NaN coordinates sample texel 0 (taps_border), no gradient
    if (!(ciy == ciy)) ciy = 0.f;
    float fx = floorf(cix), fy = floorf(ciy);
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    bool vx1 = x1 <= W - 1, vy1 = y1 <= H - 1;
    float ax = cix - fx, ay = ciy - fy, ex = (fx + 1.f) - cix, ey = (fy + 1.f) - ciy;
    const float* f = img + ((size_t)b * S + s) * C * plane;
    float* df = d_img + ((size_t)b * S + s) * C * plane;
    float gix = 0.f, giy = 0.f;
    size_t o00 = (size_t)y0 * W + x0, o01 = o00 + (vx1 ? 1 : 0), o10 = o00 + (vy1 ? W : 0), o11 = o10 + (vx1 ? 1 : 0);
    TapSet ts{x0, y0, vx1 ? x1 : x0, vy1 ? y1 : y0, ex * ey, vx1 ? ax * ey : 0.f, vy1 ? ex * ay : 0.f,
              (vx1 && vy1) ? ax * ay : 0.f};
    const bool use_win = MODE == 0 && win_begin(&box, ts, valid);
    for (int cg = 0; cg < C; cg += kWinCh) {
      const int nc = min(kWinCh, C - cg), ng = max(0, min(nc, c_grad - cg));   // channels of the pass / that need d_img
      float g[kWinCh];
      for (int c = 0; c < nc; ++c) {
        g[c] = valid ? go[cg + c] : 0.f;
        const float* fc = f + (cg + c) * plane;
        float v00 = fc[o00], v01 = vx1 ? fc[o01] : 0.f, v10 = vy1 ? fc[o10] : 0.f, v11 = (vx1 && vy1) ? fc[o11] : 0.f;
        gix += g[c] * ((v01 - v00) * ey + (v11 - v10) * ay);
        giy += g[c] * ((v10 - v00) * ex + (v11 - v01) * ax);
      }
      if (ng > 0) {
        if (use_win)
          win_scatter(acc, win, &box, ts, valid, g, ng, df + (size_t)cg * plane, plane, W);
        else if (valid)
          direct_scatter(acc, ts, g, ng, df + (size_t)cg * plane, plane, W);
      }
    }
    gix *= mx, giy *= my;
    // ix = (qx / z): the normalise / unnormalise pair cancels exactly in the derivative
    float gqx = gix / z, gqy = giy / z;
    float gqz = (qz > 1e-6f) ? -(gix * qx + giy * qy) / (z * z) : 0.f;
    float gcx = gqx * cam.K[0] + gqy * cam.K[3] + gqz * cam.K[6];
    float gcy = gqx * cam.K[1] + gqy * cam.K[4] + gqz * cam.K[7];
    float gcz = gqx * cam.K[2] + gqy * cam.K[5] + gqz * cam.K[8];
    gp[0] += gcx * cam.E[0] + gcy * cam.E[4] + gcz * cam.E[8];
    gp[1] += gcx * cam.E[1] + gcy * cam.E[5] + gcz * cam.E[9];
    gp[2] += gcx * cam.E[2] + gcy * cam.E[6] + gcz * cam.E[10];
    // ---- direction feature [ (a-b)/max(|a-b|,1e-6), a.b ],  a = unit(x - c_tar), b = unit(x - c_src)
    float xa[3], xb[3], a[3], bb[3];
    float na = 0.f, nb = 0.f;
    for (int j = 0; j < 3; ++j) {
      xa[j] = p[j] - tar_c[j], xb[j] = p[j] - cam.c[j];
      na += xa[j] * xa[j], nb += xb[j] * xb[j];
    }
    na = sqrtf(na) + 1e-6f, nb = sqrtf(nb) + 1e-6f;
    float dfv[3], nd = 0.f;
    for (int j = 0; j < 3; ++j) {
      a[j] = xa[j] / na, bb[j] = xb[j] / nb;
      dfv[j] = a[j] - bb[j];
      nd += dfv[j] * dfv[j];
    }
    nd = sqrtf(nd);
    const float gd[4] = {valid ? go[C] : 0.f, valid ? go[C + 1] : 0.f, valid ? go[C + 2] : 0.f, valid ? go[C + 3] : 0.f};
    float g_df[3];
    if (nd > 1e-6f) {  // dirn = df / |df|
      float dot = dfv[0] * gd[0] + dfv[1] * gd[1] + dfv[2] * gd[2];
      for (int j = 0; j < 3; ++j) g_df[j] = gd[j] / nd - dfv[j] * dot / (nd * nd * nd);
    } else {
      for (int j = 0; j < 3; ++j) g_df[j] = gd[j] / 1e-6f;
    }
    float ga[3], gb[3], gxa[3], gxb[3];
    for (int j = 0; j < 3; ++j) {
      ga[j] = g_df[j] + gd[3] * bb[j];
      gb[j] = -g_df[j] + gd[3] * a[j];
    }
    unit_eps_bwd(xa, 1e-6f, ga, gxa);
    unit_eps_bwd(xb, 1e-6f, gb, gxb);
    for (int j = 0; j < 3; ++j) gp[j] += gxa[j] + gxb[j];
  }
  if (valid) {
    float* o = d_xyz + ((size_t)b * P + i) * 3;
    o[0] = gp[0], o[1] = gp[1], o[2] = gp[2];
  }
}

// ---------------------------------------------------------------------------
// a7 sample_along_depth backward: d_xyz (B,N,Ns,3), d_dn (B,N,Ns) -> d_rays[..., 8:10] (B,N,2)
// (ray near / far; the volume bounds in columns 10-11 are detached, utils.py:150).
// ---------------------------------------------------------------------------
__global__ void sample_along_depth_bwd_kernel(const float* __restrict__ rays, const float* __restrict__ d_xyz,
                                              const float* __restrict__ d_dn, long nrays, int Ns, int depth_inv,
                                              float* __restrict__ d_near_far) {
  long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrays) return;
  const float* ry = rays + r * 12;
  float rn = ry[8], rf = ry[9], vn = ry[10], vf = ry[11];
  float g_rn = 0.f, g_rf = 0.f;
  for (int k = 0; k < Ns; ++k) {
    float t = Ns == 1 ? 0.5f : linspace01(k, Ns);
    float z = rn + (rf - rn) * t;
    const float* gx = d_xyz + (r * Ns + k) * 3;
    float gd = gx[0] * ry[3] + gx[1] * ry[4] + gx[2] * ry[5];
    float gz;
    if (depth_inv) {
      gz = z > 1e-6f ? -gd / (z * z) : 0.f;
      gz += -d_dn[r * Ns + k] / fmaxf(vn - vf, 1e-6f);
    } else {
      gz = gd + d_dn[r * Ns + k] / fmaxf(vf - vn, 1e-6f);
    }
    g_rn += gz * (1.f - t);
    g_rf += gz * t;
  }
  d_near_far[r * 2] = g_rn, d_near_far[r * 2 + 1] = g_rf;
}

// scatter of a bilinear (align_corners) upsample at one destination pixel
template <class Acc>
__device__ __forceinline__ void upsample_scatter(Acc& acc, float* __restrict__ g, int W, const Lerp1& ly, const Lerp1& lx,
                                                 float v) {
  acc.add(g + ly.i0 * W + lx.i0, ly.l0 * lx.l0 * v);
  acc.add(g + ly.i0 * W + lx.i1, ly.l0 * lx.l1 * v);
  acc.add(g + ly.i1 * W + lx.i0, ly.l1 * lx.l0 * v);
  acc.add(g + ly.i1 * W + lx.i1, ly.l1 * lx.l1 * v);
}

// ---------------------------------------------------------------------------
// a6 build_rays backward: d_near_far (B,N,2) (ray near / far) -> d_depth, d_std (B,hv,wv) (atomics).
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void build_rays_bwd_kernel(const float* __restrict__ rays, const float* __restrict__ depth,
                                      const float* __restrict__ std_, const float* __restrict__ near_far,
                                      const float* __restrict__ d_nf, int N, int hv, int wv, int Hr, int Wr,
                                      int depth_inv, float* __restrict__ d_depth, float* __restrict__ d_std, size_t n_out,
                                      FixedWs fixed) {
  ScatterAcc<MODE> acc_d = make_acc<MODE>(d_depth, fixed, 0, 0), acc_s = make_acc<MODE>(d_std, fixed, n_out, 1);
  int b = blockIdx.y;
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* r = rays + ((size_t)b * N + n) * 8;
  size_t hw = (size_t)hv * wv;
  int x = min(max((int)r[6], 0), Wr - 1), y = min(max((int)r[7], 0), Hr - 1);
  Lerp1 ly = upsample_axis(y, hv, Hr), lx = upsample_axis(x, wv, Wr);
  if (Hr == hv && Wr == wv) {
    ly.i0 = ly.i1 = y, lx.i0 = lx.i1 = x;
    ly.l0 = lx.l0 = 1.f, ly.l1 = lx.l1 = 0.f;
  }
  float dep = upsample_fetch(depth + b * hw, wv, ly, lx), sd = upsample_fetch(std_ + b * hw, wv, ly, lx);
  float vn = upsample_fetch(near_far + b * 2 * hw, wv, ly, lx), vf = upsample_fetch(near_far + b * 2 * hw + hw, wv, ly, lx);
  float g_rn = d_nf[((size_t)b * N + n) * 2], g_rf = d_nf[((size_t)b * N + n) * 2 + 1];
  float g_dep = 0.f, g_sd = 0.f;
  if (depth_inv) {  // rn = min(dep+sd, vn), rf = max(dep-sd, vf)
    if (!(dep + sd > vn)) g_dep += g_rn, g_sd += g_rn;
    if (!(dep - sd < vf)) g_dep += g_rf, g_sd -= g_rf;
  } else {          // rn = max(dep-sd, vn), rf = min(dep+sd, vf)
    if (!(dep - sd < vn)) g_dep += g_rn, g_sd -= g_rn;
    if (!(dep + sd > vf)) g_dep += g_rf, g_sd += g_rf;
  }
  upsample_scatter(acc_d, d_depth + b * hw, wv, ly, lx, g_dep);
  upsample_scatter(acc_s, d_std + b * hw, wv, ly, lx, g_sd);
}

// ---------------------------------------------------------------------------
// a5 depth_regression backward: d_depth, d_std (B,h,w) -> d_prob, d_values (B,D,h,w)
// ---------------------------------------------------------------------------
// DT > 0: D == DT, the DT hypotheses of a pixel are loaded once (all loads in flight together) and the passes run out
// of registers, like the forward; DT == 0: generic multi-pass form for any D.
template <int DT>
__global__ void __launch_bounds__(64) depth_regress_bwd_kernel(const float* __restrict__ prob,
                                                                const float* __restrict__ dvals,
                                                                const float* __restrict__ g_depth,
                                                                const float* __restrict__ g_std, int D, int hw,
                                                                int depth_inv, float* __restrict__ d_prob,
                                                                float* __restrict__ d_values) {
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  const float* p = prob + (size_t)b * D * hw + i;
  const float* v = dvals + (size_t)b * D * hw + i;
  const float gd = g_depth[(size_t)b * hw + i], gs = g_std[(size_t)b * hw + i];
  float* dp = d_prob + (size_t)b * D * hw + i;
  float* dvv = d_values + (size_t)b * D * hw + i;
  if constexpr (DT > 0) {
    float e[DT], raw[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) e[d] = p[(size_t)d * hw], raw[d] = v[(size_t)d * hw];
    float mx = -INFINITY;
#pragma unroll
    for (int d = 0; d < DT; ++d) mx = fmaxf(mx, e[d]);
    float den = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) e[d] = expf(e[d] - mx), den += e[d];
    float mean = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      e[d] = e[d] / den;
      mean += e[d] * (depth_inv ? 1.f / fmaxf(raw[d], 1e-6f) : raw[d]);
    }
    float var = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const float val = depth_inv ? 1.f / fmaxf(raw[d], 1e-6f) : raw[d];
      var += e[d] * (val - mean) * (val - mean);
    }
    const float gvar = var > 1e-10f ? gs / (2.f * sqrtf(var)) : 0.f;
    float dotp = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const float val = depth_inv ? 1.f / fmaxf(raw[d], 1e-6f) : raw[d];
      dotp += e[d] * (gd * val + gvar * (val - mean) * (val - mean));
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const float val = depth_inv ? 1.f / fmaxf(raw[d], 1e-6f) : raw[d];
      const float gp = gd * val + gvar * (val - mean) * (val - mean);
      dp[(size_t)d * hw] = e[d] * (gp - dotp);
      float gv = e[d] * (gd + 2.f * gvar * (val - mean));
      if (depth_inv) gv = raw[d] > 1e-6f ? -gv / (raw[d] * raw[d]) : 0.f;
      dvv[(size_t)d * hw] = gv;
    }
  } else {
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) mx = fmaxf(mx, p[(size_t)d * hw]);
    float den = 0.f;
    for (int d = 0; d < D; ++d) den += expf(p[(size_t)d * hw] - mx);
    float mean = 0.f;
    for (int d = 0; d < D; ++d) {
      float val = v[(size_t)d * hw];
      if (depth_inv) val = 1.f / fmaxf(val, 1e-6f);
      mean += expf(p[(size_t)d * hw] - mx) / den * val;
    }
    float var = 0.f;
    for (int d = 0; d < D; ++d) {
      float val = v[(size_t)d * hw];
      if (depth_inv) val = 1.f / fmaxf(val, 1e-6f);
      var += expf(p[(size_t)d * hw] - mx) / den * (val - mean) * (val - mean);
    }
    float gvar = var > 1e-10f ? gs / (2.f * sqrtf(var)) : 0.f;
    // sum_d p_d (v_d - mean) = 0, so the variance does not feed back into d_mean
    float dotp = 0.f;
    for (int d = 0; d < D; ++d) {
      float val = v[(size_t)d * hw];
      if (depth_inv) val = 1.f / fmaxf(val, 1e-6f);
      float pd = expf(p[(size_t)d * hw] - mx) / den;
      dotp += pd * (gd * val + gvar * (val - mean) * (val - mean));
    }
    for (int d = 0; d < D; ++d) {
      float raw = v[(size_t)d * hw];
      float val = depth_inv ? 1.f / fmaxf(raw, 1e-6f) : raw;
      float pd = expf(p[(size_t)d * hw] - mx) / den;
      float gp = gd * val + gvar * (val - mean) * (val - mean);
      dp[(size_t)d * hw] = pd * (gp - dotp);
      float gv = pd * (gd + 2.f * gvar * (val - mean));
      if (depth_inv) gv = raw > 1e-6f ? -gv / (raw * raw) : 0.f;
      dvv[(size_t)d * hw] = gv;
    }
  }
}

// ---------------------------------------------------------------------------
// a2 get_depth_values (cascade) backward: d_depth_values (B,D,h,w) -> d_depth, d_std (B,h0,w0) (atomics)
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void depth_values_cascade_bwd_kernel(const float* __restrict__ depth, const float* __restrict__ std_,
                                                const float* __restrict__ near_far, const float* __restrict__ g_dv,
                                                int h0, int w0, int h, int w, int D, float* __restrict__ d_depth,
                                                float* __restrict__ d_std, size_t n_out, FixedWs fixed) {
  ScatterAcc<MODE> acc_d = make_acc<MODE>(d_depth, fixed, 0, 0), acc_s = make_acc<MODE>(d_std, fixed, n_out, 1);
  int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int hw = h * w;
  if (i >= hw) return;
  int y = i / w, x = i - y * w;
  Lerp1 ly = upsample_axis(y, h0, h), lx = upsample_axis(x, w0, w);
  size_t o = (size_t)b * h0 * w0;
  float dep = upsample_fetch(depth + o, w0, ly, lx), sd = upsample_fetch(std_ + o, w0, ly, lx);
  float nf0 = upsample_fetch(near_far + o * 2, w0, ly, lx), nf1 = upsample_fetch(near_far + o * 2 + (size_t)h0 * w0, w0, ly, lx);
  float hi = dep + sd, lo = dep - sd;
  bool hi_free = !(hi > nf0), lo_free = !(lo < nf1);
  if (!hi_free) hi = nf0;
  if (!lo_free) lo = nf1;
  float g_near = 0.f, g_far = 0.f;
  float step = D > 1 ? 1.f / (float)(D - 1) : 0.f;
  for (int d = 0; d < D; ++d) {
    float t = d < D / 2 ? (float)d * step : 1.f - (float)(D - 1 - d) * step;
    float g = g_dv[((size_t)b * D + d) * hw + i];
    g_near += g * (1.f - t);
    g_far += g * t;
  }
  float g_hi = -g_near / (hi * hi), g_lo = -g_far / (lo * lo);
  float g_dep = (hi_free ? g_hi : 0.f) + (lo_free ? g_lo : 0.f);
  float g_sd = (hi_free ? g_hi : 0.f) - (lo_free ? g_lo : 0.f);
  upsample_scatter(acc_d, d_depth + o, w0, ly, lx, g_dep);
  upsample_scatter(acc_s, d_std + o, w0, ly, lx, g_sd);
}

// ---------------------------------------------------------------------------
// a3+a4 sweep backward: d_var (B,C,D,h,w) -> d_feats (B,S,C,Hs,Ws) (atomics), d_depth_values (B,D,h,w) or null.
// One thread per voxel, one workgroup per 16 x 16 tile of one plane and 8 channels: the S warped values per channel
// are recomputed (never stored by the forward), then the feature gradient of each view goes through the LDS window.
// ---------------------------------------------------------------------------
template <int CB, int S, int MODE>
__global__ void __launch_bounds__(256) sweep_bwd_kernel(const float* __restrict__ feats,
                                                         const float* __restrict__ proj,
                                                         const float* __restrict__ dv,
                                                         const float* __restrict__ g_var, int C, int Hs, int Ws, int D,
                                                         int h, int w, int tiles_x, int tiles_y,
                                                         float* __restrict__ d_feats, float* __restrict__ d_dv,
                                                         size_t n_feats, FixedWs fixed) {
  static_assert(CB <= kWinCh, "one window pass per channel block");
  __shared__ float win[kWinCh * kWinCap];
  ScatterAcc<MODE> acc_f = make_acc<MODE>(d_feats, fixed, 0, 0), acc_v = make_acc<MODE>(d_dv, fixed, n_feats, 1);
  __shared__ WinBox box;
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * CB;
  const size_t nvox = (size_t)D * h * w;
  const int tile = blockIdx.x, d = tile / (tiles_x * tiles_y), tr = tile - d * (tiles_x * tiles_y);
  const int x = (tr % tiles_x) * 16 + (threadIdx.x & 15), y = (tr / tiles_x) * 16 + (threadIdx.x >> 4);
  const bool valid = x < w && y < h;
  const size_t i = ((size_t)d * h + (valid ? y : 0)) * w + (valid ? x : 0);
  const float depth = dv[(size_t)b * nvox + i];
  const size_t plane = (size_t)Hs * Ws;
  Taps2 tp[S];
  TapSet ts[S];
  float ex[S], ey[S], ax[S], ay[S], pxs[S], pys[S], pzs[S];
  bool vx0[S], vx1[S], vy0[S], vy1[S];
  for (int s = 0; s < S; ++s) {
    const float* P = proj + ((size_t)b * S + s) * 12;
    float px = P[0] * x + P[1] * y + P[2] + P[3] / depth;
    float py = P[4] * x + P[5] * y + P[6] + P[7] / depth;
    float pz = P[8] * x + P[9] * y + P[10] + P[11] / depth;
    pxs[s] = px, pys[s] = py, pzs[s] = pz;
    float z = fmaxf(pz, 1e-6f);
    float ix = unnorm((px / z) / ((float)(Ws - 1) * 0.5f) - 1.f, Ws), iy = unnorm((py / z) / ((float)(Hs - 1) * 0.5f) - 1.f, Hs);
    tp[s] = taps_zeros(ix, iy, Ws, Hs);
    float fx = floorf(ix), fy = floorf(iy);
    ex[s] = (fx + 1.f) - ix, ey[s] = (fy + 1.f) - iy, ax[s] = ix - fx, ay[s] = iy - fy;
    int x0 = (int)fminf(fmaxf(fx, -2.f), (float)Ws), y0 = (int)fminf(fmaxf(fy, -2.f), (float)Hs);
    vx0[s] = x0 >= 0 && x0 <= Ws - 1, vx1[s] = x0 + 1 >= 0 && x0 + 1 <= Ws - 1;
    vy0[s] = y0 >= 0 && y0 <= Hs - 1, vy1[s] = y0 + 1 >= 0 && y0 + 1 <= Hs - 1;
    ts[s] = TapSet{x0, y0, x0 + 1, y0 + 1, tp[s].w00, tp[s].w01, tp[s].w10, tp[s].w11};   // weight 0 where out of range
  }
  float gix[S], giy[S], gw[S][CB];
  for (int s = 0; s < S; ++s) gix[s] = giy[s] = 0.f;
  for (int c = 0; c < CB; ++c) {
    float v[S][4], wv[S], mean = 0.f;
    for (int s = 0; s < S; ++s) {
      const float* f = feats + (((size_t)b * S + s) * C + c0 + c) * plane;
      v[s][0] = f[tp[s].o00], v[s][1] = f[tp[s].o01], v[s][2] = f[tp[s].o10], v[s][3] = f[tp[s].o11];
      wv[s] = v[s][0] * tp[s].w00 + v[s][1] * tp[s].w01 + v[s][2] * tp[s].w10 + v[s][3] * tp[s].w11;
      mean += wv[s];
    }
    mean /= (float)S;
    const float g = valid ? g_var[((size_t)b * C + c0 + c) * nvox + i] : 0.f;
    for (int s = 0; s < S; ++s) {
      const float gws = (2.f / (float)S) * g * (wv[s] - mean);  // d var / d warped_s
      gw[s][c] = gws;
      if (d_dv) {  // d warped / d (ix, iy): only in-bounds taps contribute (zeros padding)
        float a00 = (vx0[s] && vy0[s]) ? v[s][0] : 0.f, a01 = (vx1[s] && vy0[s]) ? v[s][1] : 0.f;
        float a10 = (vx0[s] && vy1[s]) ? v[s][2] : 0.f, a11 = (vx1[s] && vy1[s]) ? v[s][3] : 0.f;
        gix[s] += gws * ((a01 - a00) * ey[s] + (a11 - a10) * ay[s]);
        giy[s] += gws * ((a10 - a00) * ex[s] + (a11 - a01) * ax[s]);
      }
    }
  }
  for (int s = 0; s < S; ++s) {
    float* df = d_feats + (((size_t)b * S + s) * C + c0) * plane;
    if (MODE == 0 && win_begin(&box, ts[s], valid))
      win_scatter(acc_f, win, &box, ts[s], valid, gw[s], CB, df, plane, Ws);
    else if (valid)
      direct_scatter(acc_f, ts[s], gw[s], CB, df, plane, Ws);
  }
  if (d_dv && valid) {
    float gdepth = 0.f;
    for (int s = 0; s < S; ++s) {
      const float* P = proj + ((size_t)b * S + s) * 12;
      float z = fmaxf(pzs[s], 1e-6f);
      // ix = px / z (normalise / unnormalise cancel); p = A + T / depth
      float gpx = gix[s] / z, gpy = giy[s] / z;
      float gpz = pzs[s] > 1e-6f ? -(gix[s] * pxs[s] + giy[s] * pys[s]) / (z * z) : 0.f;
      gdepth += -(gpx * P[3] + gpy * P[7] + gpz * P[11]) / (depth * depth);
    }
    acc_v.add(d_dv + (size_t)b * nvox + i, gdepth);  // channel blocks of one voxel add up
  }
}

}  // namespace bmv

using namespace bmv;

extern "C" {

int bmv_composite_bwd(const float* raw, const float* z_vals, const float* d_rgb, const float* d_depth, long nrays,
                      int Ns, float* d_raw, bmv_stream_t stream) {
  BMV_REQUIRE(raw && z_vals && d_rgb && d_raw, "bmv_composite_bwd: null pointer");
  BMV_REQUIRE(nrays >= 0 && Ns > 0, "bmv_composite_bwd: bad shape");
  if (nrays == 0) return BMV_OK;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(cdiv(nrays, 256)), dim3(256), 0, as_stream(stream), raw, z_vals, d_rgb,
                     d_depth, nrays, Ns, d_raw);
  BMV_LAUNCH_END("bmv_composite_bwd");
}

int bmv_blend_bwd(const float* raws, const float* masks, const float* d_rgb, int B, int K, int N, int Ns,
                  float* d_raws, bmv_stream_t stream) {
  BMV_REQUIRE(raws && masks && d_rgb && d_raws, "bmv_blend_bwd: null pointer");
  BMV_REQUIRE(B > 0 && K > 0 && N >= 0 && Ns > 0, "bmv_blend_bwd: bad shape");
  if (N == 0) return BMV_OK;
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, as_stream(stream), raws, masks, d_rgb, K, N,
                     Ns, d_raws);
  BMV_LAUNCH_END("bmv_blend_bwd");
}

// tile of rays a 256-thread workgroup takes when the layout hint is usable (whole rows of rays, Ns a power of two <= 64)
static bool ray_tiles(int P, int& ray_w, int Ns, int& tw, int& th, int& tiles_x, int& nblocks) {
  tw = th = tiles_x = 0;
  nblocks = (int)cdiv(P, 256);
  if (ray_w > 0 && Ns > 0 && Ns <= 64 && (Ns & (Ns - 1)) == 0 && P % (ray_w * Ns) == 0) {
    const int nr = 256 / Ns, ray_h = P / (ray_w * Ns);
    th = 1;
    while (th * th * 4 <= nr) th *= 2;       // largest power of two with th^2 <= nr, so tw >= th
    tw = nr / th;
    tiles_x = (ray_w + tw - 1) / tw;
    nblocks = tiles_x * ((ray_h + th - 1) / th);
    return true;
  }
  ray_w = 0;
  return false;
}

static int vox_feat_bwd_impl(const float* uvd01, const float* volume, const float* d_out, int B, int P, int C, int D,
                             int h, int w, int ray_w, int Ns, float* d_volume, float* d_d01, FixedWs fixed,
                             bmv_stream_t stream, const char* name) {
  BMV_REQUIRE(uvd01 && volume && d_out && d_volume && d_d01, "%s: null pointer", name);
  BMV_REQUIRE(B > 0 && P >= 0 && C > 0 && D > 0 && h > 0 && w > 0, "%s: bad shape", name);
  if (P == 0) {      // the float form adds into a caller-zeroed buffer; the fixed-point form WRITES its output
    if (fixed.ws) {
      fixed_zero(d_volume, (size_t)B * C * D * h * w, as_stream(stream));
      BMV_LAUNCH_END(name);
    }
    return BMV_OK;
  }
  int tw, th, tiles_x, nblocks;
  ray_tiles(P, ray_w, Ns, tw, th, tiles_x, nblocks);
  launch_modes(fixed, as_stream(stream), [&](auto mode) {
    hipLaunchKernelGGL(vox_feat_bwd_kernel<decltype(mode)::value>, dim3(nblocks, B), dim3(256), 0, as_stream(stream), uvd01,
                       volume, d_out, P, C, D, h, w, ray_w, Ns, tw, th, tiles_x, d_volume, d_d01, fixed);
  });
  if (fixed.ws) fixed_finish(fixed, 0, (size_t)B * C * D * h * w, d_volume, as_stream(stream));
  BMV_LAUNCH_END(name);
}
int bmv_vox_feat_bwd(const float* uvd01, const float* volume, const float* d_out, int B, int P, int C, int D, int h,
                     int w, int ray_w, int Ns, float* d_volume, float* d_d01, bmv_stream_t stream) {
  return vox_feat_bwd_impl(uvd01, volume, d_out, B, P, C, D, h, w, ray_w, Ns, d_volume, d_d01, FixedWs{}, stream,
                           "bmv_vox_feat_bwd");
}
int bmv_vox_feat_bwd_fixed(const float* uvd01, const float* volume, const float* d_out, int B, int P, int C, int D, int h,
                           int w, int ray_w, int Ns, float* d_volume, float* d_d01, long long* workspace,
                           bmv_stream_t stream) {
  BMV_REQUIRE(workspace, "bmv_vox_feat_bwd_fixed: null workspace");
  return vox_feat_bwd_impl(uvd01, volume, d_out, B, P, C, D, h, w, ray_w, Ns, d_volume, d_d01, FixedWs{workspace}, stream,
                           "bmv_vox_feat_bwd_fixed");
}

static int img_feat_bwd_impl(const float* xyz, const float* img_feat_rgb, const float* src_exts, const float* src_ixts,
                             const float* tar_ext, float render_scale, const float* d_out, int B, int P, int S, int C,
                             int c_grad, int H, int W, int ray_w, int Ns, float* d_img, float* d_xyz, FixedWs fixed,
                             bmv_stream_t stream, const char* name) {
  BMV_REQUIRE(xyz && img_feat_rgb && src_exts && src_ixts && tar_ext && d_out && d_img && d_xyz, "%s: null pointer", name);
  BMV_REQUIRE(B > 0 && P >= 0 && S > 0 && S <= 16 && C > 0 && H > 1 && W > 1, "%s: bad shape", name);
  BMV_REQUIRE(c_grad >= 0 && c_grad <= C, "%s: c_grad=%d outside [0, %d]", name, c_grad, C);
  if (P == 0) {
    if (fixed.ws) {
      fixed_zero(d_img, (size_t)B * S * C * H * W, as_stream(stream));
      BMV_LAUNCH_END(name);
    }
    return BMV_OK;
  }
  int tw, th, tiles_x, nblocks;
  ray_tiles(P, ray_w, Ns, tw, th, tiles_x, nblocks);
  launch_modes(fixed, as_stream(stream), [&](auto mode) {
    hipLaunchKernelGGL(img_feat_bwd_kernel<decltype(mode)::value>, dim3(nblocks, B), dim3(256), 0, as_stream(stream), xyz,
                       img_feat_rgb, src_exts, src_ixts, tar_ext, render_scale, d_out, P, S, C, c_grad, H, W, ray_w, Ns,
                       tw, th, tiles_x, d_img, d_xyz, fixed);
  });
  if (fixed.ws) fixed_finish(fixed, 0, (size_t)B * S * C * H * W, d_img, as_stream(stream));
  BMV_LAUNCH_END(name);
}
int bmv_img_feat_bwd(const float* xyz, const float* img_feat_rgb, const float* src_exts, const float* src_ixts,
                     const float* tar_ext, float render_scale, const float* d_out, int B, int P, int S, int C,
                     int c_grad, int H, int W, int ray_w, int Ns, float* d_img, float* d_xyz, bmv_stream_t stream) {
  return img_feat_bwd_impl(xyz, img_feat_rgb, src_exts, src_ixts, tar_ext, render_scale, d_out, B, P, S, C, c_grad, H, W,
                           ray_w, Ns, d_img, d_xyz, FixedWs{}, stream, "bmv_img_feat_bwd");
}
int bmv_img_feat_bwd_fixed(const float* xyz, const float* img_feat_rgb, const float* src_exts, const float* src_ixts,
                           const float* tar_ext, float render_scale, const float* d_out, int B, int P, int S, int C,
                           int c_grad, int H, int W, int ray_w, int Ns, float* d_img, float* d_xyz, long long* workspace,
                           bmv_stream_t stream) {
  BMV_REQUIRE(workspace, "bmv_img_feat_bwd_fixed: null workspace");
  return img_feat_bwd_impl(xyz, img_feat_rgb, src_exts, src_ixts, tar_ext, render_scale, d_out, B, P, S, C, c_grad, H, W,
                           ray_w, Ns, d_img, d_xyz, FixedWs{workspace}, stream, "bmv_img_feat_bwd_fixed");
}

int bmv_sample_along_depth_bwd(const float* rays, const float* d_xyz, const float* d_dn, int B, int N, int Ns,
                               int depth_inv, float* d_near_far, bmv_stream_t stream) {
  BMV_REQUIRE(rays && d_xyz && d_dn && d_near_far, "bmv_sample_along_depth_bwd: null pointer");
  BMV_REQUIRE(B > 0 && N >= 0 && Ns > 0, "bmv_sample_along_depth_bwd: bad shape");
  long nrays = (long)B * N;
  if (nrays == 0) return BMV_OK;
  hipLaunchKernelGGL(sample_along_depth_bwd_kernel, dim3(cdiv(nrays, 256)), dim3(256), 0, as_stream(stream), rays, d_xyz,
                     d_dn, nrays, Ns, depth_inv, d_near_far);
  BMV_LAUNCH_END("bmv_sample_along_depth_bwd");
}

static int build_rays_bwd_impl(const float* rays, const float* depth, const float* std_, const float* near_far,
                               const float* d_near_far, int B, int N, int hv, int wv, int Hr, int Wr, int depth_inv,
                               float* d_depth, float* d_std, FixedWs fixed, bmv_stream_t stream, const char* name) {
  BMV_REQUIRE(rays && depth && std_ && near_far && d_near_far && d_depth && d_std, "%s: null pointer", name);
  BMV_REQUIRE(B > 0 && N >= 0 && hv > 0 && wv > 0 && Hr > 0 && Wr > 0, "%s: bad shape", name);
  const size_t n_out = (size_t)B * hv * wv;
  if (N == 0) {
    if (fixed.ws) {
      fixed_zero(d_depth, n_out, as_stream(stream));
      fixed_zero(d_std, n_out, as_stream(stream));
      BMV_LAUNCH_END(name);
    }
    return BMV_OK;
  }
  launch_modes(fixed, as_stream(stream), [&](auto mode) {
    hipLaunchKernelGGL(build_rays_bwd_kernel<decltype(mode)::value>, dim3(cdiv(N, 256), B), dim3(256), 0, as_stream(stream),
                       rays, depth, std_, near_far, d_near_far, N, hv, wv, Hr, Wr, depth_inv, d_depth, d_std, n_out, fixed);
  });
  if (fixed.ws) {
    fixed_finish(fixed, 0, n_out, d_depth, as_stream(stream));
    fixed_finish(fixed, n_out, n_out, d_std, as_stream(stream), 1);
  }
  BMV_LAUNCH_END(name);
}
int bmv_build_rays_bwd(const float* rays, const float* depth, const float* std_, const float* near_far,
                       const float* d_near_far, int B, int N, int hv, int wv, int Hr, int Wr, int depth_inv,
                       float* d_depth, float* d_std, bmv_stream_t stream) {
  return build_rays_bwd_impl(rays, depth, std_, near_far, d_near_far, B, N, hv, wv, Hr, Wr, depth_inv, d_depth, d_std,
                             FixedWs{}, stream, "bmv_build_rays_bwd");
}
int bmv_build_rays_bwd_fixed(const float* rays, const float* depth, const float* std_, const float* near_far,
                             const float* d_near_far, int B, int N, int hv, int wv, int Hr, int Wr, int depth_inv,
                             float* d_depth, float* d_std, long long* workspace, bmv_stream_t stream) {
  BMV_REQUIRE(workspace, "bmv_build_rays_bwd_fixed: null workspace");
  return build_rays_bwd_impl(rays, depth, std_, near_far, d_near_far, B, N, hv, wv, Hr, Wr, depth_inv, d_depth, d_std,
                             FixedWs{workspace}, stream, "bmv_build_rays_bwd_fixed");
}

int bmv_depth_regress_bwd(const float* depth_prob, const float* depth_values, const float* d_depth,
                          const float* d_std, int B, int D, int h, int w, int depth_inv, float* d_prob,
                          float* d_values, bmv_stream_t stream) {
  BMV_REQUIRE(depth_prob && depth_values && d_depth && d_std && d_prob && d_values, "bmv_depth_regress_bwd: null pointer");
  BMV_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "bmv_depth_regress_bwd: bad shape");
#define DRB(DT)                                                                                                  \
  hipLaunchKernelGGL(depth_regress_bwd_kernel<DT>, dim3(cdiv(h * w, 64), B), dim3(64), 0, as_stream(stream), depth_prob, \
                     depth_values, d_depth, d_std, D, h * w, depth_inv, d_prob, d_values)
  switch (D) {
    case 8: DRB(8); break;
    case 16: DRB(16); break;
    case 32: DRB(32); break;
    case 64: DRB(64); break;
    default: DRB(0); break;
  }
#undef DRB
  BMV_LAUNCH_END("bmv_depth_regress_bwd");
}

static int depth_values_cascade_bwd_impl(const float* depth, const float* std_, const float* near_far,
                                         const float* d_depth_values, int B, int h0, int w0, int h, int w, int D,
                                         float* d_depth, float* d_std, FixedWs fixed, bmv_stream_t stream,
                                         const char* name) {
  BMV_REQUIRE(depth && std_ && near_far && d_depth_values && d_depth && d_std, "%s: null pointer", name);
  BMV_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0 && h0 > 0 && w0 > 0, "%s: bad shape", name);
  const size_t n_out = (size_t)B * h0 * w0;
  launch_modes(fixed, as_stream(stream), [&](auto mode) {
    hipLaunchKernelGGL(depth_values_cascade_bwd_kernel<decltype(mode)::value>, dim3(cdiv(h * w, 256), B), dim3(256), 0,
                       as_stream(stream), depth, std_, near_far, d_depth_values, h0, w0, h, w, D, d_depth, d_std, n_out,
                       fixed);
  });
  if (fixed.ws) {
    fixed_finish(fixed, 0, n_out, d_depth, as_stream(stream));
    fixed_finish(fixed, n_out, n_out, d_std, as_stream(stream), 1);
  }
  BMV_LAUNCH_END(name);
}
int bmv_depth_values_cascade_bwd(const float* depth, const float* std_, const float* near_far,
                                 const float* d_depth_values, int B, int h0, int w0, int h, int w, int D,
                                 float* d_depth, float* d_std, bmv_stream_t stream) {
  return depth_values_cascade_bwd_impl(depth, std_, near_far, d_depth_values, B, h0, w0, h, w, D, d_depth, d_std,
                                       FixedWs{}, stream, "bmv_depth_values_cascade_bwd");
}
int bmv_depth_values_cascade_bwd_fixed(const float* depth, const float* std_, const float* near_far,
                                       const float* d_depth_values, int B, int h0, int w0, int h, int w, int D,
                                       float* d_depth, float* d_std, long long* workspace, bmv_stream_t stream) {
  BMV_REQUIRE(workspace, "bmv_depth_values_cascade_bwd_fixed: null workspace");
  return depth_values_cascade_bwd_impl(depth, std_, near_far, d_depth_values, B, h0, w0, h, w, D, d_depth, d_std,
                                       FixedWs{workspace}, stream, "bmv_depth_values_cascade_bwd_fixed");
}

static int sweep_variance_bwd_impl(const float* feats, const float* proj, const float* depth_values,
                                   const float* d_variance, int B, int S, int C, int Hs, int Ws, int D, int h, int w,
                                   float* d_feats, float* d_depth_values, FixedWs fixed, bmv_stream_t stream,
                                   const char* name) {
  BMV_REQUIRE(feats && proj && depth_values && d_variance && d_feats, "%s: null pointer", name);
  BMV_REQUIRE(B > 0 && C > 0 && Hs > 1 && Ws > 1 && D > 0 && h > 0 && w > 0, "%s: bad shape", name);
  if (S < 2 || S > 4 || C % 8 != 0) {
    set_error("%s: built for 2..4 views and C %% 8 == 0 (got S=%d C=%d)", name, S, C);
    return BMV_ERR_UNSUPPORTED;
  }
  const int tiles_x = (w + 15) / 16, tiles_y = (h + 15) / 16;
  const dim3 grid(tiles_x * tiles_y * D, C / 8, B);
  const size_t n_feats = (size_t)B * S * C * Hs * Ws, n_dv = d_depth_values ? (size_t)B * D * h * w : 0;
  launch_modes(fixed, as_stream(stream), [&](auto mode) {
    constexpr int M = decltype(mode)::value;
#define SB(SV)                                                                                                          \
  hipLaunchKernelGGL((sweep_bwd_kernel<8, SV, M>), grid, dim3(256), 0, as_stream(stream), feats, proj, depth_values,      \
                     d_variance, C, Hs, Ws, D, h, w, tiles_x, tiles_y, d_feats, d_depth_values, n_feats, fixed)
    if (S == 3)
      SB(3);
    else if (S == 2)
      SB(2);      // ENeRF pre-training draws 2 / 3 / 4 source views (dtu_pretrain.yaml:22-23)
    else
      SB(4);
#undef SB
  });
  if (fixed.ws) {
    fixed_finish(fixed, 0, n_feats, d_feats, as_stream(stream));
    fixed_finish(fixed, n_feats, n_dv, d_depth_values, as_stream(stream), 1);
  }
  BMV_LAUNCH_END(name);
}
int bmv_sweep_variance_bwd(const float* feats, const float* proj, const float* depth_values, const float* d_variance,
                           int B, int S, int C, int Hs, int Ws, int D, int h, int w, float* d_feats,
                           float* d_depth_values, bmv_stream_t stream) {
  return sweep_variance_bwd_impl(feats, proj, depth_values, d_variance, B, S, C, Hs, Ws, D, h, w, d_feats, d_depth_values,
                                 FixedWs{}, stream, "bmv_sweep_variance_bwd");
}
int bmv_sweep_variance_bwd_fixed(const float* feats, const float* proj, const float* depth_values,
                                 const float* d_variance, int B, int S, int C, int Hs, int Ws, int D, int h, int w,
                                 float* d_feats, float* d_depth_values, long long* workspace, bmv_stream_t stream) {
  BMV_REQUIRE(workspace, "bmv_sweep_variance_bwd_fixed: null workspace");
  return sweep_variance_bwd_impl(feats, proj, depth_values, d_variance, B, S, C, Hs, Ws, D, h, w, d_feats, d_depth_values,
                                 FixedWs{workspace}, stream, "bmv_sweep_variance_bwd_fixed");
}

// int64 words of the workspace of a *_fixed call whose scatter outputs hold n_out floats in total; zeroed by the caller
long bmv_fixed_workspace(long n_out) {
  if (n_out < 0) {
    set_error("bmv_fixed_workspace: n_out=%ld", n_out);
    return BMV_ERR_INVALID;
  }
  return (long)kFixedHeader + n_out;
}

}  // extern "C"
