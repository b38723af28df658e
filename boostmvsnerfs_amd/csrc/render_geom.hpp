// Per-sample geometry of the renderer (a6, a7, a9, a10, a14), written once as
// device functions so the stand-alone entry points and the fused render kernel
// run the same arithmetic.  Reference: lib/networks/enerf/utils.py:392-520,753-786.
#pragma once
#include "bmv_common.hpp"

namespace bmv {

struct Cam {     // one source view, as used by a10 / a14
  float E[12];   // rows 0-2 of the world->camera matrix
  float K[9];    // intrinsics with rows 0-1 scaled by render_scale   (a10)
  float Kf[9];   // full-resolution intrinsics                        (a14)
  float c[3];    // camera centre = inverse(E)[:3, 3]
};

// inverse(E)[:3,3] for a general affine E (last row 0 0 0 1): -A^-1 t via the adjugate.
__device__ __forceinline__ void camera_centre(const float* __restrict__ E, float* c) {
  float a = E[0], b = E[1], cc = E[2], d = E[4], e = E[5], f = E[6], g = E[8], h = E[9], i = E[10];
  float tx = E[3], ty = E[7], tz = E[11];
  float A = e * i - f * h, Bc = -(d * i - f * g), Cc = d * h - e * g;
  float det = a * A + b * Bc + cc * Cc;
  float inv = 1.f / det;
  float i00 = A * inv, i01 = -(b * i - cc * h) * inv, i02 = (b * f - cc * e) * inv;
  float i10 = Bc * inv, i11 = (a * i - cc * g) * inv, i12 = -(a * f - cc * d) * inv;
  float i20 = Cc * inv, i21 = -(a * h - b * g) * inv, i22 = (a * e - b * d) * inv;
  c[0] = -(i00 * tx + i01 * ty + i02 * tz);
  c[1] = -(i10 * tx + i11 * ty + i12 * tz);
  c[2] = -(i20 * tx + i21 * ty + i22 * tz);
}

__device__ __forceinline__ void load_cam(const float* __restrict__ ext, const float* __restrict__ ixt,
                                         float render_scale, Cam& cam) {
  for (int k = 0; k < 12; ++k) cam.E[k] = ext[k];
  for (int k = 0; k < 9; ++k) {
    cam.Kf[k] = ixt[k];
    cam.K[k] = k < 6 ? ixt[k] * render_scale : ixt[k];
  }
  camera_centre(ext, cam.c);
}

// a6: per-ray [near, far] and volume bounds at integer pixel (x, y) of the render image.
__device__ __forceinline__ void ray_bounds(const float* __restrict__ depth, const float* __restrict__ std_,
                                           const float* __restrict__ near_far, int hv, int wv, int Hr, int Wr,
                                           int x, int y, bool depth_inv, float& rn, float& rf, float& vn, float& vf) {
  float dep, sd;
  size_t hw = (size_t)hv * wv;
  x = min(max(x, 0), Wr - 1);
  y = min(max(y, 0), Hr - 1);
  if (Hr == hv && Wr == wv) {
    size_t o = (size_t)y * wv + x;
    dep = depth[o], sd = std_[o], vn = near_far[o], vf = near_far[hw + o];
  } else {
    Lerp1 ly = upsample_axis(y, hv, Hr), lx = upsample_axis(x, wv, Wr);
    dep = upsample_fetch(depth, wv, ly, lx);
    sd = upsample_fetch(std_, wv, ly, lx);
    vn = upsample_fetch(near_far, wv, ly, lx);
    vf = upsample_fetch(near_far + hw, wv, ly, lx);
  }
  if (depth_inv) {
    rn = dep + sd, rf = dep - sd;
    if (rn > vn) rn = vn;
    if (rf < vf) rf = vf;
  } else {
    rn = dep - sd, rf = dep + sd;
    if (rn < vn) rn = vn;
    if (rf > vf) rf = vf;
  }
}

// torch.linspace(0, 1, Ns)[k]
__device__ __forceinline__ float linspace01(int k, int Ns) {
  if (Ns <= 1) return 0.f;
  float step = 1.f / (float)(Ns - 1);
  return k < Ns / 2 ? (float)k * step : 1.f - (float)(Ns - 1 - k) * step;
}

// a7: depth of sample k, its world position and normalised volume depth coordinate.
__device__ __forceinline__ void sample_point(const float* o, const float* d, float rn, float rf, float vn, float vf,
                                             int k, int Ns, bool depth_inv, float& z, float* xyz, float& dn) {
  z = Ns == 1 ? rn + (rf - rn) * 0.5f : rn + (rf - rn) * linspace01(k, Ns);
  if (depth_inv) {
    float iz = BMV_DIV(1.f, fmaxf(z, 1e-6f));
    xyz[0] = o[0] + d[0] * iz, xyz[1] = o[1] + d[1] * iz, xyz[2] = o[2] + d[2] * iz;
    dn = BMV_DIV(vn - z, fmaxf(vn - vf, 1e-6f));
  } else {
    xyz[0] = o[0] + d[0] * z, xyz[1] = o[1] + d[1] * z, xyz[2] = o[2] + d[2] * z;
    dn = BMV_DIV(z - vn, fmaxf(vf - vn, 1e-6f));
  }
}

// a9: trilinear taps of grid_sample(5-D, zeros, align_corners=True).
struct Taps3 {
  int o[8];
  float w[8];
};
__device__ __forceinline__ Taps3 taps3_zeros(float u01, float v01, float d01, int W, int H, int D) {
  float ix = unnorm(u01 * 2.f - 1.f, W), iy = unnorm(v01 * 2.f - 1.f, H), iz = unnorm(d01 * 2.f - 1.f, D);
  Taps3 t;
  float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  bool ok = (fx >= -1.f) && (fx <= (float)(W - 1)) && (fy >= -1.f) && (fy <= (float)(H - 1)) && (fz >= -1.f) &&
            (fz <= (float)(D - 1));
  if (!ok) {
#pragma unroll
    for (int k = 0; k < 8; ++k) t.o[k] = 0, t.w[k] = 0.f;
    return t;
  }
  int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  float ax = ix - fx, ay = iy - fy, az = iz - fz;
  float ex = (fx + 1.f) - ix, ey = (fy + 1.f) - iy, ez = (fz + 1.f) - iz;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // aten order: tnw, tne, tsw, tse, bnw, bne, bsw, bse  (t = z0, n = y0, w = x0)
    int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
    int x = x0 + dx, y = y0 + dy, z = z0 + dz;
    bool in = x >= 0 && x <= W - 1 && y >= 0 && y <= H - 1 && z >= 0 && z <= D - 1;
    float wgt = (dx ? ax : ex) * (dy ? ay : ey) * (dz ? az : ez);
    t.w[k] = in ? wgt : 0.f;
    t.o[k] = in ? (z * H + y) * W + x : 0;
  }
  return t;
}
__device__ __forceinline__ float tap3_fetch_buf(__amdgpu_buffer_rsrc_t r, const Taps3& t, unsigned soff) {  // byte offsets
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) v += ld_buf(r, (unsigned)t.o[k], soff) * t.w[k];
  return v;
}
__device__ __forceinline__ float tap3_fetch_bytes(const float* __restrict__ p, const Taps3& t) {   // offsets in bytes
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) v += ld_byte_off(p, (unsigned)t.o[k]) * t.w[k];
  return v;
}
__device__ __forceinline__ float tap3_fetch(const float* __restrict__ p, const Taps3& t) {
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) v += p[(unsigned)t.o[k]] * t.w[k];
  return v;
}

// a10: projection of a world point into a source view at render resolution.
__device__ __forceinline__ Taps2 project_taps(const Cam& cam, const float* xyz, int W, int H) {
  // cam = [xyz,1] @ E^T ; q = cam @ K^T   (utils.py:763-766)
  float cx = xyz[0] * cam.E[0] + xyz[1] * cam.E[1] + xyz[2] * cam.E[2] + cam.E[3];
  float cy = xyz[0] * cam.E[4] + xyz[1] * cam.E[5] + xyz[2] * cam.E[6] + cam.E[7];
  float cz = xyz[0] * cam.E[8] + xyz[1] * cam.E[9] + xyz[2] * cam.E[10] + cam.E[11];
  float qx = cx * cam.K[0] + cy * cam.K[1] + cz * cam.K[2];
  float qy = cx * cam.K[3] + cy * cam.K[4] + cz * cam.K[5];
  float qz = cx * cam.K[6] + cy * cam.K[7] + cz * cam.K[8];
  float z = fmaxf(qz, 1e-6f);
  float gx = BMV_DIV(BMV_DIV(qx, z), (float)(W - 1)) * 2.f - 1.f;
  float gy = BMV_DIV(BMV_DIV(qy, z), (float)(H - 1)) * 2.f - 1.f;
  return taps_border(unnorm(gx, W), unnorm(gy, H), W, H);
}

// a10: ray-direction-difference feature [ (a-b)/max(|a-b|,1e-6), a.b ]
__device__ __forceinline__ void dir_feature(const float* xyz, const float* tar_c, const float* src_c, float* out4) {
  float a[3], b[3];
  float na = 0.f, nb = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    a[k] = xyz[k] - tar_c[k];
    b[k] = xyz[k] - src_c[k];
    na += a[k] * a[k];
    nb += b[k] * b[k];
  }
  na = sqrtf(na) + 1e-6f;
  nb = sqrtf(nb) + 1e-6f;
  float df[3], nd = 0.f, dot = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    a[k] = BMV_DIV(a[k], na);
    b[k] = BMV_DIV(b[k], nb);
    df[k] = a[k] - b[k];
    nd += df[k] * df[k];
    dot += a[k] * b[k];
  }
  nd = fmaxf(sqrtf(nd), 1e-6f);
  out4[0] = BMV_DIV(df[0], nd), out4[1] = BMV_DIV(df[1], nd), out4[2] = BMV_DIV(df[2], nd), out4[3] = dot;
}

// a14 get_ndc_coords (enerf/utils.py:490-508): p = K (R x + T); (p.x / p.z / (W-1), p.y / p.z / (H-1), p.z) with the
// full-resolution K and the render-resolution (W-1, H-1) the reference passes
__device__ __forceinline__ void ndc_coords(const Cam& cam, const float* xyz, float inv_w, float inv_h, float& u, float& v,
                                           float& pz) {
  float cx = xyz[0] * cam.E[0] + xyz[1] * cam.E[1] + xyz[2] * cam.E[2];
  float cy = xyz[0] * cam.E[4] + xyz[1] * cam.E[5] + xyz[2] * cam.E[6];
  float cz = xyz[0] * cam.E[8] + xyz[1] * cam.E[9] + xyz[2] * cam.E[10];
  cx += cam.E[3], cy += cam.E[7], cz += cam.E[11];
  float px = cx * cam.Kf[0] + cy * cam.Kf[1] + cz * cam.Kf[2];
  float py = cx * cam.Kf[3] + cy * cam.Kf[4] + cz * cam.Kf[5];
  pz = cx * cam.Kf[6] + cy * cam.Kf[7] + cz * cam.Kf[8];
  u = BMV_DIV(BMV_DIV(px, pz), inv_w), v = BMV_DIV(BMV_DIV(py, pz), inv_h);
}
// a14: is the point inside the viewport of this source view (full-res K, render-res W-1,H-1)?
__device__ __forceinline__ float visible(const Cam& cam, const float* xyz, float inv_w, float inv_h) {
  float u, v, pz;
  ndc_coords(cam, xyz, inv_w, inv_h, u, v, pz);
  return (u >= 0.f && u <= 1.f && v >= 0.f && v <= 1.f && pz > 0.f) ? 1.f : 0.f;
}

// a12 helper: softplus with torch defaults (beta=1, threshold=20)
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }

}  // namespace bmv
