// Weight gradient of the convolutions of the cost regularisers and of FeatureNet (lib/networks/enerf/cost_reg_net.py:
// 4-86, feature_net.py:4-36; training leg of SURVEY.md section 8 rows f1 / f2) on the fp32 matrix cores: KD x K x K taps
// with (KD, K) in {(3,3), (1,3), (1,1), (1,5)}, stride 1 or 2, a batch of B items.
//
//   G[s, b, kd, kh, kw] = sum_n sum_p small[n, s, p] * big[n, b, stride * p + (kd, kh, kw)]
//
// `small` (B, Cs, Ds, Hs, Ws) is the output-side tensor (dY of a convolution, X of a transposed convolution) and `big`
// (B, Cb, Db, Hb, Wb) the zero-padded input-side tensor (X padded by K / 2 voxels, resp. dY padded), so the taps are
// shifted views of `big` and no tap needs a boundary test.  The voxel index is the MFMA k dimension of
// v_mfma_f32_16x16x4_f32 (the layers that matter have 1-16 channels on either side; a 32 x 32 tile would idle 3/4 of
// the matrix core): lane (m, kq) of a wave holds channel m of a 16-channel block for 16 consecutive voxels of one
// (d, h) row -- the four lane groups kq take four consecutive 16-voxel groups of the flattened (row, group) list, so
// rows whose length is not a multiple of 64 waste nothing -- and 16 MFMA steps reduce 64 voxels into a 16 x 16 block
// of one tap.  A wave keeps all 27 tap blocks (108 accumulator registers), fetches tap t + 1 while the matrix core
// works on tap t and walks its items grid-stride; the four waves of a workgroup add their blocks up in LDS and write one
// partial; a second kernel sums the partials in a fixed order (deterministic, no atomics).
// MIOpen's fp32 3-D backward-weights solvers take 40-300 ms for these layers on gfx950 (scripts/probe_conv3d.py) and
// the slice-GEMM formulation this replaces needed 27 strided copies + a tall-skinny rocBLAS GEMM per layer.
#include "mlp.hpp"

namespace bmv {

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct WgradArgs {
  const float* big;
  const float* small;
  float* partials;
  int B, Cb, Db, Hb, Wb, Cs, Ds, Hs, Ws;
  int gpr, ngroups, nitems, cb_blocks;     // 16-voxel groups per row, in total (all batch items); items of 4 groups
};

template <int STRIDE, int KD, int K>
__global__ void __launch_bounds__(256, 2) conv3d_wgrad_kernel(WgradArgs a) {
  constexpr int NTAP = KD * K * K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int cs0 = (blockIdx.y / a.cb_blocks) * 16, cb0 = (blockIdx.y % a.cb_blocks) * 16;
  const bool s_ok = cs0 + m < a.Cs, b_ok = cb0 + m < a.Cb;
  const size_t s_plane = (size_t)a.Ds * a.Hs * a.Ws, b_plane = (size_t)a.Db * a.Hb * a.Wb;
  const float* __restrict__ sm0 = a.small + (size_t)(s_ok ? cs0 + m : 0) * s_plane;
  const float* __restrict__ bg0 = a.big + (size_t)(b_ok ? cb0 + m : 0) * b_plane;
  f32x4 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int item = blockIdx.x * 4 + wave; item < a.nitems; item += gridDim.x * 4) {
    const int g = item * 4 + kq;                          // this lane group's 16-voxel group
    const bool g_ok = g < a.ngroups;
    const int gc = g_ok ? g : 0;
    const int row = gc / a.gpr, w0 = (gc - row * a.gpr) * 16;
    const int nd = row / a.Hs, h = row - nd * a.Hs;       // (batch item, plane) and row
    const int n = nd / a.Ds, d = nd - n * a.Ds;
    const float* __restrict__ sm = sm0 + (size_t)n * a.Cs * s_plane;
    const float* __restrict__ bg = bg0 + (size_t)n * a.Cb * b_plane;
    const int nv = g_ok ? min(a.Ws - w0, 16) : 0;         // voxels of the group that exist
    float av[16];
    {
      const float* p = sm + ((size_t)d * a.Hs + h) * a.Ws + w0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (s_ok && 4 * q + 4 <= nv) {
          float4u t = *reinterpret_cast<const float4u*>(p + 4 * q);
          av[4 * q] = t.x, av[4 * q + 1] = t.y, av[4 * q + 2] = t.z, av[4 * q + 3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) av[4 * q + e] = (s_ok && 4 * q + e < nv) ? p[4 * q + e] : 0.f;
        }
      }
    }
    const float* brow = bg + ((size_t)(STRIDE * d) * a.Hb + STRIDE * h) * a.Wb + STRIDE * w0;
    auto fetch = [&](int tap, float (&bv)[16]) {
      const int kd = tap / (K * K), kh = (tap / K) % K, kw = tap % K;
      const float* p = brow + ((size_t)kd * a.Hb + kh) * a.Wb + kw;
      if (STRIDE == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (b_ok && 4 * q + 4 <= nv) {
            float4u v = *reinterpret_cast<const float4u*>(p + 4 * q);
            bv[4 * q] = v.x, bv[4 * q + 1] = v.y, bv[4 * q + 2] = v.z, bv[4 * q + 3] = v.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[4 * q + e] = (b_ok && 4 * q + e < nv) ? p[4 * q + e] : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {         // 4 consecutive floats = voxels 2q, 2q + 1 (elements 0 and 2)
          if (b_ok && 2 * q + 2 <= nv) {
            float4u v = *reinterpret_cast<const float4u*>(p + 4 * q);   // the padded row keeps p + 4q + 3 in bounds
            bv[2 * q] = v.x, bv[2 * q + 1] = v.z;
          } else {
            bv[2 * q] = (b_ok && 2 * q < nv) ? p[4 * q] : 0.f;
            bv[2 * q + 1] = 0.f;
          }
        }
      }
    };
    float bv[2][16];
    fetch(0, bv[0]);
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if (t + 1 < NTAP) fetch(t + 1, bv[(t + 1) & 1]);
      BMV_FENCE();
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[t & 1][j], acc[t], 0, 0, 0);
      BMV_FENCE();
    }
  }
  // accumulator register r, lane (n, kq): D[m = 4 kq + r][n].  The four waves add their blocks up in LDS one after
  // the other (fixed order), then the workgroup writes one partial.
  __shared__ float red[NTAP * 256];
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = t * 256 + (4 * kq + r) * 16 + m;
          red[o] = w == 0 ? acc[t][r] : red[o] + acc[t][r];
        }
    }
    __syncthreads();
  }
  float* __restrict__ part = a.partials + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (NTAP * 256);
  for (int i = threadIdx.x; i < NTAP * 256 / 4; i += 256)
    reinterpret_cast<float4*>(part)[i] = reinterpret_cast<const float4*>(red)[i];
}

// G (Cs, Cb, ntap): one wave per element, the workgroups' partials summed in a fixed order.
__global__ void conv3d_wgrad_finish_kernel(const float* __restrict__ partials, int nparts, int nblk, int cb_blocks,
                                           int Cs, int Cb, int ntap, float* __restrict__ G) {
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, sub = threadIdx.x & 63;
  if (idx >= Cs * Cb * ntap) return;
  const int tap = idx % ntap, b = (idx / ntap) % Cb, s = idx / (ntap * Cb);
  const int y = (s / 16) * cb_blocks + b / 16;
  const size_t off = (size_t)y * (ntap * 256) + tap * 256 + (s % 16) * 16 + (b % 16);
  float acc = 0.f;
#pragma unroll 8
  for (int w = sub; w < nparts; w += 64) acc += partials[(size_t)w * nblk * (ntap * 256) + off];
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) acc += __shfl_xor(acc, k, 64);
  if (sub == 0) G[idx] = acc;
}

static int wgrad_grid(int nitems, int nblk) {
  int g = 512 / nblk;                    // 2 workgroups per CU in total
  if (g < 1) g = 1;
  const int need = (nitems + 3) / 4;
  return need < g ? need : g;
}

}  // namespace bmv

using namespace bmv;

extern "C" {

static bool wgrad_taps_ok(int kd, int k, int stride) {
  return (kd == 3 && k == 3) || (kd == 1 && (k == 1 || k == 3) && stride == 1) || (kd == 1 && k == 5 && stride == 2);
}

long bmv_conv_wgrad_workspace(int B, int Cs, int Cb, int Ds, int Hs, int Ws, int kd, int k) {
  if (B <= 0 || Cs <= 0 || Cb <= 0 || Ds <= 0 || Hs <= 0 || Ws <= 0 || kd <= 0 || k <= 0) {
    set_error("bmv_conv_wgrad_workspace: bad shape");
    return BMV_ERR_INVALID;
  }
  const int nblk = ((Cs + 15) / 16) * ((Cb + 15) / 16);
  const long ngroups = (long)B * Ds * Hs * ((Ws + 15) / 16);
  return (long)wgrad_grid((int)((ngroups + 3) / 4), nblk) * nblk * kd * k * k * 256;
}

int bmv_conv_wgrad(const float* big, const float* small, int B, int Cb, int Db, int Hb, int Wb, int Cs, int Ds, int Hs,
                   int Ws, int kd, int k, int stride, float* workspace, float* G, bmv_stream_t stream) {
  BMV_REQUIRE(big && small && workspace && G, "bmv_conv_wgrad: null pointer");
  BMV_REQUIRE(stride == 1 || stride == 2, "bmv_conv_wgrad: stride=%d unsupported (1 or 2)", stride);
  BMV_REQUIRE(wgrad_taps_ok(kd, k, stride), "bmv_conv_wgrad: %dx%dx%d taps at stride %d are not built", kd, k, k, stride);
  BMV_REQUIRE(B > 0 && Cs > 0 && Cb > 0 && Ds > 0 && Hs > 0 && Ws > 0, "bmv_conv_wgrad: bad shape");
  BMV_REQUIRE((long)B * Ds * Hs * ((Ws + 15) / 16) < (1l << 30), "bmv_conv_wgrad: too many voxel groups");
  // every tap of every voxel must lie inside `big` (the caller pads); stride 2 reads one float past the last tap
  BMV_REQUIRE(Db >= stride * (Ds - 1) + kd && Hb >= stride * (Hs - 1) + k && Wb >= stride * (Ws - 1) + k + (stride == 2 ? 1 : 0),
              "bmv_conv_wgrad: big (%d,%d,%d) too small for small (%d,%d,%d), %dx%dx%d taps at stride %d", Db, Hb, Wb, Ds, Hs,
              Ws, kd, k, k, stride);
  WgradArgs a;
  a.big = big, a.small = small, a.partials = workspace;
  a.B = B, a.Cb = Cb, a.Db = Db, a.Hb = Hb, a.Wb = Wb, a.Cs = Cs, a.Ds = Ds, a.Hs = Hs, a.Ws = Ws;
  a.gpr = (Ws + 15) / 16, a.ngroups = B * Ds * Hs * a.gpr, a.nitems = (a.ngroups + 3) / 4, a.cb_blocks = (Cb + 15) / 16;
  const int nblk = ((Cs + 15) / 16) * a.cb_blocks;
  const int gx = wgrad_grid(a.nitems, nblk);
  const dim3 grid(gx, nblk), block(256);
  hipStream_t st = as_stream(stream);
  if (kd == 3 && stride == 1) hipLaunchKernelGGL((conv3d_wgrad_kernel<1, 3, 3>), grid, block, 0, st, a);
  else if (kd == 3) hipLaunchKernelGGL((conv3d_wgrad_kernel<2, 3, 3>), grid, block, 0, st, a);
  else if (k == 3) hipLaunchKernelGGL((conv3d_wgrad_kernel<1, 1, 3>), grid, block, 0, st, a);
  else if (k == 1) hipLaunchKernelGGL((conv3d_wgrad_kernel<1, 1, 1>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((conv3d_wgrad_kernel<2, 1, 5>), grid, block, 0, st, a);
  const int ntap = kd * k * k;
  hipLaunchKernelGGL(conv3d_wgrad_finish_kernel, dim3(cdiv((long)Cs * Cb * ntap * 64, 256)), dim3(256), 0, st, workspace, gx,
                     nblk, a.cb_blocks, Cs, Cb, ntap, G);
  BMV_LAUNCH_END("bmv_conv_wgrad");
}

}  // extern "C"
