/* libbmv -- MI355X (gfx950) kernels for the BoostMVSNeRFs rendering hot path.
 *
 * C ABI: plain device pointers (fp32, contiguous, row-major, shapes as written),
 * ints for sizes, an opaque HIP stream.  No allocation, no synchronisation and
 * no ownership transfer inside any entry point; every call only enqueues
 * kernels on `stream` (graph-capturable).  Return value: BMV_OK or a negative
 * error code; bmv_last_error() gives the message (thread-local).
 *
 * The reference has no FFI: its hot path is Python calling stock torch ops.
 * Each entry point therefore names the reference *function* it replaces
 * (file:line relative to the reference checkout); the Python binding a
 * maintainer would add is in INTEGRATION.md and boostmvsnerfs_amd/_lib.py.
 *
 * Conventions: B batch, S source views of one cost volume (3), C channels,
 * D depth planes, (h, w) volume resolution, (Hs, Ws) source-feature resolution,
 * N rays, Ns samples per ray, P = N*Ns points, K fused cost volumes.
 */
#ifndef BMV_H_
#define BMV_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bmv_stream_t; /* hipStream_t */

enum { BMV_OK = 0, BMV_ERR_INVALID = -1, BMV_ERR_LAUNCH = -2, BMV_ERR_UNSUPPORTED = -3 };

int bmv_version(void);
const char* bmv_last_error(void);

/* ---- a1  get_proj_mats                    lib/networks/enerf/utils.py:35-55
 * proj[b,s] = (K_s' E_s[:3]) inverse([K_t' E_t[:3]; 0 0 0 1]), K' rows 0-1 scaled.
 * src_exts (B,S,4,4), src_ixts (B,S,3,3), tar_ext (B,4,4), tar_ixt (B,3,3) -> proj (B,S,3,4) */
int bmv_proj_mats(const float* src_exts, const float* src_ixts, const float* tar_ext, const float* tar_ixt,
                  float src_scale, float tar_scale, int B, int S, float* proj, bmv_stream_t stream);

/* ---- a2  get_depth_values                 lib/networks/enerf/utils.py:98-153
 * level 0 (:103-111): D planes between near_far[b,0..1], uniform in disparity if depth_inv.
 * near_far (B,2) -> depth_values (B,D,h,w), near_far_out (B,2,h,w) (= 1/clamp(.,1e-6) if depth_inv) */
int bmv_depth_values_uniform(const float* near_far, int B, int D, int h, int w, int depth_inv,
                             float* depth_values, float* near_far_out, bmv_stream_t stream);
/* a1 + a2 of a whole frame in one launch (inference): the projection matrices of all L <= 4 cascade levels
 * (proj (L,B,S,3,4), level l with src_scales[l] / tar_scales[l] -- host arrays) and level 0's uniform hypotheses
 * (depth_values (B,D,h,w), near_far_out (B,2,h,w)): bmv_proj_mats x L + bmv_depth_values_uniform, same arithmetic. */
int bmv_frame_setup(const float* src_exts, const float* src_ixts, const float* tar_ext, const float* tar_ixt,
                    const float* src_scales, const float* tar_scales, int L, int B, int S, float* proj,
                    const float* near_far, int D, int h, int w, int depth_inv, float* depth_values,
                    float* near_far_out, bmv_stream_t stream);

/* cascade level (:112-153, prev level in disparity, this level in depth): bilinear
 * (align_corners) upsample of depth/std (B,h0,w0) and near_far (B,2,h0,w0) to (h,w);
 * [depth+std, depth-std] clamped to near_far, inverted, D planes uniform in depth. */
int bmv_depth_values_cascade(const float* depth, const float* std, const float* near_far, int B, int h0, int w0,
                             int h, int w, int D, float* depth_values, float* near_far_out, bmv_stream_t stream);

/* ---- a3  homo_warp                        lib/networks/enerf/utils.py:57-95
 * src_feat (B,C,Hs,Ws), proj (B,3,4), depth_values (B,D,h,w) -> warped (B,C,D,h,w),
 * grid (B,D,h,w,2) (may be NULL).  Bilinear, zeros padding, align_corners=True. */
int bmv_homo_warp_fwd(const float* src_feat, const float* proj, const float* depth_values, int B, int C, int Hs,
                      int Ws, int D, int h, int w, float* warped, float* grid, bmv_stream_t stream);

/* ---- a3+a4 build_feature_volume           lib/networks/enerf/utils.py:324-351
 * Fused plane sweep: feats, proj (B,S,3,4), depth_values (B,D,h,w)
 * -> variance (B,C,D,h,w) = sum_s(x^2)/S - (sum_s(x)/S)^2, never materialising the
 * S warped volumes.  feat_layout: 0 = feats is (B,S,C,Hs,Ws) as the reference holds it,
 * 1 = channel-last (B,S,Hs,Ws,C) (bmv_nchw_to_nhwc converts).  algo: 0 = best kernel for
 * the layout, 1 = reference-layout direct gather (feat_layout 0); feat_layout 1: 0 / 4 = LDS-staged exact windows
 * (sweep_win.hip; C in {16, 32}, 2..4 views: the training forward), 40 + i = its tuning variant i.  The inference
 * networks run bmv_sweep_variance_quad_fwd (below) instead.  (The round-1..3 experiments -- channel-last gather, corner
 * windows, split geometry, ring of windows, zero-padded windows -- were removed in round 4: all superseded.) */
int bmv_sweep_variance_fwd(const float* feats, const float* proj, const float* depth_values, int B, int S, int C,
                           int Hs, int Ws, int D, int h, int w, float* variance, int feat_layout, int algo,
                           bmv_stream_t stream);
/* Same sweep over S views PICKED from a channel-last tensor of all views: feats_all (B,n_all,Hs,Ws,C), view s of batch
 * item b is feats_all[b, view_ids[b*S + s]] (boost_enerf/network.py:178-192 builds every cost volume from a
 * triplet of the N source views; no gathered copy of the feature maps is made).  proj (B,S,3,4) is per picked view. */
int bmv_sweep_variance_views_fwd(const float* feats_all, const int* view_ids, int n_all, const float* proj,
                                 const float* depth_values, int B, int S, int C, int Hs, int Ws, int D, int h, int w,
                                 float* variance, bmv_stream_t stream);
/* ---- a3+a4 on QUAD-PLANAR features with union windows in LDS (round 4, csrc/sweep_quad.hip): the inference sweep.
 * Same function as bmv_sweep_variance_fwd (lib/networks/enerf/utils.py:57-95, :324-351) on another source layout:
 * feats_quad (B, n_views, C/4, Hs, Ws, 4) -- a record of 4 channels = 16 bytes (written directly by the convolution
 * engine, out_layout 3 of bmv_conv_fwd; bmv_to_quad_planar converts planar or channel-last maps)
 * -> variance (B,C,D,h,w) planar, as bmv_sweep_variance_fwd writes it.
 * dv_plane_uniform: 0 = depth_values (B,D,h,w); 1 = (B,D) one hypothesis per plane (cascade level 0); 2 = (B,D,h,w)
 * whose planes are constant ([b,d,0,0] is read).  view_ids (B,S) int32 or NULL: views picked from the n_all views of
 * feats_quad (n_all ignored when NULL: feats_quad holds exactly S views per item).  variant: -1 = by the source / volume
 * scale; 0.. = tuning table of sweep_quad.hip.  flags: 0; tuning ablations: 1 no fill, 2 no blend, 4 no store; bits 16-23:
 * LDS budget in 1-KB pieces for the S windows together (tests: small budgets force the global-gather fallback); bit 24
 * (round 5): `variance` is written as QUAD RECORDS (B, C/4, D, h, w, 4) -- one 16-byte store per voxel and channel quad
 * -- for bmv_conv_c4_fwd's input mode 4 (default variants only; BMV_ERR_UNSUPPORTED otherwise).
 * S in 2..4, C % 4 == 0, C <= 64. */
int bmv_sweep_variance_quad_fwd(const float* feats_quad, const int* view_ids, int n_all, const float* proj,
                                const float* depth_values, int dv_plane_uniform, int B, int S, int C, int Hs, int Ws,
                                int D, int h, int w, float* variance, int variant, int flags, bmv_stream_t stream);
/* in (n,C,H,W) planar (channels_last 0) or (n,H,W,C) (channels_last 1) -> out (n,C/4,H,W,4), C % 4 == 0 */
int bmv_to_quad_planar(const float* in, int channels_last, int n, int C, int H, int W, float* out, bmv_stream_t stream);
/* (n,C,H,W) -> (n,H,W,C), C % 4 == 0: puts the 2-D features into the sweep's channel-last layout */
int bmv_nchw_to_nhwc(const float* src, int n, int C, int H, int W, float* dst, bmv_stream_t stream);

/* ---- a5  depth_regression                 lib/networks/enerf/utils.py:722-731
 * depth_prob, depth_values (B,D,h,w) -> depth, std (B,h,w); softmax over D,
 * values inverted (1/clamp_min(v,1e-6)) first when depth_inv.
 * Honours bmv_defer_pointer for `depth` / `std` with an "ALSO" meaning: the maps are written to the given buffers (the
 * later kernels of a captured frame read those) and, in addition, to the tensors table[slot] points at when the kernel
 * runs (the frame's caller-visible depth_mvs / std outputs: no copy node at the end of the frame). */
int bmv_depth_regress_fwd(const float* depth_prob, const float* depth_values, int B, int D, int h, int w,
                          int depth_inv, float* depth, float* std, bmv_stream_t stream);

/* ---- a6  build_rays                       lib/networks/enerf/utils.py:392-422
 * rays (B,N,8) [o,d,x,y]; depth,std (B,hv,wv), near_far (B,2,hv,wv) upsampled
 * (bilinear, align_corners) to (Hr,Wr) -> rays_out (B,N,12). */
int bmv_build_rays(const float* rays, const float* depth, const float* std, const float* near_far, int B, int N,
                   int hv, int wv, int Hr, int Wr, int depth_inv, float* rays_out, bmv_stream_t stream);

/* ---- a7  sample_along_depth               lib/networks/enerf/utils.py:424-443
 * rays (B,N,12) -> world_xyz (B,N,Ns,3), uvd (B,N,Ns,3) [x,y,dnorm], z_vals (B,N,Ns) */
int bmv_sample_along_depth(const float* rays, int B, int N, int Ns, int depth_inv, float* world_xyz, float* uvd,
                           float* z_vals, bmv_stream_t stream);

/* ---- a8  unpreprocess                     lib/networks/enerf/utils.py:669-676
 * src (n,3,H,W) in [-1,1] -> out (n,3,Ho,Wo) = bilinear(align_corners)(0.5*src+0.5) */
int bmv_unpreprocess(const float* src, int n, int C, int H, int W, int Ho, int Wo, float* out, bmv_stream_t stream);

/* ---- a9  get_vox_feat                     lib/networks/enerf/utils.py:458-460
 * uvd01 (B,P,3) in [0,1]^3, volume (B,C,D,h,w) -> out (B,P,C); trilinear, zeros. */
int bmv_vox_feat(const float* uvd01, const float* volume, int B, int P, int C, int D, int h, int w, float* out,
                 bmv_stream_t stream);

/* ---- a10 get_img_feat                     lib/networks/enerf/utils.py:753-786
 * xyz (B,P,3); img_feat_rgb (B,S,C,H,W) -> out (B,P,S,C+4) (bilinear border +
 * direction feature).  src_ixts rows 0-1 are scaled by render_scale in the kernel. */
int bmv_img_feat(const float* xyz, const float* img_feat_rgb, const float* src_exts, const float* src_ixts,
                 const float* tar_ext, float render_scale, int B, int P, int S, int C, int H, int W, float* out,
                 bmv_stream_t stream);

/* ---- a11 NeRF.forward / Agg.forward       lib/networks/enerf/nerf.py:29-43, 74-89
 * Parameter pointers in the reference's state-dict order (weight (out,in), bias). */
typedef struct {
  const float *view_fc_w, *view_fc_b;     /* agg.view_fc.0    (F,4)        F = feat_ch+3 */
  const float *global_fc_w, *global_fc_b; /* agg.global_fc.0  (32,3F)                    */
  const float *agg_w_w, *agg_w_b;         /* agg.agg_w_fc.0   (1,32)                     */
  const float *fc_w, *fc_b;               /* agg.fc.0         (16,32)                    */
  const float *lr0_w, *lr0_b;             /* lr0.0            (64,24)                    */
  const float *sigma_w, *sigma_b;         /* sigma.0          (1,64)                     */
  const float *color0_w, *color0_b;       /* color.0          (64,88+F+4)                */
  const float *color2_w, *color2_b;       /* color.2          (1,64)                     */
} bmv_nerf_params;

/* floats needed for the packed (MFMA-ordered) weight blob of a given feat_ch (8 or 32) */
int bmv_nerf_blob_size(int feat_ch);
int bmv_nerf_pack_weights(const bmv_nerf_params* params, int feat_ch, float* blob, bmv_stream_t stream);
/* vox_feat (B*P,8), img_feat_rgb_dir (B*P,S,F+4) -> out (B*P,4) = [rgb, sigma]; S source views in {2, 3, 4} (the
 * reference's Agg / NeRF take any S: var / mean / softmax over the view axis, nerf.py:29-43, 74-89; ENeRF pre-trains
 * with 2, 3 and 4 views, configs/exps/pretrain/enerf/dtu_pretrain.yaml:22-23).  The blob does not depend on S. */
int bmv_nerf_mlp_fwd(const float* vox_feat, const float* img_feat_rgb_dir, const float* blob, int feat_ch, int S,
                     long npts, float* out, bmv_stream_t stream);

/* ---- a12 raw2outputs                      lib/networks/enerf/utils.py:605-637
 * raw (B*N,Ns,4), z_vals (B*N,Ns) -> rgb (B*N,3), depth (B*N), weights (B*N,Ns) (softmaxed) */
int bmv_composite_fwd(const float* raw, const float* z_vals, long nrays, int Ns, int white_bkgd, float* rgb,
                      float* depth, float* weights, bmv_stream_t stream);

/* ---- a14 get_ndc_coords / mask_viewport   lib/networks/enerf/utils.py:490-520
 * xyz (B,P,3), src_exts (B,V,4,4), src_ixts (B,V,3,3), inv_scale (W-1,H-1) -> mask (B,P) */
int bmv_mask_viewport(const float* xyz, const float* src_exts, const float* src_ixts, float inv_w, float inv_h,
                      int B, int P, int V, float* mask, bmv_stream_t stream);
/* get_ndc_coords alone (utils.py:490-508): xyz (B,P,3), ONE source view per item src_ext (B,4,4), src_ixt (B,3,3)
 * -> ndc (B,P,3) = (x/z/(W-1), y/z/(H-1), z) of K (R x + T) */
int bmv_ndc_coords(const float* xyz, const float* src_ext, const float* src_ixt, float inv_w, float inv_h, int B,
                   int P, float* ndc, bmv_stream_t stream);

/* ---- a16 raw2outputs_blend + mask normalisation
 *          lib/networks/enerf/utils.py:639-667, lib/networks/boost_enerf/network.py:163-170
 * raws (B,K,N,Ns,4), masks (B,K,N,Ns) (raw visibility fractions if normalise!=0),
 * z_vals (B,K,N,Ns) -> rgb (B,N,3), depth (B,N), weights (B,N,Ns) */
int bmv_blend_fwd(const float* raws, const float* masks, const float* z_vals, int B, int K, int N, int Ns,
                  int normalise, float* rgb, float* depth, float* weights, bmv_stream_t stream);

/* ---- fused a6..a12 (+a14): Network.render_rays
 *          lib/networks/enerf/network.py:24-43, lib/networks/boost_enerf/network.py:123-149
 * One kernel from rays to composited pixels: per-ray bounds, samples, trilinear
 * volume lookup, per-view image lookup + direction feature, MFMA MLP, composite.
 * mode 0: rgb (B,N,3), depth (B,N), weights (B,N,Ns)
 * mode 1: raw (B,N,Ns,4), z_vals (B,N,Ns), mask (B,N,Ns)   (boost path, no composite) */
typedef struct {
  const float* rays;      /* (B,N,8)                                   */
  const float* depth;     /* (B,hv,wv) regressed depth (or disparity)  */
  const float* std;       /* (B,hv,wv)                                 */
  const float* near_far;  /* (B,2,hv,wv) volume bounds                 */
  const float* volume;    /* (B,8,Dv,hv,wv) regularised feature volume */
  const float* im_feat;   /* (B,S,feat_ch,Hr,Wr)                       */
  const float* rgb_src;   /* (B,S,3,Hr,Wr); raw src_inps if rgb_affine */
  const float* src_exts;  /* (B,S,4,4)                                 */
  const float* src_ixts;  /* (B,S,3,3) full-resolution intrinsics      */
  const float* tar_ext;   /* (B,4,4)                                   */
  const float* blob;      /* packed MLP weights                        */
  int B, N, S, feat_ch, Ns, depth_inv;
  int hv, wv, Dv, Hr, Wr;
  float render_scale;
  int rgb_affine;         /* 1: rgb = 0.5*sample + 0.5 (render_scale == 1) */
  int white_bkgd;
  int mode;
  int ray_begin, ray_end; /* render rays [ray_begin, ray_end) of every batch item */
  float* out0;            /* rgb  | raw    */
  float* out1;            /* depth| z_vals */
  float* out2;            /* weights | mask */
  const int* view_ids;    /* NULL, or (B,S): im_feat / rgb_src are (B,n_all,...) and view i is view_ids[b*S + i] */
  int n_all;
  const float* im_packed; /* NULL, or (B,S|n_all,Hr,Wr,12) lookup records [ch 0 2 4 6 | ch 1 3 5 7 | r b | g 0] written by
                             bmv_fpn_smooth_fwd (feat_ch = 8): a bilinear tap of a view is then 2 loads per lane half
                             instead of 6; im_feat / rgb_src are not read */
  int vol_packed;         /* 1: `volume` is (B,Dv,hv,wv,8) records [ch 0 2 4 6 | ch 1 3 5 7] (bmv_conv_heads_fwd); needs
                             im_packed */
} bmv_render_args;
int bmv_render_rays_fwd(const bmv_render_args* args, bmv_stream_t stream);

/* ======================= backward (fine-tuning, SURVEY.md 8 "Backward contract") =======================
 * Adjoint of the forward entry point of the same name; gradient buffers that receive scatter-adds
 * (d_volume, d_img, d_depth, d_std, d_feats, d_depth_values) must be zero-initialised by the caller. */
int bmv_composite_bwd(const float* raw, const float* z_vals, const float* d_rgb, const float* d_depth /*nullable*/,
                      long nrays, int Ns, float* d_raw, bmv_stream_t stream);
int bmv_blend_bwd(const float* raws, const float* masks /*normalised*/, const float* d_rgb, int B, int K, int N, int Ns,
                  float* d_raws, bmv_stream_t stream);
/* ray_w / Ns: the optional layout hint of bmv_img_feat_bwd (0 = none) */
int bmv_vox_feat_bwd(const float* uvd01, const float* volume, const float* d_out, int B, int P, int C, int D, int h,
                     int w, int ray_w, int Ns, float* d_volume, float* d_d01, bmv_stream_t stream);
/* c_grad: only the first c_grad of the C channels receive d_img (the trailing colour channels of [features, rgb] are
 * data); ray_w / Ns: optional layout hint (0 = none) -- the P samples are Ns per ray, rays row-major over an image
 * ray_w wide -- that lets a workgroup take a compact tile of rays and pre-reduce its scatter-adds in LDS; the result
 * does not depend on it. */
int bmv_img_feat_bwd(const float* xyz, const float* img_feat_rgb, const float* src_exts, const float* src_ixts,
                     const float* tar_ext, float render_scale, const float* d_out, int B, int P, int S, int C,
                     int c_grad, int H, int W, int ray_w, int Ns, float* d_img, float* d_xyz, bmv_stream_t stream);
int bmv_sample_along_depth_bwd(const float* rays, const float* d_xyz, const float* d_dn, int B, int N, int Ns,
                               int depth_inv, float* d_near_far /*(B,N,2)*/, bmv_stream_t stream);
int bmv_build_rays_bwd(const float* rays, const float* depth, const float* std, const float* near_far,
                       const float* d_near_far, int B, int N, int hv, int wv, int Hr, int Wr, int depth_inv,
                       float* d_depth, float* d_std, bmv_stream_t stream);
int bmv_depth_regress_bwd(const float* depth_prob, const float* depth_values, const float* d_depth,
                          const float* d_std, int B, int D, int h, int w, int depth_inv, float* d_prob,
                          float* d_values, bmv_stream_t stream);
int bmv_depth_values_cascade_bwd(const float* depth, const float* std, const float* near_far,
                                 const float* d_depth_values, int B, int h0, int w0, int h, int w, int D,
                                 float* d_depth, float* d_std, bmv_stream_t stream);
/* feats in the reference layout (B,S,C,Hs,Ws); d_depth_values may be NULL (level 0: hypotheses are constants) */
int bmv_sweep_variance_bwd(const float* feats, const float* proj, const float* depth_values, const float* d_variance,
                           int B, int S, int C, int Hs, int Ws, int D, int h, int w, float* d_feats,
                           float* d_depth_values, bmv_stream_t stream);
/* The same gradient on CHANNEL-LAST tensors (round 3): feats_cl (B,S,Hs,Ws,C) in, d_feats_cl (B,S,Hs,Ws,C) accumulated
 * (zero it first) with the channel on the lane -- >= 16 consecutive channels of a tap per atomic instruction run at
 * 330 G lane-atomics/s against 116 G/s for a planar gradient and 203 G/s through an LDS window
 * (scripts/ubench/atomic_rate.hip).  S = 3, C in {16, 32}.  d_depth_values (B,D,h,w) is WRITTEN (not accumulated) or
 * NULL. */
int bmv_sweep_variance_bwd_cl(const float* feats_cl, const float* proj, const float* depth_values,
                              const float* d_variance, int B, int S, int C, int Hs, int Ws, int D, int h, int w,
                              float* d_feats_cl, float* d_depth_values, bmv_stream_t stream);

/* ---- Bit-reproducible scatter gradients (round 5; bmv_tuning "BMV_DETERMINISTIC" makes the Python host take these).
 * The entry points above add their contributions with float atomics: a texel's sum depends on the order the hardware
 * serves the adds, so gradients differ in their last bits from run to run (and training-mode batch norm amplifies
 * that).  The *_fixed twins below compute the SAME gradients with order-independent accumulation
 * (csrc/scatter.hpp): one pass finds the largest |contribution| of the launch (an atomic max: order-independent), a
 * power-of-two scale puts it just under 2^38, a second pass of the same kernel adds llrint(v * scale) with 64-bit
 * INTEGER atomics (associative), and a finish kernel writes float(q / scale) to the float gradient buffers -- which
 * then need no zero-initialisation.  Every contribution is rounded once, 38 bits below the largest one of ITS OUTPUT TENSOR
 * (a launch with two scatter outputs of different units -- d_depth + d_std, d_feats + d_depth_values -- keeps a scale per
 * output): at least as accurate as the float form.  Bound: 2^25 contributions of the largest magnitude fit a texel's int64
 * accumulator (2^63 / 2^38); the kernels of this library add at most a few thousand per texel.  Two runs on the same inputs
 * give bit-identical outputs.
 * `workspace`: bmv_fixed_workspace(n) 64-bit words, ZEROED by the caller, n = number of floats in the launch's
 * scatter outputs together (d_volume; d_img; d_depth + d_std; d_feats + d_depth_values (if not NULL); ...).
 * Reference semantics as the float twins (lib/networks/enerf/utils.py:57-95, 324-351, 392-460, 753-786 under
 * loss.backward(), lib/train/trainers/trainer.py:44-63). */
long bmv_fixed_workspace(long n_out);
int bmv_vox_feat_bwd_fixed(const float* uvd01, const float* volume, const float* d_out, int B, int P, int C, int D, int h,
                           int w, int ray_w, int Ns, float* d_volume, float* d_d01, long long* workspace,
                           bmv_stream_t stream);
int bmv_img_feat_bwd_fixed(const float* xyz, const float* img_feat_rgb, const float* src_exts, const float* src_ixts,
                           const float* tar_ext, float render_scale, const float* d_out, int B, int P, int S, int C,
                           int c_grad, int H, int W, int ray_w, int Ns, float* d_img, float* d_xyz, long long* workspace,
                           bmv_stream_t stream);
int bmv_build_rays_bwd_fixed(const float* rays, const float* depth, const float* std, const float* near_far,
                             const float* d_near_far, int B, int N, int hv, int wv, int Hr, int Wr, int depth_inv,
                             float* d_depth, float* d_std, long long* workspace, bmv_stream_t stream);
int bmv_depth_values_cascade_bwd_fixed(const float* depth, const float* std, const float* near_far,
                                       const float* d_depth_values, int B, int h0, int w0, int h, int w, int D,
                                       float* d_depth, float* d_std, long long* workspace, bmv_stream_t stream);
int bmv_sweep_variance_bwd_fixed(const float* feats, const float* proj, const float* depth_values,
                                 const float* d_variance, int B, int S, int C, int Hs, int Ws, int D, int h, int w,
                                 float* d_feats, float* d_depth_values, long long* workspace, bmv_stream_t stream);

/* ---- a11 backward (lib/networks/enerf/nerf.py:29-43, 74-89), three launches on `stream`:
 *   1. the data path: the forward of every 32-sample tile is recomputed and back-propagated on the matrix cores;
 *      d_vox (8, P) and d_img (S, IR, P) come out in [row][sample] layout (IR rows per view: the F channels padded
 *      to an even count, then the 4 direction components), and the pre-activation gradients + layer inputs go to
 *      `workspace` as per-tile matrices;
 *   2. every weight / bias gradient, dW = D_pre ACT^T, with the sample index as the MFMA k dimension;
 *   3. a deterministic reduction of the per-workgroup partials into `grads` (tensors of the reference's parameter
 *      shapes, overwritten).
 * No atomics anywhere (round 5: the two 1-wide heads whose inputs are not parked leave one partial per wave, summed
 * in a fixed order by launch 3): the call is bit-reproducible.
 * S source views in {2, 3, 4} (img_feat_rgb_dir is (P, S, F + 4); nerf.py's Agg / NeRF take any S).
 * workspace: bmv_nerf_bwd_workspace(feat_ch, S, npts) floats (about 80 KB per 32 samples for feat_ch 8, S = 3).
 * npts = 0 is rejected (nothing to write the gradients from).  IR: bmv_nerf_bwd_rows(). */
typedef struct {
  float *view_fc_w, *view_fc_b, *global_fc_w, *global_fc_b, *agg_w_w, *agg_w_b, *fc_w, *fc_b;
  float *lr0_w, *lr0_b, *sigma_w, *sigma_b, *color0_w, *color0_b, *color2_w, *color2_b;
} bmv_nerf_grads;   /* same order and shapes as bmv_nerf_params */
int bmv_nerf_bwd_blob_size(int feat_ch);
int bmv_nerf_bwd_rows(int feat_ch, int S, int* d_img_rows);
long bmv_nerf_bwd_workspace(int feat_ch, int S, long npts);
int bmv_nerf_pack_bwd_weights(const bmv_nerf_params* params, int feat_ch, float* blob, bmv_stream_t stream);
int bmv_nerf_mlp_bwd(const float* vox_feat, const float* img_feat_rgb_dir, const float* d_out, const float* blob_fwd,
                     const float* blob_bwd, int feat_ch, int S, long npts, float* workspace, float* d_vox, float* d_img,
                     const bmv_nerf_grads* grads, bmv_stream_t stream);

/* ---- f1 / f2 (training leg): weight gradient of the convolutions
 *          lib/networks/enerf/cost_reg_net.py:4-86 (Conv3d / ConvTranspose3d, 3x3x3, stride 1 or 2),
 *          lib/networks/enerf/feature_net.py:4-36 (Conv2d: 1x1, 3x3 stride 1; 5x5 stride 2)
 * G (Cs, Cb, kd,k,k): G[s,b,kz,ky,kx] = sum_n sum_p small[n,s,p] * big[n,b, stride*p + (kz,ky,kx)], the voxel index as
 * the MFMA k dimension.  small (B,Cs,Ds,Hs,Ws) is the output-side tensor (dY of a convolution, X of a transposed
 * one), big (B,Cb,Db,Hb,Wb) the input-side tensor ALREADY zero-padded by the caller so that every tap is in range:
 * Db >= stride*(Ds-1)+kd, Hb >= stride*(Hs-1)+k, Wb >= stride*(Ws-1)+k (+1 for stride 2); 2-D: Ds = Db = kd = 1.
 * Conv: dW = G(big = pad(X), small = dY); ConvTranspose3d (stride 2, padding 1, output_padding 1): dW = G(big =
 * pad(dY), small = X, stride 2).  Built: (kd,k,stride) in {(3,3,1|2), (1,3,1), (1,1,1), (1,5,2)}.
 * workspace: bmv_conv_wgrad_workspace() floats. */
long bmv_conv_wgrad_workspace(int B, int Cs, int Cb, int Ds, int Hs, int Ws, int kd, int k);
int bmv_conv_wgrad(const float* big, const float* small, int B, int Cb, int Db, int Hb, int Wb, int Cs, int Ds, int Hs,
                   int Ws, int kd, int k, int stride, float* workspace, float* G, bmv_stream_t stream);

/* ---- f1 / f2 (training leg): batch normalisation in training mode (+ ReLU) of ConvBnReLU / ConvBnReLU3D
 *          lib/networks/enerf/utils.py:10-33 (nn.BatchNorm2d / nn.BatchNorm3d under net.train())
 * x, y, dy, dx: (N, C, S) planar, S = H*W or D*H*W.  Forward: batch statistics (biased variance normalises, the
 * unbiased one updates running_var; momentum as torch: new = (1 - m) old + m batch), y = act(xhat * w + b) with
 * act(v) = v > 0 ? v : act_slope * v (1 = none, 0 = ReLU, 0.01 = InPlaceABN's leaky ReLU, mvsnerf/network.py:699-779);
 * save_mean / save_invstd (C) feed the backward.  Backward: act' is read off y (y > 0 ? 1 : act_slope).
 * workspace: C * bmv_bn_chunks(N, S) * 3 floats (forward), C * bmv_bn_chunks(N, S) * 2 + 2 C (backward).
 * weight / bias / running_* / dweight / dbias may be NULL. */
int bmv_bn_chunks(int N, long S);
int bmv_bn_train_fwd(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                     int N, int C, long S, float eps, float momentum, float act_slope, float* workspace, float* save_mean,
                     float* save_invstd, float* y, bmv_stream_t stream);
int bmv_bn_train_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* save_mean,
                     const float* save_invstd, int N, int C, long S, float act_slope, float* workspace, float* dx,
                     float* dweight, float* dbias, bmv_stream_t stream);

/* ======================= MVSNeRF backbone (lib/networks/mvsnerf) ======================= */

/* ---- a18 Network.get_proj_mats            lib/networks/mvsnerf/network.py:1070-1090
 * View 0 of each triplet is the reference view: P_0 = I, P_i = (K_i/4 E_i) inverse(K_0/4 E_0).
 * src_exts (B,S,4,4), src_ixts (B,S,3,3) -> proj (B,S,3,4) */
int bmv_mvs_proj_mats(const float* src_exts, const float* src_ixts, int B, int S, float* proj, bmv_stream_t stream);

/* ---- F.interpolate(bilinear, align_corners=False) used by a20 (network.py:913)
 * src (n,C,H,W) -> dst (n,C,h,w) */
int bmv_resize_bilinear(const float* src, int n, int C, int H, int W, int h, int w, float* dst, bmv_stream_t stream);

/* ---- a19+a20 homo_warp(pad) + build_volume_costvar_img
 *          lib/networks/mvsnerf/utils.py:580-630, network.py:887-942
 * imgs (B,S,3,h,w) (source images already resized to the feature resolution), feats (B,S,C,h,w),
 * proj (B,S,3,4), depth_values (B,D) -> volume (B, 3*S+C, D, h+2pad, w+2pad):
 *   ch 0-2 reference rgb inside the un-padded window (0 in the border: the reference leaves it
 *   uninitialised), ch 3.. warped source rgb, last C: sum(x^2)c - (sum(x)c)^2 over the zero-padded
 *   reference + warped source features, c = 1 / (1 + number of source views whose grid is inside). */
int bmv_mvs_sweep_fwd(const float* imgs, const float* feats, const float* proj, const float* depth_values, int B,
                      int S, int C, int h, int w, int D, int pad, float* volume, bmv_stream_t stream);
/* The same sweep on CHANNEL-LAST features feats_cl (B,S,h,w,C) (what MVSNeRF's FeatureNet writes on the engine path;
 * bmv_nchw_to_nhwc converts): four lanes share a voxel and read 64 contiguous bytes of a tap's record per load (round 3:
 * 298 -> 134 us at 128 planes).  imgs stay planar (B,S,3,h,w).  Same outputs up to v_rcp_f32 vs IEEE division (1 ulp of
 * the tap coordinates). */
int bmv_mvs_sweep_cl_fwd(const float* imgs, const float* feats_cl, const float* proj, const float* depth_values, int B,
                         int S, int C, int h, int w, int D, int pad, float* volume, bmv_stream_t stream);

/* ---- a25 Renderer_ours parameters          lib/networks/mvsnerf/network.py:153-229
 * D=6, W=128, input_ch=63, input_ch_feat=20, input_ch_views=3 (network.py:803-805) */
typedef struct {
  const float *pts_w[6], *pts_b[6];     /* pts_linears.{0..5}: (128,63) (128,128)x4 (128,191) */
  const float *bias_w, *bias_b;         /* pts_bias        (128,20)  */
  const float *views_w, *views_b;       /* views_linears.0 (64,131)  */
  const float *feature_w, *feature_b;   /* feature_linear  (128,128) */
  const float *alpha_w, *alpha_b;       /* alpha_linear    (1,128)   */
  const float *rgb_w, *rgb_b;           /* rgb_linear      (3,64)    */
} bmv_mvs_mlp_params;
int bmv_mvs_mlp_blob_size(void);
int bmv_mvs_mlp_pack_weights(const bmv_mvs_mlp_params* params, float* blob, bmv_stream_t stream);
/* x (npts, 86) = [embed(ndc) 63 | volume+colour feature 20 | view direction 3] -> out (npts,4) = [sigmoid rgb, relu alpha] */
int bmv_mvs_mlp_fwd(const float* x, const float* blob, long npts, float* out, bmv_stream_t stream);

/* ---- a25 under autograd: Renderer_ours.forward + its backward          lib/networks/mvsnerf/network.py:201-229
 * (replaces torch autograd through 11 nn.Linear; trained via configs/exps/finetune/mvsnerf).  Activations of the
 * forward are kept for the backward as per-tile row matrices act[tile][row][32 points] (csrc/mvs_mlp_train.hip):
 *   act      bmv_mvs_mlp_train_act_floats(npts) floats, written by _fwd, consumed AND overwritten by _bwd (one backward
 *            per forward);
 *   scratch  bmv_mvs_mlp_train_scratch_floats() floats (transposed weights of the step + weight-gradient partials);
 *   params   the parameter tensors as stored by torch (row-major (out, in)); `grads`: tensors of the same shapes that
 *            receive dL/dparam (written, not accumulated); dx (npts,86) = dL/dx.
 * Deterministic (no atomics: partial weight gradients are summed in a fixed order). */
long bmv_mvs_mlp_train_act_floats(long npts);
long bmv_mvs_mlp_train_scratch_floats(void);
int bmv_mvs_mlp_train_fwd(const float* x, const bmv_mvs_mlp_params* params, long npts, float* act, float* scratch,
                          float* out, bmv_stream_t stream);
int bmv_mvs_mlp_train_bwd(const bmv_mvs_mlp_params* params, float* act, float* scratch, const float* out,
                          const float* d_out, long npts, float* dx, const bmv_mvs_mlp_params* grads,
                          bmv_stream_t stream);

/* ---- a21..a25 fused: ray_marcher, get_ndc_coordinate, gen_dir_feature, gen_pts_feats, Embedder, MLP
 *          lib/networks/mvsnerf/network.py:945-1042, utils.py:112-146,300-383, renderer.py:111-137
 * rays (N,8): near/far are columns 6 and 7 verbatim.  volume (8,D,hp,wp) = regularised cost volume of the
 * reference view's padded frustum.  src_inps (S,3,H,W) raw images in [-1,1].  near_far: the two
 * plane-sweep bounds (device scalars, as the reference keeps them in batch['near_far']).
 * Outputs: raw (N,Ns,4), z_vals (N,Ns), mask (N,Ns) (viewport visibility fraction; NULL to skip),
 * inputs86 (N,Ns,86) (the MLP input, NULL to skip; test / API-parity hook). */
typedef struct {
  const float* rays;      /* (N,8)            */
  const float* volume;    /* (8,D,hp,wp)      */
  const float* src_inps;  /* (S,3,H,W)        */
  const float* src_exts;  /* (S,4,4)          */
  const float* src_ixts;  /* (S,3,3)          */
  const float* near_far;  /* (2,)             */
  const float* blob;      /* packed MLP weights (may be NULL when only inputs86 is wanted) */
  int N, Ns, S, D, hp, wp, H, W, pad;
  int ray_begin, ray_end;
  float* raw;
  float* z_vals;
  float* mask;
  float* inputs86;
} bmv_mvs_render_args;
int bmv_mvs_render_fwd(const bmv_mvs_render_args* args, bmv_stream_t stream);

/* ---- MVSNeRF backward (training): gradient reaches the network through the masked-variance channels of the padded
 * sweep (a19 + a20 -> the source FEATURES; d_volume is the gradient of bmv_mvs_sweep_fwd's whole output, only its last
 * C channels are read) and through the trilinear lookup of the regularised volume (a22 + a23: d_feat (N,Ns,8) = the
 * gradient of columns 63..70 of the MLP input; rays (N,8), the reference view's camera, near_far as the forward).
 * d_feats / d_volume receive scatter-adds: zero them first. */
int bmv_mvs_sweep_bwd(const float* feats, const float* proj, const float* depth_values, const float* d_volume, int B,
                      int S, int C, int h, int w, int D, int pad, float* d_feats, bmv_stream_t stream);
int bmv_mvs_vol_feat_bwd(const float* rays, const float* src_ext0, const float* src_ixt0, const float* near_far,
                         const float* d_feat, long N, int Ns, int H, int W, int D, int hp, int wp, int pad,
                         float* d_volume, bmv_stream_t stream);
/* ... and their bit-reproducible twins (see "Bit-reproducible scatter gradients" above; lib/networks/mvsnerf/network.py:
 * 887-942, utils.py:357-383 under loss.backward()) */
int bmv_mvs_sweep_bwd_fixed(const float* feats, const float* proj, const float* depth_values, const float* d_volume, int B,
                            int S, int C, int h, int w, int D, int pad, float* d_feats, long long* workspace,
                            bmv_stream_t stream);
int bmv_mvs_vol_feat_bwd_fixed(const float* rays, const float* src_ext0, const float* src_ixt0, const float* near_far,
                               const float* d_feat, long N, int Ns, int H, int W, int D, int hp, int wp, int pad,
                               float* d_volume, long long* workspace, bmv_stream_t stream);

/* ---- boost_mvsnerf calc_mask               lib/networks/boost_mvsnerf/network.py:23-45
 * rays (N,8) marched with Ns samples between columns 6 and 7; src_exts (V,4,4), src_ixts (V,3,3)
 * -> z_vals (N,Ns), mask (N,Ns) = fraction of the V views whose viewport holds the sample */
int bmv_mvs_march_mask(const float* rays, const float* src_exts, const float* src_ixts, int N, int Ns, int V,
                       float inv_w, float inv_h, float* z_vals, float* mask, bmv_stream_t stream);

/* ==== section 8(f) ranks 1-2: the convolution stacks either side of the plane sweep (inference) ==========
 * One launch per conv block of the reference (conv -> batch norm -> ReLU, lib/networks/enerf/utils.py:10-33;
 * FeatureNet lib/networks/enerf/feature_net.py:4-36; MinCostRegNet / CostRegNet
 * lib/networks/enerf/cost_reg_net.py:4-86).  The host folds the eval-mode batch-norm scale into the weights
 * and passes its shift (or the conv bias) as `bias`.
 *
 * bmv_conv_fwd: in (B,Cin,D,H,W) planar (D = 1, kd = 1 for 2-D), zero padding k/2, stride 1|2, kernels
 * (kd,k) in {(1,1),(1,3),(1,5 stride 2),(3,3)}.  act(v) = v > 0 ? v : act_slope * v (1 = none, 0 = ReLU, 0.01 =
 * InPlaceABN's leaky ReLU of the MVSNeRF stacks, mvsnerf/network.py:699-779).
 * out = act(conv(in) + bias) + skip, planar (B,Cout,Do,Ho,Wo)
 * or channel-last (B,Do,Ho,Wo,Cout) (the sweep's feature layout); skip (nullable) has out's layout.
 * out_channels_last | 16 (round 5; 3x3x3 stride 2, Cin in {4, 8, 12}, Cout <= 16: the regularisers' conv1): `in` is quad
 * records (B, Cin/4, D, H, W, 4) as bmv_conv_c4_fwd's mode 8 writes them.
 * wpack: bmv_conv_wpack_floats() floats laid out [ceil(Cout/16)][ceil(Cin/4)][tap][4][16], zero padded:
 *   wpack[t][c][tap][k][o] = weight[16 t + o][4 c + k][tap]  (tap = (kz*k + ky)*k + kx).
 * Row pairing (bmv_conv_pairs_rows() == 1: Cout <= 8, 3x3 / 3x3x3, stride 1): the 16 rows are the 8 output
 * channels of two adjacent output rows and tap = (kz*(k+1) + j)*k + kx walks the k+1 input rows j they touch:
 *   wpack[0][c][tap][kk][o]     = weight[o][4 c + kk][kz][j][kx]      (j < k,  o < 8)
 *   wpack[0][c][tap][kk][8 + o] = weight[o][4 c + kk][kz][j - 1][kx]  (j >= 1, o < 8), zero elsewhere. */
int bmv_conv_pairs_rows(int Cout, int kd, int k, int stride);
int bmv_conv_wpack_floats(int Cin, int Cout, int kd, int k, int stride);
/* wpack on the device, one launch (training repacks every step).  The blob is for a convolution with Cin inputs and
 * Cout outputs; its weight at (co, ci, tap) is read from `weight` laid out (Cout,Cin,taps), or (Cin,Cout,taps) when
 * transposed != 0, at tap index taps-1-tap when flip != 0 (transposed + flip = the data gradient of a stride-1
 * convolution run as a convolution).  for_transpose_kernel != 0: the blob of bmv_conv3d_transpose_fwd (never paired). */
int bmv_conv_pack_weights(const float* weight, int Cin, int Cout, int kd, int k, int stride, int transposed, int flip,
                          int for_transpose_kernel, float* wpack, bmv_stream_t stream);
/* out_channels_last: 0 = out planar (B,Cout,Do,Ho,Wo); 1 = channel-last (B,Do,Ho,Wo,Cout); 3 = quad-planar
 * (B,Cout/4,Do,Ho,Wo,4), the plane sweep's source layout (bmv_sweep_variance_quad_fwd; Cout % 4 == 0, no skip). */
int bmv_conv_fwd(const float* in, const float* wpack, const float* bias, const float* skip, float* out, int B, int Cin,
                 int D, int H, int W, int Cout, int kd, int k, int stride, float act_slope, int out_channels_last,
                 bmv_stream_t stream);

/* ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1, bias=False) + folded batch norm of the U-Net
 * decoders (cost_reg_net.py:23-41, 62-80: conv7 / conv9 / conv11) with the skip add fused:
 * in (B,Cin,D,H,W) -> out = act(convT(in) + bias) + skip, (B,Cout,2D,2H,2W) planar.
 * wpack as bmv_conv_fwd with wpack[t][c][tap][k][o] = weight[4 c + k][16 t + o][tap]  (torch's (Cin,Cout,3,3,3)). */
int bmv_conv3d_transpose_fwd(const float* in, const float* wpack, const float* bias, const float* skip, float* out, int B, int Cin,
                    int D, int H, int W, int Cout, float act_slope, bmv_stream_t stream);

/* FPN top-down step (feature_net.py:24-36 `_upsample_add` + lat1 / lat0):
 * out (B,C,H,W) = bilinear_x2(coarse (B,C,H/2,W/2), align_corners=True) + conv1x1(fine (B,Cf,H,W); w (C,Cf)) + bias;
 * coarse_channels_last: 0 = coarse planar; 1 = (B,H/2,W/2,C); 3 = quad-planar (B,C/4,H/2,W/2,4) (the coarsest map
 * is kept only in the plane sweep's layout) */
int bmv_fpn_topdown_fwd(const float* fine, const float* coarse, const float* w, const float* bias, float* out, int B,
                        int Cf, int C, int H, int W, int coarse_channels_last, bmv_stream_t stream);

/* feat_conv + depth_conv of a cost regulariser (cost_reg_net.py:43-44, 83-84) as one 9-channel 3x3x3 convolution whose
 * epilogue writes what the fused renderer reads: records_out (B,D,H,W,8), the 8 feature channels of a voxel as one
 * 32-byte record in MFMA row order (the caller packs the weights with output channels 0 2 4 6 1 3 5 7 8, so a record is
 * [even | odd]: a trilinear tap is one 16-byte load per lane half, bmv_render_args.vol_packed), and depth_out
 * (B,D,H,W), channel 8 (the depth logits bmv_depth_regress_fwd reads).  wpack / bias: bmv_conv_pack_weights layout for
 * (Cin, Cout = 9, 3x3x3, stride 1). */
int bmv_conv_heads_fwd(const float* in, const float* wpack, const float* bias, float* records_out, float* depth_out,
                       int B, int Cin, int D, int H, int W, bmv_stream_t stream);

/* ---- f1 / f2: stride-1 3x3x3 (kd = 3) / 3x3 (kd = 1, D = 1) convolutions with FEW output channels (Cout <= 12: the
 * regularisers' first layers and heads, lib/networks/enerf/cost_reg_net.py:4-86 conv0 / feat_conv + depth_conv;
 * FeatureNet's smooth layers, feature_net.py:17-19) on v_mfma_f32_4x4x1_16b_f32 (csrc/conv_c4.hip): 4 output channels
 * x 4 positions per block, 16 blocks per wave-instruction -- every matrix row useful for Cout = 8 (the 16-row tiles of
 * bmv_conv_fwd: 75 % with row pairing, 56 % for the 9-channel heads) at the same fp32 FMA chain per output.
 * in (B,Cin,D,H,W); wpack: bmv_conv_c4_wpack_floats(Cout, Cin, kd) floats [cin chunk of 4][tap][cout group of 4][cout][cin],
 * eval-mode batch norm folded in, zero padded; bias (4 ceil(Cout / 4)); act(v) = v > 0 ? v : slope v.
 * mode 0: out planar (B,Cout,D,H,W).  mode 2: the renderer's volume records -- out (B,D,H,W,8) = output channels 0..7
 * (the caller packs them in the record's [even | odd] order), out2 (B,D,H,W) = channel 8 (Cout = 9: the depth logits).
 * mode | 4 (round 5): `in` is QUAD RECORDS (B, Cin/4, D, H, W, 4), Cin % 4 == 0 -- what bmv_sweep_variance_quad_fwd
 * writes with flags bit 24: the regulariser's first layer stages its tile with one 16-byte load per position and chunk
 * (kd = 3, variant 0).  mode 8 (| 4): out as quad records (B, Cout/4, D, H, W, 4), Cout % 4 == 0 -- for consumers that
 * stage 16-byte records (bmv_conv_fwd's out_channels_last | 16, bmv_conv3d_transpose_c4_fwd's variant | 32).
 * variant: 0 = default tiling, 1.. = tuning. */
int bmv_conv_c4_wpack_floats(int Cout, int Cin, int kd);
int bmv_conv_c4_fwd(const float* in, const float* wpack, const float* bias, float* out, float* out2, int B, int Cin, int D,
                    int H, int W, int Cout, int kd, float slope, int mode, int variant, bmv_stream_t stream);
/* ... and the regularisers' last up-sampling step on the same blocks (cost_reg_net.py:23-41 conv11 = ConvTranspose3d(16, 8,
 * k 3, stride 2, padding 1, output_padding 1) + BatchNorm3d, then the U-Net skip add): in (B,Cin,D,H,W) -> out
 * (B,Cout,2D,2H,2W) = act(convT(in) + bias) + skip (skip nullable, layout of out), Cout <= 8.  wpack as above (kd = 3)
 * with the taps of the transposed weight (Cin,Cout,3,3,3): [cin chunk][tap][cout group][cout][cin].
 * variant | 16 (round 5): `out` is written as QUAD RECORDS (B, Cout/4, 2D, 2H, 2W, 4) (Cout = 8; `skip` stays planar) for
 * bmv_conv_c4_fwd's input mode 4; variant | 32 (with | 16): `skip` is quad records too. */
int bmv_conv3d_transpose_c4_fwd(const float* in, const float* wpack, const float* bias, const float* skip, float* out, int B, int Cin,
                     int D, int H, int W, int Cout, float slope, int variant, bmv_stream_t stream);

/* The same layers (3x3x3, stride 1, padding 1; cost_reg_net.py:4-86 conv0 / feat_conv + depth_conv) on the BF16 matrix
 * cores with THREE-PIECE fp32 operands at fp32 accuracy (round 6, csrc/conv_c4s.hip): every staged input value and every
 * weight is split into hi + mid + lo bf16 pieces (the fp32 value exactly), a product is the six piece products hh, hm, mh,
 * hl, mm, lh accumulated in fp32 (v_mfma_f32_16x16x32_bf16), what is dropped is <= 3 x 2^-24 of a product.
 * in: QUAD RECORDS (B, Cin/4, D, H, W, 4), Cin % 8 == 0 (bmv_sweep_variance_quad_fwd flags bit 24, bmv_conv3d_transpose_c4_fwd
 * variant | 16).  wsplit: bmv_conv_c4s_wsplit_ints(Cin, pair) int32 words [octet][step 3][kz 3][piece 3][lane 64][4]: lane =
 * 16 (slot % 4) + m holds, as 8 bf16, the octet's 8 input channels of matrix row m at IN-PLANE tap slot t = 4 step + lane / 16
 * of filter plane kz (a workgroup walks the input planes of its tile and keeps an octet's 27 operand quads in registers):
 *   pair = 0 (Cout <= 16): row m = output channel m, slot t < 9 = (ky, kx) = (t / 3, t % 3), slots 9..11 zero;
 *   pair = 1 (Cout == 8): row m = (output row r = m / 8, channel m % 8), slot t = (j, kx) = (t / 3, t % 3) walks the 4 input
 *            rows j the row pair touches: weight[channel][cin][kz][j - r][kx] where 0 <= j - r <= 2, else zero.
 * bias (16); act(v) = v > 0 ? v : slope v.  mode 8: out = quad records (B,Cout/4,D,H,W,4), Cout % 4 == 0; mode 2: the
 * renderer's volume records (B,D,H,W,8) of channels 0..7 + out2 (B,D,H,W) = channel 8 (as bmv_conv_c4_fwd; Cout 8 | 9).
 * (No planar form: both are 16-byte stores, and the reference layout is a strided view of either.) */
int bmv_conv_c4s_wsplit_ints(int Cin, int pair);
int bmv_conv_c4s_fwd(const float* in, const int* wsplit, const float* bias, float* out, float* out2, int B, int Cin, int D,
                     int H, int W, int Cout, int pair, float slope, int mode, bmv_stream_t stream);

/* bmv_fpn_smooth_fwd on the BF16 matrix cores with three-piece fp32 operands (round 6, csrc/fpn_s.hip; the inference
 * default): out = act(smooth0(bilinear_x2(coarse, align_corners) + lat0(fine))), feature_net.py:24-36, with the 1x1 lateral
 * convolution FOLDED into the 3x3 weights on the host (smooth0 is linear): 32 channels of up(coarse) + 8 channels of
 * `fine` = five octets for the matrix cores, no 32-channel full-resolution map anywhere.
 * fine (B,8,H,W), coarse (B,32,H/2,W/2), H and W even.  wsplit: bmv_fpn_smooth_s_wsplit_ints() int32 words
 * [octet 5][filter row ky 3][piece 3][lane 64][4]: lane = 16 kk + m holds, as 8 bf16 per piece (hi, mid, lo: the fp32
 * value exactly), the octet's 8 input channels of matrix row m = (output column parity r = m / 8, output channel m % 8) at
 * input column slot kk of a column pair: weight[channel][input channel][ky][kk - r] where 0 <= kk - r <= 2, else zero;
 * octets 0..3 = smooth0's weights on up(coarse), octet 4 = smooth0 . lat0 on fine.  btab (3, 3, 8): the bias by (row case,
 * column case) = (first / interior / last): smooth0's bias + the lateral bias through the taps that lie inside the image.
 * out (B,8,H,W) or, with rgb (B,3,H,W; may be registered with bmv_defer_pointer), packed_out (B,H,W,12) = the renderer's
 * lookup records as bmv_fpn_smooth_fwd writes them. */
int bmv_fpn_smooth_s_wsplit_ints(void);
int bmv_fpn_smooth_s_fwd(const float* fine, const float* coarse, const int* wsplit, const float* btab, float* out,
                         const float* rgb, float* packed_out, int B, int H, int W, float act_slope, bmv_stream_t stream);

/* bmv_conv0_fused_fwd with the second layer on the BF16 matrix cores (round 6, csrc/fpn_s.hip; the inference default):
 * out (B,8,H,W) = act1(conv3x3(relu(conv3x3(in (B,3,H,W); first layer) + b0); second layer) + bias), feature_net.py:8-10, batch norm
 * folded.  w0b0: the first layer as [channel 8][28] floats = its 27 weights (ci, ky, kx) + the bias (computed on the vector
 * ALU from a rolling window of image rows, ReLU).  wsplit: bmv_conv0_s_wsplit_ints() int32 words [filter row ky 3][piece 3]
 * [lane 64][4], one octet in bmv_fpn_smooth_s_fwd's x-paired layout.  bias (8).  `in` may be registered with bmv_defer_pointer. */
int bmv_conv0_s_wsplit_ints(void);
int bmv_conv0_s_fwd(const float* in, const float* w0b0, const int* wsplit, const float* bias, float* out, int B, int H, int W,
                    float slope1, bmv_stream_t stream);

/* FeatureNet's encoder convolutions on the BF16 matrix cores (round 6, csrc/conv2d_s.hip; feature_net.py:11-19, batch norm
 * folded): out (B,Cout,H/stride,W/stride) = act(conv2d(in (B,Cin,H,W); k = ks, stride, zero padding ks / 2) + bias) with
 * three-piece fp32 operands (six v_mfma_f32_16x16x32_bf16 per product group, fp32 accumulation: fp32 accuracy).  Covered:
 * (ks, stride, Cin) = (5, 2, 8 | 16) | (3, 1, 16 | 32); Cout in {16, 32}; H, W even.  Replaces bmv_conv_fwd (nd = 2) for
 * these layers.  wsplit: bmv_conv2d_s_wsplit_ints() int32 words [M tile = Cout / 16][filter row ky][step][piece 3][lane 64][4]:
 * lane = 16 kg + m holds, as 8 bf16 per piece (hi | mid | lo of the fp32 value, exactly), the 8 input channels of octet o
 * at filter column kx for output channel 16 tile + m, where (o, kx) = divmod(4 step + kg, ks) enumerates the (octet,
 * column) pairs of a filter row four to a step (pairs past Cin / 8 * ks: zeros).  bmv_conv2d_s_wsplit_ints: 0 = not covered.
 * SPLIT RECORDS: between two layers of the chain a map may travel as (B, C / 8, piece 3, H, W, 4) int32 -- the 8 channels of an
 * octet at a pixel as 8 bf16 (16 bytes) per piece, hi + mid + lo = the fp32 value exactly -- written by the producing
 * layer's epilogue (`out_records`) and staged by the consuming layer with LDS-DMA (`in_records`; `in` null): the split is
 * paid once per value instead of once per consuming wave, and the result is BIT-IDENTICAL to the planar path.  Exactly one
 * of in / in_records; out and / or out_records. */
int bmv_conv2d_s_wsplit_ints(int Cin, int Cout, int ks, int stride);
int bmv_conv2d_s_fwd(const float* in, const int* in_records, const int* wsplit, const float* bias, float* out, int* out_records,
                     int B, int Cin, int H, int W, int Cout, int ks, int stride, float act_slope, bmv_stream_t stream);

/* FeatureNet's conv2.1 + toplayer as one launch (feature_net.py:14-16): out (B,H,W,32) channel-last (out_layout 1)
 * or (B,8,H,W,4) quad-planar (out_layout 3) =
 * conv1x1(act(conv3x3(in (B,32,H,W); wpack) + bias); wpack_top) + bias_top; both packs in the bmv_conv_pack_weights
 * layout for (32, 32, k = 3 | 1, stride 1).  The 3x3 layer's output is finished in LDS by the same workgroup. */
int bmv_conv_top_fwd(const float* in, const float* wpack, const float* bias, const float* wpack_top,
                     const float* bias_top, float* out, int B, int H, int W, float act_slope, int out_layout,
                     bmv_stream_t stream);

/* FeatureNet's first block as one launch (feature_net.py:8-10: ConvBnReLU(3,8) + ConvBnReLU(8,Cout<=8), eval-mode batch
 * norm folded): out (B,Cout,H,W) = act1(conv3x3(act0(conv3x3(in (B,3,H,W); w0 (8,3,3,3)) + b0); wpack) + bias); the
 * 8-channel intermediate is computed in the second layer's tile producer and never written.  wpack / bias:
 * bmv_conv_pack_weights layout for (Cin = 8, Cout, k = 3, stride 1: row-paired); act_i = v > 0 ? v : slope_i * v. */
int bmv_conv0_fused_fwd(const float* in, const float* w0, const float* b0, const float* wpack, const float* bias,
                        float* out, int B, int Cout, int H, int W, float slope0, float slope1, bmv_stream_t stream);

/* FPN top-down step fused into the smoothing convolution that consumes it (feature_net.py:24-36, smooth0(p0) with
 * p0 = bilinear_x2(p1, align_corners=True) + lat0(c0)): out (B,Cout,H,W) = act(conv3x3(p0; wpack) + bias), p0
 * (B,C,H,W) = bilinear_x2(coarse (B,C,H/2,W/2)) + conv1x1(fine (B,8,H,W); w_lat (C,8)) + b_lat built chunk by chunk
 * in the convolution's tile producer and never written (C = 32 at full resolution is 126 MB per frame each way).
 * wpack / bias: bmv_conv_pack_weights layout for (Cin = C, Cout <= 8, k = 3, stride 1: row-paired).
 * With `packed_out` (B,H,W,12) (and rgb (B,3,H,W), Cout = 8; `out` may be null) the epilogue writes the fused renderer's
 * lookup records instead of the planar map: [ch 0 2 4 6 | ch 1 3 5 7 | r b | g 0] per pixel, where "ch i" is MFMA row
 * order -- the caller packs the weights with output channels permuted to 0 2 4 6 1 3 5 7 -- and r g b are the source
 * image's (bmv_render_args.im_packed). */
int bmv_fpn_smooth_fwd(const float* fine, const float* coarse, const float* w_lat, const float* b_lat,
                       const float* wpack, const float* bias, float* out, const float* rgb, float* packed_out, int B,
                       int Cf, int C, int Cout, int H, int W, float act_slope, bmv_stream_t stream);

/* ---- 3x3x3 stride-1 convolution on the BF16 matrix cores with SPLIT fp32 operands (round 3, csrc/conv_split.hip).
 * parts = 3: x = hi + mid + lo, the 24 mantissa bits of an fp32 number in three bf16 pieces (exact); a product is the six
 *   bf16 MFMAs hh, hm, mh, hl, mm, lh with fp32 accumulation -- the dropped terms are <= 3 x 2^-24 of it, the size of an
 *   fp32 rounding: fp32-equivalent results at 96 instead of 256 matrix-core cycles per 32 k-values.
 * parts = 2: hi + lo, three MFMAs, <= 2^-16 per product (opt-in experiment BMV_CONV_SPLIT=2; 48 cycles).
 * ConvBnReLU3D of cost_reg_net.py:4-86 with Cin % 8 == 0, Cout <= 16, W % 4 == 0; in / out planar fp32 like bmv_conv_fwd.
 * wsplit: bmv_conv3d_split_wsplit_ints(Cin, parts) int32, [octet][step 7][part][lane 64][4] (convnet.pack_conv_split). */
int bmv_conv3d_split_wsplit_ints(int Cin, int parts);
int bmv_conv3d_split_fwd(const float* in, const int* wsplit, int parts, const float* bias, float* out, int B, int Cin, int D,
                         int H, int W, int Cout, float act_slope, bmv_stream_t stream);
/* the 9-channel head convolution in that form, with bmv_conv_heads_fwd's outputs (volume records + depth logits; the
 * weights packed in record order) */
int bmv_conv3d_split_heads_fwd(const float* in, const int* wsplit, int parts, const float* bias, float* records_out,
                               float* depth_out, int B, int Cin, int D, int H, int W, bmv_stream_t stream);

/* ==== section 8(f) rank 4: target rays on the device ======================================================
 * `build_rays`, full-image branch (lib/datasets/enerf_utils.py:25-31, 62-71): tar_ext (B,4,4) world->camera,
 * tar_ixt (B,3,3) at full resolution, render scale (rows 0-1 of the intrinsics are multiplied by it, :28-31) and the
 * size h x w of the scaled image (the caller's: cv2.resize rounds, round(H*scale) x round(W*scale)) ->
 * rays (B, h*w, 8) = [origin | direction | x, y]: the layout of batch['rays_i'].  float64 inside, rounded to
 * float32 once. */
int bmv_make_rays(const float* tar_ext, const float* tar_ixt, int B, int h, int w, double scale, float* rays,
                  bmv_stream_t stream);

/* ==== section 8(d): HIP-event brackets that work inside HIP-graph capture ====================================
 * The reference times a frame with torch.cuda.synchronize brackets (run.py:117-123); the per-kernel durations of the
 * roofline figures come from HIP events on the launch stream.  bmv_event_record on a stream that is CAPTURING adds an
 * event-record node to the graph being captured (every replay stamps the event at that point); otherwise it is
 * hipEventRecord.  bmv_event_elapsed_us needs both events complete (synchronise first). */
typedef void* bmv_event_t; /* hipEvent_t */
int bmv_event_create(bmv_event_t* ev);
int bmv_event_destroy(bmv_event_t ev);
int bmv_event_record(bmv_event_t ev, bmv_stream_t stream);
int bmv_event_elapsed_us(bmv_event_t start, bmv_event_t end, float* us);
/* The producer / consumer renderer (bmv_render_rays_fwd with lookup records) hands tiles between its waves through LDS
 * mailboxes with BOUNDED waits: a wave that gives up (a lost wake-up) leaves its pixels unwritten and counts.
 * bmv_render_pc_check synchronises the device and returns BMV_ERR_LAUNCH (count in bmv_last_error) if any wave gave up
 * since the last reset; bmv_debug_render_pc_inject(1) makes workgroup 0 withhold one wake-up (tests). */
int bmv_render_pc_check(int reset);
int bmv_debug_render_pc_inject(int on);

/* ==== tuning switches of the launchers: explicit library state, never the process environment =================
 * A switch changes WHICH kernel / tile shape a launcher picks, never what is computed.  bmv_tuning_name(i) /
 * bmv_tuning_doc(i) list them (NULL past the end); a switch that is not set has the launcher's default.  May be set,
 * changed and cleared at any time from any thread (the next launch sees it).  Host frameworks that want environment
 * variables apply them through these calls (boostmvsnerfs_amd/_lib.py does, once, at load). */
int bmv_tuning_set(const char* name, int value);
int bmv_tuning_clear(const char* name);
int bmv_tuning_get(const char* name, int* value, int* is_set);
const char* bmv_tuning_name(int i);
const char* bmv_tuning_doc(int i);

/* Bind `start` / `stop` to the NEXT plane-sweep launch of this thread (bmv_sweep_variance_fwd /
 * bmv_sweep_variance_views_fwd, windowed kernel): the launch goes through hipExtLaunchKernelGGL and the events read the
 * kernel's own begin and end (a record pair around a launch reads ~2.5 us more; scripts/ubench/ext_events.hip).  Not
 * for a capturing stream.  bmv_launch_events_pending() after the call: 1 if no launch took them (they are dropped). */
int bmv_bind_next_launch(bmv_event_t start, bmv_event_t stop);
int bmv_launch_events_pending(void);

/* ---- deferred pointers: a captured frame (HIP graph) that reads the caller's tensors and writes the caller's outputs
 * A graph bakes its kernels' arguments in.  run.py hands Network.forward OTHER device tensors every frame
 * (run.py:113-123: `batch[k] = batch[k].cuda()`), so a replayed frame had to copy them into its captured buffers and
 * its results out again (31 MB per 512x640 frame).  Instead: bmv_defer_pointer(ptr, table, slot) says that the argument
 * of the NEXT launch of this thread whose value is `ptr` is to be read from the DEVICE table entry table[slot] when
 * the kernel RUNS; bmv_ptr_table_set() (a 1-workgroup launch, outside the graph) points the entries at this frame's
 * tensors before the replay.  Arguments that can be deferred: `in` of bmv_conv0_fused_fwd, `rgb` of
 * bmv_fpn_smooth_fwd, `rays` / `out0` / `out1` / `out2` of bmv_render_rays_fwd (the deferrals of one launch share one
 * table).  A registered deferral the next launch does not take fails that call with BMV_ERR_UNSUPPORTED (never a
 * silently baked pointer); bmv_deferred_pending() returns the number of untaken deferrals and drops them. */
int bmv_defer_pointer(const void* ptr, const void* const* table, int slot);
int bmv_deferred_pending(void);
/* table[slots[i]] = values[i], i < n <= 16, on `stream` */
int bmv_ptr_table_set(void* table, int n, const int* slots, const void* const* values, bmv_stream_t stream);
/* The same plus up to 8 SMALL device-to-device copies dst[i][0..counts[i]) = src[i][...] (<= 65536 floats each: the
 * cameras and near / far of a frame into the captured buffers) in ONE 1-workgroup launch: under run.py's per-frame
 * synchronize every launch in front of the replayed graph costs 10-15 us of latency whatever it moves. */
int bmv_frame_feed(void* table, int n_ptr, const int* slots, const void* const* values, int n_copy,
                   const float* const* src, float* const* dst, const int* counts, bmv_stream_t stream);
/* bmv_frame_feed as the FIRST NODE of the frame's graph: its arguments come from a ring of R messages in pinned (host
 * coherent) memory that the host fills with plain stores before each replay -- no launch in front of the replay.
 * Message n (bmv_frame_feed_msg_bytes() bytes: {u32 seq = n; i32 n_ptr, n_copy, pad; i32 slot[16]; void* value[16];
 * float* src[8]; float* dst[8]; i32 count[8]}) is read by the n-th execution of the node (state[0], device memory,
 * counts them; up to R frames may be in flight); a message whose seq is not the execution's number raises state[1]. */
int bmv_frame_feed_ring(void* table, const void* ring, unsigned* state, int R, bmv_stream_t stream);
int bmv_frame_feed_msg_bytes(void);
/* n floats from `src` to the tensor table[slot] points at WHEN THE KERNEL RUNS (a frame's small outputs, as a node of
 * the frame's own graph: nothing is left to copy after the replay); no-op when the entry points at `src` itself. */
int bmv_copy_to_slot(const float* src, const void* const* table, int slot, long n, bmv_stream_t stream);
/* ... up to 8 such copies in one launch: counts[i] floats from src[i] to the tensor table[slots[i]] points at */
int bmv_copy_to_slots(int n, const float* const* src, const void* const* table, const int* slots, const long* counts,
                      bmv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BMV_H_ */
