/*
 * diner_hip.h -- C ABI of the MI355X-native DINER render path (libdiner_hip.so).
 *
 * The reference (tancredeguillou/diner) implements this path in pure Python/PyTorch and has no
 * FFI; each entry point below therefore names the reference *Python* function it replaces
 * (paths relative to the reference root).  The Python plug-in class
 * diner_amd.NeRFRendererDGS (drop-in for src/models/nerf_renderer.py:12 NeRFRendererDGS, selected
 * by the YAML key renderer.module, src/models/diner.py:48) binds these with ctypes; the binding a
 * reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 unless stated otherwise; plain pointers and sizes
 *    only, no torch types;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing synchronises;
 *  - every function returns 0 on success, <0 on error (DINER_E_*), never aborts the process;
 *    diner_last_error() returns a thread-local message for the last failure;
 *  - tensors use the reference's own shapes: SB scenes, NV source views, NR rays per scene,
 *    NC candidates, K samples, G gaussian samples; rays [SB,NR,8] = origin(3) dir(3) near far.
 */
#ifndef DINER_HIP_H
#define DINER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DINER_OK 0
#define DINER_E_INVALID (-1)  /* bad argument (NULL pointer, unsupported size) */
#define DINER_E_LAUNCH (-2)   /* HIP launch / runtime failure */
#define DINER_E_UNSUPPORTED (-3)

#define DINER_D_LATENT 512 /* SpatialEncoder.latent_size, num_layers=4 (src/models/image_encoder.py:56) */
#define DINER_D_HIDDEN 512 /* configs/train_diner_facescape.yaml:57 */
#define DINER_D_IN 55      /* PE3 39 + viewdir 3 + PE1 13 (src/models/pixelnerf.py:18) */
#define DINER_N_BLOCKS 5
#define DINER_COMBINE_LAYER 3
#define DINER_MAP_TEXEL 8  /* floats per packed map texel: nx ny nz depth sigma 0 0 0 */

/* Arithmetic of the fusion-MLP GEMMs (everything else is fp32 in both modes):
 *  FP32  : v_mfma_f32_32x32x2_f32, bit-identical to a k-ordered fmaf chain;
 *  F16X3 : every fp32 operand split into fp16 hi+lo, 3 fp16 MFMAs per product, fp32 accumulate
 *          (fp32-grade: operand error <= 2^-23 relative, dropped lo*lo term <= 2^-22). */
#define DINER_PRECISION_FP32 0
#define DINER_PRECISION_F16X3 1

/* Per-scene state the renderer reads from the model:
 * PixelNeRF buffers (src/models/pixelnerf.py:27-30,47-51) and SpatialEncoder state
 * (src/models/image_encoder.py:92-95,214-218,271-272), with the maps re-packed once per
 * encode() by diner_pack_maps / diner_pack_latent. */
typedef struct DinerScene {
    int32_t SB, NV;          /* scenes in the batch, source views */
    int32_t H, W;            /* size of the depth / sigma / normal maps */
    int32_t h, w, C;         /* latent map size and channels (C must be DINER_D_LATENT) */
    int32_t num_freqs;       /* positional-encoding octaves (6) */
    float image_w, image_h;  /* model.image_shape = (W, H) */
    float feature_padding;   /* encoder.feature_padding in latent texels (32) */
    float freq_factor;       /* 6.28 (configs/train_diner_facescape.yaml:51) */
    const float *poses;      /* [SB,NV,4,4] world->camera */
    const float *focal;      /* [SB,NV,2] */
    const float *c;          /* [SB,NV,2] */
    const float *maps;       /* [SB,NV,H,W,8] packed by diner_pack_maps */
    const float *latent;     /* [SB,NV,h,w,C] packed by diner_pack_latent (may be NULL for the sampler) */
    const float *linz_maps;  /* [3][SB,NV,h,w,C] from diner_pack_linz_maps, or NULL: the F16X3 kernel then
                                evaluates lin_z per point like the FP32 kernel does */
} DinerScene;

/* ResnetFC parameters in the reference's nn.Linear layout, weight [out,in]
 * (src/models/resnetfc.py:72-127); input of diner_pack_mlp. */
typedef struct DinerMlpRaw {
    const float *lin_in_w, *lin_in_b;                  /* [512,55], [512] */
    const float *lin_z_w[DINER_COMBINE_LAYER], *lin_z_b[DINER_COMBINE_LAYER]; /* [512,512] */
    const float *fc0_w[DINER_N_BLOCKS], *fc0_b[DINER_N_BLOCKS];
    const float *fc1_w[DINER_N_BLOCKS], *fc1_b[DINER_N_BLOCKS];
    const float *lin_out_w, *lin_out_b;                /* [4,512], [4] */
} DinerMlpRaw;

typedef struct DinerSamplerCfg {
    int32_t n_candidates;    /* NC  (n_depth_candidates, 1000) */
    int32_t n_samples;       /* K   (n_samples, 40) */
    int32_t n_gaussian;      /* G   (n_gaussian, 15), 0 <= G <= K */
    float depth_diff_max;    /* 0.05 (src/models/nerf_renderer.py:67) */
} DinerSamplerCfg;

/* Target cameras for diner_render_image: the sampler generates each ray from its pixel instead of reading a rays tensor
 * (gen_rays fused into the sampler; src/util/cam_geometry.py:36-79). */
typedef struct DinerTargetCam {
    const float *extrinsics; /* [SB,4,4] world->camera */
    const float *intrinsics; /* [SB,3,3] */
    const float *z_near;     /* [SB] */
    const float *z_far;      /* [SB] */
    int32_t H, W;            /* target image size: rays per scene = H*W, ray r = pixel (r / W, r % W) */
} DinerTargetCam;

/* Version of THIS ABI (argument lists, struct layouts).  Bumped by every incompatible change; a binding compiled or written
 * against another value must refuse to call in: diner_version() returns the value the loaded library was built with, the
 * torch-ops extension checks it at every op entry, diner_amd/_lib.py at load time.  (2: diner_render / diner_composite gained
 * `status`.) */
#define DINER_ABI_VERSION 2

const char *diner_last_error(void);
int diner_version(void);

/* ---- adjacent per-image producers, one thread per pixel ---------- */
/* gen_rays (src/util/cam_geometry.py:36-79): extrinsics [B,4,4] world->camera, intrinsics [B,3,3],
 * z_near/z_far [B] -> rays [B,H,W,8] (pixel-centre rays: origin, unit direction, near, far). */
int diner_gen_rays(const float *extrinsics, const float *intrinsics, const float *z_near, const float *z_far,
                   int32_t B, int32_t H, int32_t W, float *rays_out, void *stream);
/* depth2normal (src/util/depth2normal.py:7-87): dmap [N,1,H,W], K [N,3,3] -> normals [N,3,H,W]
 * (central differences of the re-projected depth map + the reference's hole clean-up). */
int diner_depth2normal(const float *dmap, const float *intrinsics, int32_t N, int32_t H, int32_t W,
                       float *normals_out, void *stream);

/* ---- once per encode(): re-pack the model's maps for the kernels ----------------------- */
/* depths, depths_std [N,1,H,W], normals [N,3,H,W] (N = SB*NV) -> maps [N,H,W,8] */
int diner_pack_maps(const float *depths, const float *depths_std, const float *normals,
                    int64_t N, int32_t H, int32_t W, float *maps_out, void *stream);
/* the same with depth2normal (src/util/depth2normal.py:7-87, called at src/models/pixelnerf.py:45) fused in:
 * depths, depths_std [N,1,H,W], intrinsics [N,3,3] -> maps [N,H,W,8]; no NCHW normal tensor is materialised */
int diner_pack_maps_from_depth(const float *depths, const float *depths_std, const float *intrinsics,
                               int64_t N, int32_t H, int32_t W, float *maps_out, void *stream);
/* Upstream wire format: TransMVSNet's uint16 depth / confidence planes [N,H,W] ->
 * depth_out, std_out (and mask_out = depth > 0, optional) [N,H/stride,W/stride] fp32, nearest-subsampled by
 * `stride` (src/data/dtu.py:113-117):  depth = ((u16 * mul0) / div) * mul1,  std = std_a * conf + std_b with the
 * confidence decoded like the depth (src/data/dtu.py:100-119,220-223; src/data/facescape.py:54-56,80-91,266).
 * mesh (optional): Facescape's mesh-rendered depth for depth_type "merge" (src/data/facescape.py:96-104). */
int diner_decode_depth_u16(const uint16_t *depth, const uint16_t *conf, const uint16_t *mesh, int64_t N, int32_t H,
                           int32_t W, int32_t stride, float mul0, float div, float mul1, float std_a, float std_b,
                           float *depth_out, float *std_out, float *mask_out, void *stream);
/* latent [N,C,h,w] (NCHW, src/models/image_encoder.py:271) -> [N,h,w,C] with the channel order
 * the MLP kernel stages into LDS (C = 512) */
int diner_pack_latent(const float *latent_nchw, int64_t N, int32_t C, int32_t h, int32_t w,
                      float *latent_out, void *stream);
/* ---- once per weight version: MFMA-fragment-ordered copies of the fusion MLP (one image per
 * precision mode, both in the same buffer) ------------------------------------------------ */
int64_t diner_mlp_packed_floats(void);
int diner_pack_mlp(const DinerMlpRaw *raw, float *packed_out, void *stream);
/* ---- once per (encode, weight version): G_b = lin_z[b](latent), b = 0..2, as feature maps ----
 * lin_z (src/models/resnetfc.py:152) is linear and SpatialEncoder.index (src/models/image_encoder.py:97-127)
 * is a 4-texel convex combination, so lin_z[b](index(uv)) == bilerp(G_b)(uv) up to fp32 rounding.
 * latent_packed [N,h,w,512] from diner_pack_latent, mlp_packed from diner_pack_mlp -> out [3][N,h,w,512]. */
int diner_pack_linz_maps(const float *latent_packed, int64_t N, int32_t h, int32_t w, const float *mlp_packed,
                         float *out, void *stream);

/* ---- the hot path ---------------------------------------------------------------------- */
/* Stage entry points with the reference's stage boundaries (for stage-level parity tests and for
 * callers that use the stages on their own): */
/* NeRFRendererDGS.sample_coarse (src/models/nerf_renderer.py:39-63): rays [N,8] -> z [N,NC];
 * u_coarse [N,NC] or NULL (Philox keyed on seed). */
int diner_sample_coarse(const float *rays, int64_t N, int32_t NC, const float *u_coarse, uint64_t seed,
                        float *z_out, void *stream);
/* NeRFRendererDGS.fill_up_uniform_samples (src/models/nerf_renderer.py:367-397): z_in [N,K] with
 * 0 = empty slot -> z_out [N,K] filled and sorted; u_fill [N,K] (column i feeds the i-th empty
 * slot) or NULL. */
int diner_fill_up_uniform_samples(const float *rays, const float *z_in, int64_t N, int32_t K,
                                  const float *u_fill, uint64_t seed, float *z_out, void *stream);

/* Replaces NeRFRendererDGS.sample_coarse + sample_depthguided + fill_up_uniform_samples
 * (src/models/nerf_renderer.py:39-63, 65-284, 367-397).  One ray per wavefront.
 *   noise (parity mode, any may be NULL -> in-kernel Philox keyed on `seed`):
 *     u_coarse [SB,NR,NC] U[0,1);  n_gauss [SB,NR,G] N(0,1), row used iff the ray has a hit;
 *     u_fill [SB,NR,K] U[0,1), column i feeds the i-th empty slot of the ray.
 *   z_cand  [SB,NR,NC] optional: inject the candidates instead of computing them (tests).
 *   z_out   [SB,NR,K] sorted samples (what composite() consumes).
 *   z_dg_out [SB,NR,K] optional: samples BEFORE fill-up, kept candidates first (0 = empty),
 *            gaussian draws in the last G slots (the return value of sample_depthguided).
 *   lik_out [SB,NR,NC] optional: per-candidate likelihood max over views (:129). */
int diner_sample_depthguided(const DinerScene *scene, const float *rays, int64_t NR,
                             const DinerSamplerCfg *cfg, const float *u_coarse,
                             const float *n_gauss, const float *u_fill, const float *z_cand,
                             uint64_t seed, float *z_out, float *z_dg_out, float *lik_out,
                             void *stream);

/* Replaces the model evaluation inside composite(): points = o + z*d (:304-307) and
 * PixelNeRF.forward (src/models/pixelnerf.py:55-145) incl. PositionalEncoding.forward,
 * SpatialEncoder.index / index_depth and ResnetFC.forward.  z [SB,NR,K] -> rgbsigma [SB,NR,K,4].
 * mlp_packed from diner_pack_mlp.  scratch: device buffer of diner_render_points_scratch_floats()
 * floats (the F16X3 kernel parks per-view hidden states there until the mean over views; may be
 * NULL when that is 0); one buffer per concurrently running launch. */
int64_t diner_render_points_scratch_floats(int64_t SB, int32_t NV, int32_t precision);
int diner_render_points(const DinerScene *scene, const float *mlp_packed, const float *rays,
                        const float *z, int64_t NR, int32_t K, int32_t precision, float *scratch,
                        float *rgbsigma_out, void *stream);

/* Replaces the alpha compositing of composite() (src/models/nerf_renderer.py:299-301,341-360).
 * N rays (= SB*NR).  weights_out [N,K] optional.
 * status (optional): one device word, OR-ed with DINER_STATUS_NONFINITE when any rgb-sigma sample is inf/NaN
 * (the reference would hand NaN images on silently; in f16x3 mode it means an activation left the fp16 range,
 * |x| >= ~1e6, and the frame must be re-rendered with DINER_PRECISION_FP32).  Sticky: the caller clears it. */
#define DINER_STATUS_NONFINITE 1u
int diner_composite(const float *rays, const float *z, const float *rgbsigma, int64_t N, int32_t K,
                    int32_t white_bkgd, float *rgb_out, float *depth_out, float *weights_out,
                    uint32_t *status, void *stream);

/* Replaces NeRFRendererDGS.forward (src/models/nerf_renderer.py:399-424): the three stages
 * back to back on `stream`.  workspace: device buffer of diner_render_workspace_floats(...)
 * floats (holds z, rgbsigma and the point kernel's scratch). */
int64_t diner_render_workspace_floats(int64_t SB, int64_t NR, int32_t K, int32_t NV, int32_t precision);
int diner_render(const DinerScene *scene, const float *mlp_packed, const float *rays, int64_t NR,
                 const DinerSamplerCfg *cfg, int32_t white_bkgd, int32_t precision, const float *u_coarse,
                 const float *n_gauss, const float *u_fill, uint64_t seed, float *workspace,
                 float *rgb_out, float *depth_out, float *weights_out, uint32_t *status, void *stream);

/* Replaces the render half of DINER.predict_imgs_from_batch (src/models/diner.py:75-97): gen_rays
 * (src/util/cam_geometry.py:36-79) is evaluated INSIDE the sampler kernel -- a wave computes its ray from the pixel index and
 * the target camera, and stores it once for the two later stages -- then the three stages run as in diner_render.
 * workspace: diner_render_image_workspace_floats(...) floats (rays | z | rgbsigma | scratch).  Outputs [SB,H*W,3], [SB,H*W],
 * [SB,H*W,K]|NULL; rays_out [SB,H*W,8]|NULL also hands the generated rays to the caller. */
int64_t diner_render_image_workspace_floats(int64_t SB, int32_t H, int32_t W, int32_t K, int32_t NV, int32_t precision);
int diner_render_image(const DinerScene *scene, const float *mlp_packed, const DinerTargetCam *cam,
                       const DinerSamplerCfg *cfg, int32_t white_bkgd, int32_t precision, uint64_t seed,
                       float *workspace, float *rays_out, float *rgb_out, float *depth_out, float *weights_out,
                       uint32_t *status, void *stream);

/* ---- training path: building blocks of the forward-with-saved-activations and
 * the backward of composite (src/models/nerf_renderer.py:286-365) + PixelNeRF.forward
 * (src/models/pixelnerf.py:55-145) + ResnetFC.forward (src/models/resnetfc.py:129-159), orchestrated by
 * diner_amd/training.py exactly like autograd orchestrates the reference's ATen ops.  Gradients: MLP
 * parameters and encoder.latent (NCHW); the sampler is @torch.no_grad in the reference. ------------------ */
/* C[m][n] (+)= sum_k opA(A[m*sam + k*sak]) * opB(B[k*sbk + n*sbn]) (+ bias[n]) (* [S[m*lds + n] > 0]).
 * Each operand must be contiguous along one of its two indices; N % 4 == 0; k_chunk (0 = no split, else a
 * multiple of 32) splits the contraction over blockIdx.z (use with atomic = 1).
 * precision DINER_PRECISION_FP32: exact fp32 MFMA (amax / exp arguments ignored).
 * precision DINER_PRECISION_F16X3: each operand element is multiplied by a power of two, split into fp16
 * hi + lo, three fp16 MFMAs per product, fp32 accumulate (the arithmetic of diner_render_points' default
 * mode).  The power of two is 2^exp_a (2^exp_b), or, when amax_a (amax_b) is not NULL, the one that maps the
 * value stored there by diner_train_amax into [2^13, 2^14) -- meant for gradients, whose magnitude is
 * arbitrary while fp16 has an absolute floor of 2^-24.  C is divided by the product of the two scales. */
int diner_train_gemm(const float *A, const float *B, const float *bias, const float *S, float *C, int64_t M,
                     int32_t N, int32_t K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc,
                     int64_t lds, int32_t relu_a, int32_t relu_b, int32_t accumulate, int32_t atomic,
                     int64_t k_chunk, int32_t precision, const void *amax_a, const void *amax_b, int32_t exp_a,
                     int32_t exp_b, void *stream);
/* *amax_out (one 32-bit device word) = bit pattern of max |x[i]|, i < n (0 for an empty or all-zero tensor) */
int diner_train_amax(const float *x, int64_t n, void *amax_out, void *stream);
/* Both reductions of a gradient matrix in one pass: db[n] += sum_m dY[m*ld + n] (skipped if db is NULL) and
 * *amax_out as diner_train_amax (skipped if NULL).  N % 4 == 0 and N/4 must divide 256. */
int diner_train_colsum_amax(const float *dY, int64_t M, int32_t N, int64_t ld, float *db, void *amax_out, void *stream);
/* Weight operand of diner_train_gemm_panel: B[n][k] = (transpose ? W[k*ld + n] : W[n*ld + k]) * 2^exp, n < 512,
 * k < K, as fp16 hi / lo planes of 512 * ceil32(K) halfs each, laid out [k/32][512][32] (one k-step of the GEMM
 * is one contiguous 32-KiB piece per plane), zero-padded in k. */
int diner_train_split_panel(const float *W, int32_t K, int64_t ld, int32_t transpose, int32_t exp, void *hi, void *lo,
                            void *stream);
/* C[m][n] = addend[m*ldadd + n] + (sum_k opA(A[m*sam + k]) * B[n][k] + bias[n]) * [S[m*lds + n] > 0], n < 512:
 * the forward and dX GEMMs of the training path in f16x3 arithmetic (see diner_train_gemm) with the weights
 * arriving pre-split (no conversion work for them in the GEMM) and A split twice instead of four times.  opA = relu if relu_a,
 * scaled by 2^exp_a or by *amax_a like diner_train_gemm; exp_b must be the exponent given to
 * diner_train_split_panel.  bias, S, addend may be NULL; addend may alias C. */
int diner_train_gemm_panel(const float *A, int64_t sam, const void *Bhi, const void *Blo, const float *bias,
                           const float *S, int64_t lds, const float *addend, int64_t ldadd, float *C, int64_t ldc,
                           int64_t M, int32_t K, int32_t relu_a, const void *amax_a, int32_t exp_a, int32_t exp_b,
                           void *stream);
/* Weight operand of diner_train_gemm_core: B[n][k] = (transpose ? W[k*ld + n] : W[n*ld + k]) * 2^exp, n, k < 512, fp16 hi/lo in the
 * stream layout of the inference kernel's GEMM core (points_mlp_f16.hip, "packed weight image"): out = 512*512*2 halfs. */
int diner_train_pack_core(const float *W, int64_t ld, int32_t transpose, int32_t exp, void *out, void *stream);
/* diner_train_gemm_panel for K = 512 on the inference kernel's assembly GEMM core (same f16x3 arithmetic; A read and split once per
 * 64-row tile, the weights L2 -> registers, no workgroup barrier in the loop).  Optionally folds the two reductions of
 * diner_train_colsum_amax over the RESULT into the epilogue: colsum[n] += sum_m C[m][n], *amax_out = max(*amax_out, bits of max|C|)
 * (either may be NULL).  Reference: the Linear layers of ResnetFC (src/models/resnetfc.py:62-69) and their autograd transposes. */
int diner_train_gemm_core(const float *A, int64_t sam, const void *Wcore, const float *bias, const float *S, int64_t lds,
                          const float *addend, int64_t ldadd, float *C, int64_t ldc, int64_t M, int32_t relu_a, const void *amax_a,
                          int32_t exp_a, int32_t exp_b, float *colsum, void *amax_out, void *stream);
/* db[n] += sum_m dY[m*ld + n] */
int diner_train_colsum(const float *dY, int64_t M, int32_t N, int64_t ld, float *db, void *stream);
/* per (view, point) row = v*P + p of scene sb: in56 [R,56] (55 inputs of pixelnerf.py:128 + 0), z [R,512]
 * (bilinear latent, image_encoder.py:97-127), taps [R,8] (4 texel indices, 4 weights).  latent: the reference's
 * NCHW tensor [SB,NV,512,h,w], or (latent_is_nhwc) its diner_pack_latent copy [SB,NV,h,w,512] -- same values, a
 * wave then reads whole 256-byte pieces of a texel instead of 64 planes */
int diner_train_point_inputs(const DinerScene *scene, const float *latent, int32_t latent_is_nhwc, const float *rays,
                             const float *z, int64_t NR, int32_t K, int32_t sb, float *in56, float *zlat, float *taps,
                             void *stream);
/* dlatent_nhwc[sb][v][texel][ch] += dz[row][ch] * weight (float atomics on 256-byte contiguous rows; the
 * caller zeroes the [SB,NV,h,w,C] buffer), then diner_train_nhwc_to_nchw gives encoder.latent's layout */
int diner_train_bilinear_scatter(const float *dz, const float *taps, int64_t P, int32_t C, int32_t h, int32_t w,
                                 int32_t NV, int32_t sb, float *dlatent_nhwc, void *stream);
int diner_train_nhwc_to_nchw(const float *nhwc, int64_t N, int32_t C, int32_t h, int32_t w, float *nchw_out,
                             void *stream);
/* forward: x [NV,PC] -> mean [PC] (resnetfc.py:146-149); backward: d_mean [PC] -> dx [NV,PC] */
int diner_train_view_mean(const float *x, int64_t PC, int32_t NV, float *out, int32_t backward, void *stream);
/* forward: out [n4] -> sigmoid/relu head (pixelnerf.py:139-143); backward: d_out from d_rgbsigma */
int diner_train_head(const float *out, const float *rgbsigma, const float *d_rgbsigma, int64_t n4, float *result,
                     int32_t backward, void *stream);
/* backward of diner_composite: d_rgb [N,3], d_depth [N]|NULL, d_weights [N,K]|NULL -> d_rgbsigma [N,K,4] */
int diner_composite_backward(const float *rays, const float *z, const float *rgbsigma, const float *d_rgb,
                             const float *d_depth, const float *d_weights, int64_t N, int32_t K,
                             int32_t white_bkgd, float *d_rgbsigma, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DINER_HIP_H */
