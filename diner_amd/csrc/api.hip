// extern "C" entry points of libdiner_hip.so (declared in include/diner_hip.h): argument
// validation, error strings, stream plumbing.  No kernel code here.
#include <stdarg.h>
#include <stdio.h>

#include <atomic>

#include "common.hpp"

namespace diner {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DINER_E_LAUNCH;
    }
    return DINER_OK;
}

// ---- per-device launch state (common.hpp) ---------------------------------------------------------------------------------------
constexpr int MAX_DEVICES = 64;
static std::atomic<int> g_cus[MAX_DEVICES];                       // 0 = not asked yet
static std::atomic<int> g_lds_set[MAX_DEVICES][LDS_SLOT_COUNT];   // bytes the kernel of a slot has been raised to on a device

static int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev;
}

int device_cus()
{
    const int dev = current_device();
    if (dev < 0 || dev >= MAX_DEVICES) {      // beyond the table: ask every time
        int n = 0;
        return (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    int n = g_cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
        g_cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

int ensure_dynamic_lds(const void *kernel, int bytes, int slot)
{
    const int dev = current_device();
    const bool cached = dev >= 0 && dev < MAX_DEVICES && slot >= 0 && slot < LDS_SLOT_COUNT;
    if (cached && g_lds_set[dev][slot].load(std::memory_order_acquire) == bytes) return DINER_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return check_launch("hipFuncSetAttribute(dynamic LDS)");
    if (cached) g_lds_set[dev][slot].store(bytes, std::memory_order_release);
    return DINER_OK;
}

int launch_composite(const float *, const float *, const float *, int64_t, int, int, float *, float *, float *, unsigned int *, hipStream_t);
int launch_pack_maps(const float *, const float *, const float *, int64_t, int, int, float *, hipStream_t);
int launch_pack_latent(const float *, int64_t, int, int, int, float *, hipStream_t);
int launch_pack_mlp(const DinerMlpRaw &, float *, hipStream_t);
int64_t mlp_packed_floats();
int launch_train_gemm(const float *, const float *, const float *, const float *, float *, int64_t, int, int, int64_t, int64_t, int64_t,
                      int64_t, int64_t, int64_t, int, int, int, int, int64_t, int, const unsigned int *, const unsigned int *, int, int,
                      hipStream_t);
int launch_train_amax(const float *, int64_t, unsigned int *, hipStream_t);
int launch_train_colsum_amax(const float *, int64_t, int, int64_t, float *, unsigned int *, hipStream_t);
int launch_train_split_panel(const float *, int, int64_t, int, int, void *, void *, hipStream_t);
int launch_train_pack_core(const float *, int64_t, int, int, void *, hipStream_t);
int launch_train_gemm_core(const float *, int64_t, const void *, const float *, const float *, int64_t, const float *, int64_t, float *, int64_t,
                           int64_t, int, const unsigned int *, int, int, float *, unsigned int *, hipStream_t);
int launch_train_gemm_panel(const float *, int64_t, const void *, const void *, const float *, const float *, int64_t, const float *,
                            int64_t, float *, int64_t, int64_t, int, int, const unsigned int *, int, int, hipStream_t);
int launch_train_colsum(const float *, int64_t, int, int64_t, float *, hipStream_t);
int launch_train_point_inputs(const DinerScene &, const float *, int, const float *, const float *, int64_t, int, int, float *, float *,
                              float *, hipStream_t);
int launch_train_bilinear_scatter(const float *, const float *, int64_t, int, int, int, int, int, float *, hipStream_t);
int launch_train_view_mean(const float *, int64_t, int, float *, int, hipStream_t);
int launch_train_nhwc_to_nchw(const float *, int64_t, int, int, int, float *, hipStream_t);
int launch_train_head(const float *, const float *, const float *, int64_t, float *, int, hipStream_t);
int launch_train_composite_bwd(const float *, const float *, const float *, const float *, const float *, const float *, int64_t, int,
                               int, float *, hipStream_t);
int launch_gen_rays(const float *, const float *, const float *, const float *, int, int, int, float *, hipStream_t);
int launch_depth2normal(const float *, const float *, int, int, int, float *, hipStream_t);
int launch_pack_maps_from_depth(const float *, const float *, const float *, int, int, int, float *, hipStream_t);
int launch_decode_depth(const unsigned short *, const unsigned short *, const unsigned short *, int64_t, int, int, int, float, float, float,
                        float, float, float *, float *, float *, hipStream_t);
int launch_linz_maps(const float *, int64_t, const float *, float *, hipStream_t);
int launch_sampler(const DinerScene &, const float *, const DinerTargetCam *, float *, int64_t, const DinerSamplerCfg &, const float *,
                   const float *, const float *, const float *, uint64_t, float *, float *, float *, hipStream_t);
int launch_sample_coarse(const float *, int64_t, int, const float *, uint64_t, float *, hipStream_t);
int launch_fill_up(const float *, const float *, int64_t, int, const float *, uint64_t, float *, hipStream_t);
int launch_points_mlp(const DinerScene &, const float *, const float *, const float *, int64_t, int, float *, hipStream_t);
int64_t mlp_f16_packed_floats();
int launch_pack_mlp_f16(const DinerMlpRaw &, float *, hipStream_t);
int launch_points_mlp_f16(const DinerScene &, const float *, const float *, const float *, int64_t, int, float *, float *, hipStream_t);
int64_t points_mlp_f16_scratch_floats(int64_t SB, int NV);

static int bad(const char *msg)
{
    set_error("%s", msg);
    return DINER_E_INVALID;
}

static int check_scene(const DinerScene *s, bool need_latent)
{
    if (!s) return bad("scene is NULL");
    if (s->SB < 0 || s->NV <= 0 || s->H <= 0 || s->W <= 0) return bad("scene: bad SB/NV/H/W");
    if (!s->poses || !s->focal || !s->c || !s->maps) return bad("scene: NULL camera or map pointer");
    if (!(s->image_w > 0.f) || !(s->image_h > 0.f)) return bad("scene: bad image_shape");
    if (need_latent) {
        if (!s->latent) return bad("scene: latent is NULL");
        if (s->h <= 0 || s->w <= 0) return bad("scene: bad latent size");
    }
    return DINER_OK;
}

static int check_cfg(const DinerSamplerCfg *c)
{
    if (!c) return bad("sampler cfg is NULL");
    if (c->n_candidates < 1) return bad("n_candidates must be >= 1");
    if (c->n_samples < 1 || c->n_samples > 8192) return bad("n_samples must be in [1, 8192]");
    if (c->n_gaussian < 0 || c->n_gaussian > c->n_samples) return bad("need 0 <= n_gaussian <= n_samples");  // nerf_renderer.py:89
    return DINER_OK;
}

}  // namespace diner

using namespace diner;

extern "C" {

const char *diner_last_error(void) { return g_err; }
int diner_version(void) { return DINER_ABI_VERSION; }

int diner_gen_rays(const float *extrinsics, const float *intrinsics, const float *z_near, const float *z_far, int32_t B,
                   int32_t H, int32_t W, float *rays_out, void *stream)
{
    if (B < 0 || H <= 0 || W <= 0) return bad("gen_rays: bad size");
    if (B > 0 && (!extrinsics || !intrinsics || !z_near || !z_far || !rays_out)) return bad("gen_rays: NULL pointer");
    return launch_gen_rays(extrinsics, intrinsics, z_near, z_far, B, H, W, rays_out, (hipStream_t)stream);
}

int diner_depth2normal(const float *dmap, const float *intrinsics, int32_t N, int32_t H, int32_t W, float *normals_out,
                       void *stream)
{
    if (N < 0 || H <= 0 || W <= 0) return bad("depth2normal: bad size");
    if (N > 0 && (!dmap || !intrinsics || !normals_out)) return bad("depth2normal: NULL pointer");
    return launch_depth2normal(dmap, intrinsics, N, H, W, normals_out, (hipStream_t)stream);
}

int diner_pack_maps(const float *depths, const float *depths_std, const float *normals, int64_t N, int32_t H,
                    int32_t W, float *maps_out, void *stream)
{
    if (!depths || !depths_std || !normals || !maps_out) return bad("pack_maps: NULL pointer");
    if (N < 0 || H <= 0 || W <= 0) return bad("pack_maps: bad size");
    return launch_pack_maps(depths, depths_std, normals, N, H, W, maps_out, (hipStream_t)stream);
}

int diner_pack_maps_from_depth(const float *depths, const float *depths_std, const float *intrinsics, int64_t N, int32_t H,
                               int32_t W, float *maps_out, void *stream)
{
    if (!depths || !depths_std || !intrinsics || !maps_out) return bad("pack_maps_from_depth: NULL pointer");
    if (N < 0 || N > 0x7fffffff || H <= 0 || W <= 0) return bad("pack_maps_from_depth: bad size");
    return launch_pack_maps_from_depth(depths, depths_std, intrinsics, (int)N, H, W, maps_out, (hipStream_t)stream);
}

int diner_decode_depth_u16(const uint16_t *depth, const uint16_t *conf, const uint16_t *mesh, int64_t N, int32_t H, int32_t W, int32_t stride,
                           float mul0, float div, float mul1, float std_a, float std_b, float *depth_out, float *std_out, float *mask_out,
                           void *stream)
{
    if (!depth || !conf || !depth_out || !std_out) return bad("decode_depth_u16: NULL pointer");
    if (N < 0 || H <= 0 || W <= 0 || stride <= 0 || H % stride || W % stride) return bad("decode_depth_u16: bad size (stride must divide H and W)");
    if (!(div != 0.0f)) return bad("decode_depth_u16: div must be non-zero");
    return launch_decode_depth(depth, conf, mesh, N, H, W, stride, mul0, div, mul1, std_a, std_b, depth_out, std_out, mask_out,
                               (hipStream_t)stream);
}

int diner_pack_latent(const float *latent_nchw, int64_t N, int32_t C, int32_t h, int32_t w, float *latent_out,
                      void *stream)
{
    if (!latent_nchw || !latent_out) return bad("pack_latent: NULL pointer");
    if (N < 0 || h <= 0 || w <= 0) return bad("pack_latent: bad size");
    return launch_pack_latent(latent_nchw, N, C, h, w, latent_out, (hipStream_t)stream);
}

int64_t diner_mlp_packed_floats(void) { return mlp_packed_floats() + mlp_f16_packed_floats(); }

int diner_pack_mlp(const DinerMlpRaw *raw, float *packed_out, void *stream)
{
    if (!raw || !packed_out) return bad("pack_mlp: NULL pointer");
    const float *const *p = (const float *const *)raw;
    for (size_t i = 0; i < sizeof(DinerMlpRaw) / sizeof(float *); ++i)
        if (!p[i]) return bad("pack_mlp: NULL weight pointer");
    const int rc = launch_pack_mlp(*raw, packed_out, (hipStream_t)stream);
    if (rc) return rc;
    return launch_pack_mlp_f16(*raw, packed_out + mlp_packed_floats(), (hipStream_t)stream);
}

int diner_pack_linz_maps(const float *latent_packed, int64_t N, int32_t h, int32_t w, const float *mlp_packed, float *out,
                         void *stream)
{
    if (!latent_packed || !mlp_packed || !out) return bad("pack_linz_maps: NULL pointer");
    if (N < 0 || h <= 0 || w <= 0) return bad("pack_linz_maps: bad size");
    return launch_linz_maps(latent_packed, N * h * w, mlp_packed, out, (hipStream_t)stream);
}

int diner_sample_coarse(const float *rays, int64_t N, int32_t NC, const float *u_coarse, uint64_t seed, float *z_out,
                        void *stream)
{
    if (N < 0 || NC < 1) return bad("sample_coarse: bad N / NC");
    if (N > 0 && (!rays || !z_out)) return bad("sample_coarse: NULL pointer");
    return launch_sample_coarse(rays, N, NC, u_coarse, seed, z_out, (hipStream_t)stream);
}

int diner_fill_up_uniform_samples(const float *rays, const float *z_in, int64_t N, int32_t K, const float *u_fill,
                                  uint64_t seed, float *z_out, void *stream)
{
    if (N < 0 || K < 1 || K > 8192) return bad("fill_up: bad N / K");
    if (N > 0 && (!rays || !z_in || !z_out)) return bad("fill_up: NULL pointer");
    return launch_fill_up(rays, z_in, N, K, u_fill, seed, z_out, (hipStream_t)stream);
}

int diner_sample_depthguided(const DinerScene *scene, const float *rays, int64_t NR, const DinerSamplerCfg *cfg,
                             const float *u_coarse, const float *n_gauss, const float *u_fill, const float *z_cand,
                             uint64_t seed, float *z_out, float *z_dg_out, float *lik_out, void *stream)
{
    int rc;
    if ((rc = check_scene(scene, false)) || (rc = check_cfg(cfg))) return rc;
    if (NR < 0) return bad("NR < 0");
    if (NR > 0 && scene->SB > 0 && (!rays || !z_out)) return bad("sample_depthguided: NULL rays / z_out");
    return launch_sampler(*scene, rays, nullptr, nullptr, NR, *cfg, u_coarse, n_gauss, u_fill, z_cand, seed, z_out, z_dg_out, lik_out,
                          (hipStream_t)stream);
}

int64_t diner_render_points_scratch_floats(int64_t SB, int32_t NV, int32_t precision)
{
    return precision == DINER_PRECISION_F16X3 ? points_mlp_f16_scratch_floats(SB, NV) : 0;
}

int diner_render_points(const DinerScene *scene, const float *mlp_packed, const float *rays, const float *z,
                        int64_t NR, int32_t K, int32_t precision, float *scratch, float *rgbsigma_out, void *stream)
{
    int rc;
    if ((rc = check_scene(scene, true))) return rc;
    if (NR < 0 || K < 1) return bad("render_points: bad NR / K");
    if (!mlp_packed) return bad("render_points: mlp_packed is NULL");
    if (NR > 0 && scene->SB > 0 && (!rays || !z || !rgbsigma_out)) return bad("render_points: NULL rays / z / out");
    if (precision == DINER_PRECISION_FP32) return launch_points_mlp(*scene, mlp_packed, rays, z, NR, K, rgbsigma_out, (hipStream_t)stream);
    if (precision == DINER_PRECISION_F16X3)
        return launch_points_mlp_f16(*scene, mlp_packed + mlp_packed_floats(), rays, z, NR, K, scratch, rgbsigma_out,
                                     (hipStream_t)stream);
    return bad("render_points: unknown precision");
}

int diner_composite(const float *rays, const float *z, const float *rgbsigma, int64_t N, int32_t K,
                    int32_t white_bkgd, float *rgb_out, float *depth_out, float *weights_out, uint32_t *status, void *stream)
{
    if (N < 0 || K < 1) return bad("composite: bad N / K");
    if (N > 0 && (!rays || !z || !rgbsigma || !rgb_out || !depth_out)) return bad("composite: NULL pointer");
    return launch_composite(rays, z, rgbsigma, N, K, white_bkgd, rgb_out, depth_out, weights_out, status, (hipStream_t)stream);
}

/* ---- training path building blocks (train.hip) ------------------------------------------------ */
int diner_train_gemm(const float *A, const float *B, const float *bias, const float *S, float *C, int64_t M, int32_t N, int32_t K,
                     int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc, int64_t lds, int32_t relu_a, int32_t relu_b,
                     int32_t accumulate, int32_t atomic, int64_t k_chunk, int32_t precision, const void *amax_a, const void *amax_b,
                     int32_t exp_a, int32_t exp_b, void *stream)
{
    if (!A || !B || !C) return bad("train_gemm: NULL pointer");
    if (precision != DINER_PRECISION_FP32 && precision != DINER_PRECISION_F16X3) return bad("train_gemm: unknown precision");
    if (M < 0 || N <= 0 || K <= 0 || (N & 3) || (k_chunk & 31)) return bad("train_gemm: bad size (N % 4, k_chunk % 32 must be 0)");
    if (exp_a < -60 || exp_a > 60 || exp_b < -60 || exp_b > 60) return bad("train_gemm: scale exponent out of range");
    return launch_train_gemm(A, B, bias, S, C, M, N, K, sam, sak, sbk, sbn, ldc, lds, relu_a, relu_b, accumulate, atomic, k_chunk,
                             precision, (const unsigned int *)amax_a, (const unsigned int *)amax_b, exp_a, exp_b, (hipStream_t)stream);
}

int diner_train_amax(const float *x, int64_t n, void *amax_out, void *stream)
{
    if (!x || !amax_out || n < 0) return bad("train_amax: bad argument");
    return launch_train_amax(x, n, (unsigned int *)amax_out, (hipStream_t)stream);
}

int diner_train_colsum_amax(const float *dY, int64_t M, int32_t N, int64_t ld, float *db, void *amax_out, void *stream)
{
    if (!dY || M < 0 || N <= 0 || (N & 3) || 256 % (N / 4) || (ld & 3)) return bad("train_colsum_amax: bad argument (N % 4 == 0 and (N/4) | 256)");
    if (!db && !amax_out) return bad("train_colsum_amax: nothing to compute");
    return launch_train_colsum_amax(dY, M, N, ld, db, (unsigned int *)amax_out, (hipStream_t)stream);
}

int diner_train_split_panel(const float *W, int32_t K, int64_t ld, int32_t transpose, int32_t exp, void *hi, void *lo, void *stream)
{
    if (!W || !hi || !lo || K <= 0 || ld <= 0) return bad("train_split_panel: bad argument");
    if (exp < -60 || exp > 60) return bad("train_split_panel: scale exponent out of range");
    return launch_train_split_panel(W, K, ld, transpose, exp, hi, lo, (hipStream_t)stream);
}

int diner_train_gemm_panel(const float *A, int64_t sam, const void *Bhi, const void *Blo, const float *bias, const float *S, int64_t lds,
                           const float *addend, int64_t ldadd, float *C, int64_t ldc, int64_t M, int32_t K, int32_t relu_a,
                           const void *amax_a, int32_t exp_a, int32_t exp_b, void *stream)
{
    if (!A || !Bhi || !Blo || !C) return bad("train_gemm_panel: NULL pointer");
    if (M < 0 || K < 4 || (K & 3) || (sam & 3) || (ldc < DINER_D_HIDDEN)) return bad("train_gemm_panel: bad size (K % 4, sam % 4 must be 0)");
    if (((uintptr_t)A & 15) || ((uintptr_t)Bhi & 15) || ((uintptr_t)Blo & 15)) return bad("train_gemm_panel: operands must be 16-byte aligned");
    if (exp_a < -60 || exp_a > 60 || exp_b < -60 || exp_b > 60) return bad("train_gemm_panel: scale exponent out of range");
    return launch_train_gemm_panel(A, sam, Bhi, Blo, bias, S, lds, addend, ldadd, C, ldc, M, K, relu_a, (const unsigned int *)amax_a, exp_a,
                                   exp_b, (hipStream_t)stream);
}

int diner_train_pack_core(const float *W, int64_t ld, int32_t transpose, int32_t exp, void *out, void *stream)
{
    if (!W || !out) return bad("train_pack_core: NULL pointer");
    if (ld < DINER_D_HIDDEN || exp < -60 || exp > 60) return bad("train_pack_core: bad leading dimension / exponent");
    return launch_train_pack_core(W, ld, transpose, exp, out, (hipStream_t)stream);
}

int diner_train_gemm_core(const float *A, int64_t sam, const void *Wcore, const float *bias, const float *S, int64_t lds, const float *addend,
                          int64_t ldadd, float *C, int64_t ldc, int64_t M, int32_t relu_a, const void *amax_a, int32_t exp_a, int32_t exp_b,
                          float *colsum, void *amax_out, void *stream)
{
    if (!A || !Wcore || !C) return bad("train_gemm_core: NULL pointer");
    if (M < 0 || sam < DINER_D_HIDDEN || (sam & 3) || ldc < DINER_D_HIDDEN || (ldc & 3) || (S && (lds & 3)) || (addend && (ldadd & 3)))
        return bad("train_gemm_core: bad size (K = N = 512; leading dimensions % 4 must be 0)");
    if (((uintptr_t)A & 15) || ((uintptr_t)Wcore & 15) || ((uintptr_t)C & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)S & 15) || ((uintptr_t)addend & 15))
        return bad("train_gemm_core: operands must be 16-byte aligned");
    if (exp_a < -60 || exp_a > 60 || exp_b < -60 || exp_b > 60) return bad("train_gemm_core: scale exponent out of range");
    return launch_train_gemm_core(A, sam, Wcore, bias, S, lds, addend, ldadd, C, ldc, M, relu_a, (const unsigned int *)amax_a, exp_a, exp_b, colsum,
                                  (unsigned int *)amax_out, (hipStream_t)stream);
}

int diner_train_colsum(const float *dY, int64_t M, int32_t N, int64_t ld, float *db, void *stream)
{
    if (!dY || !db || M < 0 || N <= 0) return bad("train_colsum: bad argument");
    return launch_train_colsum(dY, M, N, ld, db, (hipStream_t)stream);
}

int diner_train_point_inputs(const DinerScene *scene, const float *latent, int32_t latent_is_nhwc, const float *rays, const float *z,
                             int64_t NR, int32_t K, int32_t sb, float *in56, float *zlat, float *taps, void *stream)
{
    int rc;
    if ((rc = check_scene(scene, false))) return rc;
    if (!latent || !rays || !z || !in56 || !zlat || !taps) return bad("train_point_inputs: NULL pointer");
    if (scene->C != DINER_D_LATENT || scene->h <= 0 || scene->w <= 0 || sb < 0 || sb >= scene->SB) return bad("train_point_inputs: bad scene");
    return launch_train_point_inputs(*scene, latent, latent_is_nhwc, rays, z, NR, K, sb, in56, zlat, taps, (hipStream_t)stream);
}

int diner_train_bilinear_scatter(const float *dz, const float *taps, int64_t P, int32_t C, int32_t h, int32_t w, int32_t NV, int32_t sb,
                                 float *dlatent_nhwc, void *stream)
{
    if (!dz || !taps || !dlatent_nhwc) return bad("train_bilinear_scatter: NULL pointer");
    return launch_train_bilinear_scatter(dz, taps, P, C, h, w, NV, sb, dlatent_nhwc, (hipStream_t)stream);
}

int diner_train_nhwc_to_nchw(const float *nhwc, int64_t N, int32_t C, int32_t h, int32_t w, float *nchw_out, void *stream)
{
    if (!nhwc || !nchw_out || N < 0 || C <= 0 || h <= 0 || w <= 0) return bad("train_nhwc_to_nchw: bad argument");
    return launch_train_nhwc_to_nchw(nhwc, N, C, h, w, nchw_out, (hipStream_t)stream);
}

int diner_train_view_mean(const float *x, int64_t PC, int32_t NV, float *out, int32_t backward, void *stream)
{
    if (!x || !out || NV < 1) return bad("train_view_mean: bad argument");
    return launch_train_view_mean(x, PC, NV, out, backward, (hipStream_t)stream);
}

int diner_train_head(const float *out, const float *rgbsigma, const float *d_rgbsigma, int64_t n4, float *result, int32_t backward,
                     void *stream)
{
    if (!out || !result || (backward && (!rgbsigma || !d_rgbsigma))) return bad("train_head: NULL pointer");
    return launch_train_head(out, rgbsigma, d_rgbsigma, n4, result, backward, (hipStream_t)stream);
}

int diner_composite_backward(const float *rays, const float *z, const float *rgbsigma, const float *d_rgb, const float *d_depth,
                             const float *d_weights, int64_t N, int32_t K, int32_t white_bkgd, float *d_rgbsigma, void *stream)
{
    if (N < 0 || K < 1) return bad("composite_backward: bad N / K");
    if (N > 0 && (!rays || !z || !rgbsigma || !d_rgb || !d_rgbsigma)) return bad("composite_backward: NULL pointer");
    return launch_train_composite_bwd(rays, z, rgbsigma, d_rgb, d_depth, d_weights, N, K, white_bkgd, d_rgbsigma, (hipStream_t)stream);
}

int64_t diner_render_workspace_floats(int64_t SB, int64_t NR, int32_t K, int32_t NV, int32_t precision)
{
    return SB * NR * (int64_t)K * 5 + diner_render_points_scratch_floats(SB, NV, precision);
}

int diner_render(const DinerScene *scene, const float *mlp_packed, const float *rays, int64_t NR,
                 const DinerSamplerCfg *cfg, int32_t white_bkgd, int32_t precision, const float *u_coarse, const float *n_gauss,
                 const float *u_fill, uint64_t seed, float *workspace, float *rgb_out, float *depth_out,
                 float *weights_out, uint32_t *status, void *stream)
{
    int rc;
    if ((rc = check_scene(scene, true)) || (rc = check_cfg(cfg))) return rc;
    if (NR < 0) return bad("NR < 0");
    if (NR == 0 || scene->SB == 0) return DINER_OK;
    if (!workspace) return bad("render: workspace is NULL");
    const int64_t N = (int64_t)scene->SB * NR;
    float *z = workspace, *rgbsigma = workspace + N * cfg->n_samples, *scratch = workspace + N * cfg->n_samples * 5;
    if ((rc = diner_sample_depthguided(scene, rays, NR, cfg, u_coarse, n_gauss, u_fill, nullptr, seed, z, nullptr,
                                       nullptr, stream)))
        return rc;
    if ((rc = diner_render_points(scene, mlp_packed, rays, z, NR, cfg->n_samples, precision, scratch, rgbsigma, stream))) return rc;
    return diner_composite(rays, z, rgbsigma, N, cfg->n_samples, white_bkgd, rgb_out, depth_out, weights_out, status, stream);
}

int64_t diner_render_image_workspace_floats(int64_t SB, int32_t H, int32_t W, int32_t K, int32_t NV, int32_t precision)
{
    const int64_t NR = (int64_t)H * W;
    return SB * NR * 8 + diner_render_workspace_floats(SB, NR, K, NV, precision);
}

int diner_render_image(const DinerScene *scene, const float *mlp_packed, const DinerTargetCam *cam, const DinerSamplerCfg *cfg,
                       int32_t white_bkgd, int32_t precision, uint64_t seed, float *workspace, float *rays_out, float *rgb_out,
                       float *depth_out, float *weights_out, uint32_t *status, void *stream)
{
    int rc;
    if ((rc = check_scene(scene, true)) || (rc = check_cfg(cfg))) return rc;
    if (!cam || !cam->extrinsics || !cam->intrinsics || !cam->z_near || !cam->z_far) return bad("render_image: NULL camera");
    if (cam->H <= 0 || cam->W <= 0) return bad("render_image: bad image size");
    if (scene->SB == 0) return DINER_OK;
    if (!workspace) return bad("render_image: workspace is NULL");
    const int64_t NR = (int64_t)cam->H * cam->W, N = (int64_t)scene->SB * NR;
    float *rays = rays_out ? rays_out : workspace;                 // generated by the sampler, read by the two later stages
    float *z = workspace + N * 8, *rgbsigma = z + N * cfg->n_samples, *scratch = z + N * cfg->n_samples * 5;
    if ((rc = launch_sampler(*scene, nullptr, cam, rays, NR, *cfg, nullptr, nullptr, nullptr, nullptr, seed, z, nullptr, nullptr,
                             (hipStream_t)stream)))
        return rc;
    if ((rc = diner_render_points(scene, mlp_packed, rays, z, NR, cfg->n_samples, precision, scratch, rgbsigma, stream))) return rc;
    return diner_composite(rays, z, rgbsigma, N, cfg->n_samples, white_bkgd, rgb_out, depth_out, weights_out, status, stream);
}

}  // extern "C"
