// torch operator registration for the render path (north_star: "via a torch C++/HIP extension"): `torch.ops.diner.render`
// and `torch.ops.diner.render_image` = NeRFRendererDGS.forward (reference src/models/nerf_renderer.py:399-424) and the render half
// of DINER.predict_imgs_from_batch (src/models/diner.py:75-97) as dispatcher ops on the caller's current stream.
// No arithmetic lives here: the ops check their tensors and call the C ABI of libdiner_hip.so (include/diner_hip.h) -- the same
// entry points the ctypes binding uses, so the two bindings cannot drift.  Built by __graft_entry__.build() (g++ against the torch
// headers; host code only), loaded with torch.ops.load_library by diner_amd/ops.py.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/diner_hip.h"

namespace {

const float *fp(const at::Tensor &t, const char *name)
{
    TORCH_CHECK(t.is_cuda(), "diner: ", name, " must be a GPU tensor");
    TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous(), "diner: ", name, " must be contiguous float32");
    return t.data_ptr<float>();
}

void *stream_of(const at::Tensor &t) { return (void *)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

// the ctypes binding's mapping (diner_amd/_lib.py check()): INVALID -> ValueError, UNSUPPORTED -> NotImplementedError, else RuntimeError
void check_rc(int rc, const char *what)
{
    if (rc == DINER_OK) return;
    TORCH_CHECK_VALUE(rc != DINER_E_INVALID, what, ": ", diner_last_error());
    TORCH_CHECK_NOT_IMPLEMENTED(rc != DINER_E_UNSUPPORTED, what, ": ", diner_last_error());
    TORCH_CHECK(false, what, ": ", diner_last_error(), " (rc=", rc, ")");
}

// The extension is a separate file from libdiner_hip.so: one compiled against another ABI version (a stale build next to a fresh
// library) would pass a wrong argument list.  Checked at every op entry (one call + compare).
void check_abi()
{
    TORCH_CHECK(diner_version() == DINER_ABI_VERSION, "diner: libdiner_torch_ops.so was built for ABI version ", DINER_ABI_VERSION,
                " but libdiner_hip.so is version ", diner_version(), ": rebuild the extension (diner_amd.ops.build(force=True))");
}

int64_t abi_version() { return DINER_ABI_VERSION; }

DinerScene scene_of(const at::Tensor &maps, const at::Tensor &poses, const at::Tensor &focal, const at::Tensor &c, const at::Tensor &latent,
                    const c10::optional<at::Tensor> &linz, double image_w, double image_h, double feature_padding, int64_t num_freqs,
                    double freq_factor)
{
    TORCH_CHECK(maps.dim() == 5 && maps.size(4) == 8, "diner: maps must be [SB,NV,H,W,8] (diner_pack_maps)");
    TORCH_CHECK(latent.dim() == 5, "diner: latent must be [SB,NV,h,w,C] (diner_pack_latent)");
    TORCH_CHECK(latent.size(0) == maps.size(0) && latent.size(1) == maps.size(1), "diner: maps / latent batch mismatch");
    DinerScene s;
    s.SB = (int32_t)maps.size(0); s.NV = (int32_t)maps.size(1); s.H = (int32_t)maps.size(2); s.W = (int32_t)maps.size(3);
    s.h = (int32_t)latent.size(2); s.w = (int32_t)latent.size(3); s.C = (int32_t)latent.size(4);
    s.num_freqs = (int32_t)num_freqs;
    s.image_w = (float)image_w; s.image_h = (float)image_h; s.feature_padding = (float)feature_padding; s.freq_factor = (float)freq_factor;
    s.poses = fp(poses, "poses"); s.focal = fp(focal, "focal"); s.c = fp(c, "c");
    s.maps = fp(maps, "maps"); s.latent = fp(latent, "latent");
    s.linz_maps = linz.has_value() ? fp(*linz, "linz_maps") : nullptr;
    return s;
}

DinerSamplerCfg cfg_of(int64_t n_candidates, int64_t n_samples, int64_t n_gaussian, double depth_diff_max)
{
    DinerSamplerCfg c;
    c.n_candidates = (int32_t)n_candidates; c.n_samples = (int32_t)n_samples; c.n_gaussian = (int32_t)n_gaussian;
    c.depth_diff_max = (float)depth_diff_max;
    return c;
}

// rays [SB,NR,8] -> rgb [SB,NR,3], depth [SB,NR], weights [SB,NR,K] (empty unless want_weights)
std::tuple<at::Tensor, at::Tensor, at::Tensor> render(const at::Tensor &maps, const at::Tensor &poses, const at::Tensor &focal, const at::Tensor &c,
                                                      const at::Tensor &latent, const c10::optional<at::Tensor> &linz, const at::Tensor &mlp,
                                                      const at::Tensor &rays, double image_w, double image_h, double feature_padding,
                                                      int64_t num_freqs, double freq_factor, int64_t n_candidates, int64_t n_samples,
                                                      int64_t n_gaussian, double depth_diff_max, bool white_bkgd, int64_t precision, int64_t seed,
                                                      bool want_weights, const c10::optional<at::Tensor> &status)
{
    check_abi();
    const DinerScene s = scene_of(maps, poses, focal, c, latent, linz, image_w, image_h, feature_padding, num_freqs, freq_factor);
    const DinerSamplerCfg cfg = cfg_of(n_candidates, n_samples, n_gaussian, depth_diff_max);
    TORCH_CHECK(rays.dim() == 3 && rays.size(2) == 8 && rays.size(0) == s.SB, "diner: rays must be [SB,NR,8]");   // nerf_renderer.py:412
    const int64_t NR = rays.size(1);
    const auto opt = rays.options();
    at::Tensor ws = at::empty({diner_render_workspace_floats(s.SB, NR, cfg.n_samples, s.NV, (int32_t)precision)}, opt);
    at::Tensor rgb = at::empty({s.SB, NR, 3}, opt), depth = at::empty({s.SB, NR}, opt);
    at::Tensor weights = want_weights ? at::empty({s.SB, NR, n_samples}, opt) : at::empty({0}, opt);
    uint32_t *st = status.has_value() ? (uint32_t *)status->data_ptr() : nullptr;
    check_rc(diner_render(&s, fp(mlp, "mlp_packed"), fp(rays, "rays"), NR, &cfg, white_bkgd ? 1 : 0, (int32_t)precision, nullptr, nullptr, nullptr,
                          (uint64_t)seed, ws.data_ptr<float>(), rgb.data_ptr<float>(), depth.data_ptr<float>(),
                          want_weights ? weights.data_ptr<float>() : nullptr, st, stream_of(rays)),
             "diner_render");
    return {rgb, depth, weights};
}

// target cameras -> rgb [SB,H*W,3], depth [SB,H*W]; the rays are generated inside the sampler kernel (cam_geometry.py:36-79)
std::tuple<at::Tensor, at::Tensor> render_image(const at::Tensor &maps, const at::Tensor &poses, const at::Tensor &focal, const at::Tensor &c,
                                                const at::Tensor &latent, const c10::optional<at::Tensor> &linz, const at::Tensor &mlp,
                                                const at::Tensor &extrinsics, const at::Tensor &intrinsics, const at::Tensor &z_near,
                                                const at::Tensor &z_far, int64_t H, int64_t W, double image_w, double image_h,
                                                double feature_padding, int64_t num_freqs, double freq_factor, int64_t n_candidates,
                                                int64_t n_samples, int64_t n_gaussian, double depth_diff_max, bool white_bkgd, int64_t precision,
                                                int64_t seed, const c10::optional<at::Tensor> &status)
{
    check_abi();
    const DinerScene s = scene_of(maps, poses, focal, c, latent, linz, image_w, image_h, feature_padding, num_freqs, freq_factor);
    const DinerSamplerCfg cfg = cfg_of(n_candidates, n_samples, n_gaussian, depth_diff_max);
    TORCH_CHECK(extrinsics.numel() == (int64_t)s.SB * 16 && intrinsics.numel() == (int64_t)s.SB * 9 && z_near.numel() == s.SB && z_far.numel() == s.SB,
                "diner: target camera tensors must be [SB,4,4], [SB,3,3], [SB], [SB]");
    DinerTargetCam cam;
    cam.extrinsics = fp(extrinsics, "extrinsics"); cam.intrinsics = fp(intrinsics, "intrinsics");
    cam.z_near = fp(z_near, "z_near"); cam.z_far = fp(z_far, "z_far");
    cam.H = (int32_t)H; cam.W = (int32_t)W;
    const auto opt = extrinsics.options();
    at::Tensor ws = at::empty({diner_render_image_workspace_floats(s.SB, cam.H, cam.W, cfg.n_samples, s.NV, (int32_t)precision)}, opt);
    at::Tensor rgb = at::empty({s.SB, H * W, 3}, opt), depth = at::empty({s.SB, H * W}, opt);
    uint32_t *st = status.has_value() ? (uint32_t *)status->data_ptr() : nullptr;
    check_rc(diner_render_image(&s, fp(mlp, "mlp_packed"), &cam, &cfg, white_bkgd ? 1 : 0, (int32_t)precision, (uint64_t)seed, ws.data_ptr<float>(),
                                nullptr, rgb.data_ptr<float>(), depth.data_ptr<float>(), nullptr, st, stream_of(extrinsics)),
             "diner_render_image");
    return {rgb, depth};
}

}  // namespace

TORCH_LIBRARY(diner, m)
{
    m.def("abi_version() -> int", &abi_version);   // the DINER_ABI_VERSION this extension was compiled against
    m.def("render(Tensor maps, Tensor poses, Tensor focal, Tensor c, Tensor latent, Tensor? linz_maps, Tensor mlp_packed, Tensor rays, "
          "float image_w, float image_h, float feature_padding, int num_freqs, float freq_factor, int n_candidates, int n_samples, "
          "int n_gaussian, float depth_diff_max, bool white_bkgd, int precision, int seed, bool want_weights, Tensor? status) "
          "-> (Tensor, Tensor, Tensor)");
    m.def("render_image(Tensor maps, Tensor poses, Tensor focal, Tensor c, Tensor latent, Tensor? linz_maps, Tensor mlp_packed, "
          "Tensor extrinsics, Tensor intrinsics, Tensor z_near, Tensor z_far, int H, int W, float image_w, float image_h, "
          "float feature_padding, int num_freqs, float freq_factor, int n_candidates, int n_samples, int n_gaussian, float depth_diff_max, "
          "bool white_bkgd, int precision, int seed, Tensor? status) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(diner, CUDA, m)
{
    m.impl("render", &render);
    m.impl("render_image", &render_image);
}
