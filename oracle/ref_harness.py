"""TEST INFRASTRUCTURE -- runs ONLY in the build container (never on the GPU box, never from
the product path).

Imports the *unmodified* reference render path from ``/root/reference`` on the CPU, so it can
(1) pin the C restatement in ``oracle/diner_oracle.c`` and (2) generate the golden vectors that
are committed under ``tests/golden`` (``oracle/gen_golden.py``).

The reference path needs three third-party modules that are absent here and carry no
arithmetic of the path: ``dotmap`` (attribute dict), ``imageio`` (video
writer, unused) and ``torchvision`` (``Normalize`` + the ResNet34 trunk, which is *not* on the
hot path).  They are stubbed in ``sys.modules`` before the import; no reference file is copied
or modified.
"""
from __future__ import annotations

import contextlib
import sys
import types
from types import SimpleNamespace as NS

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


# ----------------------------------------------------------------------------------------
# stubs for the three missing non-arithmetic modules
# ----------------------------------------------------------------------------------------
class _DotMap(dict):
    def __init__(self, **kw):
        super().__init__(**kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Normalize(torch.nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.mean = torch.tensor(mean).view(-1, 1, 1)
        self.std = torch.tensor(std).view(-1, 1, 1)

    def forward(self, x):
        return (x - self.mean) / self.std


class _FakeTrunk(torch.nn.Module):
    """Shape-compatible stand-in for torchvision.models.resnet34: the CNN trunk runs once per
    image in ``encode`` and is out of scope; the harness sets
    ``encoder.latent`` directly, so this only has to construct."""

    def __init__(self, pretrained=False, norm_layer=None):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = torch.nn.Identity()
        self.relu = torch.nn.ReLU()
        self.maxpool = torch.nn.Identity()
        self.layer1 = self.layer2 = self.layer3 = self.layer4 = torch.nn.Identity()
        self.fc = self.avgpool = torch.nn.Identity()


def _pil_to_tensor(pic):
    a = np.array(pic, copy=True)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.dtype == np.uint16:
        a = a.astype(np.int32)
    return torch.from_numpy(a).permute(2, 0, 1).contiguous()


def _resize_nearest(x, size, interpolation=None):
    assert interpolation in (None, "nearest")
    return torch.nn.functional.interpolate(x, size=list(size), mode="nearest")


def install_stubs():
    if "dotmap" not in sys.modules:
        m = types.ModuleType("dotmap")
        m.DotMap = _DotMap
        sys.modules["dotmap"] = m
    if "imageio" not in sys.modules:
        sys.modules["imageio"] = types.ModuleType("imageio")
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        trf = types.ModuleType("torchvision.transforms.functional")
        mo = types.ModuleType("torchvision.models")
        tr.Normalize = _Normalize
        tr.functional = trf
        # dataset readers (src/data/dtu.py, facescape.py; used only by the wire-format / pose goldens): pil_to_tensor is a pure
        # dtype/layout conversion (PIL image -> integer tensor [C,H,W], values unchanged; a 16-bit PNG is widened to int32 since
        # torch has no arithmetic on uint16), resize(NEAREST) on a tensor is what torchvision itself calls
        # (torch.nn.functional.interpolate(mode="nearest")), save_image is unused on the path
        trf.resize = _resize_nearest
        trf.pil_to_tensor = _pil_to_tensor
        tr.InterpolationMode = NS(NEAREST="nearest")
        tu = types.ModuleType("torchvision.utils")
        tu.save_image = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
        mo.resnet34 = _FakeTrunk
        tv.transforms, tv.models, tv.utils = tr, mo, tu
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.utils": tu,
                            "torchvision.transforms.functional": trf, "torchvision.models": mo})
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    install_stubs()
    import matplotlib
    matplotlib.use("Agg")
    from src.models.nerf_renderer import NeRFRendererDGS  # noqa
    from src.models.pixelnerf import PixelNeRF  # noqa
    return NS(NeRFRendererDGS=NeRFRendererDGS, PixelNeRF=PixelNeRF)


# ----------------------------------------------------------------------------------------
# model construction from a synthetic scene
# ----------------------------------------------------------------------------------------
def build_model(scene, weights, image_padding=None, dtype=torch.float32):
    """Reference ``PixelNeRF`` with the scene's maps/cameras and the given MLP weights.
    ``scene`` is a ``synthetic.synth.Scene``; ``weights`` a dict from ``make_mlp_weights``."""
    ref = import_reference()
    image_padding = 2 * scene.feature_padding if image_padding is None else image_padding
    nerf = ref.PixelNeRF(
        poscode_conf=NS(kwargs=dict(num_freqs=6, freq_factor=6.28, include_input=True)),
        encoder_conf=NS(module="src.models.image_encoder.SpatialEncoder",
                        kwargs=dict(image_padding=image_padding, padding_pe=4, pretrained=False)),
        mlp_fine_conf=NS(module="src.models.resnetfc.ResnetFC",
                         kwargs=dict(n_blocks=5, d_hidden=512, combine_layer=3,
                                     combine_type="average")))
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    missing = nerf.mlp_fine.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    enc = nerf.encoder
    enc.depths, enc.depths_std, enc.normals = t(scene.depths), t(scene.depths_std), t(scene.normals)
    enc.nviews, enc.nobjects = scene.NV, scene.poses.shape[0]
    if scene.latent is not None:
        enc.latent = t(scene.latent)
    nerf.poses, nerf.focal, nerf.c = t(scene.poses), t(scene.focal), t(scene.c)
    nerf.image_shape = t(scene.image_shape)
    assert enc.feature_padding == scene.feature_padding
    nerf = nerf.to(dtype)
    # non-persistent buffers set above are plain attributes -> .to() does not touch them
    return nerf.eval()


# ----------------------------------------------------------------------------------------
# noise record / replay and internal-tensor capture
# ----------------------------------------------------------------------------------------
@contextlib.contextmanager
def replay_noise(rand_queue=None, randn_queue=None, record=None):
    """Feed ``torch.rand_like``/``torch.randn_like`` from queues (lists of tensors, consumed in
    call order); anything not queued falls through to the real generator.  ``record`` (a dict)
    receives the list of every draw."""
    real_rand, real_randn = torch.rand_like, torch.randn_like
    rq = list(rand_queue or [])
    nq = list(randn_queue or [])

    def _mk(real, q, tag):
        def f(x, *a, **k):
            if q:
                v = q.pop(0)
                assert tuple(v.shape) == tuple(x.shape), (tag, tuple(v.shape), tuple(x.shape))
                v = v.to(x.dtype)
            else:
                v = real(x, *a, **k)
            if record is not None:
                record.setdefault(tag, []).append(v.clone())
            return v
        return f

    torch.rand_like, torch.randn_like = _mk(real_rand, rq, "rand"), _mk(real_randn, nq, "randn")
    try:
        yield
    finally:
        torch.rand_like, torch.randn_like = real_rand, real_randn


@contextlib.contextmanager
def capture_likelihoods(store):
    """Record the per-view likelihood tensor (input of the ``torch.max`` over views,
    src/models/nerf_renderer.py:129) and the cumprod input (:132) of one sampler call."""
    real_max, real_cumprod = torch.max, torch.cumprod

    def _max(x, *a, **k):
        store["pt_likelihood_views"] = x.detach().clone()
        return real_max(x, *a, **k)

    def _cumprod(x, *a, **k):
        out = real_cumprod(x, *a, **k)
        store["cumprod_in"], store["cumprod_out"] = x.detach().clone(), out.detach().clone()
        return out

    torch.max, torch.cumprod = _max, _cumprod
    try:
        yield
    finally:
        torch.max, torch.cumprod = real_max, real_cumprod


def run_reference(nerf, rays, K, NC, G, noise, white_bkgd=True, want_internals=True):
    """Run the reference renderer stage by stage with dense replayed noise
    (``noise`` = (u_coarse [NR,NC], n_gauss [NR,G], u_fill [NR,K]) as numpy).  SB must be 1.
    Returns a dict of numpy arrays (all stage outputs)."""
    ref = import_reference()
    dtype = nerf.poses.dtype
    rays_t = torch.from_numpy(rays).to(dtype)
    SB, NR, _ = rays_t.shape
    assert SB == 1
    u_coarse, n_gauss, u_fill = [torch.from_numpy(x).to(dtype) for x in noise]
    rend = ref.NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G,
                               white_bkgd=white_bkgd)
    out = {}
    with torch.no_grad():
        # pass A: which rays have any non-zero likelihood (decides the shape of the randn draw)
        with replay_noise(rand_queue=[u_coarse]):
            zA = rend.sample_depthguided(rays_t, nerf, n_samples=K, n_candidates=NC, n_gaussian=0)
        hit = (zA[0, :, 0] != 0)
        store = {}
        with replay_noise(rand_queue=[u_coarse], randn_queue=[n_gauss[hit]]), \
                capture_likelihoods(store):
            z_dg = rend.sample_depthguided(rays_t, nerf, n_samples=K, n_candidates=NC, n_gaussian=G)
        with replay_noise(rand_queue=[u_coarse]):
            z_cand = rend.sample_coarse(rays_t, n_coarse=NC)
        # fill-up: the compact draw is the row-major list of missing slots
        zs = z_dg.sort(dim=-1).values.view(-1, K)
        miss = zs == 0
        m = miss.sum(-1)
        col_rank = torch.cumsum(miss.int(), dim=-1) - 1  # rank of each missing slot in its ray
        u_compact = u_fill[torch.where(miss)[0], col_rank[miss]]
        with replay_noise(rand_queue=[u_compact]):
            z_fill = rend.fill_up_uniform_samples(z_dg.clone(), rays_t)
        cap = {}
        h = nerf.mlp_fine.register_forward_pre_hook(lambda mod, args: cap.setdefault("x", args[0].detach().clone()))
        points = rays_t[..., None, :3] + z_fill.unsqueeze(-1) * rays_t[..., None, 3:6]
        viewdirs = rays_t[..., None, 3:6].expand(-1, -1, K, -1)
        rgbsigma = nerf(points.reshape(SB, NR * K, 3), viewdirs=viewdirs.reshape(SB, NR * K, 3))
        h.remove()
        weights, rgb, depth = rend.composite(nerf, rays_t, z_fill)
    out.update(hit=hit.numpy(), z_cand=z_cand.numpy(), z_dg=z_dg.numpy(), n_missing=m.numpy(),
               z_fill=z_fill.numpy(), rgbsigma=rgbsigma.reshape(SB, NR, K, 4).numpy(),
               weights=weights.numpy(), rgb=rgb.numpy(), depth=depth.numpy())
    if want_internals:
        lv = store["pt_likelihood_views"]  # [SB,NV,1,NR*NC]
        out["pt_likelihood"] = lv.max(dim=1).values.reshape(SB, NR, NC).numpy()
        out["cumprod_out"] = store["cumprod_out"].numpy()
        out["mlp_input"] = cap["x"].numpy()  # [SB,NV,NR*K,567]
    return out
