"""Seeded synthetic scenes for the DINER render path (numpy only, no torch, no reference).

There are no datasets or checkpoints offline, so parity tests, the smoke test and the
benchmark all run on the synthetic scene SURVEY.md §8(d) specifies: NV source cameras on
a circle around a sphere, analytic z-depth maps, random latent feature maps and
kaiming-initialised fusion-MLP weights.  Everything here is a pure function of its
seeds (legacy ``numpy.random.RandomState`` streams, which numpy keeps frozen), so the
golden fixtures under ``tests/golden`` store only seeds + small tensors and are rebuilt
bit-identically on the GPU box.

Tensor layouts are the reference's (so they can be assigned straight onto the
reference's ``PixelNeRF``/``SpatialEncoder`` attributes):

* ``poses``       [SB,NV,4,4]  world->camera extrinsics  (src/models/pixelnerf.py:47)
* ``focal``/``c`` [SB,NV,2]    (fx,fy)/(cx,cy)           (src/models/pixelnerf.py:48-49)
* ``image_shape`` [2]          (W,H)                     (src/models/pixelnerf.py:50-51)
* ``depths``/``depths_std`` [SB,NV,1,H,W], ``normals`` [SB,NV,3,H,W]
                                                          (src/models/image_encoder.py:214-216)
* ``latent``      [SB,NV,C,h,w]                          (src/models/image_encoder.py:271-272)
* ``rays``        [SB,NR,8] = origin, unit direction, near, far
                                                          (src/util/cam_geometry.py:36-79)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

F32 = np.float32

# conf2std(0): the sigma_depth a dataset assigns to a zero-confidence (background) pixel
# (reference src/data/facescape.py:54-56, src/data/dtu.py:68-70)
BG_SIGMA = {"facescape": F32(1.649e-2), "dtu": F32(3.2818e-2)}
NEAR_FAR = {"facescape": (F32(1.0), F32(2.5)), "dtu": (F32(0.321), F32(1.204))}


# ----------------------------------------------------------------------------------------
# cameras
# ----------------------------------------------------------------------------------------
def look_at_origin_w2c(yaw: float, radius: float) -> np.ndarray:
    """World->camera 4x4 (OpenCV convention: +x right, +y down, +z forward) of a camera on
    the circle of ``radius`` in the world xz-plane, rotated by ``yaw`` and looking at 0."""
    centre = np.array([np.sin(yaw), 0.0, -np.cos(yaw)], dtype=np.float64) * radius
    fwd = -centre / radius
    down = np.array([0.0, 1.0, 0.0])
    right = np.cross(down, fwd)
    r_c2w = np.stack([right, down, fwd], axis=1)
    r_w2c = r_c2w.T
    ext = np.eye(4, dtype=np.float64)
    ext[:3, :3] = r_w2c
    ext[:3, 3] = -r_w2c @ centre
    return ext.astype(F32)


def intrinsics(W: int, H: int) -> np.ndarray:
    k = np.eye(3, dtype=F32)
    k[0, 0] = k[1, 1] = F32(1.2 * W)
    k[0, 2] = F32(W / 2)
    k[1, 2] = F32(H / 2)
    return k


def gen_rays(extrinsics: np.ndarray, K: np.ndarray, W: int, H: int, near, far) -> np.ndarray:
    """Pixel-centre camera rays [H,W,8]; numpy restatement of the producer of ``rays``
    (reference src/util/cam_geometry.py:36-79).  Not on the hot path: its output is an
    *input* of the render path, so it only has to be a valid ray set."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ys, xs = np.meshgrid(np.arange(0.5, H, 1, dtype=F32), np.arange(0.5, W, 1, dtype=F32),
                         indexing="ij")
    d = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], axis=-1).astype(F32)
    d = d / np.sqrt((d * d).sum(-1, keepdims=True))
    r_c2w = extrinsics[:3, :3].T
    d_w = (d.reshape(-1, 3) @ r_c2w.T).reshape(H, W, 3).astype(F32)
    o = (-r_c2w @ extrinsics[:3, 3]).astype(F32)
    rays = np.empty((H, W, 8), dtype=F32)
    rays[..., :3] = o
    rays[..., 3:6] = d_w
    rays[..., 6] = near
    rays[..., 7] = far
    return rays


# ----------------------------------------------------------------------------------------
# depth / normal maps
# ----------------------------------------------------------------------------------------
def sphere_zdepth(extrinsics: np.ndarray, K: np.ndarray, W: int, H: int, radius: float,
                  centre=(0.0, 0.0, 0.0)) -> np.ndarray:
    """z-depth (camera-space z of the first hit) of a sphere seen through a pinhole
    camera, 0 on background -- the quantity the sampler compares with ``xyz_cam.z``
    (reference src/models/nerf_renderer.py:114,122)."""
    fx, fy, cx, cy = [np.float64(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2])]
    ys, xs = np.meshgrid(np.arange(0.5, H, 1), np.arange(0.5, W, 1), indexing="ij")
    d_c = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], axis=-1)  # z = 1
    ext = extrinsics.astype(np.float64)
    r_c2w = ext[:3, :3].T
    o = -r_c2w @ ext[:3, 3] - np.asarray(centre, dtype=np.float64)
    d_w = d_c @ r_c2w.T
    a = (d_w * d_w).sum(-1)
    b = 2.0 * (d_w * o).sum(-1)
    c = (o * o).sum() - radius * radius
    disc = b * b - 4 * a * c
    hit = disc > 0
    s = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0.0))) / (2 * a), 0.0)
    s = np.where(s > 0, s, 0.0)
    return s.astype(F32)


def depth2normal(dmap: np.ndarray, K: np.ndarray) -> np.ndarray:
    """Normal maps [N,3,H,W] from z-depth maps [N,1,H,W] by central differences plus the
    reference's hole clean-up; numpy restatement of reference src/util/depth2normal.py:7-87
    (a per-image step *before* the hot path)."""
    N, _, H, W = dmap.shape
    out = np.zeros((N, 3, H, W), dtype=F32)
    ys, xs = np.meshgrid(np.arange(0.5, H, 1, dtype=F32), np.arange(0.5, W, 1, dtype=F32),
                         indexing="ij")
    for n in range(N):
        k = K[n]
        rx = (xs - k[0, 2]) / k[0, 0]
        ry = (ys - k[1, 2]) / k[1, 1]
        rays = np.stack([rx, ry, np.ones_like(rx)], axis=-1).astype(F32)
        pts = rays * dmap[n, 0][..., None]
        pts = np.pad(pts, ((1, 1), (1, 1), (0, 0)), mode="edge")
        down, up = pts[2:, 1:-1], pts[:-2, 1:-1]
        right, left = pts[1:-1, 2:], pts[1:-1, :-2]
        vdiff = down - up
        hdiff = right - left
        nrm = np.cross(vdiff, hdiff).astype(F32)
        with np.errstate(invalid="ignore", divide="ignore"):
            nrm = nrm / np.sqrt((nrm * nrm).sum(-1, keepdims=True))
        off_y = np.zeros((H, W), dtype=np.int64)
        off_x = np.zeros((H, W), dtype=np.int64)
        off_y += -1 * (down[..., 0] == 0)
        off_y += 1 * (up[..., 0] == 0)
        off_x += -1 * (right[..., 0] == 0)
        off_x += 1 * (left[..., 0] == 0)
        mask = (off_y != 0) | (off_x != 0)
        iy, ix = np.nonzero(mask)
        ny = np.clip(iy + off_y[iy, ix], 0, H - 1)
        nx = np.clip(ix + off_x[iy, ix], 0, W - 1)
        src = nrm[ny, nx].copy()
        nrm[iy, ix] = src
        nrm[dmap[n, 0] == 0] = 0
        out[n] = nrm.transpose(2, 0, 1)
    return out


# ----------------------------------------------------------------------------------------
# fusion-MLP weights (ResnetFC d_in=55, d_latent=512, d_hidden=512, 5 blocks, combine 3)
# ----------------------------------------------------------------------------------------
def mlp_param_shapes(d_in=55, d_latent=512, d_hidden=512, d_out=4, n_blocks=5,
                     combine_layer=3) -> Dict[str, tuple]:
    """state_dict keys/shapes of the reference fusion MLP (src/models/resnetfc.py:72-127)."""
    shapes = {"lin_in.weight": (d_hidden, d_in), "lin_in.bias": (d_hidden,),
              "lin_out.weight": (d_out, d_hidden), "lin_out.bias": (d_out,)}
    for b in range(n_blocks):
        shapes[f"blocks.{b}.fc_0.weight"] = (d_hidden, d_hidden)
        shapes[f"blocks.{b}.fc_0.bias"] = (d_hidden,)
        shapes[f"blocks.{b}.fc_1.weight"] = (d_hidden, d_hidden)
        shapes[f"blocks.{b}.fc_1.bias"] = (d_hidden,)
    for b in range(min(combine_layer, n_blocks)):
        shapes[f"lin_z.{b}.weight"] = (d_hidden, d_latent)
        shapes[f"lin_z.{b}.bias"] = (d_hidden,)
    return shapes


def make_mlp_weights(seed: int = 1, bias_scale: float = 0.0, **dims) -> Dict[str, np.ndarray]:
    """Kaiming-normal (fan-in, gain sqrt 2) weights for every Linear *including* ``fc_1``
    (which the reference zero-initialises, src/models/resnetfc.py:47 -- a zero ``fc_1``
    would make every block an identity and hide errors).  ``bias_scale``>0 draws non-zero
    biases so the bias path is exercised too."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in sorted(mlp_param_shapes(**dims).items()):
        if name.endswith("weight"):
            std = np.sqrt(2.0 / shape[1])
            out[name] = (rs.standard_normal(shape) * std).astype(F32)
        else:
            out[name] = (rs.standard_normal(shape) * bias_scale).astype(F32)
    return out


# ----------------------------------------------------------------------------------------
# whole scene
# ----------------------------------------------------------------------------------------
@dataclass
class Scene:
    H: int
    W: int
    NV: int
    C: int
    feature_padding: int
    near: float
    far: float
    white_bkgd: bool
    poses: np.ndarray          # [1,NV,4,4]
    focal: np.ndarray          # [1,NV,2]
    c: np.ndarray              # [1,NV,2]
    image_shape: np.ndarray    # [2] = (W,H)
    depths: np.ndarray         # [1,NV,1,H,W]
    depths_std: np.ndarray     # [1,NV,1,H,W]
    normals: np.ndarray        # [1,NV,3,H,W]
    latent: Optional[np.ndarray]  # [1,NV,C,h,w] or None (generate on device instead)
    target_extrinsics: np.ndarray  # [4,4]
    target_intrinsics: np.ndarray  # [3,3]
    meta: dict = field(default_factory=dict)

    @property
    def latent_hw(self):
        return (self.H + 4 * self.feature_padding) // 2, (self.W + 4 * self.feature_padding) // 2

    def target_rays(self, Ht: Optional[int] = None, Wt: Optional[int] = None,
                    crop: Optional[tuple] = None, focal_scale: float = 1.0) -> np.ndarray:
        """rays [1,NR,8] of the target view (full image, or a (y0,x0,h,w) crop).
        ``focal_scale`` < 1 widens the field of view so that rays leave the source images
        (exercises the border / zero / exponential paddings of the map look-ups)."""
        Ht = Ht or self.H
        Wt = Wt or self.W
        k = self.target_intrinsics.copy()
        if (Ht, Wt) != (self.H, self.W):
            k = intrinsics(Wt, Ht)
        k[0, 0] *= F32(focal_scale)
        k[1, 1] *= F32(focal_scale)
        rays = gen_rays(self.target_extrinsics, k, Wt, Ht, self.near, self.far)
        if crop is not None:
            y0, x0, h, w = crop
            rays = rays[y0:y0 + h, x0:x0 + w]
        return np.ascontiguousarray(rays.reshape(1, -1, 8))


def make_scene(H: int, W: int, NV: int, *, seed: int = 0, dataset: str = "facescape",
               bg_sigma_zero: bool = False, C: int = 512, feature_padding: int = 32,
               with_latent: bool = True, sphere_radius: float = 0.45,
               target_yaw: float = 0.1, latent_scale: float = 1.0) -> Scene:
    """The synthetic scene of SURVEY.md §8(d).  ``feature_padding`` is in latent texels
    (reference default 32 = image_padding 64 / conv1 stride 2, src/models/image_encoder.py:57-58);
    the latent map is (H/2 + 2*fp) x (W/2 + 2*fp)."""
    near, far = NEAR_FAR[dataset]
    # camera circle: 1.75 for the Facescape near/far (1.0/2.5); same ratio for DTU
    cam_radius = 1.75 * float(near + far) / 3.5
    sph_radius = sphere_radius * float(near + far) / 3.5
    yaws = np.linspace(-0.5, 0.5, NV) if NV > 1 else np.array([0.0])
    poses = np.stack([look_at_origin_w2c(y, cam_radius) for y in yaws])[None]  # [1,NV,4,4]
    k = intrinsics(W, H)
    focal = np.tile(np.array([k[0, 0], k[1, 1]], dtype=F32), (1, NV, 1))
    c = np.tile(np.array([k[0, 2], k[1, 2]], dtype=F32), (1, NV, 1))
    depths = np.stack([sphere_zdepth(poses[0, v], k, W, H, sph_radius) for v in range(NV)])
    depths = depths[None, :, None]  # [1,NV,1,H,W]
    rs = np.random.RandomState(seed)
    std = (0.004 + 0.004 * rs.random_sample(depths.shape)).astype(F32)
    bg = F32(0.0) if bg_sigma_zero else BG_SIGMA[dataset]
    depths_std = np.where(depths > 0, std, bg).astype(F32)
    normals = depth2normal(depths[0], np.tile(k[None], (NV, 1, 1)))[None]
    h, w = H // 2 + 2 * feature_padding, W // 2 + 2 * feature_padding
    latent = None
    if with_latent:
        latent = (np.random.RandomState(seed + 1000).standard_normal((1, NV, C, h, w))
                  * latent_scale).astype(F32)
    return Scene(H=H, W=W, NV=NV, C=C, feature_padding=feature_padding, near=float(near),
                 far=float(far), white_bkgd=(dataset == "facescape"), poses=poses.astype(F32),
                 focal=focal, c=c, image_shape=np.array([W, H], dtype=F32), depths=depths,
                 depths_std=depths_std, normals=normals, latent=latent,
                 target_extrinsics=look_at_origin_w2c(target_yaw, cam_radius),
                 target_intrinsics=k,
                 meta=dict(seed=seed, dataset=dataset, bg_sigma_zero=bg_sigma_zero,
                           cam_radius=cam_radius, sphere_radius=sph_radius))


def make_noise(NR: int, NC: int, G: int, K: int, seed: int = 2):
    """Dense noise tensors for parity mode: u_coarse [NR,NC] ~U[0,1), n_gauss [NR,G] ~N(0,1),
    u_fill [NR,K] ~U[0,1) (row r, column j = the j-th missing slot of ray r)."""
    rs = np.random.RandomState(seed)
    return (rs.random_sample((NR, NC)).astype(F32), rs.standard_normal((NR, G)).astype(F32),
            rs.random_sample((NR, K)).astype(F32))
