"""HIP versions of the two per-image producers next to the render path,
with the reference's own function signatures so they can replace them in place:

* :func:`gen_rays`      -- reference ``src/util/cam_geometry.py:36-79``
* :func:`depth2normal`  -- reference ``src/util/depth2normal.py:7-87``
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check


def _f(t):
    return t.detach().to(torch.float32).contiguous()


def _st(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


@torch.no_grad()
def gen_rays(extrinsics, intrinsics, W, H, z_near, z_far):
    """extrinsics [B,4,4], intrinsics [B,3,3], z_near/z_far [B] -> rays [B,H,W,8]
    (origin, unit direction, near, far; pixel centres, OpenCV convention)."""
    e, k, zn, zf = _f(extrinsics), _f(intrinsics), _f(z_near).reshape(-1), _f(z_far).reshape(-1)
    if not e.is_cuda:
        raise RuntimeError("diner_amd.glue.gen_rays runs on the GPU only")
    B = e.shape[0]
    out = torch.empty((B, int(H), int(W), 8), dtype=torch.float32, device=e.device)
    check(_lib.lib().diner_gen_rays(e.data_ptr(), k.data_ptr(), zn.data_ptr(), zf.data_ptr(), B, int(H), int(W),
                                    out.data_ptr(), _st(e.device)), "diner_gen_rays")
    return out


@torch.no_grad()
def depth2normal(dmap, K):
    """dmap [N,1,H,W], K [N,3,3] -> normals [N,3,H,W]."""
    d, k = _f(dmap), _f(K)
    if not d.is_cuda:
        raise RuntimeError("diner_amd.glue.depth2normal runs on the GPU only")
    N, _, H, W = d.shape
    out = torch.empty((N, 3, H, W), dtype=torch.float32, device=d.device)
    check(_lib.lib().diner_depth2normal(d.data_ptr(), k.data_ptr(), N, H, W, out.data_ptr(), _st(d.device)),
          "diner_depth2normal")
    return out


@torch.no_grad()
def pack_maps_from_depth(depths, depths_std, intrinsics):
    """depth2normal fused into the renderer's map packing: depths, depths_std [SB,NV,1,H,W], intrinsics
    [SB,NV,3,3] -> packed maps [SB,NV,H,W,8] (nx ny nz depth | sigma 0 0 0), the layout ``DinerScene.maps`` takes."""
    d, s, k = _f(depths), _f(depths_std), _f(intrinsics)
    SB, NV, _, H, W = d.shape
    out = torch.empty((SB, NV, H, W, 8), dtype=torch.float32, device=d.device)
    check(_lib.lib().diner_pack_maps_from_depth(d.data_ptr(), s.data_ptr(), k.data_ptr(), SB * NV, H, W, out.data_ptr(),
                                                _st(d.device)), "diner_pack_maps_from_depth")
    return out
