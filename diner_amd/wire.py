"""Upstream wire format: TransMVSNet's uint16 depth / confidence planes -> the ``depths`` /
``depths_std`` tensors ``PixelNeRF.encode`` takes, decoded on the GPU (``diner_decode_depth_u16``).

Mirrors the arithmetic of the reference's dataset readers (PNG decoding itself stays with PIL on the host):

* DTU        ``src/data/dtu.py:100-119`` (``read_depth``) and ``:68-70,220-223`` (``conf2std``)
* Facescape  ``src/data/facescape.py:80-104`` (``read_depth``, depth types) and ``:54-56,266``

Parity: the reference holds no fixture for these readers and its ``pil_to_tensor`` dependency is absent here, so the
numpy restatement ``oracle/wire_oracle.py`` is checked against hand-computed values only ("parity unpinned").
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check

SCALE_FACTOR = 1e-4                       # dtu.py:103, facescape.py:82
DTU_TRAIN_SCALE = 0.7 / 872.0             # dtu.py:105 (python double, applied as a float32 scalar)
DTU_CONF2STD = (-2.5679e-2, 3.2818e-2)    # dtu.py:68-70
FACESCAPE_CONF2STD = (-1.582e-2, 1.649e-2)  # facescape.py:54-56


def _u16(t, dev):
    if isinstance(t, np.ndarray):
        assert t.dtype == np.uint16
        t = torch.from_numpy(t.view(np.int16))   # same bits; torch kernels never touch the values
    assert t.dtype in (torch.int16, torch.uint16)
    return t.to(dev).contiguous()


@torch.no_grad()
def decode_depth_u16(depth, conf, *, div, mul1, conf2std, stride=1, mesh=None, device="cuda:0", want_mask=False):
    """depth, conf (, mesh): uint16 planes [N,H,W] (numpy uint16 or torch int16/uint16 holding the same bits) ->
    depth [N,1,h,w], depth_std [N,1,h,w] (, mask [N,1,h,w]) float32 on ``device``."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("diner_amd.wire runs on the GPU only")
    d, c = _u16(depth, dev), _u16(conf, dev)
    m = None if mesh is None else _u16(mesh, dev)
    N, H, W = d.shape
    assert tuple(c.shape) == (N, H, W) and (m is None or tuple(m.shape) == (N, H, W))
    h, w = H // stride, W // stride
    out_d = torch.empty((N, 1, h, w), dtype=torch.float32, device=dev)
    out_s = torch.empty_like(out_d)
    out_m = torch.empty_like(out_d) if want_mask else None
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(_lib.lib().diner_decode_depth_u16(d.data_ptr(), c.data_ptr(), None if m is None else m.data_ptr(), N, H, W, int(stride),
                                            np.float32(SCALE_FACTOR), np.float32(div), np.float32(mul1), np.float32(conf2std[0]),
                                            np.float32(conf2std[1]), out_d.data_ptr(), out_s.data_ptr(),
                                            None if out_m is None else out_m.data_ptr(), st), "diner_decode_depth_u16")
    return (out_d, out_s, out_m) if want_mask else (out_d, out_s)


def decode_dtu(depth, conf, scale_factor, downsample=1.0, **kw):
    """``DTUDataSet.read_depth`` on the prediction PNGs + ``conf2std`` (dtu.py:100-119,220-223); ``downsample`` must be 1/k."""
    stride = int(round(1.0 / downsample))
    assert abs(stride * downsample - 1.0) < 1e-9, "nearest downsample is exact for 1/k only"
    return decode_depth_u16(depth, conf, div=DTU_TRAIN_SCALE, mul1=scale_factor, conf2std=DTU_CONF2STD, stride=stride, **kw)


def decode_facescape(depth, conf, mesh=None, **kw):
    """``FacescapeDataSet.read_depth`` (depth_type "original", or "merge" when ``mesh`` is given) + ``conf2std``
    (facescape.py:80-104,266)."""
    return decode_depth_u16(depth, conf, div=1.0, mul1=1.0, conf2std=FACESCAPE_CONF2STD, mesh=mesh, **kw)
