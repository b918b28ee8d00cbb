"""TEST INFRASTRUCTURE (never imported by the product path): numpy float32 restatement of the reference's depth /
confidence readers for the TransMVSNet uint16 planes (one float32 rounding per tensor operation, python scalars applied as
float32).  Pinned bit-exactly by ``tests/golden/wire.npz``: outputs of the reference's own ``read_depth`` / ``conf2std`` on
uint16 PNGs written with PIL (``oracle/gen_golden.py --wire-only``; ``pil_to_tensor`` stubbed as a pure dtype/layout
conversion, ``oracle/ref_harness.py``) -- tests/test_wire.py.

* ``dtu_read_depth``        src/data/dtu.py:100-119   (PNG branch; NEAREST resize for downsample = 1/k)
* ``dtu_conf2std``          src/data/dtu.py:68-70,220-223
* ``facescape_read_depth``  src/data/facescape.py:80-104
* ``facescape_conf2std``    src/data/facescape.py:54-56,266
"""
import numpy as np

F = np.float32


def dtu_read_depth(u16, scale_factor, stride=1):
    d = u16.astype(F) * F(1e-4)                 # :104  pil_to_tensor(..).float() * SCALE_FACTOR
    d = d / F(0.7 / 872.0)                      # :105
    d = d[..., ::stride, ::stride]              # :113-117 NEAREST, out = in * downsample
    mask = (d > 0).astype(F)                    # :118
    d = d * F(scale_factor)                     # :119
    return d, mask


def dtu_conf2std(x):
    return F(-2.5679e-2) * x + F(3.2818e-2)     # :68-70


def facescape_read_depth(depth_u16, conf_u16, mesh_u16=None, depth_type=None):
    """depth_type: 'original' (default without a mesh plane), 'merge' (default with one) or 'mesh'"""
    pred_mvs = depth_u16.astype(F) * F(1e-4)    # :90
    conf_mvs = conf_u16.astype(F) * F(1e-4)     # :91
    if mesh_u16 is None:                        # depth_type == 'original' :93-94
        return pred_mvs, conf_mvs
    pred = mesh_u16.astype(F) * F(1e-4)         # :82
    conf = np.where(pred == 0, F(0.0), F(0.8))  # :83-85
    if depth_type == "mesh":                    # :95-96
        return pred, conf
    pred = np.where((pred == 0) & (pred_mvs != 0), pred_mvs, pred)   # :98-100
    conf = np.where((conf == 0) & (conf_mvs != 0), conf_mvs, conf)   # :101-103
    return pred, conf


def facescape_conf2std(x):
    return F(-1.582e-2) * x + F(1.649e-2)       # :54-56
