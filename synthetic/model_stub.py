"""A minimal stand-in for the reference's ``PixelNeRF`` model object.

The renderer plug-in only *reads* state from the model it is handed:
camera buffers, the encoder's maps and the fusion-MLP parameters.  On the GPU box (where the
reference does not exist) tests, the smoke test and the benchmark need an object with the same
attribute surface; this module builds one from a synthetic scene.  Attribute names and shapes
follow reference src/models/pixelnerf.py:12-53, src/models/image_encoder.py:14-95,
src/models/resnetfc.py:72-127 and src/models/positional_encoding.py:9-31, so the real model
and this stand-in are interchangeable as the ``model`` argument of ``NeRFRendererDGS.forward``.
It owns no arithmetic: calling it raises.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn


class _Block(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.fc_0 = nn.Linear(d, d)
        self.fc_1 = nn.Linear(d, d)


class _ResnetFCState(nn.Module):
    def __init__(self, d_in=55, d_out=4, n_blocks=5, d_latent=512, d_hidden=512, combine_layer=3):
        super().__init__()
        self.lin_in = nn.Linear(d_in, d_hidden)
        self.lin_out = nn.Linear(d_hidden, d_out)
        self.blocks = nn.ModuleList([_Block(d_hidden) for _ in range(n_blocks)])
        self.lin_z = nn.ModuleList([nn.Linear(d_latent, d_hidden) for _ in range(min(combine_layer, n_blocks))])
        self.activation = nn.ReLU()
        self.n_blocks, self.d_latent, self.d_in, self.d_out, self.d_hidden = n_blocks, d_latent, d_in, d_out, d_hidden
        self.combine_layer, self.combine_type = combine_layer, "average"

    def forward(self, *a, **k):
        raise RuntimeError("model stub: the MLP is evaluated by the HIP kernels, not by PyTorch")


class _PEState(nn.Module):
    def __init__(self, num_freqs=6, d_in=3, freq_factor=6.28, include_input=True):
        super().__init__()
        self.num_freqs, self.d_in, self.include_input = num_freqs, d_in, include_input
        self.freqs = freq_factor * 2.0 ** torch.arange(0, num_freqs)
        self.d_out = num_freqs * 2 * d_in + (d_in if include_input else 0)


class _EncoderState(nn.Module):
    def __init__(self, feature_padding):
        super().__init__()
        self.index_interp, self.index_padding = "bilinear", "border"
        self.feature_padding = float(feature_padding)
        self.latent_size = 512
        self.latent = self.depths = self.depths_std = self.normals = None
        self.nviews = self.nobjects = None


class PixelNeRFState(nn.Module):
    def __init__(self, feature_padding=32, freq_factor=6.28):
        super().__init__()
        self.poscode = _PEState(d_in=3, freq_factor=freq_factor)
        self.depthcode = _PEState(d_in=1, freq_factor=freq_factor)
        self.encoder = _EncoderState(feature_padding)
        self.mlp_fine = _ResnetFCState()
        self.poses = self.focal = self.c = self.image_shape = None

    def forward(self, *a, **k):
        raise RuntimeError("model stub: points are evaluated by diner_amd.NeRFRendererDGS")


def model_from_scene(scene, weights, device="cuda", latent: torch.Tensor | None = None) -> PixelNeRFState:
    """Build the stand-in from a ``synthetic.synth.Scene`` and a ``make_mlp_weights`` dict.
    ``latent`` may be passed as a device tensor [SB,NV,C,h,w] for scenes generated on the GPU."""
    m = PixelNeRFState(feature_padding=scene.feature_padding)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    m.mlp_fine.load_state_dict(sd, strict=True)
    m = m.to(device)
    for p in m.parameters():
        p.requires_grad_(False)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    m.poses, m.focal, m.c, m.image_shape = t(scene.poses), t(scene.focal), t(scene.c), t(scene.image_shape)
    e = m.encoder
    e.depths, e.depths_std, e.normals = t(scene.depths), t(scene.depths_std), t(scene.normals)
    e.latent = latent if latent is not None else (t(scene.latent) if scene.latent is not None else None)
    e.nviews, e.nobjects = scene.NV, scene.poses.shape[0]
    return m.eval()
