"""diner_amd -- MI355X-native render path of DINER (depth-guided sampler, multi-view fusion MLP,
volume compositing) as hand-written HIP kernels behind the reference's renderer plug-in API.

``renderer.module: diner_amd.NeRFRendererDGS`` in a reference YAML config selects it.
Importing :class:`NeRFRendererDGS` requires the built HIP library (no fallback).
"""
from .renderer import NeRFRendererDGS, RenderOutput  # noqa: F401

__all__ = ["NeRFRendererDGS", "RenderOutput"]
