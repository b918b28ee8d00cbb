"""Differentiable render path for training.

The reference trains through ``NeRFRendererDGS.composite`` and ``PixelNeRF.forward`` with PyTorch autograd
(``DINER.calc_losses``, reference src/models/diner.py:217-290; the sampler is ``@torch.no_grad``,
src/models/nerf_renderer.py:65).  Here the same graph is evaluated by the HIP building blocks of
``diner_amd/csrc/train.hip`` (one exact fp32-MFMA GEMM kernel + small per-point kernels, all through the C ABI);
this module only orchestrates them the way autograd orchestrates ATen ops: a forward that keeps every layer's
input, and a hand-written backward producing gradients for the fusion-MLP parameters and ``encoder.latent``.
PyTorch supplies buffers, clones and the autograd hook -- no arithmetic of the path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check

HID = 512
K_CHUNK = 4096  # rows per split of the weight-gradient GEMMs (multiple of 16)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# f16x3 mode: static power-of-two scales of the fp16 hi/lo split (include/diner_hip.h, diner_train_gemm):
# activations 2^-4 (what the inference kernel carries, fp16 overflow at |x| >= 1e6), weights 2^4;
# gradients are scaled by their measured maximum (diner_train_amax), their magnitude being arbitrary.
EXP_ACT, EXP_W = -4, 4


def _gemm(A, B, bias, S, Cm, M, N, K, sam, sak, sbk, sbn, ldc, lds, relu_a=0, relu_b=0, accumulate=0, atomic=0, k_chunk=0,
          prec=0, amax_a=None, amax_b=None, exp_a=0, exp_b=0):
    check(_lib.lib().diner_train_gemm(_p(A), _p(B), _p(bias), _p(S), _p(Cm), M, N, K, sam, sak, sbk, sbn, ldc, lds, relu_a, relu_b,
                                      accumulate, atomic, k_chunk, prec, _p(amax_a), _p(amax_b), exp_a, exp_b, _st(Cm.device)),
          "diner_train_gemm")


def amax_of(t, prec):
    """Device word holding max|t| for the f16x3 GEMMs that take ``t`` as a gradient operand (None in fp32 mode)."""
    if prec == 0:
        return None
    out = torch.empty(1, dtype=torch.int32, device=t.device)
    assert t.is_contiguous()
    check(_lib.lib().diner_train_amax(_p(t), t.numel(), _p(out), _st(t.device)), "diner_train_amax")
    return out


def linear_fwd(X, W, b, out, relu_in=False, accumulate=False, prec=0):
    """out[M,N] (+)= relu?(X[M,K]) W[N,K]^T + b."""
    M, K = X.shape
    N = W.shape[0]
    _gemm(X, W, b, None, out, M, N, K, X.stride(0), 1, 1, W.stride(0), out.stride(0), 0, relu_a=int(relu_in), accumulate=int(accumulate),
          prec=prec, exp_a=EXP_ACT, exp_b=EXP_W)


def linear_bwd_x(dY, W, mask_src, out, accumulate=False, prec=0, amax=None):
    """out[M,K] (+)= (dY[M,N] W[N,K]) * [mask_src > 0]."""
    M, N = dY.shape
    K = W.shape[1]
    _gemm(dY, W, None, mask_src, out, M, K, N, dY.stride(0), 1, W.stride(0), 1, out.stride(0),
          0 if mask_src is None else mask_src.stride(0), accumulate=int(accumulate), prec=prec, amax_a=amax, exp_b=EXP_W)


def linear_bwd_w(dY, X, dW, db, relu_x=False, prec=0, amax=None):
    """dW[N,K] += dY[M,N]^T relu?(X[M,K]);  db[N] += sum_m dY."""
    M, N = dY.shape
    K = X.shape[1]
    _gemm(dY, X, None, None, dW, N, K, M, 1, dY.stride(0), X.stride(0), 1, dW.stride(0), 0, relu_b=int(relu_x), atomic=1, k_chunk=K_CHUNK,
          prec=prec, amax_a=amax, exp_b=EXP_ACT)
    if db is not None:
        check(_lib.lib().diner_train_colsum(_p(dY), M, N, dY.stride(0), _p(db), _st(dY.device)), "diner_train_colsum")


def _mlp_params(mlp):
    ps = [mlp.lin_in.weight, mlp.lin_in.bias]
    for b in range(3):
        ps += [mlp.lin_z[b].weight, mlp.lin_z[b].bias]
    for b in range(5):
        ps += [mlp.blocks[b].fc_0.weight, mlp.blocks[b].fc_0.bias, mlp.blocks[b].fc_1.weight, mlp.blocks[b].fc_1.bias]
    ps += [mlp.lin_out.weight, mlp.lin_out.bias]
    return ps


class _RenderFn(torch.autograd.Function):
    """(latent, *mlp_params) -> (rgb, depth, weights) for fixed rays / samples."""

    @staticmethod
    def forward(ctx, renderer, scene, rays, z, latent, *params):
        L = _lib.lib()
        dev = rays.device
        st = _st(dev)
        SB, NR, K = z.shape
        NV, P = scene.NV, NR * K
        R = NV * P
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        lat = latent.detach().to(torch.float32).contiguous()
        prm = [p.detach().to(torch.float32).contiguous() for p in params]
        w_in56 = torch.zeros((HID, 56), dtype=torch.float32, device=dev)
        w_in56[:, :55] = prm[0]
        rgbsigma = f(SB, NR, K, 4)
        prec = _lib.PRECISIONS[renderer.precision]
        saved = []
        for sb in range(SB):
            in56, zl, taps = f(R, 56), f(R, HID), f(R, 8)
            check(L.diner_train_point_inputs(C.byref(scene), _p(lat), _p(rays), _p(z), NR, K, sb, _p(in56), _p(zl), _p(taps), st),
                  "diner_train_point_inputs")
            x = f(R, HID)
            linear_fwd(in56, w_in56, prm[1], x, prec=prec)                                  # resnetfc.py:139
            xs, nets = [], []
            for b in range(3):
                xb = x.clone()
                linear_fwd(zl, prm[2 + 2 * b], prm[3 + 2 * b], xb, accumulate=True, prec=prec)      # :152-153
                net = f(R, HID)
                linear_fwd(xb, prm[8 + 4 * b], prm[9 + 4 * b], net, relu_in=True, prec=prec)        # :62
                x = xb.clone()
                linear_fwd(net, prm[10 + 4 * b], prm[11 + 4 * b], x, relu_in=True, accumulate=True, prec=prec)  # :63,:69
                xs.append(xb)
                nets.append(net)
            xbar = f(P, HID)
            check(L.diner_train_view_mean(_p(x), P * HID, NV, _p(xbar), 0, st), "diner_train_view_mean")   # :146-149
            xbars, pnets = [], []
            for b in range(3, 5):
                net = f(P, HID)
                linear_fwd(xbar, prm[8 + 4 * b], prm[9 + 4 * b], net, relu_in=True, prec=prec)
                nxt = xbar.clone()
                linear_fwd(net, prm[10 + 4 * b], prm[11 + 4 * b], nxt, relu_in=True, accumulate=True, prec=prec)
                xbars.append(xbar)
                pnets.append(net)
                xbar = nxt
            out = f(P, 4)
            linear_fwd(xbar, prm[28], prm[29], out, relu_in=True, prec=prec)                # :158
            check(L.diner_train_head(_p(out), None, None, P * 4, _p(rgbsigma[sb]), 0, st), "diner_train_head")  # pixelnerf.py:139-143
            saved.append((in56, zl, taps, xs, nets, xbars, pnets, xbar, out))
        N = SB * NR
        rgb, depth, weights = f(SB, NR, 3), f(SB, NR), f(SB, NR, K)
        check(L.diner_composite(_p(rays), _p(z), _p(rgbsigma), N, K, int(bool(renderer.white_bkgd)), _p(rgb), _p(depth), _p(weights), st),
              "diner_composite")
        ctx.renderer, ctx.scene, ctx.rays, ctx.z, ctx.rgbsigma = renderer, scene, rays, z, rgbsigma
        ctx.saved_acts, ctx.prm, ctx.w_in56, ctx.lat_shape = saved, prm, w_in56, tuple(latent.shape)
        ctx.keep = (lat,)
        ctx.prec = prec
        return rgb, depth, weights

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_weights):
        L = _lib.lib()
        scene, rays, z, rgbsigma, prm = ctx.scene, ctx.rays, ctx.z, ctx.rgbsigma, ctx.prm
        dev = rays.device
        st = _st(dev)
        prec = ctx.prec
        SB, NR, K = z.shape
        NV, P = scene.NV, NR * K
        R = NV * P
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        d_rgb, d_depth, d_weights = c(d_rgb), c(d_depth), c(d_weights)
        if d_rgb is None:
            d_rgb = torch.zeros((SB, NR, 3), dtype=torch.float32, device=dev)
        d_rgbsigma = f(SB, NR, K, 4)
        check(L.diner_composite_backward(_p(rays), _p(z), _p(rgbsigma), _p(d_rgb), _p(d_depth), _p(d_weights), SB * NR, K,
                                         int(bool(ctx.renderer.white_bkgd)), _p(d_rgbsigma), st), "diner_composite_backward")
        g = [torch.zeros_like(p) for p in prm]          # parameter gradients (fp32, accumulated atomically)
        g_in56 = torch.zeros_like(ctx.w_in56)
        SBl, NVl, Cl, hl, wl = ctx.lat_shape
        d_lat_nhwc = torch.zeros((SBl, NVl, hl, wl, Cl), dtype=torch.float32, device=dev)
        for sb in range(SB):
            in56, zl, taps, xs, nets, xbars, pnets, xbar5, out = ctx.saved_acts[sb]
            d_out = f(P, 4)
            check(L.diner_train_head(_p(out), _p(rgbsigma[sb]), _p(d_rgbsigma[sb]), P * 4, _p(d_out), 1, st), "diner_train_head(bwd)")
            a_out = amax_of(d_out, prec)
            linear_bwd_w(d_out, xbar5, g[28], g[29], relu_x=True, prec=prec, amax=a_out)
            d_x = f(P, HID)
            linear_bwd_x(d_out, prm[28], xbar5, d_x, prec=prec, amax=a_out)
            for i, b in ((1, 4), (0, 3)):                                         # post-mean blocks, reversed
                d_net = f(P, HID)
                a_x = amax_of(d_x, prec)
                linear_bwd_x(d_x, prm[10 + 4 * b], pnets[i], d_net, prec=prec, amax=a_x)
                linear_bwd_w(d_x, pnets[i], g[10 + 4 * b], g[11 + 4 * b], relu_x=True, prec=prec, amax=a_x)
                d_prev = d_x.clone()
                a_net = amax_of(d_net, prec)
                linear_bwd_x(d_net, prm[8 + 4 * b], xbars[i], d_prev, accumulate=True, prec=prec, amax=a_net)
                linear_bwd_w(d_net, xbars[i], g[8 + 4 * b], g[9 + 4 * b], relu_x=True, prec=prec, amax=a_net)
                d_x = d_prev
            d_xv = f(R, HID)
            check(L.diner_train_view_mean(_p(d_x), P * HID, NV, _p(d_xv), 1, st), "diner_train_view_mean(bwd)")
            d_zl = torch.zeros((R, HID), dtype=torch.float32, device=dev)
            a_xv = amax_of(d_xv, prec)
            for b in (2, 1, 0):                                                   # per-view blocks, reversed
                d_net = f(R, HID)
                linear_bwd_x(d_xv, prm[10 + 4 * b], nets[b], d_net, prec=prec, amax=a_xv)
                linear_bwd_w(d_xv, nets[b], g[10 + 4 * b], g[11 + 4 * b], relu_x=True, prec=prec, amax=a_xv)
                d_xs = d_xv.clone()
                a_net = amax_of(d_net, prec)
                linear_bwd_x(d_net, prm[8 + 4 * b], xs[b], d_xs, accumulate=True, prec=prec, amax=a_net)
                linear_bwd_w(d_net, xs[b], g[8 + 4 * b], g[9 + 4 * b], relu_x=True, prec=prec, amax=a_net)
                a_xv = amax_of(d_xs, prec)
                linear_bwd_w(d_xs, zl, g[2 + 2 * b], g[3 + 2 * b], prec=prec, amax=a_xv)  # lin_z[b]
                linear_bwd_x(d_xs, prm[2 + 2 * b], None, d_zl, accumulate=True, prec=prec, amax=a_xv)
                d_xv = d_xs
            linear_bwd_w(d_xv, in56, g_in56, g[1], prec=prec, amax=a_xv)           # lin_in
            check(L.diner_train_bilinear_scatter(_p(d_zl), _p(taps), P, HID, scene.h, scene.w, NV, sb, _p(d_lat_nhwc), st),
                  "diner_train_bilinear_scatter")
        d_lat = torch.empty(ctx.lat_shape, dtype=torch.float32, device=dev)
        check(L.diner_train_nhwc_to_nchw(_p(d_lat_nhwc), SBl * NVl, Cl, hl, wl, _p(d_lat), st), "diner_train_nhwc_to_nchw")
        g[0] = g_in56[:, :55].contiguous()
        return (None, None, None, None, d_lat) + tuple(g)


def render_with_grad(renderer, model, rays, z, scene):
    """rgb, depth, weights = composite(model, rays, z) with gradients to the MLP parameters and encoder.latent."""
    params = _mlp_params(model.mlp_fine)
    return _RenderFn.apply(renderer, scene, rays, z, model.encoder.latent, *params)
