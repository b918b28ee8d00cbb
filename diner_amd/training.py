"""Differentiable render path for training.

The reference trains through ``NeRFRendererDGS.composite`` and ``PixelNeRF.forward`` with PyTorch autograd
(``DINER.calc_losses``, reference src/models/diner.py:217-290; the sampler is ``@torch.no_grad``,
src/models/nerf_renderer.py:65).  Here the same graph is evaluated by the HIP building blocks of
``diner_amd/csrc/train.hip`` (MFMA GEMM kernels in the renderer's precision + small per-point kernels, all through
the C ABI); this module only orchestrates them the way autograd orchestrates ATen ops: a forward that keeps every
layer's input, and a hand-written backward producing gradients for the fusion-MLP parameters and
``encoder.latent``.  PyTorch supplies buffers and the autograd hook -- no arithmetic of the path.
"""
from __future__ import annotations

import ctypes as C
import weakref

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


USE_CORE = True   # 512 x 512 layers: the inference kernel's assembly GEMM core (train_core.hip) instead of the panel kernel


class CorePack:
    """A 512 x 512 weight in the stream layout of the GEMM core (``diner_train_pack_core``)."""
    __slots__ = ("data",)

    def __init__(self, data):
        self.data = data


def split_panel(W, transpose, prec):
    """Weight operand of the pre-split-weight GEMMs (f16x3 mode): B[n][k] = W[n][k] (forward) or W[k][n] (transpose: the dX
    GEMM), 512 rows.  K = 512: a ``CorePack`` (GEMM core of the inference kernel); other K: fp16 hi/lo planes of the panel
    kernel; None when neither applies (fp32 mode, not 512 columns)."""
    n_rows, K = (W.shape[1], W.shape[0]) if transpose else (W.shape[0], W.shape[1])
    if prec == 0 or n_rows != HID:
        return None
    if USE_CORE and K == HID and W.stride(1) == 1:
        data = torch.empty(2 * HID * HID, dtype=torch.float16, device=W.device)
        check(_lib.lib().diner_train_pack_core(_p(W), W.stride(0), int(transpose), EXP_W, _p(data), _st(W.device)), "diner_train_pack_core")
        return CorePack(data)
    kpad = (K + 31) // 32 * 32
    planes = torch.empty((2, HID * kpad), dtype=torch.float16, device=W.device)
    check(_lib.lib().diner_train_split_panel(_p(W), K, W.stride(0), int(transpose), EXP_W, _p(planes[0]), _p(planes[1]), _st(W.device)),
          "diner_train_split_panel")
    return planes


class PanelCache:
    """fp16 hi/lo panels of the MLP weights, kept per (parameter object, ``_version``, orientation): re-split only after an
    optimizer step (or any other in-place update) -- gradient accumulation / several renders per step reuse them.  Weak
    references: the cache keeps no parameter alive and a new tensor at a recycled address is a different object."""

    def __init__(self):
        self._e = {}

    def get(self, key, src, W, transpose, prec):
        ent = self._e.get((key, transpose))
        if ent is not None and ent[0]() is src and ent[1] == src._version and ent[2] == prec:
            return ent[3]
        planes = split_panel(W, transpose, prec)
        self._e[(key, transpose)] = (weakref.ref(src), src._version, prec, planes)
        return planes


def _panel(A, planes, bias, S, addend, out, relu_a, amax, exp_a, colsum=None, amax_out=None):
    M, K = A.shape
    if isinstance(planes, CorePack):
        check(_lib.lib().diner_train_gemm_core(_p(A), A.stride(0), _p(planes.data), _p(bias), _p(S), 0 if S is None else S.stride(0),
                                               _p(addend), 0 if addend is None else addend.stride(0), _p(out), out.stride(0), M, int(relu_a),
                                               _p(amax), exp_a, EXP_W, _p(colsum), _p(amax_out), _st(out.device)), "diner_train_gemm_core")
        return
    assert colsum is None and amax_out is None
    check(_lib.lib().diner_train_gemm_panel(_p(A), A.stride(0), _p(planes[0]), _p(planes[1]), _p(bias), _p(S), 0 if S is None else S.stride(0),
                                            _p(addend), 0 if addend is None else addend.stride(0), _p(out), out.stride(0), M, K, int(relu_a),
                                            _p(amax), exp_a, EXP_W, _st(out.device)), "diner_train_gemm_panel")


def colsum_amax(dY, db, prec):
    """db += column sums of dY (the bias gradient) and, in f16x3 mode, the device word with max|dY| -- one pass."""
    M, N = dY.shape
    out = torch.empty(1, dtype=torch.int32, device=dY.device) if prec else None
    check(_lib.lib().diner_train_colsum_amax(_p(dY), M, N, dY.stride(0), _p(db), _p(out), _st(dY.device)), "diner_train_colsum_amax")
    return out


def linear_fwd(X, W, b, out, relu_in=False, addend=None, prec=0, panel=None):
    """out[M,N] = addend + relu?(X[M,K]) W[N,K]^T + b   (addend may be ``out`` itself)."""
    M, K = X.shape
    N = W.shape[0]
    if panel is not None:
        return _panel(X, panel, b, None, addend, out, relu_in, None, EXP_ACT)
    if addend is not None and addend is not out:
        out.copy_(addend)
    _gemm(X, W, b, None, out, M, N, K, X.stride(0), 1, 1, W.stride(0), out.stride(0), 0, relu_a=int(relu_in), accumulate=int(addend is not None),
          prec=prec, exp_a=EXP_ACT, exp_b=EXP_W)


def linear_bwd_x(dY, W, mask_src, out, addend=None, prec=0, amax=None, panel=None):
    """out[M,K] = addend + (dY[M,N] W[N,K]) * [mask_src > 0]   (addend may be ``out`` itself)."""
    M, N = dY.shape
    K = W.shape[1]
    if panel is not None:
        return _panel(dY, panel, None, mask_src, addend, out, False, amax, 0)
    if addend is not None and addend is not out:
        out.copy_(addend)
    _gemm(dY, W, None, mask_src, out, M, K, N, dY.stride(0), 1, W.stride(0), 1, out.stride(0),
          0 if mask_src is None else mask_src.stride(0), accumulate=int(addend is not None), prec=prec, amax_a=amax, exp_b=EXP_W)


def linear_bwd_x_reduced(dY, W, mask_src, out, dbs, addend=None, prec=0, amax=None, panel=None):
    """``linear_bwd_x`` followed by ``colsum_amax`` of its result: every ``db`` of ``dbs`` += column sums of ``out``; returns the
    max|out| word.  On the GEMM core both reductions ride in the GEMM's epilogue (no extra pass over ``out``)."""
    if not isinstance(panel, CorePack):
        linear_bwd_x(dY, W, mask_src, out, addend=addend, prec=prec, amax=amax, panel=panel)
        word = colsum_amax(out, dbs[0], prec)
        for db in dbs[1:]:
            check(_lib.lib().diner_train_colsum(_p(out), out.shape[0], out.shape[1], out.stride(0), _p(db), _st(out.device)), "diner_train_colsum")
        return word
    word = torch.zeros(1, dtype=torch.int32, device=out.device)
    cs = dbs[0] if len(dbs) == 1 else torch.zeros(HID, dtype=torch.float32, device=out.device)
    _panel(dY, panel, None, mask_src, addend, out, False, amax, 0, colsum=cs, amax_out=word)
    if len(dbs) > 1:
        for db in dbs:   # db += cs (the colsum kernel on a one-row matrix)
            check(_lib.lib().diner_train_colsum(_p(cs), 1, HID, HID, _p(db), _st(out.device)), "diner_train_colsum")
    return word


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
        SBl, NVl, Cl, hl, wl = lat.shape
        lat_nhwc = torch.empty((SBl, NVl, hl, wl, Cl), dtype=torch.float32, device=dev)   # coalesced texel reads for the gather
        check(L.diner_pack_latent(_p(lat), SBl * NVl, Cl, hl, wl, _p(lat_nhwc), st), "diner_pack_latent")
        prm = [p.detach().to(torch.float32).contiguous() for p in params]
        # backward() re-reads these tensors (they alias the live parameters): remember their versions, as
        # ctx.save_for_backward would, so that an in-place update between forward and backward is an error instead of a
        # silently wrong dX (optimizer.step / EMA copy_ under gradient accumulation)
        ctx.versions = [(weakref.ref(p), p._version) for p in params] + [(weakref.ref(latent), latent._version)]
        w_in56 = torch.zeros((HID, 56), dtype=torch.float32, device=dev)
        w_in56[:, :55] = prm[0]
        rgbsigma = f(SB, NR, K, 4)
        prec = _lib.PRECISIONS[renderer.precision]
        saved = []
        cache = renderer.__dict__.setdefault("_panel_cache", PanelCache())
        wp = {i: cache.get(i, params[i], prm[i], False, prec) for i in (2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26)}
        wp[0] = cache.get(0, params[0], w_in56, False, prec)
        for sb in range(SB):
            in56, zl, taps = f(R, 56), f(R, HID), f(R, 8)
            check(L.diner_train_point_inputs(C.byref(scene), _p(lat_nhwc), 1, _p(rays), _p(z), NR, K, sb, _p(in56), _p(zl), _p(taps), st),
                  "diner_train_point_inputs")
            x = f(R, HID)
            linear_fwd(in56, w_in56, prm[1], x, prec=prec, panel=wp[0])                          # resnetfc.py:139
            xs, nets = [], []
            for b in range(3):
                xb = f(R, HID)
                linear_fwd(zl, prm[2 + 2 * b], prm[3 + 2 * b], xb, addend=x, prec=prec, panel=wp[2 + 2 * b])          # :152-153
                net = f(R, HID)
                linear_fwd(xb, prm[8 + 4 * b], prm[9 + 4 * b], net, relu_in=True, prec=prec, panel=wp[8 + 4 * b])      # :62
                x = f(R, HID)
                linear_fwd(net, prm[10 + 4 * b], prm[11 + 4 * b], x, relu_in=True, addend=xb, prec=prec, panel=wp[10 + 4 * b])  # :63,:69
                xs.append(xb)
                nets.append(net)
            xbar = f(P, HID)
            check(L.diner_train_view_mean(_p(x), P * HID, NV, _p(xbar), 0, st), "diner_train_view_mean")   # :146-149
            xbars, pnets = [], []
            for b in range(3, 5):
                net = f(P, HID)
                linear_fwd(xbar, prm[8 + 4 * b], prm[9 + 4 * b], net, relu_in=True, prec=prec, panel=wp[8 + 4 * b])
                nxt = f(P, HID)
                linear_fwd(net, prm[10 + 4 * b], prm[11 + 4 * b], nxt, relu_in=True, addend=xbar, prec=prec, panel=wp[10 + 4 * b])
                xbars.append(xbar)
                pnets.append(net)
                xbar = nxt
            out = f(P, 4)
            linear_fwd(xbar, prm[28], prm[29], out, relu_in=True, prec=prec)                # :158
            check(L.diner_train_head(_p(out), None, None, P * 4, _p(rgbsigma[sb]), 0, st), "diner_train_head")  # pixelnerf.py:139-143
            saved.append((in56, zl, taps, xs, nets, xbars, pnets, xbar, out))
        N = SB * NR
        rgb, depth, weights = f(SB, NR, 3), f(SB, NR), f(SB, NR, K)
        check(L.diner_composite(_p(rays), _p(z), _p(rgbsigma), N, K, int(bool(renderer.white_bkgd)), _p(rgb), _p(depth), _p(weights), None, st),
              "diner_composite")
        ctx.renderer, ctx.scene, ctx.rays, ctx.z, ctx.rgbsigma = renderer, scene, rays, z, rgbsigma
        ctx.saved_acts, ctx.prm, ctx.w_in56, ctx.lat_shape = saved, prm, w_in56, tuple(latent.shape)
        ctx.keep = (lat,)
        ctx.prec = prec
        ctx.params = params
        return rgb, depth, weights

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_weights):
        L = _lib.lib()
        for ref, ver in ctx.versions:
            t = ref()
            if t is None or t._version != ver:
                raise RuntimeError("diner_amd.training: one of the variables needed for gradient computation (an MLP parameter or "
                                   "encoder.latent) has been modified by an inplace operation between forward and backward")
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
        cache = ctx.renderer.__dict__.setdefault("_panel_cache", PanelCache())
        wt = {i: cache.get(i, ctx.params[i], prm[i], True, prec) for i in (2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28)}
        for sb in range(SB):
            in56, zl, taps, xs, nets, xbars, pnets, xbar5, out = ctx.saved_acts[sb]
            d_out = f(P, 4)
            check(L.diner_train_head(_p(out), _p(rgbsigma[sb]), _p(d_rgbsigma[sb]), P * 4, _p(d_out), 1, st), "diner_train_head(bwd)")
            a_out = colsum_amax(d_out, g[29], prec)
            linear_bwd_w(d_out, xbar5, g[28], None, relu_x=True, prec=prec, amax=a_out)
            d_x = f(P, HID)
            a_x = linear_bwd_x_reduced(d_out, prm[28], xbar5, d_x, [g[27]], prec=prec, amax=a_out, panel=wt[28])
            for i, b in ((1, 4), (0, 3)):                                         # post-mean blocks, reversed
                d_net = f(P, HID)                                                  # (a_x, g[11 + 4b]: reduced by the GEMM that produced d_x)
                a_net = linear_bwd_x_reduced(d_x, prm[10 + 4 * b], pnets[i], d_net, [g[9 + 4 * b]], prec=prec, amax=a_x, panel=wt[10 + 4 * b])
                linear_bwd_w(d_x, pnets[i], g[10 + 4 * b], None, relu_x=True, prec=prec, amax=a_x)
                d_prev = f(P, HID)
                if b == 4:
                    a_prev = linear_bwd_x_reduced(d_net, prm[8 + 4 * b], xbars[i], d_prev, [g[11 + 4 * (b - 1)]], addend=d_x, prec=prec, amax=a_net,
                                                  panel=wt[8 + 4 * b])
                else:
                    linear_bwd_x(d_net, prm[8 + 4 * b], xbars[i], d_prev, addend=d_x, prec=prec, amax=a_net, panel=wt[8 + 4 * b])
                    a_prev = None
                linear_bwd_w(d_net, xbars[i], g[8 + 4 * b], None, relu_x=True, prec=prec, amax=a_net)
                d_x, a_x = d_prev, a_prev
            d_xv = f(R, HID)
            check(L.diner_train_view_mean(_p(d_x), P * HID, NV, _p(d_xv), 1, st), "diner_train_view_mean(bwd)")
            a_xv = colsum_amax(d_xv, g[11 + 4 * 2], prec)
            d_zl = f(R, HID)          # written by the first (b = 2) lin_z backward, accumulated by the other two
            for b in (2, 1, 0):                                                   # per-view blocks, reversed
                d_net = f(R, HID)
                a_net = linear_bwd_x_reduced(d_xv, prm[10 + 4 * b], nets[b], d_net, [g[9 + 4 * b]], prec=prec, amax=a_xv, panel=wt[10 + 4 * b])
                linear_bwd_w(d_xv, nets[b], g[10 + 4 * b], None, relu_x=True, prec=prec, amax=a_xv)
                d_xs = f(R, HID)
                # d_xs is the dY of lin_z[b] AND of the previous block's fc_1 (the residual stream): one reduction, two bias gradients
                a_xs = linear_bwd_x_reduced(d_net, prm[8 + 4 * b], xs[b], d_xs, [g[3 + 2 * b]] + ([g[11 + 4 * (b - 1)]] if b else []), addend=d_xv,
                                            prec=prec, amax=a_net, panel=wt[8 + 4 * b])
                linear_bwd_w(d_net, xs[b], g[8 + 4 * b], None, relu_x=True, prec=prec, amax=a_net)
                linear_bwd_w(d_xs, zl, g[2 + 2 * b], None, prec=prec, amax=a_xs)          # lin_z[b]
                linear_bwd_x(d_xs, prm[2 + 2 * b], None, d_zl, addend=None if b == 2 else d_zl, prec=prec, amax=a_xs, panel=wt[2 + 2 * b])
                d_xv, a_xv = d_xs, a_xs
            linear_bwd_w(d_xv, in56, g_in56, None, prec=prec, amax=a_xs)           # lin_in (its d_xv is lin_z[0]'s d_xs)
            check(L.diner_train_bilinear_scatter(_p(d_zl), _p(taps), P, HID, scene.h, scene.w, NV, sb, _p(d_lat_nhwc), st),
                  "diner_train_bilinear_scatter")
        d_lat = torch.empty(ctx.lat_shape, dtype=torch.float32, device=dev)
        check(L.diner_train_nhwc_to_nchw(_p(d_lat_nhwc), SBl * NVl, Cl, hl, wl, _p(d_lat), st), "diner_train_nhwc_to_nchw")
        g[0] = g_in56[:, :55].contiguous()
        g[1] = g[3].clone()  # lin_in's bias sees the same dY as lin_z[0]'s: x = lin_in(..) + lin_z[0](z)
        return (None, None, None, None, d_lat) + tuple(g)


def render_with_grad(renderer, model, rays, z, scene):
    """rgb, depth, weights = composite(model, rays, z) with gradients to the MLP parameters and encoder.latent."""
    params = _mlp_params(model.mlp_fine)
    return _RenderFn.apply(renderer, scene, rays, z, model.encoder.latent, *params)
