"""``NeRFRendererDGS`` -- the MI355X-native drop-in for the reference renderer plug-in.

Mirrors ``src/models/nerf_renderer.py:12-430`` of tancredeguillou/diner: same constructor
kwargs, same mutable attributes (``n_samples``/``n_gaussian`` are re-assigned after loading a
checkpoint, ``python_scripts/create_prediction_folder.py:49-52``), same
``forward(model, rays, want_weights)`` contract and the same stage methods.  Selecting it is a
one-line YAML change (``renderer.module: diner_amd.NeRFRendererDGS``, resolved by
``src/util/import_helper.py:16-24`` at ``src/models/diner.py:48``).

All arithmetic runs in the hand-written HIP kernels of ``libdiner_hip.so`` through the C ABI in
``include/diner_hip.h``; PyTorch only owns device memory and the stream.  There is no CPU or
eager fallback: without the library the import of this module's dependencies fails.

The module holds no parameters or buffers (reference checkpoints load with ``strict=True``); it
keeps an identity-keyed cache of re-packed copies of the model's maps and MLP weights: an entry is
valid only while the very tensor OBJECTS it was packed from are alive (weak references, compared with
``is``) and their ``_version`` counters are unchanged -- a fresh tensor that the caching allocator
happens to place at a freed tensor's address is a different object and invalidates the entry, as does
every ``encode()`` of the reference, which re-binds ``encoder.latent/depths/...`` to new tensors
(src/models/image_encoder.py:214-218,271-272).  No strong reference to the sources is kept.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import DinerMlpRaw, DinerSamplerCfg, DinerScene, check


class RenderOutput(dict):
    """Attribute dict standing in for ``dotmap.DotMap`` (reference nerf_renderer.py:421-430)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Sources:
    """Identity + version of the tensors a packed copy was made from (SURVEY.md §8(b) ownership row): weak
    references only, so the cache neither extends a source's lifetime nor can be fooled by address reuse."""

    __slots__ = ("refs", "versions", "shapes")

    def __init__(self, tensors):
        self.refs = [weakref.ref(t) for t in tensors]
        self.versions = [t._version for t in tensors]
        self.shapes = [(tuple(t.shape), t.dtype, t.device) for t in tensors]

    def valid_for(self, tensors) -> bool:
        return len(tensors) == len(self.refs) and all(
            r() is t and v == t._version and sh == (tuple(t.shape), t.dtype, t.device)
            for r, v, sh, t in zip(self.refs, self.versions, self.shapes, tensors))


class _FiniteGuard:
    """The non-finite status of the frames a renderer has issued (its own object so that a ``weakref.finalize`` of the renderer can
    run the last check without keeping the renderer alive).  The compositing kernel ORs DINER_STATUS_NONFINITE into the device word;
    a copy to pinned host memory + an event follow every call; ``poll`` examines the copies that have completed (all, if ``wait``)."""

    unreported = []          # messages of non-finite frames found where nothing could be raised (a finalizer): raised by the next call

    def __init__(self):
        self.status = None   # device int32 word
        self.pending = []    # [(event, pinned host word)] oldest first
        self.free = []       # examined (event, word) pairs, reused by after_launch
        self.precision = "f16x3"

    def word(self, dev):
        if self.status is None or self.status.device != dev:
            self.status = torch.zeros(1, dtype=torch.int32, device=dev)
            self.pending = []
        return self.status

    def message(self):
        return ("diner_amd.NeRFRendererDGS: a rendered rgb-sigma sample was inf/NaN" +
                (" -- in precision='f16x3' an MLP activation left the fp16 range (|x| >= ~1e6, see DESIGN.md); set "
                 "renderer.precision = 'fp32' (exact fp32 MFMA) for this model" if self.precision == "f16x3" else
                 " -- the model itself produces non-finite values for these inputs"))

    def poll(self, wait=False, keep=0):
        """examine the completed copies; with ``wait`` block for all but the ``keep`` youngest; raise if a frame went non-finite"""
        if _FiniteGuard.unreported:
            msg, _FiniteGuard.unreported[:] = _FiniteGuard.unreported[0], []
            raise RuntimeError(msg + " (found when an earlier renderer was collected: its last frames had not been examined)")
        bad = False
        while self.pending and ((wait and len(self.pending) > keep) or self.pending[0][0].query()):
            ev, host = self.pending.pop(0)
            ev.synchronize()
            bad |= bool(int(host[0]) & 1)
            if len(self.free) < 8:
                self.free.append((ev, host))
        if bad:
            self.pending = []
            self.status.zero_()
            raise RuntimeError(self.message())

    def after_launch(self, dev):
        # pinned host words and events are recycled (a pinned allocation per call costs tens of microseconds of host time:
        # 64 calls per 512 x 512 image in the reference's 4096-ray chunks)
        ev, host = self.free.pop() if self.free else (torch.cuda.Event(), torch.empty(1, dtype=torch.int32, pin_memory=True))
        host.copy_(self.status, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        self.pending.append((ev, host))

    def finalize(self):
        """weakref.finalize of the renderer (garbage collection, or interpreter exit): the frames nobody examined.  An exception
        cannot leave a finalizer, so: at interpreter exit the process ends with a non-zero status and the message on stderr; earlier,
        the message is parked and raised by the next call of any renderer."""
        import sys
        if not self.pending:
            return
        try:
            bad = False
            for ev, host in self.pending:
                ev.synchronize()
                bad |= bool(int(host[0]) & 1)
            self.pending = []
        except Exception:      # the HIP runtime is already gone: nothing left to examine with
            return
        if bad:
            if _EXITING[0]:
                sys.stderr.write("RuntimeError: " + self.message() + " (found at interpreter exit: the last frames were never examined; "
                                 "call renderer.check_finite() after the last forward())\n")
                sys.stderr.flush()
                import os
                os._exit(70)
            _FiniteGuard.unreported.append(self.message())


_EXITING = [False]


def _mark_exit():
    _EXITING[0] = True


import atexit  # noqa: E402
# weakref.finalize callbacks run from an atexit hook registered when the FIRST finalize object is created; atexit runs hooks
# last-in-first-out, so this one (registered at import, i.e. before any renderer exists ... but possibly after another module's
# finalize) is re-registered by every renderer constructor to be sure it runs before them
atexit.register(_mark_exit)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class NeRFRendererDGS(torch.nn.Module):
    """NeRF renderer with depth-guided sampling (reference nerf_renderer.py:12-37).

    :param n_samples: samples per ray
    :param n_depth_candidates: stratified candidates that are short-listed by surface likelihood
    :param n_gaussian: samples drawn from the gaussian fitted to the occlusion-aware likelihoods
    :param eval_batch_size: kept for signature compatibility (the fused kernel needs no chunking)
    :param white_bkgd: white instead of black background
    """

    def __init__(self, n_samples=40, n_depth_candidates=1000, n_gaussian=15, eval_batch_size=100000,
                 white_bkgd=True):
        super().__init__()
        self.n_samples = n_samples
        self.n_depth_candidates = n_depth_candidates
        self.n_gaussian = n_gaussian
        self.eval_batch_size = eval_batch_size
        self.white_bkgd = white_bkgd
        self.seed = 0            # base seed of the in-kernel Philox generator (perf mode)
        # arithmetic of the fusion-MLP GEMMs (build-only extra): "f16x3" = fp32 operands split into fp16
        # hi+lo, 3 fp16 MFMAs per product, fp32 accumulate (fp32-grade, ~3x faster); "fp32" = fp32 MFMA
        self.precision = "f16x3"
        # profiling hook (bench.py): when a list, forward() launches the three stage kernels through
        # their own C-ABI entry points (exactly what diner_render does internally) and appends
        # (ev0, ev1, ev2, ev3) torch.cuda.Events bracketing sampler | points+MLP | compositing
        self.stage_events = None
        self._calls = 0
        self._maps_key = self._maps_pack = None      # packed depth/sigma/normal maps + cameras
        self._latent_key = self._latent_pack = None  # packed NHWC latent
        self._linz_key = self._linz_pack = None      # lin_z[b](latent) feature maps (f16x3 mode)
        # f16x3 mode: hoist lin_z from per point to per latent texel (linear map and bilinear interpolation
        # commute; diner_pack_linz_maps).  Costs 3x the latent's memory per encode(); set False to keep
        # lin_z as per-point GEMMs.
        # Binding of the two whole-path entry points: "torch_ops" = torch.ops.diner.render / render_image (diner_amd/ops.py, the
        # torch C++ extension north_star names), "ctypes" = the C ABI directly.  Same C functions either way; the build-only
        # arguments (replayed noise, stage events) always take the ctypes route.  Default: the ops when the extension is built.
        from . import ops as _ops
        self.binding = "torch_ops" if _ops.available() else "ctypes"
        self.linz_maps = True
        # Memory budget of that hoist: the lin_z maps are 3x the latent (2.5 GB at the headline config, 16 GB for a 1024^2 x 8-view
        # encode).  Above this many bytes the maps are NOT built and lin_z stays a per-point gather + GEMM (the kernel's LINZ = false
        # instantiation: same results, ~1.3x the frame time); None = no limit.  See memory_report().
        self.linz_maps_max_bytes = 64 << 30
        # informational (bench.py's executed-FLOP count): the f16x3 kernel applies block 2's fc_1 once to the mean over views
        self.fc1_on_mean = True
        self._mlp_key = None
        self._mlp_pack = None
        self._latent_gen = self._mlp_gen = 0         # bumped by every re-pack; the lin_z maps depend on both
        # Non-finite guard.  The compositing kernel ORs DINER_STATUS_NONFINITE into a device word when an rgb-sigma
        # sample is inf/NaN (in f16x3 mode: an MLP activation beyond the fp16 range, |x| >= ~1e6).  The word is copied
        # to pinned host memory behind every call.  No frame can leave a program unexamined:
        #   * a call that carries more than `finite_sync_rays` rays (anything larger than the reference's 4096-ray chunk:
        #     a whole image / a large batch) or asks for the weights is examined before forward() returns (the wait is
        #     microseconds against >= 30 ms of rendering); render_image always is;
        #   * smaller chunks ("deferred"): forward() examines every copy that has completed and blocks only for frames
        #     older than the two youngest -- the host never stalls the GPU, and at most two chunks are unexamined at any time;
        #   * those last chunks are examined by check_finite(), by the next call of any renderer, and by a weakref.finalize of
        #     this module: at garbage collection the finding is raised by the next call, at interpreter exit the process ends
        #     with status 70 and the message on stderr (tests/test_gpu_edge.py).
        # "sync" waits at the end of every call; "off" never looks.
        self.finite_check = "deferred"
        self.finite_sync_rays = 4096
        self._guard = _FiniteGuard()
        atexit.unregister(_mark_exit)
        self._finalizer = weakref.finalize(self, self._guard.finalize)   # (registers weakref's own atexit hook the first time)
        atexit.register(_mark_exit)                  # ... and this one after it: atexit is LIFO, so the flag is set before the finalizers run

    # ------------------------------------------------------------------------------------------
    # model -> packed device state (cached)
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _validate_model(model):
        enc, mlp = model.encoder, model.mlp_fine
        if getattr(enc, "index_interp", "bilinear") != "bilinear" or getattr(enc, "index_padding", "border") != "border":
            raise NotImplementedError("only index_interp='bilinear', index_padding='border' (image_encoder.py:24-25)")
        if getattr(mlp, "combine_type", "average") != "average":
            raise NotImplementedError("only combine_type='average' (resnetfc.py:9-14)")
        if not isinstance(getattr(mlp, "activation", torch.nn.ReLU()), torch.nn.ReLU):
            raise NotImplementedError("only ReLU activations (beta=0, resnetfc.py:124-127)")
        dims = (mlp.d_in, mlp.d_latent, mlp.d_hidden, mlp.d_out, mlp.n_blocks, mlp.combine_layer)
        if dims != (55, 512, 512, 4, 5, 3):
            raise NotImplementedError(f"fusion MLP dims {dims} unsupported; kernels are built for (55,512,512,4,5,3)")
        for pe in (model.poscode, model.depthcode):
            if pe.num_freqs != 6 or not pe.include_input:
                raise NotImplementedError("positional encoding must be num_freqs=6, include_input=True")

    def _scene(self, model, need_latent=True, packed_mlp=None) -> Tuple[DinerScene, tuple]:
        enc = model.encoder
        dev = enc.depths.device
        msrc = [model.poses, model.focal, model.c, model.image_shape, enc.depths, enc.depths_std, enc.normals]
        if self._maps_key is None or not self._maps_key.valid_for(msrc):
            SB, NV, _, H, W = enc.depths.shape
            maps = torch.empty((SB, NV, H, W, 8), dtype=torch.float32, device=dev)
            d, s, n = _f32c(enc.depths), _f32c(enc.depths_std), _f32c(enc.normals)
            check(_lib.lib().diner_pack_maps(_ptr(d), _ptr(s), _ptr(n), SB * NV, H, W, _ptr(maps), _stream(dev)),
                  "diner_pack_maps")
            poses = _f32c(model.poses)
            if poses.shape[-2:] != (4, 4):  # accept [.., 3, 4] poses
                full = torch.zeros((*poses.shape[:-2], 4, 4), dtype=torch.float32, device=dev)
                full[..., :3, :] = poses[..., :3, :]
                full[..., 3, 3] = 1
                poses = full.contiguous()
            torch.cuda.current_stream(dev).synchronize()  # d/s/n may be temporaries
            ishape = [float(v) for v in model.image_shape.detach().float().cpu()]  # (W, H), pixelnerf.py:50-51
            self._maps_pack, self._maps_key = (maps, poses, _f32c(model.focal), _f32c(model.c), ishape), _Sources(msrc)
        if need_latent:
            if self._latent_key is None or not self._latent_key.valid_for([enc.latent]):
                lat = _f32c(enc.latent)
                SB, NV, Cc, h, w = lat.shape
                latent = torch.empty((SB, NV, h, w, Cc), dtype=torch.float32, device=dev)
                check(_lib.lib().diner_pack_latent(_ptr(lat), SB * NV, Cc, h, w, _ptr(latent), _stream(dev)),
                      "diner_pack_latent")
                torch.cuda.current_stream(dev).synchronize()
                self._latent_pack, self._latent_key = latent, _Sources([enc.latent])
                self._latent_gen += 1
        maps, poses, focal, c, ishape = self._maps_pack
        latent = self._latent_pack if need_latent else None
        linz = None
        linz_bytes = 3 * latent.numel() * 4 if latent is not None else 0
        if (need_latent and packed_mlp is not None and self.linz_maps and self.precision == "f16x3"
                and (self.linz_maps_max_bytes is None or linz_bytes <= self.linz_maps_max_bytes)):
            zkey = (self._latent_gen, self._mlp_gen)   # generations of the two packs the maps were computed from
            if zkey != self._linz_key:
                SB, NV, h, w, Cc = latent.shape
                out = torch.empty((3, SB, NV, h, w, Cc), dtype=torch.float32, device=dev)
                check(_lib.lib().diner_pack_linz_maps(_ptr(latent), SB * NV, h, w, _ptr(packed_mlp), _ptr(out), _stream(dev)),
                      "diner_pack_linz_maps")
                self._linz_pack, self._linz_key = out, zkey
            linz = self._linz_pack
        SB, NV, H, W, _ = maps.shape
        sc = DinerScene()
        sc.SB, sc.NV, sc.H, sc.W = SB, NV, H, W
        if latent is not None:
            assert latent.shape[:2] == (SB, NV)  # image_encoder.py:105
            sc.h, sc.w, sc.C = latent.shape[2], latent.shape[3], latent.shape[4]
        sc.image_w, sc.image_h = ishape
        sc.feature_padding = float(model.encoder.feature_padding)
        sc.num_freqs = int(model.poscode.num_freqs)
        sc.freq_factor = float(model.poscode.freqs[0])
        sc.poses, sc.focal, sc.c = poses.data_ptr(), focal.data_ptr(), c.data_ptr()
        sc.maps = maps.data_ptr()
        sc.latent = latent.data_ptr() if latent is not None else None
        sc.linz_maps = linz.data_ptr() if linz is not None else None
        return sc, (maps, poses, focal, c, latent, linz)

    def memory_report(self, model=None, rays_per_call=None, n_views=None):
        """Bytes of device memory this renderer holds / will take, so that the appetite is a number and not a surprise:
        ``cached`` = what the pack caches hold right now (packed maps, NHWC latent, lin_z maps, packed MLP); with ``rays_per_call``
        (and ``n_views``, default: the cached scene's) also ``per_call`` = workspace + outputs of one inference ``forward`` and
        ``training_step`` = the activations the differentiable path keeps alive between forward and backward
        (diner_amd/training.py: every layer's input in fp32, 24 GB for 4096 rays x 40 samples x 4 views)."""
        nb = lambda t: 0 if t is None else t.numel() * t.element_size()
        maps = self._maps_pack[0] if self._maps_pack is not None else None
        rep = {"cached": {"maps": nb(maps), "latent_nhwc": nb(self._latent_pack), "linz_maps": nb(self._linz_pack), "mlp_packed": nb(self._mlp_pack)}}
        rep["cached"]["total"] = sum(rep["cached"].values())
        rep["linz_maps_max_bytes"] = self.linz_maps_max_bytes
        if rays_per_call is not None:
            NV = n_views if n_views is not None else (maps.shape[1] if maps is not None else 1)
            K, P = int(self.n_samples), int(rays_per_call) * int(self.n_samples)
            ws = int(_lib.lib().diner_render_workspace_floats(1, int(rays_per_call), K, NV, _lib.PRECISIONS[self.precision])) * 4
            rep["per_call"] = {"workspace": ws, "outputs": int(rays_per_call) * (4 + K) * 4}
            rows = NV * P
            # forward keeps: in56 + zlat + taps per (view, point); x, net of the 3 per-view blocks; the post-mean tensors per point
            rep["training_step"] = {"per_view_rows": rows,
                                    "saved_activations": rows * 4 * (56 + 512 + 8 + 2 * 3 * 512) + P * 4 * (2 * 2 * 512 + 512 + 4),
                                    "note": "peak = saved activations + ~8 row-matrices [rows,512] fp32 of forward/backward temporaries "
                                            "(measured: 24.5 GB at 4096 rays x 40 samples x 4 views, tools/bench_train.py)"}
        return rep

    def _mlp(self, model) -> torch.Tensor:
        mlp = model.mlp_fine
        params = [mlp.lin_in.weight, mlp.lin_in.bias, mlp.lin_out.weight, mlp.lin_out.bias]
        for b in range(3):
            params += [mlp.lin_z[b].weight, mlp.lin_z[b].bias]
        for b in range(5):
            params += [mlp.blocks[b].fc_0.weight, mlp.blocks[b].fc_0.bias, mlp.blocks[b].fc_1.weight, mlp.blocks[b].fc_1.bias]
        if self._mlp_key is None or not self._mlp_key.valid_for(params):
            keep = [_f32c(p) for p in params]
            raw = DinerMlpRaw()
            raw.lin_in_w, raw.lin_in_b, raw.lin_out_w, raw.lin_out_b = [t.data_ptr() for t in keep[:4]]
            for b in range(3):
                raw.lin_z_w[b], raw.lin_z_b[b] = keep[4 + 2 * b].data_ptr(), keep[5 + 2 * b].data_ptr()
            for b in range(5):
                o = 10 + 4 * b
                raw.fc0_w[b], raw.fc0_b[b] = keep[o].data_ptr(), keep[o + 1].data_ptr()
                raw.fc1_w[b], raw.fc1_b[b] = keep[o + 2].data_ptr(), keep[o + 3].data_ptr()
            dev = keep[0].device
            packed = torch.empty(int(_lib.lib().diner_mlp_packed_floats()), dtype=torch.float32, device=dev)
            check(_lib.lib().diner_pack_mlp(C.byref(raw), _ptr(packed), _stream(dev)), "diner_pack_mlp")
            torch.cuda.current_stream(dev).synchronize()  # `keep` may be temporaries: finish before they die
            self._mlp_pack, self._mlp_key = packed, _Sources(params)
            self._mlp_gen += 1
        return self._mlp_pack

    # ------------------------------------------------------------------------------------------
    # non-finite guard
    # ------------------------------------------------------------------------------------------
    def _status_word(self, dev) -> Optional[torch.Tensor]:
        if self.finite_check == "off":
            return None
        return self._guard.word(dev)

    def _poll_status(self, wait=False, keep=0):
        """Examine the completed status copies (all but the ``keep`` youngest if ``wait``); raise if a frame went non-finite."""
        self._guard.precision = self.precision
        self._guard.poll(wait=wait, keep=keep)

    def _after_launch(self, dev, sync=False):
        if self._guard.status is None or self.finite_check == "off":
            return
        self._guard.after_launch(dev)
        if sync or self.finite_check == "sync":
            self._poll_status(wait=True)
        else:
            self._poll_status(wait=True, keep=2)     # never more than two small chunks unexamined; no stall: the GPU is two calls ahead

    def check_finite(self):
        """Wait for every render issued so far and raise ``RuntimeError`` if one produced inf/NaN samples."""
        self._poll_status(wait=True)

    @staticmethod
    def _check_rays(rays):
        assert len(rays.shape) == 3 and rays.shape[-1] == 8  # nerf_renderer.py:412
        if not rays.is_cuda:
            raise RuntimeError("diner_amd.NeRFRendererDGS runs on the GPU only (rays are on %s)" % rays.device)
        return _f32c(rays)

    def _next_seed(self) -> int:
        self._calls += 1
        return (int(self.seed) * 0x9E3779B97F4A7C15 + self._calls) & 0xFFFFFFFFFFFFFFFF

    def _cfg(self, n_samples, n_candidates, n_gaussian, depth_diff_max=0.05) -> DinerSamplerCfg:
        assert n_samples >= n_gaussian  # nerf_renderer.py:89
        cfg = DinerSamplerCfg()
        cfg.n_candidates, cfg.n_samples, cfg.n_gaussian = int(n_candidates), int(n_samples), int(n_gaussian)
        cfg.depth_diff_max = float(depth_diff_max)
        return cfg

    # ------------------------------------------------------------------------------------------
    # stages (reference signatures; keyword-only extras are build-only)
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_coarse(self, rays, n_coarse=None, *, u_coarse=None):
        """Stratified candidates (reference nerf_renderer.py:39-63): rays [SB,B,8] -> [SB,B,Kc]."""
        n_coarse = n_coarse if n_coarse else self.n_coarse
        r = self._check_rays(rays.reshape(1, -1, 8))
        N = r.shape[1]
        z = torch.empty((N, n_coarse), dtype=torch.float32, device=r.device)
        u = None if u_coarse is None else _f32c(u_coarse).reshape(N, n_coarse)
        check(_lib.lib().diner_sample_coarse(_ptr(r), N, n_coarse, _ptr(u), self._next_seed(), _ptr(z), _stream(r.device)),
              "diner_sample_coarse")
        return z.view(*rays.shape[:-1], n_coarse)

    @torch.no_grad()
    def sample_depthguided(self, rays, model, n_samples, n_candidates, depth_diff_max=0.05, n_gaussian=None, *,
                           noise: Optional[Sequence[Optional[torch.Tensor]]] = None, z_cand=None,
                           return_internals=False):
        """Depth-guided short-list + gaussian samples (reference nerf_renderer.py:65-284).
        Returns z [SB,NR,n_samples] *before* fill-up (0 = empty slot), like the reference.
        ``noise=(u_coarse, n_gauss, u_fill)`` dense tensors for parity runs."""
        n_gaussian = n_gaussian if n_gaussian is not None else self.n_gaussian
        out = self._sample(rays, model, n_samples, n_candidates, n_gaussian, depth_diff_max, noise, z_cand,
                           want_dg=True, want_lik=return_internals)
        return (out["z_dg"], out) if return_internals else out["z_dg"]

    def _sample(self, rays, model, K, NC, G, depth_diff_max, noise, z_cand, want_dg=False, want_lik=False):
        r = self._check_rays(rays)
        SB, NR, _ = r.shape
        sc, _keep = self._scene(model, need_latent=False)
        assert SB == sc.SB
        cfg = self._cfg(K, NC, G, depth_diff_max)
        dev = r.device
        u_c = n_g = u_f = None
        if noise is not None:
            u_c, n_g, u_f = [None if t is None else _f32c(t).to(dev) for t in noise]
            if u_c is not None: assert u_c.numel() == SB * NR * NC
            if n_g is not None: assert n_g.numel() == SB * NR * G
            if u_f is not None: assert u_f.numel() == SB * NR * K
        zc = None if z_cand is None else _f32c(z_cand).to(dev)
        z = torch.empty((SB, NR, K), dtype=torch.float32, device=dev)
        z_dg = torch.empty_like(z) if want_dg else None
        lik = torch.empty((SB, NR, NC), dtype=torch.float32, device=dev) if want_lik else None
        check(_lib.lib().diner_sample_depthguided(C.byref(sc), _ptr(r), NR, C.byref(cfg), _ptr(u_c), _ptr(n_g), _ptr(u_f),
                                                  _ptr(zc), self._next_seed(), _ptr(z), _ptr(z_dg), _ptr(lik),
                                                  _stream(dev)), "diner_sample_depthguided")
        return dict(z=z, z_dg=z_dg, likelihood=lik)

    @torch.no_grad()
    def fill_up_uniform_samples(self, z_samples, rays, *, u_fill=None):
        """Uniform fill-up of the empty slots + sort (reference nerf_renderer.py:367-397)."""
        r = self._check_rays(rays)
        z = _f32c(z_samples)
        K = z.shape[-1]
        N = r.shape[0] * r.shape[1]
        out = torch.empty_like(z)
        u = None if u_fill is None else _f32c(u_fill).to(r.device)
        check(_lib.lib().diner_fill_up_uniform_samples(_ptr(r), _ptr(z), N, K, _ptr(u), self._next_seed(), _ptr(out),
                                                       _stream(r.device)), "diner_fill_up_uniform_samples")
        return out

    def render_points(self, model, rays, z_samp):
        """rgb-sigma of the sample points (the ``model(points, viewdirs)`` calls of composite(),
        reference nerf_renderer.py:304-339 + pixelnerf.py:55-145): -> [SB,B,K,4]."""
        self._require_no_grad(model)
        r = self._check_rays(rays)
        z = _f32c(z_samp)
        SB, NR, K = z.shape
        packed = self._mlp(model)
        sc, _keep = self._scene(model, need_latent=True, packed_mlp=packed)
        assert SB == sc.SB  # pixelnerf.py:68
        out = torch.empty((SB, NR, K, 4), dtype=torch.float32, device=r.device)
        prec = _lib.PRECISIONS[self.precision]
        n_scr = int(_lib.lib().diner_render_points_scratch_floats(SB, sc.NV, prec))
        scr = torch.empty(n_scr, dtype=torch.float32, device=r.device) if n_scr else None
        check(_lib.lib().diner_render_points(C.byref(sc), _ptr(packed), _ptr(r), _ptr(z), NR, K, prec, _ptr(scr),
                                             _ptr(out), _stream(r.device)),
              "diner_render_points")
        return out

    def composite(self, model, rays, z_samp, *, rgbsigma=None, _sync=False):
        """Alpha compositing (reference nerf_renderer.py:286-365) -> (weights, rgb, depth)."""
        r = self._check_rays(rays)
        z = _f32c(z_samp)
        SB, NR, K = z.shape
        if rgbsigma is None and torch.is_grad_enabled() and (any(p.requires_grad for p in model.mlp_fine.parameters())
                                                             or model.encoder.latent.requires_grad):
            out = self._forward_train(model, rays, True, z_samples=z).fine   # differentiable like the reference's composite
            return out.weights, out.rgb, out.depth
        if rgbsigma is None:
            rgbsigma = self.render_points(model, rays, z)
        c = _f32c(rgbsigma)
        dev = r.device
        weights = torch.empty((SB, NR, K), dtype=torch.float32, device=dev)
        rgb = torch.empty((SB, NR, 3), dtype=torch.float32, device=dev)
        depth = torch.empty((SB, NR), dtype=torch.float32, device=dev)
        self._poll_status()
        check(_lib.lib().diner_composite(_ptr(r), _ptr(z), _ptr(c), SB * NR, K, int(bool(self.white_bkgd)), _ptr(rgb),
                                         _ptr(depth), _ptr(weights), _ptr(self._status_word(dev)), _stream(dev)), "diner_composite")
        self._after_launch(dev, sync=_sync)
        return weights, rgb, depth

    @staticmethod
    def _require_no_grad(model):
        if torch.is_grad_enabled() and any(p.requires_grad for p in model.mlp_fine.parameters()):
            raise RuntimeError(
                "render_points() is the fused inference kernel and carries no autograd graph: call forward()/composite() "
                "(they switch to the differentiable training path of diner_amd/training.py) or wrap it in torch.no_grad().")

    # ------------------------------------------------------------------------------------------
    def forward(self, model, rays, want_weights=False, *, noise=None, z_samples=None):
        """Reference nerf_renderer.py:399-424.
        :param model: ``PixelNeRF`` (encode() already called)
        :param rays: [SB,B,8] = origin, direction, near, far
        :param want_weights: also return the compositing weights [SB,B,K]
        :param noise: (build-only) dense ``(u_coarse, n_gauss, u_fill)`` replacing the in-kernel RNG
        :param z_samples: (build-only) inject sorted samples [SB,B,K] and skip the sampler
        :return: ``out.fine.rgb`` [SB,B,3], ``out.fine.depth`` [SB,B], ``out.fine.weights`` iff requested
        """
        assert len(rays.shape) == 3
        self._validate_model(model)
        if torch.is_grad_enabled() and (any(p.requires_grad for p in model.mlp_fine.parameters()) or model.encoder.latent.requires_grad):
            return self._forward_train(model, rays, want_weights, noise=noise, z_samples=z_samples)
        with torch.no_grad():
            r = self._check_rays(rays)
            SB, NR, _ = r.shape
            K = int(self.n_samples)
            dev = r.device
            big = want_weights or SB * NR > int(self.finite_sync_rays)   # examined before this call returns (see __init__)
            if z_samples is not None:
                w_, rgb, depth = self.composite(model, r, z_samples, _sync=big)
                weights = w_ if want_weights else None
            else:
                packed = self._mlp(model)
                sc, _keep = self._scene(model, need_latent=True, packed_mlp=packed)
                assert SB == sc.SB
                cfg = self._cfg(K, self.n_depth_candidates, self.n_gaussian)
                u_c = n_g = u_f = None
                if noise is not None:
                    u_c, n_g, u_f = [None if t is None else _f32c(t).to(dev) for t in noise]
                prec = _lib.PRECISIONS[self.precision]
                L, st, seed = _lib.lib(), _stream(dev), self._next_seed()
                self._poll_status()
                status = _ptr(self._status_word(dev))
                use_ops = self.stage_events is None and noise is None and self.binding == "torch_ops"
                if not use_ops:    # (the op allocates its own workspace and outputs: never both sets at once)
                    ws = torch.empty(int(L.diner_render_workspace_floats(SB, NR, K, sc.NV, prec)), dtype=torch.float32, device=dev)
                    rgb = torch.empty((SB, NR, 3), dtype=torch.float32, device=dev)
                    depth = torch.empty((SB, NR), dtype=torch.float32, device=dev)
                    weights = torch.empty((SB, NR, K), dtype=torch.float32, device=dev) if want_weights else None
                if use_ops:
                    from . import ops as _ops
                    maps_t, poses_t, focal_t, c_t, latent_t, linz_t = _keep
                    rgb, depth, w_ = _ops.load().render(maps_t, poses_t, focal_t, c_t, latent_t, linz_t, packed, r, sc.image_w, sc.image_h,
                                                        sc.feature_padding, sc.num_freqs, sc.freq_factor, cfg.n_candidates, cfg.n_samples,
                                                        cfg.n_gaussian, cfg.depth_diff_max, bool(self.white_bkgd), prec, seed - (1 << 64) if seed >= (1 << 63) else seed,
                                                        bool(want_weights), self._status_word(dev))
                    weights = w_ if want_weights else None
                elif self.stage_events is None:
                    check(L.diner_render(C.byref(sc), _ptr(packed), _ptr(r), NR, C.byref(cfg), int(bool(self.white_bkgd)),
                                         prec, _ptr(u_c), _ptr(n_g), _ptr(u_f), seed, _ptr(ws), _ptr(rgb), _ptr(depth),
                                         _ptr(weights), status, st), "diner_render")
                else:
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    z, c, scr = ws[:SB * NR * K], ws[SB * NR * K:SB * NR * K * 5], ws[SB * NR * K * 5:]
                    ev[0].record()
                    check(L.diner_sample_depthguided(C.byref(sc), _ptr(r), NR, C.byref(cfg), _ptr(u_c), _ptr(n_g), _ptr(u_f),
                                                     None, seed, _ptr(z), None, None, st), "diner_sample_depthguided")
                    ev[1].record()
                    check(L.diner_render_points(C.byref(sc), _ptr(packed), _ptr(r), _ptr(z), NR, K, prec,
                                                _ptr(scr) if scr.numel() else None, _ptr(c), st),
                          "diner_render_points")
                    ev[2].record()
                    check(L.diner_composite(_ptr(r), _ptr(z), _ptr(c), SB * NR, K, int(bool(self.white_bkgd)), _ptr(rgb),
                                            _ptr(depth), _ptr(weights), status, st), "diner_composite")
                    ev[3].record()
                    self.stage_events.append(ev)
                self._after_launch(dev, sync=big)
        return RenderOutput(fine=self._format_outputs(weights, rgb, depth, want_weights=want_weights))

    @torch.no_grad()
    def render_image(self, model, target_extrinsics, target_intrinsics, H, W, z_near, z_far, return_depth=False):
        """The render half of ``DINER.predict_imgs_from_batch`` (reference src/models/diner.py:75-97) without the
        ray-batch loop and without a rays tensor round trip: ``gen_rays`` (src/util/cam_geometry.py:36-79) is evaluated
        inside the sampler kernel (``diner_render_image``), the whole target image is ONE launch per stage (no 4096-ray
        chunks, no ``torch.cat``), output in the reference's image layout.  Bit-identical to ``forward(gen_rays(...))``.
        :param target_extrinsics: [SB,4,4] world->cam;  target_intrinsics: [SB,3,3];  z_near, z_far: [SB] or scalars
        :return: rgb [SB,3,H,W] (, depth [SB,1,H,W])"""
        self._validate_model(model)
        dev = target_extrinsics.device
        SB = target_extrinsics.shape[0]
        E, Ki = _f32c(target_extrinsics), _f32c(target_intrinsics)
        zn = torch.as_tensor(z_near, dtype=torch.float32, device=dev).expand(SB).contiguous()
        zf = torch.as_tensor(z_far, dtype=torch.float32, device=dev).expand(SB).contiguous()
        packed = self._mlp(model)
        sc, _keep = self._scene(model, need_latent=True, packed_mlp=packed)
        assert SB == sc.SB
        K = int(self.n_samples)
        cfg = self._cfg(K, self.n_depth_candidates, self.n_gaussian)
        cam = _lib.DinerTargetCam()
        cam.extrinsics, cam.intrinsics, cam.z_near, cam.z_far = E.data_ptr(), Ki.data_ptr(), zn.data_ptr(), zf.data_ptr()
        cam.H, cam.W = int(H), int(W)
        prec = _lib.PRECISIONS[self.precision]
        L = _lib.lib()
        self._poll_status()
        seed = self._next_seed()
        if self.binding == "torch_ops":
            from . import ops as _ops
            maps_t, poses_t, focal_t, c_t, latent_t, linz_t = _keep
            rgb, depth = _ops.load().render_image(maps_t, poses_t, focal_t, c_t, latent_t, linz_t, packed, E, Ki, zn, zf, int(H), int(W), sc.image_w,
                                                  sc.image_h, sc.feature_padding, sc.num_freqs, sc.freq_factor, cfg.n_candidates, cfg.n_samples,
                                                  cfg.n_gaussian, cfg.depth_diff_max, bool(self.white_bkgd), prec,
                                                  seed - (1 << 64) if seed >= (1 << 63) else seed, self._status_word(dev))
        else:
            ws = torch.empty(int(L.diner_render_image_workspace_floats(SB, int(H), int(W), K, sc.NV, prec)), dtype=torch.float32, device=dev)
            rgb = torch.empty((SB, H * W, 3), dtype=torch.float32, device=dev)
            depth = torch.empty((SB, H * W), dtype=torch.float32, device=dev)
            check(L.diner_render_image(C.byref(sc), _ptr(packed), C.byref(cam), C.byref(cfg), int(bool(self.white_bkgd)), prec, seed,
                                       _ptr(ws), None, _ptr(rgb), _ptr(depth), None, _ptr(self._status_word(dev)), _stream(dev)), "diner_render_image")
        self._after_launch(dev, sync=self.finite_check != "off")    # once per frame: a NaN image never leaves this function
        rgb = rgb.view(SB, H, W, 3).permute(0, 3, 1, 2)
        if return_depth:
            return rgb, depth.view(SB, H, W, 1).permute(0, 3, 1, 2)
        return rgb

    def _forward_train(self, model, rays, want_weights, noise=None, z_samples=None):
        """Training path (reference DINER.calc_losses, src/models/diner.py:217-290): sampler under no_grad
        (src/models/nerf_renderer.py:65), then the differentiable point evaluation + compositing of
        diner_amd/training.py (HIP building blocks; gradients to the MLP parameters and encoder.latent)."""
        from . import training
        r = self._check_rays(rays)
        SB, NR, _ = r.shape
        K = int(self.n_samples)
        with torch.no_grad():
            if z_samples is not None:
                z = _f32c(z_samples)
            else:
                z = self._sample(r, model, K, self.n_depth_candidates, self.n_gaussian, 0.05, noise, None)["z"]
            sc, _keep = self._scene(model, need_latent=False)
        assert SB == sc.SB
        lat = model.encoder.latent
        assert lat.shape[:2] == (sc.SB, sc.NV) and lat.shape[2] == 512
        sc.C, sc.h, sc.w = int(lat.shape[2]), int(lat.shape[3]), int(lat.shape[4])
        rgb, depth, weights = training.render_with_grad(self, model, r, z, sc)
        return RenderOutput(fine=self._format_outputs(weights, rgb, depth, want_weights=want_weights))

    # alias asked for by the north_star text; the reference itself has no render_rays
    render_rays = forward

    def _format_outputs(self, weights, rgb, depth, want_weights):
        out = RenderOutput(rgb=rgb, depth=depth)
        if want_weights:
            out.weights = weights
        return out
