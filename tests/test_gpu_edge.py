"""GPU edge cases of the render path, checked against the CPU oracle on the same seeded inputs
(no golden needed: the oracle is pinned by tests/test_oracle_golden.py): batches of scenes, 1 and
8 source views, every candidate-count kernel variant, K = 256, ragged tails, empty inputs,
n_gaussian = 0 / = K, rays that miss everything, chunk invariance.  All through the plug-in class
and therefore the C ABI."""
import copy

import numpy as np
import pytest
import torch

from synthetic import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def make(H=24, W=24, NV=3, seed=0, **kw):
    sc = synth.make_scene(H, W, NV, seed=seed, feature_padding=4, **kw)
    w = synth.make_mlp_weights(seed + 1, bias_scale=0.1)
    return sc, w


def run_gpu(sc, w, rays, K, NC, G, noise, dev, precision="f16x3", want_weights=True):
    from diner_amd import NeRFRendererDGS
    from synthetic.model_stub import model_from_scene
    m = model_from_scene(sc, w, device=dev)
    r = NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G, white_bkgd=sc.white_bkgd)
    r.precision = precision
    with torch.no_grad():
        out = r(m, T(rays, dev), want_weights=want_weights, noise=None if noise is None else tuple(None if n is None else T(n, dev) for n in noise))
    return out, r, m


def run_oracle(sc, w, rays, K, NC, G, noise):
    from oracle.oracle import Oracle
    return Oracle(sc, w).render(rays, NC, K, G, noise, white_bkgd=sc.white_bkgd)


def agree(out_rgb, ref_rgb, frac=0.97, tol=1e-4):
    ok = np.abs(out_rgb - ref_rgb).max(-1) <= tol
    assert ok.mean() >= frac, f"{(~ok).sum()} of {ok.size} rays differ by more than {tol}"


@pytest.mark.parametrize("NV", [1, 2, 8])
def test_number_of_views(dev, NV):
    sc, w = make(NV=NV, seed=20 + NV)
    rays = sc.target_rays()[:, ::5]
    K, NC, G = 24, 300, 8
    noise = synth.make_noise(rays.shape[1], NC, G, K, seed=1)
    out, _, _ = run_gpu(sc, w, rays, K, NC, G, noise, dev)
    ref = run_oracle(sc, w, rays, K, NC, G, noise)
    agree(out.fine.rgb.cpu().numpy()[0], ref["rgb"])


@pytest.mark.parametrize("NC,K,G", [(64, 8, 2), (250, 16, 4), (1000, 256, 96), (1500, 40, 15), (2048, 32, 8), (4096, 24, 6)])
def test_candidate_and_sample_counts(dev, NC, K, G):
    """NC selects the sampler's candidates-per-lane variant (4 / 16 / 32); K = 256 is cfg5's sample count."""
    sc, w = make(seed=30)
    rays = sc.target_rays()[:, ::9]
    noise = synth.make_noise(rays.shape[1], NC, G, K, seed=2)
    out, _, _ = run_gpu(sc, w, rays, K, NC, G, noise, dev)
    ref = run_oracle(sc, w, rays, K, NC, G, noise)
    z = out.fine.weights  # shape check only
    assert tuple(z.shape) == (1, rays.shape[1], K)
    agree(out.fine.rgb.cpu().numpy()[0], ref["rgb"], frac=0.95)


def test_unsupported_candidate_count_raises(dev):
    sc, w = make(seed=31)
    rays = sc.target_rays()[:, :4]
    with pytest.raises(NotImplementedError):
        run_gpu(sc, w, rays, 8, 4097, 2, None, dev)


# NV = 3: the view-sequential form of the f16x3 point kernel; NV = 4, 8: its views-in-tile form (one / two view groups per sub-tile)
@pytest.mark.parametrize("NV", [3, 4, 8])
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_batch_of_two_scenes(dev, precision, NV):
    """SB = 2 (the reference trains with batches of scenes, pixelnerf.py:68): every scene of the batch must
    equal its own single-scene render."""
    a, w = make(seed=40, NV=NV)
    b, _ = make(seed=41, NV=NV, bg_sigma_zero=True)
    both = copy.copy(a)
    for name in ("poses", "focal", "c", "depths", "depths_std", "normals", "latent"):
        setattr(both, name, np.concatenate([getattr(a, name), getattr(b, name)], 0))
    ra, rb = a.target_rays()[:, ::7], b.target_rays()[:, 3::7]
    n = min(ra.shape[1], rb.shape[1])
    rays = np.concatenate([ra[:, :n], rb[:, :n]], 0)
    K, NC, G = 16, 200, 5
    na, nb = synth.make_noise(n, NC, G, K, seed=3), synth.make_noise(n, NC, G, K, seed=4)
    noise = tuple(np.stack([x, y]) for x, y in zip(na, nb))
    out, _, _ = run_gpu(both, w, rays, K, NC, G, noise, dev, precision=precision)
    rgb = out.fine.rgb.cpu().numpy()
    assert rgb.shape == (2, n, 3)
    agree(rgb[0], run_oracle(a, w, ra[:, :n], K, NC, G, na)["rgb"])
    agree(rgb[1], run_oracle(b, w, rb[:, :n], K, NC, G, nb)["rgb"])


@pytest.mark.parametrize("NV", [3, 4, 8])
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_ragged_tail_and_tiny_inputs(dev, precision, NV):
    """Point counts that are not a multiple of the 64-point tile (nor of the 16-point sub-tile of the views-in-tile form), down to
    a single ray."""
    sc, w = make(seed=50, NV=NV)
    K, NC, G = 40, 200, 15  # reference defaults for K and G (configs/train_diner_facescape.yaml:61-66)
    for n_rays in (1, 7, 33):
        rays = sc.target_rays()[:, 100:100 + n_rays]
        noise = synth.make_noise(n_rays, NC, G, K, seed=5)
        out, _, _ = run_gpu(sc, w, rays, K, NC, G, noise, dev, precision=precision)
        ref = run_oracle(sc, w, rays, K, NC, G, noise)
        np.testing.assert_allclose(out.fine.rgb.cpu().numpy()[0], ref["rgb"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(out.fine.depth.cpu().numpy()[0], ref["depth"], rtol=0, atol=2e-4)


def test_empty_ray_batch(dev):
    sc, w = make(seed=51)
    rays = sc.target_rays()[:, :0]
    out, _, _ = run_gpu(sc, w, rays, 8, 64, 2, None, dev)
    assert tuple(out.fine.rgb.shape) == (1, 0, 3) and tuple(out.fine.depth.shape) == (1, 0)
    assert tuple(out.fine.weights.shape) == (1, 0, 8)


@pytest.mark.parametrize("G", [0, 12])
def test_gaussian_count_extremes(dev, G):
    """n_gaussian = 0 (no gaussian draws, nerf_renderer.py:181) and n_gaussian = n_samples (nothing short-listed)."""
    sc, w = make(seed=52)
    K, NC = 12, 128
    rays = sc.target_rays()[:, ::11]
    noise = synth.make_noise(rays.shape[1], NC, max(G, 1), K, seed=6)
    noise = (noise[0], noise[1][:, :G] if G else np.zeros((rays.shape[1], 0), np.float32), noise[2])
    out, _, _ = run_gpu(sc, w, rays, K, NC, G, noise if G else (noise[0], None, noise[2]), dev)
    ref = run_oracle(sc, w, rays, K, NC, G, (noise[0], noise[1] if G else np.zeros((rays.shape[1], 1), np.float32), noise[2]))
    agree(out.fine.rgb.cpu().numpy()[0], ref["rgb"])


def test_rays_that_miss_every_surface(dev):
    """No candidate has a likelihood: all K samples come from the uniform fill-up (nerf_renderer.py:376-396)."""
    sc, w = make(seed=53)
    rays = sc.target_rays()[:, ::13].copy()
    rays[..., 3:6] = -rays[..., 3:6]  # look away from the object
    K, NC, G = 16, 100, 4
    noise = synth.make_noise(rays.shape[1], NC, G, K, seed=7)
    from diner_amd import NeRFRendererDGS
    from synthetic.model_stub import model_from_scene
    from oracle.oracle import Oracle
    m = model_from_scene(sc, w, device=dev)
    r = NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G)
    z = r._sample(T(rays, dev), m, K, NC, G, 0.05, tuple(T(n, dev) for n in noise), None)["z"].cpu().numpy()[0]
    orc = Oracle(sc, w)
    zc = orc.sample_coarse(rays, NC, noise[0])
    zd = orc.sample_depthguided(rays, zc, K, G, noise[1])
    assert not zd.any(), "scene construction: these rays should have no likelihood"
    np.testing.assert_array_equal(z, orc.fill_up(rays, zd, noise[2]))


@pytest.mark.parametrize("linz", [True, False])
def test_views_in_tile_equals_view_sequential(dev, linz, monkeypatch):
    """The two forms of the f16x3 point kernel on the same inputs (NV = 4, 8; lin_z maps and per-point lin_z GEMMs): the views-in-tile
    form sums the views in the reference's order like the view-sequential one, so the two agree to the last bits of fp32 (not bit for
    bit: NV = 8 parks a partial sum per group of 4 views).  The sequential form is selected per PROCESS (DINER_F16_NO_VIT is read once), so
    it runs in a child process."""
    import subprocess, sys, tempfile
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.test_gpu_edge import make, T
from diner_amd import NeRFRendererDGS
from synthetic.model_stub import model_from_scene
dev = torch.device("cuda:0")
outs = {}
for NV in (4, 8):
    sc, w = make(seed=60 + NV, NV=NV)
    rays = sc.target_rays()[:, ::5]
    K = 24
    z = np.sort(np.random.RandomState(NV).uniform(sc.near, sc.far, (1, rays.shape[1], K)).astype(np.float32), -1)
    m = model_from_scene(sc, w, device=dev)
    r = NeRFRendererDGS(n_samples=K, n_depth_candidates=64, n_gaussian=4)
    r.linz_maps = %s
    with torch.no_grad():
        outs[str(NV)] = r.render_points(m, T(rays, dev), T(z, dev)).cpu().numpy()
np.savez(sys.argv[1], **outs)
""" % (str(root), "True" if linz else "False")
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for name, env in (("vit", {}), ("seq", {"DINER_F16_NO_VIT": "1"})):
            import os
            e = dict(os.environ); e.update(env)
            out = str(Path(td) / (name + ".npz"))
            p = subprocess.run([sys.executable, "-c", code, out], capture_output=True, text=True, timeout=600, cwd=str(root), env=e)
            assert p.returncode == 0, p.stderr[-2000:]
            res[name] = dict(np.load(out))
    for k in res["vit"]:
        a, b = res["vit"][k], res["seq"][k]
        assert np.isfinite(a).all() and np.isfinite(b).all()
        d = np.abs(a - b)
        assert float(d[..., :3].max()) <= 2e-6 and float((d[..., 3] / np.maximum(1.0, np.abs(b[..., 3]))).max()) <= 2e-6, (k, float(d.max()))


def test_chunk_invariance(dev):
    """Rendering a ray batch in two calls equals one call (sub-batch invariance)."""
    sc, w = make(seed=54)
    rays = sc.target_rays()[:, ::3]
    n = rays.shape[1]
    K, NC, G = 16, 200, 5
    noise = synth.make_noise(n, NC, G, K, seed=8)
    full, _, _ = run_gpu(sc, w, rays, K, NC, G, noise, dev, precision="fp32")
    h = n // 2 + 3
    a, _, _ = run_gpu(sc, w, rays[:, :h], K, NC, G, tuple(x[:h] for x in noise), dev, precision="fp32")
    b, _, _ = run_gpu(sc, w, rays[:, h:], K, NC, G, tuple(x[h:] for x in noise), dev, precision="fp32")
    np.testing.assert_array_equal(torch.cat([a.fine.rgb, b.fine.rgb], 1).cpu().numpy(), full.fine.rgb.cpu().numpy())
    np.testing.assert_array_equal(torch.cat([a.fine.depth, b.fine.depth], 1).cpu().numpy(), full.fine.depth.cpu().numpy())


def test_cache_invalidation_on_new_encode(dev):
    """The packed copies are keyed on tensor identity/version: writing new maps in place must be picked up."""
    sc, w = make(seed=55)
    rays = sc.target_rays()[:, ::8]
    K, NC, G = 16, 200, 5
    noise = synth.make_noise(rays.shape[1], NC, G, K, seed=9)
    out1, r, m = run_gpu(sc, w, rays, K, NC, G, noise, dev)
    with torch.no_grad():
        m.encoder.latent.mul_(0.5)  # in-place change -> _version bump
        out2 = r(m, T(rays, dev), noise=tuple(T(n, dev) for n in noise))
    sc2 = copy.copy(sc)
    sc2.latent = sc.latent * np.float32(0.5)
    ref2 = run_oracle(sc2, w, rays, K, NC, G, noise)
    agree(out2.fine.rgb.cpu().numpy()[0], ref2["rgb"])
    assert not torch.equal(out1.fine.rgb, out2.fine.rgb)


def test_cache_follows_rebound_tensors(dev):
    """The reference re-binds ``encoder.latent/depths/...`` to FRESH tensors on every encode()
    (src/models/image_encoder.py:214-218,271-272).  A fresh tensor may land on the address the caching allocator just
    freed, with ``_version`` 0 again -- the pack cache must key on the tensor object, not on its address."""
    sc, w = make(seed=56)
    rays = sc.target_rays()[:, ::8]
    K, NC, G = 16, 200, 5
    noise = synth.make_noise(rays.shape[1], NC, G, K, seed=9)
    out, r, m = run_gpu(sc, w, rays, K, NC, G, noise, dev)
    nz = tuple(T(n, dev) for n in noise)
    reused = 0
    prev = out.fine.rgb.clone()
    for it in range(1, 5):
        scale = np.float32(1.0 + 0.5 * it)
        old_ptr = m.encoder.latent.data_ptr()
        m.encoder.latent = None                      # the caller drops the old latent first ...
        torch.cuda.synchronize()
        fresh = T(sc.latent * scale, dev)            # ... and the new one usually lands on its address
        reused += int(fresh.data_ptr() == old_ptr and fresh._version == 0)
        m.encoder.latent = fresh
        with torch.no_grad():
            o = r(m, T(rays, dev), noise=nz)
        sc2 = copy.copy(sc)
        sc2.latent = sc.latent * scale
        agree(o.fine.rgb.cpu().numpy()[0], run_oracle(sc2, w, rays, K, NC, G, noise)["rgb"])
        assert not torch.equal(o.fine.rgb, prev)
        prev = o.fine.rgb.clone()
    # same for the depth maps (new surface -> different samples)
    old_ptr = m.encoder.depths.data_ptr()
    sc3 = copy.copy(sc2)
    sc3.depths = (sc.depths * np.float32(1.01)).astype(np.float32)
    m.encoder.depths = None
    torch.cuda.synchronize()
    m.encoder.depths = T(sc3.depths, dev)
    reused += int(m.encoder.depths.data_ptr() == old_ptr)
    with torch.no_grad():
        o = r(m, T(rays, dev), noise=nz)
    agree(o.fine.rgb.cpu().numpy()[0], run_oracle(sc3, w, rays, K, NC, G, noise)["rgb"])
    print(f"address reuse happened in {reused} of 5 re-bindings")


def test_cache_holds_no_strong_reference(dev):
    import gc
    import weakref
    sc, w = make(seed=57)
    rays = sc.target_rays()[:, ::16]
    out, r, m = run_gpu(sc, w, rays, 8, 64, 2, None, dev)
    ref = weakref.ref(m.encoder.latent)
    m.encoder.latent = None
    gc.collect()
    assert ref() is None, "the renderer's pack cache kept the source latent alive"


def test_linz_maps_budget_and_memory_report(dev):
    """The lin_z feature maps (3x the latent) are only built within ``linz_maps_max_bytes``; beyond it lin_z stays a per-point
    GEMM with the same results, and ``memory_report()`` says what the renderer holds."""
    sc, w = make(seed=60)
    rays = sc.target_rays()[:, 50:114]
    K, NC, G = 16, 128, 5
    noise = synth.make_noise(rays.shape[1], NC, G, K, seed=6)
    from diner_amd import NeRFRendererDGS
    from synthetic.model_stub import model_from_scene
    m = model_from_scene(sc, w, device=dev)
    outs = []
    for limit in (64 << 30, 0):
        r = NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G, white_bkgd=sc.white_bkgd)
        r.linz_maps_max_bytes = limit
        with torch.no_grad():
            outs.append(r(m, T(rays, dev), noise=tuple(T(n, dev) for n in noise)).fine.rgb.cpu().numpy())
        rep = r.memory_report(m, rays_per_call=4096)
        assert (rep["cached"]["linz_maps"] > 0) == (limit > 0)
        if limit:
            assert rep["cached"]["linz_maps"] == 3 * rep["cached"]["latent_nhwc"]
        assert rep["cached"]["total"] == sum(v for k, v in rep["cached"].items() if k != "total")
        assert rep["per_call"]["workspace"] > 0 and rep["training_step"]["saved_activations"] > 0
    np.testing.assert_allclose(outs[0], outs[1], rtol=0, atol=1e-4)


def test_second_device_uses_its_own_launch_state():
    """the raised dynamic-LDS limit of a kernel and the CU count that sizes the persistent grids belong to the
    DEVICE (diner_amd/csrc/api.hip: device_cus / ensure_dynamic_lds, cached per device ordinal), not to the process: a process that
    renders on cuda:1 after cuda:0 must get the same frame there.  Needs two visible GPUs (the 1-GPU boxes skip it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("only one GPU visible")
    from diner_amd import NeRFRendererDGS
    from synthetic.model_stub import model_from_scene
    sc, w = make(seed=3)
    rays = sc.target_rays()[:, ::5]
    K, NC, G = 24, 300, 8
    noise = synth.make_noise(rays.shape[1], NC, G, K, seed=1)
    outs = []
    for d in (0, 1, 0):
        dv = torch.device("cuda", d)
        with torch.cuda.device(dv):
            m = model_from_scene(sc, w, device=dv)
            for prec in ("f16x3", "fp32"):
                r = NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G, white_bkgd=sc.white_bkgd)
                r.precision = prec
                with torch.no_grad():
                    o = r(m, T(rays, dv), want_weights=True, noise=tuple(T(n, dv) for n in noise))
                outs.append((d, prec, o.fine.rgb.cpu(), o.fine.depth.cpu()))
    for d, prec, rgb, depth in outs[2:]:
        ref = next(o for o in outs if o[1] == prec)
        assert torch.equal(rgb, ref[2]) and torch.equal(depth, ref[3]), (d, prec)
