"""GPU parity: the HIP path (through the C ABI, via the plug-in class) against
(i) the golden vectors the unmodified reference produced on the CPU (tests/golden) and
(ii) the CPU oracle on the same seeded inputs.

Bars (north_star): <= 1e-4 abs on RGB and on sigma (sigma is unbounded above: the bar is applied
relative to max(1, sigma/12), SURVEY.md §7 "FP32 parity"); sample positions within a few fp32 ulps;
discontinuous decisions (nearest texel, masks, top-K cut) may flip only where the deciding
quantity is within rounding of its threshold (same rule as tests/test_oracle_golden.py).
"""
import numpy as np
import pytest
import torch

from tests.test_oracle_golden import _check_rgbsigma, soft_shortlist_mismatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


_models = {}


def model_for(g, dev):
    from synthetic.model_stub import model_from_scene
    if g.name not in _models:
        _models[g.name] = model_from_scene(g.scene, g.weights, device=dev)
    return _models[g.name]


PRECISIONS = ["fp32", "f16x3", "f16x3-gemm"]  # f16x3 = lin_z hoisted to feature maps; -gemm = per-point lin_z GEMMs


def renderer_for(g, precision="f16x3"):
    from diner_amd import NeRFRendererDGS
    r = NeRFRendererDGS(n_samples=g.K, n_depth_candidates=g.NC, n_gaussian=g.G, white_bkgd=g.scene.white_bkgd)
    r.precision = precision.split("-")[0]
    r.linz_maps = not precision.endswith("-gemm")
    return r


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_library_loaded_is_in_tree():
    from diner_amd import _lib
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert str(_lib.LIB_PATH) in maps


def test_sample_coarse(golden, dev):
    r = renderer_for(golden)
    z = r.sample_coarse(T(golden.rays, dev), n_coarse=golden.NC, u_coarse=T(golden.noise[0], dev)).cpu().numpy()[0]
    np.testing.assert_allclose(z, golden["z_cand"], rtol=0, atol=2.5e-7 * float(golden.rays[0, 0, 7]))


def test_likelihood_and_shortlist(golden, dev):
    r = renderer_for(golden)
    m = model_for(golden, dev)
    noise = tuple(T(n, dev)[None] for n in golden.noise)
    z_dg, internals = r.sample_depthguided(T(golden.rays, dev), m, golden.K, golden.NC, n_gaussian=golden.G, noise=noise,
                                           z_cand=T(golden["z_cand"], dev)[None], return_internals=True)
    L = internals["likelihood"].cpu().numpy()[0]
    ref = golden["pt_likelihood"]
    np.testing.assert_allclose(L, ref, rtol=0, atol=1.2e-7)  # erf: ocml vs Sleef, <= 2 ulp near 1
    flips = (L == 0) != (ref == 0)
    assert flips.mean() <= 2e-3 and max(L[flips].max(initial=0), ref[flips].max(initial=0)) <= 1.2e-7
    z = z_dg.cpu().numpy()[0]
    keep = golden.K - golden.G
    assert soft_shortlist_mismatch(z, golden["z_dg"], golden["z_cand"], L, keep, soft=2.4e-7) == []
    solid, firm = L.max(-1) > 1e-5, golden.firm_rays
    zg, rg = np.sort(z[:, keep:], -1), np.sort(golden["z_dg"][:, keep:], -1)
    np.testing.assert_allclose(zg[solid & firm], rg[solid & firm], rtol=0, atol=2e-5)
    # rays with erf-last-ulp candidates: one "non-zero" weight more or less changes M in the weighted std's (M-1)/M
    np.testing.assert_allclose(zg[solid & ~firm], rg[solid & ~firm], rtol=0, atol=1e-3)
    assert np.array_equal(np.any(z[:, :keep] != 0, axis=1) | (golden.K == golden.G), golden["hit"] | (golden.K == golden.G))


def test_fill_up(golden, dev):
    r = renderer_for(golden)
    z = r.fill_up_uniform_samples(T(golden["z_dg"], dev)[None], T(golden.rays, dev), u_fill=T(golden.noise[2], dev))
    np.testing.assert_array_equal(z.cpu().numpy()[0], golden["z_fill"])


@pytest.mark.parametrize("precision", PRECISIONS)
def test_points_rgbsigma_vs_reference(golden, dev, precision):
    r = renderer_for(golden, precision)
    m = model_for(golden, dev)
    with torch.no_grad():
        out = r.render_points(m, T(golden.rays, dev), T(golden["z_fill"], dev)[None]).cpu().numpy()[0]
    _check_rgbsigma(out, golden["rgbsigma"])


def test_points_rgbsigma_repeated_renders(golden, dev):
    """The same golden 24 times in a row, per-point lin_z GEMMs (the mode in which the FLAT-store hazard of round 3 showed: a rare
    wrong sample in about one render of three): every render within the parity bar, and all of them bit-identical."""
    r = renderer_for(golden, "f16x3-gemm")
    m = model_for(golden, dev)
    rays, z = T(golden.rays, dev), T(golden["z_fill"], dev)[None]
    first = None
    with torch.no_grad():
        for _ in range(24):
            out = r.render_points(m, rays, z).cpu().numpy()[0]
            _check_rgbsigma(out, golden["rgbsigma"])
            if first is None:
                first = out
            np.testing.assert_array_equal(out, first)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_points_rgbsigma_vs_oracle(golden, dev, precision):
    from oracle.oracle import Oracle
    r = renderer_for(golden, precision)
    m = model_for(golden, dev)
    rays = golden.rays[0]
    z = golden["z_fill"]
    with torch.no_grad():
        out = r.render_points(m, T(golden.rays, dev), T(z, dev)[None]).cpu().numpy()[0]
    xyz = rays[:, None, :3] + z[..., None] * rays[:, None, 3:6]
    dirs = np.broadcast_to(rays[:, None, 3:6], xyz.shape)
    orc = Oracle(golden.scene, golden.weights).points_forward(xyz.reshape(-1, 3), dirs.reshape(-1, 3)).reshape(out.shape)
    # fp32 mode: same fp32 fma-chain order in both; f16x3: fp32-grade split products.
    # Either way expect agreement well below the 1e-4 bar.
    _check_rgbsigma(out, orc)
    d = np.abs(out - orc)
    print(f"{golden.name} {precision}: max|drgb|={d[..., :3].max():.2e} max|dsigma|={d[..., 3].max():.2e} (sigma max {orc[..., 3].max():.1f})")
    assert d[..., :3].max() < 3e-5


def test_composite(golden, dev):
    r = renderer_for(golden)
    w, rgb, depth = r.composite(None, T(golden.rays, dev), T(golden["z_fill"], dev)[None], rgbsigma=T(golden["rgbsigma"], dev)[None])
    np.testing.assert_allclose(w.cpu().numpy()[0], golden["weights"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(rgb.cpu().numpy()[0], golden["rgb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(depth.cpu().numpy()[0], golden["depth"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_forward_with_injected_samples(golden, dev, precision):
    """north_star parity clause: RGB within 1e-4 of the reference renderer for identical samples."""
    r = renderer_for(golden, precision)
    m = model_for(golden, dev)
    with torch.no_grad():
        out = r(m, T(golden.rays, dev), want_weights=True, z_samples=T(golden["z_fill"], dev)[None])
    np.testing.assert_allclose(out.fine.rgb.cpu().numpy()[0], golden["rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.fine.depth.cpu().numpy()[0], golden["depth"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.fine.weights.cpu().numpy()[0], golden["weights"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_forward_end_to_end_replayed_noise(golden, dev, precision):
    r = renderer_for(golden, precision)
    m = model_for(golden, dev)
    noise = tuple(T(n, dev)[None] for n in golden.noise)
    with torch.no_grad():
        out = r(m, T(golden.rays, dev), want_weights=True, noise=noise)
    rgb, depth = out.fine.rgb.cpu().numpy()[0], out.fine.depth.cpu().numpy()[0]
    # z candidates differ by <= 1 ulp from the golden's (linspace form): isolated rays may pick a
    # different texel / candidate.  Compare the rays whose colour agrees and bound the others.
    ok = np.abs(rgb - golden["rgb"]).max(-1) <= 1e-4
    firm = golden.firm_rays  # rays without erf-last-ulp candidates (tests/conftest.py)
    assert ok[firm].mean() >= 0.95, f"{(~ok[firm]).sum()} of {firm.sum()} firm rays differ"
    np.testing.assert_allclose(depth[ok], golden["depth"][ok], rtol=0, atol=2e-4)
    w = out.fine.weights.cpu().numpy()[0]
    assert np.all(w >= 0) and np.all(w.sum(-1) <= 1 + 1e-5)
    # The non-firm rays: an erf-last-ulp candidate is "non-zero" or not depending on the math library, one kept sample more
    # or less changes m and with it EVERY uniform fill-up sample of the ray (z_j = near + (j+U)(far-near)/m,
    # nerf_renderer.py:376-396), and on these scenes' random radiance field a shifted sample set is a different integral.
    # Not asserted to 1e-4 (the reference disagrees with itself across erf implementations there): reported, and bounded
    # loosely so that a systematic error on these rays (the majority of surface rays at the headline parameters) would show.
    nf = ~firm
    if nf.any():
        d, dd = np.abs(rgb - golden["rgb"]).max(-1)[nf], np.abs(depth - golden["depth"])[nf]
        same = np.abs(np.sort(out.fine.weights.cpu().numpy()[0], -1) - np.sort(golden["weights"], -1)).max(-1)[nf] <= 1e-4
        print(f"{golden.name} {precision}: non-firm rays {nf.sum()}/{nf.size}: |drgb| median {np.median(d):.2e} p90 "
              f"{np.percentile(d, 90):.2e} max {d.max():.2e}; |ddepth| median {np.median(dd):.2e} p90 {np.percentile(dd, 90):.2e} "
              f"max {dd.max():.2e}; rays with the golden's weights {same.mean():.2f}")
        assert np.median(d) <= 0.05 and np.median(dd) <= 0.05 * float(golden.rays[0, 0, 7])


def test_forward_perf_mode_properties(golden, dev):
    """In-kernel Philox noise: no reference to compare with, so check the invariants of the path
    (SURVEY.md §4): finite output, colour in [0,1] (white bkgd keeps it there), weights >= 0 with
    sum <= 1, determinism for a fixed seed, and ray-permutation equivariance."""
    r = renderer_for(golden)
    m = model_for(golden, dev)
    rays = T(golden.rays, dev)
    with torch.no_grad():
        r.seed, r._calls = 7, 0
        a = r(m, rays, want_weights=True)
        r.seed, r._calls = 7, 0
        b = r(m, rays, want_weights=True)
    assert torch.equal(a.fine.rgb, b.fine.rgb) and torch.equal(a.fine.depth, b.fine.depth)
    rgb, w = a.fine.rgb.cpu().numpy(), a.fine.weights.cpu().numpy()
    assert np.isfinite(rgb).all() and rgb.min() >= -1e-5 and rgb.max() <= 1 + 1e-5
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()
