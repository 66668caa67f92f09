"""CPU tests (-m "not gpu"): the oracle against its committed golden vectors and against the independent dense
autograd oracle; host-side camera conventions.  No GPU, no HIP compute calls."""
import os

import numpy as np
import pytest
import torch

import cases
from golden import make_golden
from sigman_release_amd import cameras, synthetic

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name", make_golden.GOLDEN_CASES)
def test_oracle_reproduces_golden(name, oracle):
    """Bit-exact for the integer artefacts AND the forward images: in its default (reproducible) mode the forward uses no libm function whose
    last bit could differ between hosts -- sqrt and division are correctly rounded, the exponent is an explicit FMA chain, 2^x an fp64
    polynomial of correctly rounded steps (gsplat_ref.c header).  Gradients (fp64 sums over OpenMP-scheduled tiles) to 1e-5 of their largest entry."""
    want = np.load(os.path.join(HERE, "golden", f"{name}.npz"))
    got = make_golden.make(name)
    assert sorted(got) == sorted(want.files), name
    for k in ("radii", "rect", "tiles_touched", "keys", "point_list", "ranges", "n_contrib"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=f"{name}:{k}")
    for k in ("color", "depth", "alpha", "final_T"):
        np.testing.assert_array_equal(got[k].view(np.uint32), want[k].view(np.uint32), err_msg=f"{name}:{k}")
    for k in want.files:
        if k.startswith("g_"):
            scale = max(np.abs(want[k]).max(), 1e-20)
            assert np.abs(got[k] - want[k]).max() / scale <= 1e-5, f"{name}:{k}"


def test_exp2_cr_is_correctly_rounded(oracle):
    """ref_exp2_cr (the fp64 polynomial whose steps preprocess.hip repeats) against numpy's exp2 in double, rounded to fp32: identical on
    two million exponents over the range the alpha test can reach and beyond, plus exact powers of two."""
    rng = np.random.default_rng(5)
    x = np.concatenate([-rng.random(1_000_000, np.float32) * 9.0, -rng.random(1_000_000, np.float32) * 40.0,
                        -np.arange(0, 127, dtype=np.float32), -np.float32(2.0) ** -np.arange(1, 60, dtype=np.float32)]).astype(np.float32)
    got = oracle.exp2_cr(x)
    want = np.exp2(x.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int((got != want).sum())
    assert np.array_equal(oracle.exp2_cr(-np.arange(0, 20, dtype=np.float32)), (0.5 ** np.arange(0, 20)).astype(np.float32))


def test_alpha_test_is_a_threshold_on_the_exponent(oracle):
    """What the HIP kernels rely on: for every opacity there is ONE fp32 p* <= 0 with  (alpha test passes at p)  <=>  (p >= p*)  for all
    p <= 0.  ref_alpha_threshold finds it by bisection; here the equivalence is checked against the DIRECT evaluation of the test (the one
    the oracle's renderer uses) at p*, at its neighbours, and at random exponents -- for opacities from below 1/255 to above 1, the
    threshold itself, 0.99 and its neighbours."""
    rng = np.random.default_rng(11)
    c = np.float32(1.0) / np.float32(255.0)
    op = np.concatenate([rng.random(200_000, np.float32), np.float32(10.0) ** -(rng.random(50_000, np.float32) * 2.5).astype(np.float32),
                         np.nextafter(c, np.float32(0), dtype=np.float32)[None], c[None], np.nextafter(c, np.float32(1), dtype=np.float32)[None],
                         np.array([0.0, 1.0, 0.99, np.nextafter(np.float32(0.99), np.float32(1)), np.nextafter(np.float32(0.99), np.float32(0)), 1.5, 250.0, np.nan, -0.3], np.float32)]).astype(np.float32)
    ps = oracle.alpha_threshold(op)
    never = ~(op >= c)                                     # (NaN included)
    assert np.isposinf(ps[never]).all() and np.isfinite(ps[~never]).all() and (ps[~never] <= 0).all()
    o, p = op[~never], ps[~never]
    assert oracle.alpha_test(o, p).all()                                                        # p* passes
    below = np.nextafter(p, np.float32(-np.inf), dtype=np.float32)
    assert not oracle.alpha_test(o, below).any()                                                # its lower neighbour does not
    for k in range(4):                                                                          # monotone: random exponents on either side
        q = (-rng.random(o.size, np.float32) * 9.0).astype(np.float32)
        np.testing.assert_array_equal(oracle.alpha_test(o, q), q >= p)
    assert not oracle.alpha_test(op[never], np.zeros(int(never.sum()), np.float32)).any()       # never-passing opacities fail even at p = 0


@pytest.mark.parametrize("name", ["cloud_precomp", "opaque_stack", "humanoid_20k_256"])
def test_reproducible_and_published_exponent_agree_up_to_threshold_decisions(name, oracle):
    """The oracle's two arithmetic forms of the per-visit exponent (gsplat_ref.c header: 0 = explicit FMA chain on the pre-scaled conic +
    correctly rounded exp2, the default the HIP kernels are pinned against; 1 = the published expression left to right + libm expf) are the
    same function up to a few ulp: identical integer artefacts, images equal to 1e-5 except where an alpha within ~1e-6 of 1/255 (or a T
    within rounding of 1e-4) falls on the other side -- at most a handful of pixels, each moved by less than one alpha step."""
    inp, st = cases.CASES[name]()
    sv = cases.single_view(st)
    H, W = st["image_height"], st["image_width"]
    gC, gD, gA = cases.grads_for(H, W)
    res = []
    try:
        for mode in (0, 1):
            oracle.set_alpha_mode(mode)
            r = oracle.forward(**inp, **sv)
            res.append((r, oracle.backward(r, gC, gD, gA)))
    finally:
        oracle.set_alpha_mode(0)
    (r0, g0), (r1, g1) = res
    np.testing.assert_array_equal(r0.keys, r1.keys)
    np.testing.assert_array_equal(r0.ranges, r1.ranges)
    off = np.zeros((H, W), bool)
    for a, b in ((r0.color, r1.color), (r0.depth, r1.depth), (r0.alpha, r1.alpha)):
        e = np.abs(a - b)
        off |= (e > 1e-5).reshape(-1, H, W).any(0)
        assert e.max() <= 4.0 / 255.0
    flips = int((r0.n_contrib != r1.n_contrib).sum())
    assert off.sum() <= 4 and flips <= 4, (int(off.sum()), flips)
    for k in g0:
        scale = max(np.abs(g0[k]).max(), 1e-20)
        assert np.abs(g0[k] - g1[k]).max() / scale <= (1e-5 if off.sum() == 0 else 2e-2), k


def _dense_vs_c(oracle, inp, st, H, W):
    from oracle import dense_oracle
    sv = cases.single_view(st)
    r = oracle.forward(**inp, **sv)
    gC, gD, gA = cases.grads_for(H, W)
    g = oracle.backward(r, gC, gD, gA)
    tin = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in inp.items()}
    m2d = torch.zeros(r.P, 3, dtype=torch.float64, requires_grad=True)
    out = dense_oracle.render(**tin, means2D=m2d, **sv)
    loss = (out["color"] * torch.tensor(gC, dtype=torch.float64)).sum() + (out["depth"] * torch.tensor(gD, dtype=torch.float64)).sum() \
        + (out["alpha"] * torch.tensor(gA, dtype=torch.float64)).sum()
    loss.backward()
    return r, g, out, tin, m2d


@pytest.mark.parametrize("name", ["cloud_precomp", "cloud_sh3", "cull_and_clamp", "opaque_stack", "cloud_precomp_ragged"])
def test_c_oracle_matches_dense_autograd(name, oracle):
    """The hand-derived backward of gsplat_ref.c against torch.autograd through an independent dense evaluation."""
    inp, st = cases.CASES[name]()
    H, W = st["image_height"], st["image_width"]
    r, g, out, tin, m2d = _dense_vs_c(oracle, inp, st, H, W)
    assert out["margin"] > 1e-9, "pick another seed: a discrete decision sits on its threshold"
    np.testing.assert_array_equal(out["radii"].numpy(), r.radii)
    np.testing.assert_array_equal(out["rect"].numpy(), r.rect)
    np.testing.assert_array_equal(out["n_contrib"].numpy().astype(np.uint32), r.n_contrib)
    for k, ref_img in (("color", r.color), ("depth", r.depth), ("alpha", r.alpha)):
        assert np.abs(out[k].detach().numpy() - ref_img).max() <= 2e-5, k
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert rel(g["means3D"], tin["means3D"].grad.numpy()) <= 2e-4
    assert rel(g["means2D"], m2d.grad.numpy()) <= 2e-4
    assert rel(g["opacities"].ravel(), tin["opacities"].grad.numpy().ravel()) <= 2e-4
    if "colors_precomp" in inp:
        assert rel(g["colors_precomp"], tin["colors_precomp"].grad.numpy()) <= 2e-4
        assert rel(g["cov3D_precomp"], tin["cov3D_precomp"].grad.numpy()) <= 2e-4
    else:
        assert rel(g["sh"], tin["shs"].grad.numpy()) <= 2e-4
        assert rel(g["scales"], tin["scales"].grad.numpy()) <= 2e-4
        assert rel(g["rotations"], tin["rotations"].grad.numpy()) <= 2e-4


def test_edge_cases_are_exercised(oracle):
    """The case set really hits the discrete rules it claims to hit."""
    inp, st = cases.cull_and_clamp()
    r = oracle.forward(**inp, **cases.single_view(st))
    assert 0 < (r.radii > 0).sum() < r.P                                   # z <= 0.2 cull active
    V = st["viewmatrix"][0].reshape(16)
    pv = inp["means3D"] @ np.array([[V[0], V[1], V[2]], [V[4], V[5], V[6]], [V[8], V[9], V[10]]]) + V[12:15]
    vis = r.radii > 0
    assert (np.abs(pv[vis, 0] / pv[vis, 2]) > 1.3 * st["tanfovx"]).any()   # frustum clamp active on a visible Gaussian
    inp, st = cases.opaque_stack()
    r = oracle.forward(**inp, **cases.single_view(st))
    assert r.final_T.min() < 2e-4 and (r.n_contrib < (r.ranges[:, 1] - r.ranges[:, 0]).max()).any()   # stop rule fired
    assert (inp["opacities"] >= 0.99).any()                                # 0.99 cap reachable


def test_properties(oracle):
    """Size-independent properties: background closure, alpha range, zero-opacity and culled Gaussians inert."""
    inp, st = cases.cloud_precomp(P=500, H=64, W=64, seed=21)
    sv = cases.single_view(st)
    r = oracle.forward(**inp, **sv)
    # colour with bg = C + T*bg ; alpha = 1 - T up to rounding
    sv0 = dict(sv); sv0["bg"] = np.zeros(3, np.float32)
    r0 = oracle.forward(**inp, **sv0)
    np.testing.assert_allclose(r.color, r0.color + r.final_T[None] * sv["bg"][:, None, None], atol=1e-6)
    np.testing.assert_allclose(r.alpha[0], 1.0 - r.final_T, atol=2e-6)
    assert r.alpha.min() >= 0 and r.alpha.max() <= 1.0
    # zero-opacity Gaussians contribute nothing and get zero colour gradient
    inp2 = dict(inp); op = inp["opacities"].copy(); op[::2] = 0.0; inp2["opacities"] = op
    r2 = oracle.forward(**inp2, **sv)
    keep = {k: (v[1::2] if k != "opacities" else v[1::2]) for k, v in inp.items()}
    r3 = oracle.forward(**keep, **sv)
    np.testing.assert_allclose(r2.color, r3.color, atol=1e-6)
    g = oracle.backward(r2, *cases.grads_for(64, 64))
    assert np.abs(g["colors_precomp"][::2]).max() == 0.0
    # permutation invariance (no exact depth ties in this seed)
    perm = np.random.default_rng(0).permutation(500)
    rp = oracle.forward(**{k: v[perm] for k, v in inp.items()}, **sv)
    np.testing.assert_allclose(rp.color, r.color, atol=1e-6)
    np.testing.assert_array_equal(rp.radii, r.radii[perm])


def test_empty_input(oracle):
    inp, st = cases.cloud_precomp(P=1)
    e = {k: v[:0] for k, v in inp.items()}
    r = oracle.forward(**e, **cases.single_view(st))
    assert r.R == 0 and np.allclose(r.color, 1.0) and r.alpha.max() == 0


def test_camera_rig_matches_reference_calibration():
    """Analytic rig == the 90 (R,T) pairs of core/dataset/camera_full_calibration.json (committed as data fixture)."""
    d = np.load(os.path.join(HERE, "golden", "camera_rig.npz"))
    for i in range(90):
        w = cameras.rig_w2c(i)
        assert np.abs(w[:3, :3] - d["R"][i]).max() < 2e-6 and np.abs(w[:3, 3] - d["T"][i]).max() < 2e-6


def test_projection_matrix_constants():
    """P00 = P11 = 2.1484375, P02 = P12 = 0, P32 = 1, z mapping from znear/zfar; tanfov = 512/1100."""
    P = cameras.projection_matrix()
    assert P[0, 0] == np.float32(2.1484375) and P[1, 1] == np.float32(2.1484375)
    assert P[0, 2] == 0 and P[1, 2] == 0 and P[3, 2] == 1
    assert abs(P[2, 2] - 100 / 99.9) < 1e-6 and abs(P[2, 3] + 10 / 99.9) < 1e-6
    assert abs(cameras.TAN_HALF_FOV - 512 / 1100) < 1e-9
    cv, cvp, cp = cameras.make_cameras([30])
    assert np.allclose(cp[0], [0, 0, 2.5], atol=1e-6)
    p = np.array([0.1, 0.2, 0.0, 1.0], np.float32) @ cv[0]             # row vector times cam_view = w2c^T
    assert np.allclose(p[:3], [0.1, -0.2, 2.5], atol=1e-6)


def test_synthetic_humanoid_statistics():
    g = synthetic.humanoid(20000, 1)
    p = g["position"]
    assert -0.90 < p[:, 0].min() < -0.80 and 0.80 < p[:, 0].max() < 0.90
    assert -1.05 < p[:, 1].min() < -0.95 and 0.70 < p[:, 1].max() < 0.80
    assert np.abs(p[:, 2]).max() < 0.16
    R = g["cov3d"]
    assert np.allclose(R @ np.transpose(R, (0, 2, 1)), np.eye(3)[None], atol=1e-5)
    cov = synthetic.covariance_from_gaussians(g)
    assert cov.shape == (20000, 6) and np.isfinite(cov).all() and (cov[:, [0, 3, 5]] > 0).all()
