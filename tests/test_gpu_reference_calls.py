"""-m gpu: the hot path driven the way the REFERENCE drives it.

  * test_reference_render_loop_under_autocast: the B x V loop of /root/reference/core/gaussians/gs.py:62-117 restated in shape
    (per-subject `.contiguous().float()`, distCUDA2 -> clamp_min -> sqrt -> repeat -> detach, get_covariance out of PyTorch ops,
    one GaussianRasterizationSettings + GaussianRasterizer per view, keyword call with `zeros_like` means2D inside
    `torch.autocast("cuda", dtype=torch.bfloat16)` as accelerate's `mixed_precision: bf16` (configs/training.yaml:10) arranges,
    clamp, stack, view) -- imported through the `diff_gaussian_rasterization` / `simple_knn` shims, i.e. with gs.py UNCHANGED --
    against the batched `GaussianRenderer.render` and against the CPU oracle, forward and backward.
  * debug=True / prefiltered=True semantics of the settings tuple (SURVEY 8b "Error conventions").
  * overflow bookkeeping of the sync-free mode under the reference's call pattern (B*V forwards before one backward).
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import cases
from sigman_release_amd import cameras, synthetic

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _get_covariance(scaling, rotation):
    """gs.py:17-38 out of plain PyTorch ops (what autograd differentiates in the reference)."""
    L = torch.zeros_like(rotation)
    L[:, 0, 0], L[:, 1, 1], L[:, 2, 2] = scaling[:, 0], scaling[:, 1], scaling[:, 2]
    full = rotation @ (L ** 2) @ rotation.permute(0, 2, 1)
    out = torch.zeros((full.shape[0], 6), dtype=torch.float, device=full.device)
    for k, (i, j) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
        out[:, k] = full[:, i, j]
    return out


def _reference_render(opt, gaussians, cam_view, cam_view_proj, cam_pos, bg_color, autocast_dtype):
    """The loop of gs.py:49-117, through the import shims (same module names the reference imports at gs.py:6-11)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from simple_knn._C import distCUDA2
    device = gaussians["position"].device
    B, V = cam_view.shape[:2]
    tan_half_fov = np.tan(0.5 * opt.FoVy)
    images, alphas = [], []
    for b in range(B):
        means3D = gaussians["position"][b].contiguous().float()
        opacity = gaussians["opacity"][b].contiguous().float()
        scales = gaussians["scale"][b].contiguous().float()
        cov3D = gaussians["cov3d"][b].contiguous().float()
        rgbs = gaussians["rgb"][b].contiguous().float()
        dist2 = torch.clamp_min(distCUDA2(means3D), 0.0000001)
        scales_ = torch.sqrt(dist2)[..., None].repeat(1, 3).detach()
        cov3D = _get_covariance((scales + 1) * scales_, cov3D).reshape(-1, 6)
        for v in range(V):
            raster_settings = GaussianRasterizationSettings(
                image_height=opt.output_size_h, image_width=opt.output_size_w, tanfovx=tan_half_fov, tanfovy=tan_half_fov, bg=bg_color,
                scale_modifier=0.5, viewmatrix=cam_view[b, v].float(), projmatrix=cam_view_proj[b, v].float(), sh_degree=0,
                campos=cam_pos[b, v].float(), prefiltered=False, debug=False)
            rasterizer = GaussianRasterizer(raster_settings=raster_settings)
            with torch.autocast("cuda", dtype=autocast_dtype, enabled=True):
                rendered_image, radii, rendered_depth, rendered_alpha = rasterizer(
                    means3D=means3D, means2D=torch.zeros_like(means3D, dtype=torch.float32, device=device), shs=None,
                    colors_precomp=rgbs, opacities=opacity, cov3D_precomp=cov3D)
            images.append(rendered_image.clamp(0, 1))
            alphas.append(rendered_alpha)
    H, W = opt.output_size_h, opt.output_size_w
    return {"image": torch.stack(images, dim=0).view(B, V, 3, H, W), "alpha": torch.stack(alphas, dim=0).view(B, V, 1, H, W)}


@pytest.mark.parametrize("autocast_dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_reference_render_loop_under_autocast(oracle, autocast_dtype):
    """gs.py:62-117 through the shims inside an autocast region with HALF-PRECISION producer tensors (the decoder's outputs under
    accelerate's mixed precision, autoencoder.py:294-345): the rasterizer must still compute in fp32 -- outputs fp32, equal to the
    batched renderer and to the oracle on the fp32 values of the same inputs -- and route gradients back in the producers' dtype."""
    from sigman_release_amd.renderer import GaussianRenderer
    dev = _dev()
    B, V, P, H, W = 2, 3, 6000, 128, 128
    views = [(30, 37, 65), (45, 0, 85)]
    subj = [synthetic.humanoid(P, 60 + b) for b in range(B)]
    # what the producer hands over: half-precision tensors (values rounded to the autocast dtype)
    half = {k: torch.from_numpy(np.stack([s[k] for s in subj])).to(dev).to(autocast_dtype) for k in ("position", "opacity", "scale", "cov3d", "rgb")}
    g_loop = {k: v.clone().requires_grad_(True) for k, v in half.items()}
    g_batch = {k: v.clone().requires_grad_(True) for k, v in half.items()}
    cams = [cameras.make_cameras(v) for v in views]
    cam_view = torch.from_numpy(np.stack([c[0] for c in cams])).to(dev)
    cam_view_proj = torch.from_numpy(np.stack([c[1] for c in cams])).to(dev)
    cam_pos = torch.from_numpy(np.stack([c[2] for c in cams])).to(dev)
    opt = SimpleNamespace(FoVy=cameras.FOVY, output_size_h=H, output_size_w=W)
    bg = torch.tensor([1, 1, 1], dtype=torch.float32, device=dev)
    out = _reference_render(opt, g_loop, cam_view, cam_view_proj, cam_pos, bg, autocast_dtype)
    assert out["image"].dtype is torch.float32 and out["alpha"].dtype is torch.float32
    with torch.autocast("cuda", dtype=autocast_dtype, enabled=True):
        ref_b = GaussianRenderer(opt).render(g_batch, cam_view, cam_view_proj, cam_pos)
    assert ref_b["image"].dtype is torch.float32
    gsum = torch.randn(B, V, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    (out["image"] * gsum).sum().backward()
    (ref_b["image"] * gsum).sum().backward()
    torch.cuda.synchronize()
    # ---- loop == batched renderer (different launch shapes of the same kernels; the 3-NN / covariance ops differ: PyTorch vs fused)
    assert (out["image"] - ref_b["image"]).abs().max() <= 2e-5 and (out["alpha"] - ref_b["alpha"]).abs().max() <= 2e-5
    for k in ("position", "opacity", "scale", "cov3d", "rgb"):
        a, b = g_loop[k].grad, g_batch[k].grad
        assert a is not None and a.dtype is autocast_dtype and a.shape == half[k].shape, k
        # gradients come back rounded to the producer's dtype: compare at that resolution
        tol = (2e-2 if autocast_dtype is torch.bfloat16 else 4e-3) * float(b.float().abs().max())
        assert float((a.float() - b.float()).abs().max()) <= tol, k
    # ---- loop == oracle on the fp32 values of the half-precision inputs
    for b in range(B):
        s32 = {k: half[k][b].float().cpu().numpy() for k in half}
        dist2 = synthetic.nn_dist2_cpu(s32["position"])
        cov = synthetic.covariance_from_gaussians(dict(position=s32["position"], scale=s32["scale"], cov3d=s32["cov3d"]), dist2)
        for v in range(V):
            r = oracle.forward(s32["position"], s32["opacity"].reshape(P), colors_precomp=s32["rgb"], cov3D_precomp=cov,
                               viewmatrix=cams[b][0][v], projmatrix=cams[b][1][v], campos=cams[b][2][v], bg=np.ones(3, np.float32),
                               tanfovx=cameras.TAN_HALF_FOV, tanfovy=cameras.TAN_HALF_FOV, image_height=H, image_width=W)
            assert np.abs(out["image"][b, v].detach().cpu().numpy() - np.clip(r.color, 0, 1)).max() <= 1e-4
            assert np.abs(out["alpha"][b, v].detach().cpu().numpy() - r.alpha).max() <= 1e-4


def test_reference_loop_fp32_gradients_match_oracle(oracle):
    """The same loop with fp32 producers (autocast still on): per-subject gradients of every input against the oracle, with
    dL/dcov3D chained to dL/dscale through the reference's own PyTorch ops by autograd."""
    dev = _dev()
    B, V, P, H, W = 1, 4, 5000, 112, 96
    views = [(30, 45, 0, 85)]
    subj = [synthetic.humanoid(P, 70)]
    g = {k: torch.from_numpy(np.stack([s[k] for s in subj])).to(dev).requires_grad_(True) for k in ("position", "opacity", "scale", "cov3d", "rgb")}
    cams = [cameras.make_cameras(v) for v in views]
    cam_view, cam_view_proj, cam_pos = [torch.from_numpy(np.stack([c[i] for c in cams])).to(dev) for i in range(3)]
    opt = SimpleNamespace(FoVy=cameras.FOVY, output_size_h=H, output_size_w=W)
    bg = torch.tensor([1, 1, 1], dtype=torch.float32, device=dev)
    out = _reference_render(opt, g, cam_view, cam_view_proj, cam_pos, bg, torch.bfloat16)
    gsum = torch.randn(B, V, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(9)) / (H * W)
    (out["image"] * gsum).sum().backward()
    torch.cuda.synchronize()
    s = subj[0]
    dist2 = synthetic.nn_dist2_cpu(s["position"])
    cov = synthetic.covariance_from_gaussians(s, dist2)
    acc = None
    for v in range(V):
        r = oracle.forward(s["position"], s["opacity"].reshape(P), colors_precomp=s["rgb"], cov3D_precomp=cov, viewmatrix=cams[0][0][v],
                           projmatrix=cams[0][1][v], campos=cams[0][2][v], bg=np.ones(3, np.float32), tanfovx=cameras.TAN_HALF_FOV,
                           tanfovy=cameras.TAN_HALF_FOV, image_height=H, image_width=W)
        gc = gsum[0, v].cpu().numpy() * ((r.color >= 0) & (r.color <= 1))
        gr = oracle.backward(r, gc.astype(np.float32))
        acc = gr if acc is None else {k: acc[k] + gr[k] for k in acc}
    for nm, got, want in (("position", g["position"].grad[0], acc["means3D"]), ("opacity", g["opacity"].grad[0], acc["opacities"]),
                          ("rgb", g["rgb"].grad[0], acc["colors_precomp"])):
        err = np.abs(got.cpu().numpy().reshape(want.shape) - want).max() / max(np.abs(want).max(), 1e-20)
        assert err <= 1e-4, (nm, err)
    s2 = torch.from_numpy(s["scale"]).double().requires_grad_(True)
    R2 = torch.from_numpy(s["cov3d"]).double().requires_grad_(True)
    scale = (s2 + 1) * torch.sqrt(torch.clamp_min(torch.from_numpy(dist2).double(), 1e-7))[:, None]
    Lm = torch.zeros_like(R2)
    Lm[:, 0, 0], Lm[:, 1, 1], Lm[:, 2, 2] = scale[:, 0], scale[:, 1], scale[:, 2]
    full = R2 @ (Lm ** 2) @ R2.permute(0, 2, 1)
    packed = torch.stack([full[:, 0, 0], full[:, 0, 1], full[:, 0, 2], full[:, 1, 1], full[:, 1, 2], full[:, 2, 2]], 1)
    (packed * torch.from_numpy(acc["cov3D_precomp"]).double()).sum().backward()
    for nm, got, want in (("scale", g["scale"].grad[0], s2.grad), ("cov3d", g["cov3d"].grad[0], R2.grad)):
        err = (got.cpu().double() - want).abs().max() / want.abs().max()
        assert err < 2e-4, (nm, float(err))


def test_many_forwards_before_one_backward_keep_their_own_counters():
    """The reference issues B*V (= 64) per-view forwards before ONE backward (gs.py:62-109 then train_vae.py:166).  In explicit
    sync-free mode every pending forward must keep its own pinned count slot: 70 forwards (more than one pool chunk), one of them too
    small for its capacity -> exactly that one is reported, with its own numbers, and the others' gradients are intact."""
    from sigman_release_amd import rasterizer as R
    dev = _dev()
    inp, st = cases.humanoid(P=3000, H=96, W=96, seed=21)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = {k: t(v)[None].requires_grad_(True) for k, v in inp.items()}
    base = R.BatchedRasterizationSettings(96, 96, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]), t(st["projmatrix"]), 0,
                                          t(st["campos"]), 1)
    with torch.no_grad():
        Rn = R.forward_debug(d["means3D"].detach(), d["opacities"].detach(), colors_precomp=d["colors_precomp"].detach(),
                             cov3D_precomp=d["cov3D_precomp"].detach(), settings=base)["num_rendered"]
    bad = 41
    imgs, early = [], []
    i = 0
    while i < 70:
        cap = Rn // 3 if i == bad else Rn + 100 + i                       # distinct capacities: a mixed-up slot would show
        try:
            imgs.append(R.rasterize_gaussians_batched(d["means3D"], None, None, d["colors_precomp"], d["opacities"][..., None], None, None,
                                                      d["cov3D_precomp"], base._replace(max_rendered=cap))[0])
            i += 1
        except RuntimeError as e:                                          # the early report of forward `bad`, raised by a LATER forward
            early.append((i, str(e)))                                      # (that forward did not run: it is re-issued)
            torch.cuda.synchronize()
    assert len(early) <= 1 and all(k > bad and f"exceeds max_rendered {Rn // 3}" in m and "EARLIER" in m for k, m in early), early
    good = sum(x.sum() for k, x in enumerate(imgs) if k != bad)
    good.backward()                                                        # 69 backwards: none of them may raise
    torch.cuda.synchronize()
    g69 = d["means3D"].grad.clone()
    one = R.rasterize_gaussians_batched(d["means3D"], None, None, d["colors_precomp"], d["opacities"][..., None], None, None,
                                        d["cov3D_precomp"], base)[0]
    d["means3D"].grad = None
    one.sum().backward()
    assert (g69 - 69 * d["means3D"].grad).abs().max() <= 1e-4 * g69.abs().max()
    with pytest.raises(RuntimeError, match=f"exceeds max_rendered {Rn // 3}"):
        imgs[bad].sum().backward()                                         # its own backward reports it (again): the data IS truncated
    torch.cuda.synchronize()


def test_overflow_is_reported_without_a_backward():
    """an explicit-capacity forward that never sees a backward must not return a silently truncated image."""
    from sigman_release_amd import rasterizer as R
    dev = _dev()
    inp, st = cases.humanoid(P=3000, H=96, W=96, seed=22)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = {k: t(v)[None] for k, v in inp.items()}
    base = R.BatchedRasterizationSettings(96, 96, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]), t(st["projmatrix"]), 0,
                                          t(st["campos"]), 1)
    args = lambda dd: (dd["means3D"], None, None, dd["colors_precomp"], dd["opacities"][..., None], None, None, dd["cov3D_precomp"])
    with torch.no_grad():
        Rn = R.forward_debug(d["means3D"], d["opacities"], colors_precomp=d["colors_precomp"], cov3D_precomp=d["cov3D_precomp"], settings=base)["num_rendered"]
        # (a) inference: raises in the call itself
        with pytest.raises(RuntimeError, match="exceeds max_rendered"):
            R.rasterize_gaussians_batched(*args(d), base._replace(max_rendered=Rn // 2))
        ok = R.rasterize_gaussians_batched(*args(d), base._replace(max_rendered=Rn + 10))[0]
    # (b) a forward that wants gradients but whose output is dropped: reported by the next forward / an explicit check
    dg = {k: v.clone().requires_grad_(True) for k, v in d.items()}
    trunc = R.rasterize_gaussians_batched(*args(dg), base._replace(max_rendered=Rn // 2))[0]
    del trunc
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="exceeds max_rendered"):
        R.rasterize_gaussians_batched(*args(dg), base._replace(max_rendered=Rn + 10))
    again = R.rasterize_gaussians_batched(*args(dg), base._replace(max_rendered=Rn + 10))[0]      # reported once; the path is usable again
    assert torch.equal(again.detach(), ok)
    dg2 = {k: v.clone().requires_grad_(True) for k, v in d.items()}
    R.rasterize_gaussians_batched(*args(dg2), base._replace(max_rendered=Rn // 2))
    with pytest.raises(RuntimeError, match="exceeds max_rendered"):
        R.check_pending_overflows(block=True)
    R.check_pending_overflows(block=True)


def test_debug_and_prefiltered_flags(tmp_path, monkeypatch):
    """debug=True: sync after every kernel, same results, and a failing call leaves snapshot_fw.dump behind
    (upstream's behaviour); prefiltered=True with a point behind the near plane is an error (upstream traps)."""
    from sigman_release_amd import _cabi
    from sigman_release_amd import rasterizer as R
    import ctypes as C
    dev = _dev()
    monkeypatch.chdir(tmp_path)
    inp, st = cases.cull_and_clamp()                                        # camera inside the cloud: some points have z <= 0.2
    H, W = st["image_height"], st["image_width"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    sv = cases.single_view(st)
    P = inp["means3D"].shape[0]

    def run(debug, prefiltered=False, means=None):
        d = {k: t(v).requires_grad_(True) for k, v in inp.items()}
        rs = R.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(sv["viewmatrix"]), t(sv["projmatrix"]), 0,
                                             t(sv["campos"]), prefiltered, debug)
        m3 = d["means3D"] if means is None else means
        color, radii, depth, alpha = R.GaussianRasterizer(rs)(means3D=m3, means2D=torch.zeros(P, 3, device=dev), opacities=d["opacities"].reshape(P, 1),
                                                              colors_precomp=d["colors_precomp"], cov3D_precomp=d["cov3D_precomp"])
        (color * color).sum().backward()
        torch.cuda.synchronize()
        return color.detach(), d["cov3D_precomp"].grad

    c0, g0 = run(False)
    for _ in range(4):
        c1, g1 = run(True)
        assert torch.equal(c0, c1) and torch.equal(g0, g1)
    assert _cabi.lib().sgr_set_debug(0) == 0, "the debug flag leaked out of the call"
    with pytest.raises(RuntimeError, match="prefiltered"):
        run(False, prefiltered=True)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        run(True, means=torch.zeros(P, 4, device=dev))
    assert os.path.exists(tmp_path / "snapshot_fw.dump")
    snap = torch.load(tmp_path / "snapshot_fw.dump", weights_only=False)
    assert snap[0].shape == (P, 4) and snap[0].device.type == "cpu"


def test_loss_gradient_at_exact_zero_and_one():
    """torch.clamp's backward mask is inclusive -- pixels exactly at 0.0 / 1.0 pass the gradient."""
    from sigman_release_amd.losses import clamped_l1_loss
    dev = _dev()
    H, W = 8, 12
    vals = torch.tensor([0.0, 1.0, -0.0, 0.5, 1.5, -0.25, 1.0, 0.0], device=dev)
    color = vals.repeat(3 * H * W // 8).reshape(1, 3, H, W).clone().requires_grad_(True)
    target = torch.full((1, 3, H, W), 0.25, device=dev)
    mask = (torch.rand(1, 1, H, W, device=dev) > 0.3).float()
    ref_in = color.detach().clone().requires_grad_(True)
    want = 0.7 * ((ref_in.clamp(0, 1) - target) * mask).abs().sum()
    want.backward()
    got = clamped_l1_loss(color, target, mask, 0.7)
    got.backward()
    assert abs(float(got) - float(want)) <= 1e-5 * abs(float(want))
    assert torch.equal(color.grad, ref_in.grad)
    assert float(color.grad[0, 0, 0, 0].abs()) > 0 or float(mask[0, 0, 0, 0]) == 0      # x == 0.0 exactly: gradient passes


def test_count_published_by_copy_is_never_read_torn():
    """Large sync-free launches (> 2048 preprocess workgroups) publish the instance count through an async device-to-host copy instead of
    a kernel store; a copy engine may write the 8-byte word piecewise, so the host must wait for the event instead of polling the word
    (seen as 'num_rendered 9223370937365261137 exceeds max_rendered' in the 2-rank C4 bench).  Automatic-capacity mode polls after every
    forward: 40 forwards must all agree with exact mode."""
    from sigman_release_amd import rasterizer as R
    dev = _dev()
    V, P, H = 12, 60_000, 256
    g = synthetic.humanoid(P, 5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cv, cvp, cp = cameras.make_cameras([30, 37, 45, 53, 65, 85, 0, 8, 10, 20, 40, 50])
    st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), V)
    args = (t(g["position"])[None], None, None, t(g["rgb"])[None], t(g["opacity"])[None], None, None, t(synthetic.covariance_from_gaussians(g))[None])
    with torch.no_grad():
        want = R.rasterize_gaussians_batched(*args, st)[0]
        for it in range(40):
            got = R.rasterize_gaussians_batched(*args, st._replace(max_rendered=-1))[0]
            assert torch.equal(got, want), it
    R.check_pending_overflows(block=True)


@pytest.mark.parametrize("name", ["humanoid_20k_256", "cloud_sh3", "cull_and_clamp"])
def test_cpp_autograd_node_equals_python_node(name):
    """The single-view upstream-signature op exists twice above the same C ABI: as a C++ autograd node (csrc/torch_node.cpp, the default:
    the reference calls it once per view and the Python node's host time exceeded the kernels') and as the Python node
    (rasterizer._RasterizeGaussians).  Same outputs and gradients, bit for bit, over repeated calls (exact first call, sync-free later)."""
    from sigman_release_amd import _cabi
    from sigman_release_amd import rasterizer as R
    dev = _dev()
    node = _cabi.torch_node()
    assert node is not None, "lib/sgr_torch_node.so is not built (make -C sigman_release_amd/csrc)"
    inp, st = cases.CASES[name]()
    H, W = st["image_height"], st["image_width"]
    gC, gD, gA = cases.grads_for(H, W)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    sv = cases.single_view(st)
    rs = R.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), st["scale_modifier"], t(sv["viewmatrix"]), t(sv["projmatrix"]),
                                         st["sh_degree"], t(sv["campos"]), False, False)
    P = inp["means3D"].shape[0]

    def run():
        d = {k: t(v).requires_grad_(True) for k, v in inp.items()}
        means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
        color, radii, depth, alpha = R.GaussianRasterizer(rs)(means3D=d["means3D"], means2D=means2D, opacities=d["opacities"].reshape(P, 1), shs=d.get("shs"),
                                                              colors_precomp=d.get("colors_precomp"), scales=d.get("scales"), rotations=d.get("rotations"),
                                                              cov3D_precomp=d.get("cov3D_precomp"))
        ((color * t(gC)).sum() + (depth * t(gD)).sum() + (alpha * t(gA)).sum()).backward()
        torch.cuda.synchronize()
        return [color.detach(), radii, depth.detach(), alpha.detach(), means2D.grad] + [d[k].grad for k in sorted(d)]

    try:
        a = [run() for _ in range(3)]
        _cabi._node = None                                                  # force the Python node
        b = [run() for _ in range(3)]
    finally:
        _cabi._node = node
    for x in a[1:] + b:
        for u, v in zip(a[0], x):
            assert torch.equal(u, v)


_BUSY = {}


def _busy(dev):
    """Queue ~50 ms of GPU work on the current stream (fp32 matmuls): whatever is queued behind it has NOT run when the host gets its next turn."""
    x = _BUSY.get("x")
    if x is None:
        x = _BUSY["x"] = torch.randn(8192, 8192, device=dev)
    for _ in range(6):
        _BUSY["y"] = x @ x


def _cpp_node_case(dev):
    from sigman_release_amd import rasterizer as R
    inp, st = cases.humanoid(P=7000, H=144, W=144, seed=41)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    sv = cases.single_view(st)
    rs = R.GaussianRasterizationSettings(144, 144, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(sv["viewmatrix"]), t(sv["projmatrix"]), 0, t(sv["campos"]),
                                         False, False)
    P = 7000
    m, o, c = t(inp["means3D"]), t(inp["opacities"]).reshape(P, 1), t(inp["colors_precomp"])
    cov = t(inp["cov3D_precomp"])

    def fwd(cov_, grad=False):
        mm = m.clone().requires_grad_(grad)
        out = R.GaussianRasterizer(rs)(means3D=mm, means2D=torch.zeros_like(m), opacities=o, colors_precomp=c, cov3D_precomp=cov_)
        return out, mm

    def exact(cov_):
        bst = R.BatchedRasterizationSettings(144, 144, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(sv["viewmatrix"])[None], t(sv["projmatrix"])[None], 0,
                                             t(sv["campos"])[None], 1)
        with torch.no_grad():
            return R.rasterize_gaussians_batched(m[None], None, None, c[None], o[None], None, None, cov_[None], bst)[0][0]
    return fwd, exact, cov, P


def test_cpp_node_default_never_returns_a_truncated_image():
    """Default policy of the C++ node (the zero-change drop-in path): the instance count is checked inside every call and a forward that does
    not fit its automatic capacity is re-rendered exactly before anything is returned -- however long the shape has been stable, and however
    much larger the subject suddenly gets.  No error, no warning, the exact image and finite gradients."""
    import warnings
    from sigman_release_amd import _cabi
    dev = _dev()
    node = _cabi.torch_node()
    assert node is not None
    node.reset()
    node.set_count_check("inline")
    fwd, exact, cov, P = _cpp_node_case(dev)
    first = fwd(cov)[0][0].clone()
    assert torch.equal(first, exact(cov))
    for i in range(14):
        assert torch.equal(fwd(cov)[0][0], first)
    assert not node.key_state(0, P, 144, 144)[3], "the default policy must never defer the count check"
    big = cov * 25.0                                                          # > 2x the tile instances of everything seen so far
    want = exact(big)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _busy(dev)                                         # the GPU is busy: the count is NOT there when the launches are queued
        got, mm = fwd(big, grad=True)
        assert torch.equal(got[0], want)
        got[0].sum().backward()
        assert torch.isfinite(mm.grad).all()
        assert torch.equal(fwd(cov)[0][0], first)
    # the pinned count slots are pooled process-wide: many forwards before one backward, repeated, do not keep creating slots
    created0 = node.slot_stats()[0]
    for step in range(4):
        outs = [fwd(cov, grad=True) for _ in range(24)]
        sum(o[0][0].sum() for o in outs).backward()
    torch.cuda.synchronize()
    assert node.slot_stats()[0] - created0 <= 1, node.slot_stats()
    node.reset()


def test_cpp_node_deferred_count_check():
    """OPT-IN steady state of the C++ node (SIGMAN_COUNT_CHECK=deferred / set_count_check("deferred")): once a shape's capacity has been stable
    for 8 calls the instance count is no longer waited for inside the call (the host runs ahead of the GPU like in the batched path); results
    stay identical; a forward that does not fit and whose count is already visible when its launches are queued is re-rendered exactly on
    the spot; one whose count arrives later is reported by its own backward (RuntimeError) or check_pending(), an unrelated next forward only
    warns; the capacity is re-learned."""
    from sigman_release_amd import _cabi
    dev = _dev()
    node = _cabi.torch_node()
    assert node is not None
    node.reset()
    node.set_count_check("deferred")
    try:
        fwd, exact, cov, P = _cpp_node_case(dev)
        first = fwd(cov)[0][0].clone()
        for i in range(14):
            assert torch.equal(fwd(cov)[0][0], first)
        node.check_pending()
        capacity, max_count, stable, deferred = node.key_state(0, P, 144, 144)
        assert deferred and capacity >= 2 * max_count, (capacity, max_count, stable, deferred)
        # gradients through a deferred forward
        (out, mm) = fwd(cov, grad=True)
        out[0].sum().backward()
        assert torch.isfinite(mm.grad).all()
        big = cov * 25.0
        # the count is visible when the call ends (idle GPU, everything drained first): repaired on the spot, exact image
        torch.cuda.synchronize()
        got = fwd(big)[0][0]
        torch.cuda.synchronize()
        if node.key_state(0, P, 144, 144)[3]:                                # (only if the count did not make it in time: then it is a late one)
            with pytest.warns(UserWarning, match="earlier deferred forward"):
                fwd(cov)
        else:
            assert torch.equal(got, exact(big))
        # a forward that needs > 2x the largest count seen while the GPU is busy (count NOT visible in time): the next call only warns ...
        node.reset()                                                            # (the repaired forward above taught the capacity the big count)
        for i in range(14):
            fwd(cov)
        node.check_pending()
        assert node.key_state(0, P, 144, 144)[3]
        _busy(dev)
        fwd(big)
        torch.cuda.synchronize()
        with pytest.warns(UserWarning, match="earlier deferred forward"):
            fwd(cov)
        assert torch.equal(fwd(cov)[0][0], first)                                # re-learning: exact again, same image
        for i in range(12):
            fwd(cov)
        node.check_pending()
        assert node.key_state(0, P, 144, 144)[3]
        # ... its own backward raises
        _busy(dev)
        (out, mm) = fwd(big, grad=True)
        with pytest.raises(RuntimeError, match="EARLIER forward"):
            out[0].sum().backward()
        torch.cuda.synchronize()
        # ... and so does check_pending()
        for i in range(12):
            fwd(cov)
        node.check_pending()
        _busy(dev)
        fwd(big)
        with pytest.raises(RuntimeError, match="EARLIER forward"):
            node.check_pending()
    finally:
        node.set_count_check("inline")
        node.reset()


def _batched_l1_inputs(dev, S, V, P=5000, H=128, W=144, seed=5):
    from sigman_release_amd import rasterizer as R
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    hosts = [synthetic.humanoid(P, seed + s) for s in range(S)]
    base = {"means3D": torch.stack([t(g["position"]) for g in hosts]), "rgb": torch.stack([t(g["rgb"]) for g in hosts]),
            "opacity": torch.stack([t(g["opacity"].reshape(P, 1)) for g in hosts]),
            "cov3D": torch.stack([t(synthetic.covariance_from_gaussians(g)) for g in hosts])}
    cv, cvp, cp = cameras.make_cameras([int(v) for v in np.random.default_rng(seed).choice(90, V, replace=False)] * S)
    mk = lambda cap, da=None: R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0,
                                                             t(cp), V, False, cap, da)
    target = torch.rand(S * V, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    return base, mk, target


@pytest.mark.parametrize("S,V,mode", [(1, 1, "loss"), (1, 1, "loss*1.7"), (1, 2, "loss+color"), (2, 4, "loss"), (2, 4, "loss*1.7"), (1, 2, "depth+alpha")])
def test_cpp_batched_l1_node_equals_python_node(S, V, mode):
    """rasterize_l1_loss_batched routes the reference's input flavour in the explicit sync-free mode to the C++ node (csrc/torch_node.cpp):
    same loss, per-view losses, images and gradients as the Python node, bit for bit.  The Python node always runs the UNFUSED chain
    (compositing, loss kernel, compositing backward, gather); at these sizes (<= 2048 tiles) the C++ node takes the FUSED single-view step
    (loss shares + dL/dcolor from the compositing kernel, the compositing backward of dL/dloss = 1 queued by the forward call, csrc/render.hip
    FusedL1): the gather multiplies the upstream scalar in, so gradients are bit-identical for an upstream gradient of 1 and equal to the
    last bits otherwise; with a second gradient into the colour (or depth / alpha) the node falls back to the unfused backward."""
    from sigman_release_amd import _cabi, rasterizer as R
    if _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    base, mk, target = _batched_l1_inputs(dev, S, V)
    st = mk(400000, True if mode == "depth+alpha" else None)
    res = []
    for impl in ("python", "cpp"):
        d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        args = (d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st, target, None, 0.37)
        out = R._RasterizeL1Batched.apply(*args) if impl == "python" else R.rasterize_l1_loss_batched(*args)
        if impl == "cpp":
            assert type(out[0].grad_fn).__name__ != "_RasterizeL1BatchedBackward", "the call did not reach the C++ node"
        loss, per_view, color, radii, depth, alpha = out
        total = loss * (1.0 if mode == "loss" else 1.7)
        if mode == "loss+color":
            total = total + (color * color).sum() * 0.01
        if mode == "depth+alpha":
            total = total + (depth * 0.3).sum() + (alpha * alpha).sum() * 0.2
        total.backward()
        torch.cuda.synchronize()
        res.append([x.detach().clone() for x in (loss, per_view, color, radii, depth, alpha)] + [d[k].grad.clone() for k in ("means3D", "rgb", "opacity", "cov3D")])
    for i, (a, b) in enumerate(zip(*res)):
        assert a.shape == b.shape
        if i < 2:       # loss / per-view losses: float atomics over the pixels, equal up to the order of the additions
            assert torch.allclose(a, b, rtol=1e-5, atol=0.0)
        elif i >= 6 and mode == "loss*1.7":     # the fused step scales the gathered sums, the unfused chain dL/dcolor
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-7 * float(b.abs().max())), (i, float((a - b).abs().max()), float(b.abs().max()))
        else:
            assert torch.equal(a, b), i
    R.check_pending_overflows(True)


@pytest.mark.parametrize("P,H,W,V,use_mask,sort_mode", [(20000, 256, 256, 1, True, 3), (6000, 128, 144, 2, False, 3), (3000, 100, 90, 1, True, 3),
                                                        (6000, 128, 144, 2, True, 4)])
def test_fused_single_view_step_equals_unfused_chain(P, H, W, V, use_mask, sort_mode):
    """The fused single-view step (SgrL1Epilogue.fuse_backward: loss shares + dL/dcolor inside the segment-parallel compositing kernel, the
    background's share pre-filled by extra workgroups of the preprocess launch, the bucket backward queued behind the compositing kernel by the
    forward call with the loss reduction in its spare workgroup, the caller's backward only gathers) against the unfused chain through the
    SAME C++ node (sgr_set_fused_step 0): images, radii and gradients bit for bit with dL/dloss = 1, the loss to the order of its additions.
    A second backward on the same forward (gather only, twice) repeats the first; a backward with a gradient into the colour as well falls
    back to the unfused backward and is bit-identical to the unfused chain's; the fused loss is bitwise reproducible from run to run.
    sort_mode 4: the view-segmented binning flavour (plain work order written by its register-sort launch's spare workgroup) instead of the
    single-view path (class-major work order, empty tiles never visited by the compositing kernel)."""
    from sigman_release_amd import _cabi, rasterizer as R
    if _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    base, mk, target = _batched_l1_inputs(dev, 1, V, P=P, H=H, W=W, seed=11)
    mask = (torch.rand(V, 1, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) > 0.3).float() if use_mask else None
    st = mk(600000)
    L = _cabi.lib()
    res = {}
    try:
        assert L.sgr_set_sort_mode(sort_mode) == 0
        for fused in (0, 1, 1):
            L.sgr_set_fused_step(fused)
            for flavour in ("loss", "loss+color"):
                d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
                loss, per_view, color, radii, depth, alpha = R.rasterize_l1_loss_batched(d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st,
                                                                                         target, mask, 0.37)
                total = loss if flavour == "loss" else loss + (color * color).sum() * 0.01
                total.backward(retain_graph=(flavour == "loss"))
                g1 = [d[k].grad.clone() for k in ("means3D", "rgb", "opacity", "cov3D")]
                if flavour == "loss":
                    for k in d:
                        d[k].grad = None
                    total.backward()
                    g2 = [d[k].grad.clone() for k in ("means3D", "rgb", "opacity", "cov3D")]
                    for a, b in zip(g1, g2):
                        assert torch.equal(a, b), "a second backward on the same forward must repeat the first"
                torch.cuda.synchronize()
                if (fused, flavour) in res:         # the second fused run: everything repeats bit for bit, the loss included (no atomics)
                    for a, b in zip(res[(fused, flavour)], [x.detach() for x in (loss, per_view, color, radii, depth, alpha)] + g1):
                        assert torch.equal(a, b)
                res[(fused, flavour)] = [x.detach().clone() for x in (loss, per_view, color, radii, depth, alpha)] + g1
    finally:
        L.sgr_set_fused_step(1)
        L.sgr_set_sort_mode(3)
    for flavour in ("loss", "loss+color"):
        for i, (a, b) in enumerate(zip(res[(0, flavour)], res[(1, flavour)])):
            if i < 2:
                assert torch.allclose(a, b, rtol=1e-5, atol=0.0), (flavour, i, a, b)
            else:
                assert torch.equal(a, b), (flavour, i, float((a.float() - b.float()).abs().max()))
    assert float(res[(1, "loss")][6].abs().max()) > 0.0
    R.check_pending_overflows(True)


def test_fused_step_ordinary_backward_then_gather_only_backward():
    """One fused forward, then (1) an ORDINARY backward through the colour output (another upstream gradient: the compositing backward runs
    again, into scratch records with their own flags) and (2) the gather-only backward of the step's own loss.  The second must read the
    fused records through the flags the FUSED pass wrote (SgrForwardState.off_flags_fused): with one shared flags array the ordinary
    backward's flags decided which fused records the gather saw (records dropped, or uninitialised slots read).  Both orders must give the
    gradients of a fresh forward, bit for bit (the reference's calculate_adaptive_weight calls autograd.grad twice on one graph)."""
    from sigman_release_amd import _cabi, rasterizer as R
    if _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    base, mk, target = _batched_l1_inputs(dev, 1, 1, P=8000, H=128, W=128, seed=5)
    # a mask that switches the loss off on half of the image: there the L1 gradient is zero (no fused record, flag 0) while the colour
    # gradient below is not (the ordinary backward writes a record and sets its flag)
    mask = torch.zeros(1, 1, 128, 128, device=dev)
    mask[..., :, :64] = 1.0
    st = mk(300000)
    names = ("means3D", "rgb", "opacity", "cov3D")

    def run(order):
        d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        loss, per_view, color, radii, depth, alpha = R.rasterize_l1_loss_batched(d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st,
                                                                                 target, mask, 0.37)
        other = (color * color).sum() * 0.01
        out = {}
        for which in order:
            for k in d:
                d[k].grad = None
            (loss if which == "loss" else other).backward(retain_graph=True)
            out[which] = [d[k].grad.clone() for k in names]
        torch.cuda.synchronize()
        return out

    a = run(("loss", "color"))
    b = run(("color", "loss"))
    c = run(("color", "loss", "color", "loss"))
    assert float(a["loss"][0].abs().max()) > 0.0 and float(a["color"][0].abs().max()) > 0.0
    for which in ("loss", "color"):
        for x, y, z in zip(a[which], b[which], c[which]):
            assert torch.equal(x, y), (which, float((x - y).abs().max()))
            assert torch.equal(x, z), (which, float((x - z).abs().max()))
    R.check_pending_overflows(True)


@pytest.mark.parametrize("flavour,cap", [("sh+scales", 300000), ("colors+cov", 0), ("sh+scales", 0)])
def test_fused_single_view_step_other_input_flavours_and_exact_mode(flavour, cap):
    """The fused single-view step is decided below the C ABI (sgr_rasterize_forward_l1 with SgrL1Epilogue.fuse_backward), for every input
    flavour and capacity mode; the C++ node only takes colours + covariances in the sync-free mode.  Through the Python node with
    rasterizer.FUSE_STEP_IN_PYTHON_NODE: spherical harmonics + scales / rotations, and the exact mode (max_rendered = 0: preprocess runs before
    the image blob exists, so the background is not pre-filled and the compositing kernel's own empty-tile workgroups write its loss shares
    and dL/dcolor) -- loss, images and gradients against the same node unfused, bit for bit at dL/dloss = 1."""
    from sigman_release_amd import rasterizer as R
    import cases
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if flavour == "sh+scales":
        inp, st = cases.cloud_sh(P=2000, H=96, W=112, seed=6, views=(30, 53))
    else:
        inp, st = cases.humanoid(P=5000, H=128, W=144, seed=4, views=(30, 65))
    nv = np.asarray(st["viewmatrix"]).reshape(-1, 16).shape[0]
    bst = R.BatchedRasterizationSettings(st["image_height"], st["image_width"], st["tanfovx"], st["tanfovy"], t(st["bg"]), float(st["scale_modifier"]),
                                         t(np.asarray(st["viewmatrix"]).reshape(nv, 16)), t(np.asarray(st["projmatrix"]).reshape(nv, 16)), int(st["sh_degree"]),
                                         t(np.asarray(st["campos"]).reshape(nv, 3)), nv, False, cap)
    H, W = st["image_height"], st["image_width"]
    g = torch.Generator(device=dev).manual_seed(9)
    target = torch.rand(nv, 3, H, W, device=dev, generator=g)
    mask = (torch.rand(nv, 1, H, W, device=dev, generator=g) > 0.25).float()
    res = []
    try:
        for fuse in (False, True):
            R.FUSE_STEP_IN_PYTHON_NODE = fuse
            d = {k: t(v)[None].requires_grad_(True) for k, v in inp.items()}
            op = d["opacities"][..., None] if d["opacities"].dim() == 2 else d["opacities"]
            out = R._RasterizeL1Batched.apply(d["means3D"], None, d.get("shs"), d.get("colors_precomp"), op, d.get("scales"), d.get("rotations"),
                                              d.get("cov3D_precomp"), bst, target, mask, 0.5)
            out[0].backward()
            torch.cuda.synchronize()
            res.append([x.detach().clone() for x in out] + [d[k].grad.clone() for k in sorted(d)])
    finally:
        R.FUSE_STEP_IN_PYTHON_NODE = False
    assert len(res[0]) > 8 and float(res[1][0]) > 0.0
    for i, (a, b) in enumerate(zip(*res)):
        if i < 2:
            assert torch.allclose(a, b, rtol=1e-5, atol=0.0), (i, a, b)
        else:
            assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))
    R.check_pending_overflows(True)


@pytest.mark.parametrize("fused", [1, 0])
def test_fused_single_view_step_with_nothing_visible(fused):
    """Every Gaussian culled (behind the camera): the step returns the background's loss, the background image and zero gradients -- in the
    sync-free mode the fused step still runs (the buffers are sized by the capacity, the true count is zero): pre-filled background, a
    compositing launch and a bucket backward with nothing to do, the loss from the pre-filled shares alone."""
    from sigman_release_amd import _cabi, rasterizer as R
    if _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    base, mk, target = _batched_l1_inputs(dev, 1, 2, P=500, H=70, W=100, seed=2)
    base["means3D"] = base["means3D"] + torch.tensor([0.0, 0.0, 50.0], device=dev)       # (camera rig looks at the origin from ~2.5 m: all behind / beyond it)
    st = mk(50000)
    L = _cabi.lib()
    try:
        L.sgr_set_fused_step(fused)
        d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        mask = (torch.rand(2, 1, 70, 100, device=dev) > 0.5).float()
        loss, per_view, color, radii, depth, alpha = R.rasterize_l1_loss_batched(d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st,
                                                                                 target, mask, 0.25)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        L.sgr_set_fused_step(1)
    if int(radii.sum()) != 0:
        pytest.skip("the shifted subject is still visible from this rig")
    bg = st.bg[None, :, None, None].expand(2, 3, 70, 100)
    assert torch.equal(color.detach(), bg) and float(alpha.detach().abs().max()) == 0.0 and float(depth.detach().abs().max()) == 0.0
    want = 0.25 * ((bg.clamp(0, 1) - target) * mask).abs().double().sum()
    assert abs(float(loss.detach()) - float(want)) <= 1e-5 * float(want)
    assert torch.allclose(per_view.double().sum(), loss.detach().double(), rtol=1e-6)
    for k in d:
        assert d[k].grad is not None and float(d[k].grad.abs().sum()) == 0.0, k
    R.check_pending_overflows(True)


def test_randomised_fused_step_slice():
    """A slice of tools/fuzz_fused_step.py under the driver: ~1 200 random small scenes (one to four views, odd sizes, with / without mask, backgrounds
    beyond [0, 1], random capacity head-room) through the rasterizer + masked L1 node with the fused step on, off, on: images, radii, gradients bit
    for bit, the fused loss identical run to run.  (This sweep found the one nondeterminism the forward ever had: a transmittance that crossed 1e-4
    within rounding could be stopped by two segments of the segment-parallel forward -- csrc/render.hip, phase 2.)"""
    import importlib.util
    import os
    from sigman_release_amd import _cabi
    if _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    spec = importlib.util.spec_from_file_location("fuzz_fused_step", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_fused_step.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # (seed, scenes): with the forward of before the fix scene 120 of seed 14, 414 of seed 2 and 594 of seed 19 differed between two runs in one pixel
    # (13 of 14 attempts on the old build; 29 000 scenes and these three sequences are clean on the fixed one)
    for seed, scenes in ((14, 160), (2, 450), (19, 620)):
        n, n_empty = mod.run(seconds=60.0, seed=seed, max_scenes=scenes)
        assert n == scenes, (seed, n)


@pytest.mark.parametrize("impl", ["cpp", "python"])
def test_allocation_failure_is_an_error_not_a_fault(impl, monkeypatch):
    """A forward whose buffers do not fit the GPU (a capacity of 3e9 tile instances: > 1 TB of checkpoints) raises a RuntimeError that names the
    allocation -- before anything is launched on the blob that could not be had -- and the next ordinary call works.  The allocator callbacks used to
    let the out-of-memory exception escape: ctypes then handed the library an uninitialised pointer (kernels launched on it: a memory fault; found by
    tools/fuzz_determinism.py with a scene of 300 000 splats at ten times their size), the C++ node unwound through the C ABI with its count slot."""
    from sigman_release_amd import _cabi, rasterizer as R
    if impl == "cpp" and _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built")
    if impl == "python":
        monkeypatch.setattr(_cabi, "_node", None)
    dev = _dev()
    base, mk, target = _batched_l1_inputs(dev, 1, 2, P=800, H=64, W=80, seed=3)
    def call(cap):
        d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        out = R.rasterize_l1_loss_batched(d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], mk(cap), target, None, 1.0)
        out[0].backward()
        torch.cuda.synchronize()
        return out[0].detach().clone(), d["means3D"].grad.clone()
    good = call(200000)
    with pytest.raises(RuntimeError, match="(?i)allocat"):
        call(3_000_000_000)
    again = call(200000)
    assert torch.allclose(good[0], again[0], rtol=1e-6) and torch.equal(good[1], again[1])
    R.check_pending_overflows(True)


def test_cpp_batched_l1_node_overflow_and_no_grad():
    """A forward that does not fit its explicit capacity raises from its own backward; without a backward, from check_pending_overflows();
    under torch.no_grad() from the forward itself -- the Python node's behaviour."""
    from sigman_release_amd import _cabi, rasterizer as R
    if _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    base, mk, target = _batched_l1_inputs(dev, 1, 2)
    call = lambda d, st: R.rasterize_l1_loss_batched(d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st, target, None, 1.0)
    small = mk(1000)
    d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    loss = call(d, small)[0]
    with pytest.raises(RuntimeError, match="exceeds max_rendered 1000"):
        loss.backward()
    d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    call(d, small)                                          # no backward
    with pytest.raises(RuntimeError, match="EARLIER forward"):
        R.check_pending_overflows(True)
    with torch.no_grad(), pytest.raises(RuntimeError, match="exceeds max_rendered 1000"):
        call(base, small)
    d = {k: v.clone().requires_grad_(True) for k, v in base.items()}          # leaves that require grad, but under no_grad: still "now"
    with torch.no_grad(), pytest.raises(RuntimeError, match="this forward"):
        call(d, small)
    # and a fitting capacity keeps working afterwards, many forwards deep
    ok = mk(400000)
    for _ in range(300):
        d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        call(d, ok)[0].backward()
    R.check_pending_overflows(True)
    assert np.isfinite(float(d["means3D"].grad.abs().sum()))


def test_cpp_batched_l1_node_two_threads_two_streams():
    """Two Python threads drive the batched node on their own streams at the same time (forward on the caller's thread, backward on the
    autograd engine's): every step of each thread reproduces that thread's single-threaded result bit for bit."""
    import threading
    from sigman_release_amd import _cabi, rasterizer as R
    if _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    cases_ = [_batched_l1_inputs(dev, 1, 2, P=20000, H=192, W=192, seed=3), _batched_l1_inputs(dev, 1, 1, P=50000, H=256, W=256, seed=9)]

    def run(case, n, out, stream):
        base, mk, target = case
        st = mk(600000)
        try:
            with torch.cuda.stream(stream):
                res = None
                for _ in range(n):
                    d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
                    o = R.rasterize_l1_loss_batched(d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st, target, None, 1e-3)
                    o[0].backward()
                    cur = [o[2].detach().clone()] + [d[k].grad.clone() for k in ("means3D", "rgb", "opacity", "cov3D")]
                    if res is not None:
                        assert all(torch.equal(a, b) for a, b in zip(res, cur))
                    res = cur
                stream.synchronize()
            out.append(res)
        except BaseException as e:          # (an assertion in a thread would otherwise vanish)
            out.append(e)

    ref = [[], []]
    for i, c in enumerate(cases_):
        run(c, 2, ref[i], torch.cuda.current_stream())
    outs = [[], []]
    ths = [threading.Thread(target=run, args=(cases_[i], 60, outs[i], torch.cuda.Stream())) for i in range(2)]
    [th.start() for th in ths]
    [th.join() for th in ths]
    torch.cuda.synchronize()
    for i in range(2):
        assert not isinstance(outs[i][0], BaseException), outs[i][0]
        assert all(torch.equal(a, b) for a, b in zip(ref[i][0], outs[i][0]))
    R.check_pending_overflows(True)


def test_cpp_batched_l1_node_input_forms():
    """Input forms the Python node accepts, through the C++ node: opacities as [S,P] (no trailing 1), non-contiguous colours, half-precision
    leaves (cast to fp32 inside, gradients come back in the leaf's dtype) -- same numbers as the Python node."""
    from sigman_release_amd import _cabi, rasterizer as R
    if _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    base, mk, target = _batched_l1_inputs(dev, 2, 2)
    st = mk(400000)
    S, P = base["means3D"].shape[:2]
    wide = torch.zeros(S, P, 6, device=dev)
    wide[..., ::2] = base["rgb"]
    forms = {
        "opacity_2d": lambda: dict(base, opacity=base["opacity"].reshape(S, P)),
        "non_contiguous_rgb": lambda: dict(base, rgb=wide[..., ::2]),
        "bf16_leaves": lambda: {k: v.to(torch.bfloat16) for k, v in base.items()},
    }
    for name, make in forms.items():
        res = []
        for impl in ("python", "cpp"):
            src = make()
            d = {k: v.detach().clone().requires_grad_(True) if k != "rgb" or name != "non_contiguous_rgb" else v.detach().requires_grad_(True) for k, v in src.items()}
            args = (d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st, target, None, 0.5)
            out = R._RasterizeL1Batched.apply(*args) if impl == "python" else R.rasterize_l1_loss_batched(*args)
            out[0].backward()
            torch.cuda.synchronize()
            res.append([out[2].detach().clone()] + [d[k].grad.clone() for k in ("means3D", "rgb", "opacity", "cov3D")])
            for k in ("means3D", "rgb", "opacity", "cov3D"):
                assert d[k].grad.shape == d[k].shape and d[k].grad.dtype == d[k].dtype, (name, impl, k)
        for a, b in zip(*res):
            assert torch.equal(a, b), name
    R.check_pending_overflows(True)


@pytest.mark.parametrize("S,V,cap,mode", [(1, 1, -1, "color"), (2, 3, -1, "color+depth+alpha"), (1, 2, 400000, "color"), (2, 2, 0, "color+depth+alpha")])
def test_cpp_batched_rasterize_node_equals_python_node(S, V, cap, mode):
    """rasterize_gaussians_batched routes the reference's input flavour to the C++ node (csrc/torch_node.cpp, RenderBatchedNode) in all three
    capacity modes (automatic / explicit / exact): images, radii and gradients equal the Python node's bit for bit -- also on the second and
    third call of the automatic mode, which are the sync-free ones."""
    from sigman_release_amd import _cabi, rasterizer as R
    node = _cabi.torch_node()
    if node is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    base, mk, _target = _batched_l1_inputs(dev, S, V)
    st = mk(cap, True if "depth" in mode else None)
    node.reset_batched()
    R._auto_capacity.clear()
    gsum = torch.randn(S * V, 3, st.image_height, st.image_width, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    res = {"python": [], "cpp": []}
    for rep in range(3):
        for impl in ("python", "cpp"):
            d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            args = (d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st)
            out = R._RasterizeGaussiansBatched.apply(*args) if impl == "python" else R.rasterize_gaussians_batched(*args)
            if impl == "cpp":
                assert type(out[0].grad_fn).__name__ != "_RasterizeGaussiansBatchedBackward", "the call did not reach the C++ node"
            color, radii, depth, alpha = out
            total = (color * gsum).sum()
            if "depth" in mode:
                total = total + (depth * 0.3).sum() + (alpha * alpha).sum() * 0.2
            total.backward()
            torch.cuda.synchronize()
            res[impl].append([x.detach().clone() for x in (color, radii, depth, alpha)] + [d[k].grad.clone() for k in ("means3D", "rgb", "opacity", "cov3D")])
    for rep in range(3):
        for i, (a, b) in enumerate(zip(res["python"][rep], res["cpp"][rep])):
            assert a.shape == b.shape and torch.equal(a, b), (rep, i)
        for a, b in zip(res["cpp"][0], res["cpp"][rep]):
            assert torch.equal(a, b), rep
    R.check_pending_overflows(True)


def test_cpp_batched_rasterize_node_capacity_policy():
    """Automatic mode: a scene that outgrows the remembered capacity is re-rendered exactly inside the call (never a truncated image, never an
    error); explicit mode: too small a capacity raises from the backward / at once under no_grad, like the Python node."""
    from sigman_release_amd import _cabi, rasterizer as R
    node = _cabi.torch_node()
    if node is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    base, mk, _t = _batched_l1_inputs(dev, 1, 2)
    node.reset_batched()
    call = lambda d, st: R.rasterize_gaussians_batched(d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st)
    auto = mk(-1)
    small = {k: (v * 0.35 if k == "cov3D" else v) for k, v in base.items()}                 # smaller splats: fewer tile instances
    with torch.no_grad():
        call(small, auto)                                                                  # learns the small scene's capacity
        exact_big = call(base, mk(0))
        got_big = call(base, auto)                                                         # does not fit: must come back complete
        again = call(base, auto)                                                           # now sync-free with the re-learned capacity
    for a, b, c in zip(exact_big, got_big, again):
        assert torch.equal(a, b) and torch.equal(a, c)
    tiny = mk(1000)
    d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    color = call(d, tiny)[0]
    with pytest.raises(RuntimeError, match="exceeds max_rendered 1000"):
        color.sum().backward()
    with torch.no_grad(), pytest.raises(RuntimeError, match="exceeds max_rendered 1000"):
        call(base, tiny)
    R.check_pending_overflows(True)


def test_cpp_render_node_equals_python_render(monkeypatch):
    """GaussianRenderer.render through the C++ node (3-NN + covariance + rasterizer + clamp in one autograd node) == the Python path
    (dist_cuda2, _Cov3D, the Python rasterizer node, torch's clamp): image, alpha and the gradients of all five leaves, bit for bit; bf16
    leaves get their gradients back in bf16."""
    from types import SimpleNamespace
    from sigman_release_amd import _cabi
    from sigman_release_amd.renderer import GaussianRenderer
    if _cabi.torch_node() is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    B, V, P, H, W = 2, 3, 6000, 112, 96
    subj = [synthetic.humanoid(P, 70 + b) for b in range(B)]
    host = {k: np.stack([s[k] for s in subj]) for k in ("position", "opacity", "scale", "cov3d", "rgb")}
    host["rgb"] = host["rgb"] * 1.3 - 0.1                                                  # some colours leave [0, 1]: the clamp and its mask matter
    cams = [cameras.make_cameras(v) for v in [(30, 37, 65), (45, 0, 85)]]
    cam = [torch.from_numpy(np.stack([c[i] for c in cams])).to(dev) for i in range(3)]
    rend = GaussianRenderer(SimpleNamespace(FoVy=cameras.FOVY, output_size_h=H, output_size_w=W))
    gsum = torch.randn(B, V, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    for dtype in (torch.float32, torch.bfloat16):
        res = []
        for impl in ("python", "cpp"):
            if impl == "python":
                monkeypatch.setattr(_cabi, "_node", None)
            else:
                monkeypatch.undo()
            g = {k: torch.from_numpy(v).to(dev).to(dtype).requires_grad_(True) for k, v in host.items()}
            out = rend.render(g, *cam)
            assert out["image"].shape == (B, V, 3, H, W) and out["alpha"].shape == (B, V, 1, H, W)
            assert float(out["image"].min()) >= 0.0 and float(out["image"].max()) <= 1.0
            ((out["image"] * gsum).sum() + (out["alpha"] * 0.1).sum()).backward()
            torch.cuda.synchronize()
            for k in g:
                assert g[k].grad is not None and g[k].grad.dtype == dtype and g[k].grad.shape == g[k].shape, (impl, k)
            res.append([out["image"].detach().clone(), out["alpha"].detach().clone()] + [g[k].grad.clone() for k in ("position", "opacity", "scale", "cov3d", "rgb")])
        monkeypatch.undo()
        for i, (a, b) in enumerate(zip(*res)):
            assert torch.equal(a, b), (str(dtype), i)


def test_cpp_batched_l1_node_lazy_count_wait():
    """set_count_wait("lazy" / "lazy:N") (what bench.py opts into): a backward only LOOKS at its forward's instance count; one that has not arrived
    is waited for by the thread's forward after next (N = 1) / by the forward N + 1 later at the latest.  Same results as the default; an overflow is
    reported by the forward's own backward if the count was there, else by one of the next N + 1 forwards ("EARLIER forward") or by
    check_pending_overflows() -- never lost."""
    from sigman_release_amd import _cabi, rasterizer as R
    node = _cabi.torch_node()
    if node is None:
        pytest.skip("sgr_torch_node.so not built / SIGMAN_PY_NODE=1")
    dev = _dev()
    base, mk, target = _batched_l1_inputs(dev, 1, 2)
    call = lambda d, st: R.rasterize_l1_loss_batched(d["means3D"], None, None, d["rgb"], d["opacity"], None, None, d["cov3D"], st, target, None, 1.0)
    ok, small = mk(400000), mk(1000)
    res = {}
    try:
        for bad in ("", "eager", "lazy:", "lazy:0", "lazy:17", "lazy:x", "lazy:4 "):
            with pytest.raises(RuntimeError, match="set_count_wait"):
                node.set_count_wait(bad)
        for mode in ("own", "lazy", "lazy:4"):
            node.set_count_wait(mode)
            for _ in range(40):                                              # many steps deep: pending entries are recycled
                d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
                out = call(d, ok)
                out[0].backward()
            torch.cuda.synchronize()
            res[mode] = [out[2].detach().clone()] + [d[k].grad.clone() for k in ("means3D", "rgb", "opacity", "cov3D")]
            R.check_pending_overflows(True)
        for a, b, c in zip(res["own"], res["lazy"], res["lazy:4"]):
            assert torch.equal(a, b) and torch.equal(a, c)
        for depth, lazy_mode in ((1, "lazy"), (4, "lazy:4")):
            node.set_count_wait(lazy_mode)
            seen = []
            d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            loss = call(d, small)[0]                                             # does not fit
            try:
                loss.backward()
            except RuntimeError as e:
                seen.append(str(e))
            for _ in range(depth + 2):                                           # the report comes from one of the next forwards at the latest
                if seen:
                    break
                try:
                    d = {k: v.clone().requires_grad_(True) for k, v in base.items()}
                    call(d, ok)[0].backward()
                except RuntimeError as e:
                    seen.append(str(e))
            assert len(seen) == 1 and "exceeds max_rendered 1000" in seen[0], seen
            torch.cuda.synchronize()
            R.check_pending_overflows(True)                                      # nothing left behind
    finally:
        node.set_count_wait("own")
