"""-m gpu: F1 + F2 (sgr_preprocess_forward) on their own.
  * the view group of F1 (how many views a workgroup walks with its Gaussians in registers, sgr_set_preprocess_view_group) must not
    change a single bit of any artefact -- also when a group straddles two subjects (the Gaussians are reloaded at the boundary) and
    when the last group is short;
  * F2's exclusive scan of the per-workgroup tile counts (one workgroup per 8192 counts, each summing what lies before its tile) against
    numpy's cumsum of the counts recomputed from F1's own radii / rect outputs, for counts of 1..3 entries (no whole group of four),
    counts around the 8192-entry tile boundary and several tiles."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(S, P, seed, dev):
    from sigman_release_amd import synthetic
    hosts = [synthetic.humanoid(P, seed + s) for s in range(S)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return (torch.stack([t(g["position"]) for g in hosts]), torch.stack([t(g["opacity"].reshape(P)) for g in hosts]),
            torch.stack([t(g["rgb"]) for g in hosts]), torch.stack([t(synthetic.covariance_from_gaussians(g)) for g in hosts]))


def _settings(views, S, H, W, dev, vps):
    from sigman_release_amd import cameras
    from sigman_release_amd import rasterizer as R
    cv, cvp, cp = cameras.make_cameras(list(views) * S)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return R.BatchedRasterizationSettings(H, W, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 1.0, t(cv), t(cvp), 0, t(cp), vps)


def test_view_group_does_not_change_a_bit():
    from sigman_release_amd import _cabi
    from sigman_release_amd import rasterizer as R
    dev = torch.device("cuda", 0)
    S, views, P = 2, (30, 65, 0), 5000                                # 6 view slots, 3 per subject: groups of 2, 4 and 5 straddle the subjects
    m, o, c, cov = _scene(S, P, 21, dev)
    st = _settings(views, S, 304, 272, dev, len(views))
    L = _cabi.lib()
    base = None
    try:
        for group in (1, 2, 3, 4, 5, 8, 0):
            L.sgr_set_preprocess_view_group(group)
            out = R.forward_debug(m, o, colors_precomp=c, cov3D_precomp=cov, settings=st)
            torch.cuda.synchronize()
            got = {k: out[k].detach().cpu().numpy().copy() for k in ("rec", "rect", "radii", "keys", "point_list", "ranges", "color", "final_T", "n_contrib")}
            got["num_rendered"] = np.array(out["num_rendered"])
            if base is None:
                base = got
                assert int(base["num_rendered"]) > 0
                continue
            for k in base:
                np.testing.assert_array_equal(got[k].view(np.uint8) if got[k].dtype.kind == "f" else got[k],
                                              base[k].view(np.uint8) if base[k].dtype.kind == "f" else base[k], err_msg=f"{k}, view group {group}")
    finally:
        L.sgr_set_preprocess_view_group(0)


@pytest.mark.parametrize("P,n_views", [(200, 1), (300, 1), (700, 1), (1100, 1), (300, 3), (70_000, 31), (600_000, 4), (100_000, 50)])
def test_block_count_scan(P, n_views):
    from sigman_release_amd import _cabi
    from sigman_release_amd import rasterizer as R
    dev = torch.device("cuda", 0)
    L = _cabi.lib()
    H = W = 256
    m, o, c, cov = _scene(1, P, 5, dev)
    views = [(7 * k) % 90 for k in range(n_views)]
    st = _settings(views, 1, H, W, dev, n_views)
    pb = R._make_problem(m, o, c, None, cov, None, None, st)
    nbx = int(L.sgr_preprocess_blocks_per_view(P))
    n = nbx * n_views
    rec = torch.empty(n_views * P * 16, dtype=torch.float32, device=dev)
    radii = torch.empty(n_views * P, dtype=torch.int32, device=dev)
    rect = torch.empty(n_views * P * 4, dtype=torch.int32, device=dev)
    offs = torch.full((2 * (n + 1),), -1, dtype=torch.int32, device=dev)
    nr = torch.zeros(4, dtype=torch.int64, device=dev)
    rc = L.sgr_preprocess_forward(C.byref(pb), rec.data_ptr(), radii.data_ptr(), rect.data_ptr(), None, offs.data_ptr(), nr.data_ptr(), 0, None)
    assert rc == 0, L.sgr_last_error()
    torch.cuda.synchronize()
    rd = radii.cpu().numpy().reshape(n_views, P)
    rc4 = rect.cpu().numpy().view(np.uint32).reshape(n_views, P, 4)
    w = (rc4[..., 1] & 0xFFFF).astype(np.int64) - (rc4[..., 0] & 0xFFFF)
    h = (rc4[..., 1] >> 16).astype(np.int64) - (rc4[..., 0] >> 16)
    cnt = np.where(rd > 0, w * h, 0)
    pad = np.zeros((n_views, nbx * 256), np.int64)
    pad[:, :P] = cnt
    sums = pad.reshape(n_views, nbx, 256).sum(-1).reshape(-1)
    want = np.concatenate([[0], np.cumsum(sums)])
    got = offs.cpu().numpy().view(np.uint32)[: n + 1].astype(np.int64)
    np.testing.assert_array_equal(got, want)
    nrh = nr.cpu().numpy()
    assert int(nrh[0]) == int(want[-1]) and int(nrh[1]) == 0 and int(nrh[2]) == int(want[-1])
    if P >= 70_000:
        assert n > 8192 and int(want[-1]) > 0
