"""-m gpu: sgr_bin (F3 emission + F4 sort + F5 ranges) called DIRECTLY through the C ABI on synthetic rect records, against a
numpy restatement of the published binning step (duplicateWithKeys -> stable radix sort on (tile | depth bits) -> identifyTileRanges;
restated in oracle/gsplat_ref.c from the same source).  Every sort flavour must give the same bit-exact keys, values and ranges -- also
for depth keys the real pipeline rarely produces: massive ties, the smallest / largest positive floats, denormal floats (the register
sort of the view-segmented flavour orders (depth bits, value) composites as positive doubles: a high word below 0x00100000 is a
DENORMAL double, one at 0x7F7FFFFF a huge one)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(n_views, P, size, seed, depth_kind, big_rects):
    rng = np.random.default_rng(seed)
    T = size // 16
    radii = np.where(rng.random((n_views, P)) < 0.8, 3, 0).astype(np.int32)
    w = rng.integers(1, 5, (n_views, P))
    h = rng.integers(1, 5, (n_views, P))
    if big_rects:
        big = rng.random((n_views, P)) < 0.01
        w = np.where(big, rng.integers(8, T + 1, (n_views, P)), w)
        h = np.where(big, rng.integers(8, T + 1, (n_views, P)), h)
    minx = (rng.random((n_views, P)) * (T - w + 1)).astype(np.int64)
    miny = (rng.random((n_views, P)) * (T - h + 1)).astype(np.int64)
    if big_rects == "hot1":                                                      # ONE tile receives 3/4 of everything: > 131 072 entries (64 buckets of > 2048)
        big = np.zeros((n_views, P), bool)
        hot = rng.random((n_views, P)) < 0.75
        w = np.where(hot, 1, w); h = np.where(hot, 1, h)
        minx = np.where(hot, 5, (rng.random((n_views, P)) * (T - w + 1)).astype(np.int64))
        miny = np.where(hot, 3, (rng.random((n_views, P)) * (T - h + 1)).astype(np.int64))
    elif big_rects:                                                              # a hot 6x6-tile window: tile lists of several thousand entries
        hot = rng.random((n_views, P)) < 0.3
        minx = np.where(hot & ~big, 10 + rng.integers(0, 3, (n_views, P)), minx)
        miny = np.where(hot & ~big, 7 + rng.integers(0, 3, (n_views, P)), miny)
    if depth_kind == "normal":
        depth = rng.uniform(0.2, 50.0, (n_views, P)).astype(np.float32).view(np.uint32)
    elif depth_kind == "ties":
        depth = rng.choice(np.array([0.25, 1.0, 1.0000001, 2.5, 7.0], np.float32), (n_views, P)).view(np.uint32)
    elif depth_kind == "band_outlier":
        # nearly everything inside a band of a few thousand ulps, 0.2 % outliers a hundred times farther away: the depth range of a tile is
        # stretched 1e5-fold beyond where its entries are (the distribution sort's bins adapt to the density, or the tile is declined)
        depth = rng.uniform(2.5, 2.5005, (n_views, P)).astype(np.float32)
        far = rng.random((n_views, P)) < 0.002
        depth = np.where(far, rng.uniform(50.0, 300.0, (n_views, P)).astype(np.float32), depth).view(np.uint32)
    elif depth_kind == "two_planes":
        # two thin surfaces (front / back of a body part), each a few hundred distinct depth values wide: thousands of small tie groups
        depth = np.where(rng.random((n_views, P)) < 0.5, 2.3, 2.7).astype(np.float32) + (rng.integers(0, 300, (n_views, P)) * np.float32(2.4e-7)).astype(np.float32)
        depth = depth.astype(np.float32).view(np.uint32)
    else:
        pool = np.array([1, 2, 0x7FFFF, 0xFFFFF, 0x00100000, 0x00800000, 0x3F800000, 0x3F800001, 0x7F7FFFFF, 0x7F7FFFFE, 0x7EFFFFFF], np.uint32)
        depth = np.where(rng.random((n_views, P)) < 0.5, rng.choice(pool, (n_views, P)),
                         rng.integers(1, 0x7F7FFFFF, (n_views, P), dtype=np.int64).astype(np.uint32)).astype(np.uint32)
    rect = np.zeros((n_views, P, 4), np.uint32)
    rect[..., 0] = (minx | (miny << 16)).astype(np.uint32)
    rect[..., 1] = ((minx + w) | ((miny + h) << 16)).astype(np.uint32)
    rect[..., 2] = depth
    cnt = np.where(radii > 0, w * h, 0).astype(np.int64)
    # ---- expected emission (view-major, Gaussian order, rect row-major), stable sort, ranges
    keys, vals = [], []
    for v in range(n_views):
        for i in np.nonzero(cnt[v])[0]:
            ys, xs = np.meshgrid(np.arange(miny[v, i], miny[v, i] + h[v, i]), np.arange(minx[v, i], minx[v, i] + w[v, i]), indexing="ij")
            tile = (v * T * T + ys * T + xs).reshape(-1).astype(np.uint64)
            keys.append((tile << np.uint64(32)) | np.uint64(depth[v, i]))
            vals.append(np.full(tile.size, v * P + i, np.uint32))
    keys, vals = np.concatenate(keys), np.concatenate(vals)
    order = np.argsort(keys, kind="stable")
    skeys, svals = keys[order], vals[order]
    tiles_total = n_views * T * T
    tid = (skeys >> np.uint64(32)).astype(np.int64)
    ranges = np.zeros((tiles_total, 2), np.uint32)
    occ = np.unique(tid)
    ranges[occ, 0] = np.searchsorted(tid, occ, "left")
    ranges[occ, 1] = np.searchsorted(tid, occ, "right")
    first = np.zeros((n_views, P), np.int64)
    first.reshape(-1)[:] = np.concatenate([[0], np.cumsum(cnt.reshape(-1))[:-1]])
    return dict(radii=radii, rect=rect, cnt=cnt, keys=skeys, vals=svals, ranges=ranges, first=first, T=T,
                longest=int((ranges[:, 1] - ranges[:, 0]).max()))


@pytest.mark.parametrize("n_views,P,size,depth_kind,big_rects", [
    (1, 3000, 256, "normal", False),
    (1, 40000, 512, "extreme", True),          # one view, <= 2^19 instances: the single-view path (no tile pass), tiles of every length class
    (3, 5000, 512, "ties", True),
    (4, 60000, 512, "extreme", True),
    (2, 6000, 1024, "extreme", False),
    (2, 50000, 1024, "ties", True),            # 4096 tiles per view, many 8192-key chunks per view: the staged tile pass at full width
    (1, 230000, 256, "ties", "hot1"),          # one tile of > 131 072 entries with five distinct depths: the deep kernels must decline it
    (1, 230000, 256, "extreme", "hot1"),
    (1, 230000, 256, "normal", "hot1"),        # ... and with spread-out depths: nine windows of one tile, nine workgroups
    (3, 90000, 512, "normal", True),           # three views with long tiles, depths without ties: both deep instantiations next to the register classes
    (1, 230000, 256, "band_outlier", "hot1"),  # a 140 000-entry tile whose depth range is 1e5 times wider than the band its entries sit in
    (2, 60000, 512, "band_outlier", True),
    (1, 230000, 256, "two_planes", "hot1"),    # ... and one made of two thin surfaces with small tie groups everywhere
    (3, 90000, 512, "two_planes", True),
])
def test_sgr_bin_direct_all_flavours(n_views, P, size, depth_kind, big_rects):
    from sigman_release_amd import _cabi
    L = _cabi.lib()
    dev = torch.device("cuda", 0)
    case = _build(n_views, P, size, 11 + n_views, depth_kind, big_rects)
    R = int(case["cnt"].sum())
    if big_rects == "hot1":
        assert case["longest"] > 131072, case["longest"]
    if big_rects and P >= 40000:
        assert case["longest"] > 4096, case["longest"]                           # the multi-wave classes of the register sort are exercised
    nbx = int(L.sgr_preprocess_blocks_per_view(P))
    n = nbx * n_views
    pad = np.zeros((n_views, nbx * 256), np.int64)
    pad[:, :P] = case["cnt"]
    sums = pad.reshape(n_views, nbx, 256).sum(-1).reshape(-1)
    offs = np.zeros(2 * (n + 1), np.uint32)
    offs[1:n + 1] = np.cumsum(sums)
    offs[n + 1:2 * n + 1] = sums
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dummy = torch.zeros(max(P, 1) * 16, dtype=torch.float32, device=dev)
    cams = torch.zeros(n_views * 16, dtype=torch.float32, device=dev)
    pb = _cabi.SgrProblem(P=P, n_views=n_views, views_per_subject=n_views, H=size, W=size, sh_degree=0, M=0, tanfovx=1.0, tanfovy=1.0,
                          scale_modifier=1.0, means3D=dummy.data_ptr(), opacities=dummy.data_ptr(), colors_precomp=dummy.data_ptr(), shs=None,
                          cov3D_precomp=dummy.data_ptr(), scales=None, rotations=None, viewmatrix=cams.data_ptr(), projmatrix=cams.data_ptr(),
                          campos=cams.data_ptr(), bg=cams.data_ptr())
    tiles_total = n_views * case["T"] ** 2
    ws_bytes = int(L.sgr_bin_workspace_bytes(R, tiles_total))
    try:
        # (flavour, deep mode): 4 with the deep mode forced on = the long tiles (all tiles, for one or two views) go through the LDS
        # distribution sort; tiles with massive exact depth ties must come back through the generic path
        # (5, 1 << 8) / (5, 2 << 8): flavour 5 with the window cap of the single-view path's distribution sort lowered to 1 / 2 (bits 8..15 of the deep
        # mode): tiles of more than 3968 / 7936 entries are listed ONCE, marked "whole", and sorted on the spot by one workgroup -- the path a
        # tile of more than 64 windows (254 000 entries) takes in production, exercised here at sizes the other flavours are checked at
        # (flavours 0 = onesweep and 2 = LDS-segmented were removed in round 5: the automatic choice could hardly reach them)
        # (4, 1 | 2 << 16): deep mode with the collect launch (the default for one or two views), (4, 1 | 1 << 16): with the five-launch tile pass
        for mode, deep in ((1, 0), (4, 2), (4, 1), (4, 1 | (2 << 16)), (4, 1 | (1 << 16)), (5, 0), (5, 1 << 8), (5, 2 << 8), (3, 0)):
            radii, rect, boff = t(case["radii"]), t(case["rect"]), t(offs)
            ka, kb = (torch.zeros(R, dtype=torch.int64, device=dev) for _ in range(2))
            va, vb = (torch.zeros(R, dtype=torch.int32, device=dev) for _ in range(2))
            ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
            rg = torch.full((tiles_total, 2), 0x7FFFFFFF, dtype=torch.int32, device=dev)
            in_b = C.c_int32(-1)
            assert L.sgr_set_sort_mode(mode) == 0
            L.sgr_set_sort_deep(deep)
            rc = L.sgr_bin(C.byref(pb), radii.data_ptr(), rect.data_ptr(), boff.data_ptr(), R, None, ka.data_ptr(), kb.data_ptr(),
                           va.data_ptr(), vb.data_ptr(), ws.data_ptr(), ws_bytes, rg.data_ptr(), C.byref(in_b), None)
            assert rc == 0, L.sgr_last_error()
            torch.cuda.synchronize()
            k = (kb if in_b.value else ka).cpu().numpy().view(np.uint64)
            v = (vb if in_b.value else va).cpu().numpy().view(np.uint32)
            np.testing.assert_array_equal(k, case["keys"], err_msg=f"sorted keys, flavour {mode} deep {deep}")
            np.testing.assert_array_equal(v, case["vals"], err_msg=f"point list, flavour {mode} deep {deep}")
            got = rg.cpu().numpy().view(np.uint32)
            occ = case["ranges"][:, 1] > case["ranges"][:, 0]
            np.testing.assert_array_equal(got[occ], case["ranges"][occ], err_msg=f"tile ranges, flavour {mode}")
            assert bool((got[~occ, 1] == got[~occ, 0]).all()), f"empty tiles must have empty ranges, flavour {mode}"
            seen = case["radii"] > 0
            np.testing.assert_array_equal(rect.cpu().numpy()[..., 3][seen], case["first"][seen].astype(np.uint32),
                                          err_msg=f"first tile-instance index per Gaussian, flavour {mode}")
    finally:
        L.sgr_set_sort_mode(3)
        L.sgr_set_sort_deep(0)
