"""Seeded synthetic Gaussian sets for the benchmark configs of BASELINE.json (host side, numpy).

No SMPL-X asset may be shipped, so the "SMPL-X-anchored" subject is a procedural humanoid whose
statistics match the reference's template: T-pose bounding box
x in [-0.87,0.87], y in [-1.01,0.78], z in [-0.15,0.15], surface area ~2 m^2, one Gaussian per surface
sample whose rotation is the local tangent frame (as core/modules/deformers/utils_smplx.py:88-111
builds it per face) times a Rodrigues rotation with axis-angle ~ U(-pi/2,pi/2)^3
(core/modules/autoencoder.py:309,455-486), raw scale = 2*sigmoid(N(0,1))-1 (autoencoder.py:308),
opacity = sigmoid(N(2,1)) (autoencoder.py:305), rgb ~ U(0,1).

Everything returned is what /root/reference/core/gaussians/gs.py:56-60 calls the `gaussians` dict
(for ONE subject): position [P,3], opacity [P,1], scale [P,3] in (-1,1), cov3d [P,3,3] (rotation),
rgb [P,3]; all float32.
"""
from __future__ import annotations

import numpy as np

# (a, b, radius): capsules of the procedural humanoid, already translated like smplx_deformer.py:84
_CAPSULES = [
    ((0.0, -0.05, 0.0), (0.0, 0.45, 0.0), 0.13),     # torso
    ((0.0, 0.65, 0.0), (0.0, 0.65, 0.0), 0.11),      # head (sphere)
    ((0.18, 0.42, 0.0), (0.83, 0.42, 0.0), 0.045),   # left arm
    ((-0.18, 0.42, 0.0), (-0.83, 0.42, 0.0), 0.045),  # right arm
    ((0.09, -0.08, 0.0), (0.09, -0.95, 0.0), 0.065),  # left leg
    ((-0.09, -0.08, 0.0), (-0.09, -0.95, 0.0), 0.065),  # right leg
]


def _rodrigues(aa: np.ndarray) -> np.ndarray:
    """axis-angle [P,3] -> rotation [P,3,3] (same formula as autoencoder.py:455-486)."""
    theta = np.linalg.norm(aa, axis=1, keepdims=True)
    k = aa / np.maximum(theta, 1e-8)
    K = np.zeros((aa.shape[0], 3, 3), aa.dtype)
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    s, c = np.sin(theta)[:, :, None], np.cos(theta)[:, :, None]
    return np.eye(3, dtype=aa.dtype)[None] + s * K + (1 - c) * (K @ K)


def _sample_capsule_surface(rng, n, a, b, r):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    L = np.linalg.norm(b - a)
    axis = (b - a) / L if L > 0 else np.array([0.0, 1.0, 0.0])
    # orthonormal frame (axis, e1, e2)
    tmp = np.array([0.0, 0.0, 1.0]) if abs(axis[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
    e1 = np.cross(axis, tmp); e1 /= np.linalg.norm(e1)
    e2 = np.cross(axis, e1)
    area_cyl, area_sph = 2 * np.pi * r * L, 4 * np.pi * r * r
    on_cyl = rng.random(n) < area_cyl / (area_cyl + area_sph)
    pos = np.zeros((n, 3)); nrm = np.zeros((n, 3))
    # cylinder part
    m = int(on_cyl.sum())
    h, phi = rng.random(m) * L, rng.random(m) * 2 * np.pi
    nc = np.cos(phi)[:, None] * e1 + np.sin(phi)[:, None] * e2
    pos[on_cyl] = a + h[:, None] * axis + r * nc
    nrm[on_cyl] = nc
    # spherical caps: uniform on sphere, assigned to the cap on its side of the axis
    m2 = n - m
    v = rng.normal(size=(m2, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    side = (v @ axis) >= 0
    pos[~on_cyl] = np.where(side[:, None], b, a) + r * v
    nrm[~on_cyl] = v
    return pos, nrm, area_cyl + area_sph


def _tangent_frames(nrm: np.ndarray) -> np.ndarray:
    """R0 = [t1 t2 n] per point (columns), right-handed."""
    up = np.where(np.abs(nrm[:, 1:2]) < 0.95, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t1 = np.cross(up, nrm); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    return np.stack([t1, t2, nrm], axis=2)


def _finish(rng, pos, R0, P):
    aa = rng.uniform(-np.pi / 2, np.pi / 2, size=(P, 3))
    rot = R0 @ _rodrigues(aa)
    scale_raw = 2.0 / (1.0 + np.exp(-rng.normal(size=(P, 3)))) - 1.0
    opacity = 1.0 / (1.0 + np.exp(-rng.normal(loc=2.0, size=(P, 1))))
    rgb = rng.random((P, 3))
    f = np.float32
    return dict(position=pos.astype(f), opacity=opacity.astype(f), scale=scale_raw.astype(f),
                cov3d=rot.astype(f), rgb=rgb.astype(f))


def _morton_order(pos: np.ndarray) -> np.ndarray:
    """Indices that sort points along a 30-bit Morton curve over their bounding box (a spatially coherent order)."""
    lo, hi = pos.min(0), pos.max(0)
    q = np.minimum(((pos - lo) / np.maximum(hi - lo, 1e-12) * 1024.0).astype(np.uint64), 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")


def humanoid(P: int, seed: int, order: str = "random") -> dict:
    """Config C2/C3/C4 subject: P Gaussians on the procedural humanoid surface.
    order = "random" (default: the Gaussians of the whole subject in a random permutation -- the pessimistic case for every gather) or
    "template": part by part, each part along a Morton curve -- neighbours in memory are neighbours in space, as the faces of the reference's
    subdivided template mesh are (/root/reference/core/modules/deformers/utils_smplx.py:145-162: one Gaussian per face, in face order)."""
    rng = np.random.default_rng(seed)
    areas = []
    for a, b, r in _CAPSULES:
        L = np.linalg.norm(np.subtract(b, a))
        areas.append(2 * np.pi * r * L + 4 * np.pi * r * r)
    areas = np.array(areas); counts = rng.multinomial(P, areas / areas.sum())
    pos, nrm = [], []
    for (a, b, r), n in zip(_CAPSULES, counts):
        p, nn, _ = _sample_capsule_surface(rng, int(n), a, b, r)
        pos.append(p); nrm.append(nn)
    if order == "template":
        for k in range(len(pos)):
            o = _morton_order(pos[k])
            pos[k], nrm[k] = pos[k][o], nrm[k][o]
        pos, nrm = np.concatenate(pos), np.concatenate(nrm)
    else:
        pos, nrm = np.concatenate(pos), np.concatenate(nrm)
        perm = rng.permutation(P)
        pos, nrm = pos[perm], nrm[perm]
    return _finish(rng, pos, _tangent_frames(nrm), P)


def humanoid_layers(P: int, seed: int, layers: int = 10, sigma: float = 0.02) -> dict:
    """Config C5 stress: a P/layers humanoid replicated `layers` times with N(0, sigma) jitter (deep tile lists)."""
    base = humanoid(P // layers, seed)
    rng = np.random.default_rng(seed + 1000)
    out = {}
    for k, v in base.items():
        out[k] = np.concatenate([v] * layers, 0)
    out["position"] = (out["position"] + rng.normal(scale=sigma, size=out["position"].shape)).astype(np.float32)
    out["rgb"] = rng.random(out["rgb"].shape).astype(np.float32)
    return out


def random_cloud(P: int, seed: int) -> dict:
    """Config C1: positions ~ U([-0.8,0.8]^3), random rotations; use with `isotropic_world_scale`."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-0.8, 0.8, size=(P, 3))
    q = rng.normal(size=(P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    r, x, y, z = q.T
    R0 = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
                   np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
                   np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], 1)
    g = _finish(rng, pos, R0, P)
    g["cov3d"] = R0.astype(np.float32)
    g["world_scale"] = np.exp(rng.uniform(np.log(5e-3), np.log(5e-2), size=(P, 1))).repeat(3, 1).astype(np.float32)
    return g


def nn_dist2_cpu(position: np.ndarray) -> np.ndarray:
    """Host stand-in for simple_knn.distCUDA2 (gs.py:70): mean squared distance to the 3 nearest OTHER points."""
    from scipy.spatial import cKDTree
    d, _ = cKDTree(position.astype(np.float64)).query(position.astype(np.float64), k=4)
    return (d[:, 1:4] ** 2).mean(1).astype(np.float32)


def covariance_from_gaussians(g: dict, dist2: np.ndarray | None = None) -> np.ndarray:
    """numpy restatement of gs.py:70-73 + gs.py:17-38: cov3D [P,6] = strip_lowerdiag(R diag(s)^2 R^T)."""
    if "world_scale" in g:
        s = g["world_scale"].astype(np.float32)
    else:
        if dist2 is None:
            dist2 = nn_dist2_cpu(g["position"])
        nn = np.sqrt(np.maximum(dist2, 1e-7)).astype(np.float32)[:, None]
        s = (g["scale"] + 1.0) * nn
    R = g["cov3d"].astype(np.float32)
    L2 = np.zeros_like(R); L2[:, 0, 0], L2[:, 1, 1], L2[:, 2, 2] = s[:, 0] ** 2, s[:, 1] ** 2, s[:, 2] ** 2
    S = R @ L2 @ np.transpose(R, (0, 2, 1))
    return np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
