"""Camera conventions of the hot path's callers (host side, numpy/torch, no GPU needed).

Restates what the reference's data layer feeds the renderer:
  * /root/reference/core/dataset/dataloader_VAE.py:42-60   intrinsics K (fx=fy=1100, cx=cy=512 @ 1024^2)
  * dataloader_VAE.py:218-246  getProjectionMatrix (OpenCV-style, +z forward, z in [0,1])
  * dataloader_VAE.py:126-131  w2c = [R|T] from camera_full_calibration.json
  * dataloader_VAE.py:207-213  cam_view = w2c^T, cam_view_proj = cam_view @ P^T, cam_pos = inv(w2c)[:3,3]
The rig itself (core/dataset/camera_full_calibration.json) is 3 rings x 30 azimuths on a sphere of
radius 2.5 m looking at the origin (elevations -30, 0, +45 degrees, 12 degree steps); it is
regenerated analytically here and checked against tests/golden/camera_rig.npz.
"""
from __future__ import annotations

import math

import numpy as np

FOVY = 0.8712626851529752          # core/model_config/VAE.py:32-33
TAN_HALF_FOV = math.tan(0.5 * FOVY)  # = 512/1100 = 0.46545454...
ZNEAR, ZFAR = 0.1, 100.0           # VAE.py:35-37
RIG_RADIUS = 2.5
RIG_ELEVATIONS_DEG = (-30.0, 0.0, 45.0)
RIG_AZIMUTHS = 30
TRAIN_VIEW_IDS = (30, 37, 45, 53, 65, 85)   # dataloader_VAE.py:77


def projection_matrix(znear: float = ZNEAR, zfar: float = ZFAR, fx: float = 1100.0, fy: float = 1100.0,
                      cx: float = 512.0, cy: float = 512.0, img_w: int = 1024, img_h: int = 1024) -> np.ndarray:
    """Row-major 4x4 P with P[3,2]=1 (dataloader_VAE.py:218-246, K branch). float32 like torch.zeros(4,4)."""
    near_fx, near_fy = znear / fx, znear / fy
    left, right = -(img_w - cx) * near_fx, cx * near_fx
    bottom, top = (cy - img_h) * near_fy, cy * near_fy
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def rig_w2c(view_id: int) -> np.ndarray:
    """4x4 world->camera (OpenCV: x right, y down, z forward) of rig camera `view_id` in [0, 90)."""
    ring, k = divmod(int(view_id), RIG_AZIMUTHS)
    el = math.radians(RIG_ELEVATIONS_DEG[ring])
    az = math.radians(12.0 * k)
    c = RIG_RADIUS * np.array([math.cos(el) * math.sin(az), math.sin(el), math.cos(el) * math.cos(az)], np.float64)
    z = -c / np.linalg.norm(c)                 # forward: towards the origin
    up = np.array([0.0, 1.0, 0.0])
    y = -(up - np.dot(up, z) * z)              # image-down = -(world up projected off the view axis)
    y /= np.linalg.norm(y)
    x = np.cross(y, z)
    R = np.stack([x, y, z], 0)
    w2c = np.eye(4, dtype=np.float64)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ c
    return w2c.astype(np.float32)


def make_cameras(view_ids, proj: np.ndarray | None = None):
    """-> (cam_view [V,4,4], cam_view_proj [V,4,4], cam_pos [V,3]) float32, exactly as dataloader_VAE.py:207-213."""
    if proj is None:
        proj = projection_matrix()
    proj_t = proj.T.astype(np.float32)                       # dataloader_VAE.py:59-60 stores the transpose
    w2c = np.stack([rig_w2c(v) for v in view_ids], 0)        # [V,4,4]
    cam_view = np.ascontiguousarray(np.transpose(w2c, (0, 2, 1)))
    cam_view_proj = (cam_view @ proj_t).astype(np.float32)
    cam_pos = np.linalg.inv(w2c.astype(np.float64))[:, :3, 3].astype(np.float32)
    return cam_view, np.ascontiguousarray(cam_view_proj), np.ascontiguousarray(cam_pos)
