"""Seeded synthetic plane scenes for tests and the benchmark.

A scene mimics example/calib_1 of the reference: a depth camera on a robot
flange looks at one plane from several flange poses.  For every pose the cloud
is a jittered pixel grid back-projected onto the plane, expressed in the camera
frame, with Gaussian noise along the ray, stored as float32 -- the shape of the
reference's voxel-downsampled `cloud_downsampled[j]` sets
(src/camera_calibration_optimization.cpp:196-205).  Only numpy is used; no
objective value is computed here.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

DEFAULT_SEED = 20251111

# ground truth, calib_1-like (example/config/config_calib_1.xml:12)
POSE_TRUE = np.array([0.0204504, 0.111258, 0.0505291, -3.15376, -0.0280023, 1.58043])
PLANE_TRUE = np.array([0.0067, 0.0137, 0.9999, -0.145])
# initial guess = truth + (+6,-9,+3) mm, (+2,-1,+3) deg: the optimum is interior
# to the shipped +-20 mm / +-5 deg grid
INIT_OFFSET = np.array([0.006, -0.009, 0.003, np.deg2rad(2.0), np.deg2rad(-1.0), np.deg2rad(3.0)])


def _rot(axis: int, a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    if axis == 2:
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    if axis == 1:
        return np.array([[c, 0, s], [0, 1.0, 0], [-s, 0, c]])
    return np.array([[1.0, 0, 0], [0, c, -s], [0, s, c]])


def pose_matrix(pose) -> np.ndarray:
    """ZYX pose vector -> 4x4 (plain numpy; only used to build scenes)."""
    T = np.eye(4)
    T[:3, :3] = _rot(2, pose[3]) @ _rot(1, pose[4]) @ _rot(0, pose[5])
    T[:3, 3] = pose[:3]
    return T


@dataclass
class SyntheticScene:
    clouds: List[np.ndarray]          # per cloud (N,3) float32, camera frame
    base_T_flange: np.ndarray         # (C,4,4) float64
    plane: np.ndarray                 # float32[4], unit normal
    pose_true: np.ndarray             # float64[6]
    pose_init: np.ndarray             # float64[6]
    seed: int = DEFAULT_SEED
    meta: dict = field(default_factory=dict)

    @property
    def n_clouds(self) -> int:
        return len(self.clouds)

    @property
    def n_points(self) -> int:
        return int(sum(c.shape[0] for c in self.clouds))


def make_scene(n_clouds: int, n_points: int, seed: int = DEFAULT_SEED, noise: float = 0.0005,
               ragged: bool = False) -> SyntheticScene:
    """n_points per cloud (ragged=True varies it by up to -30 % per cloud)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = PLANE_TRUE[:3] / np.linalg.norm(PLANE_TRUE[:3])
    d = PLANE_TRUE[3] / np.linalg.norm(PLANE_TRUE[:3])
    F = pose_matrix(POSE_TRUE)
    Finv = np.linalg.inv(F)
    # an orthonormal basis of the plane
    e1 = np.cross(n, [1.0, 0.0, 0.0]); e1 /= np.linalg.norm(e1)
    e2 = np.cross(n, e1)

    clouds, mats = [], []
    for _ in range(n_clouds):
        # point on the plane the camera looks at
        g_xy = np.array([0.55, -0.40]) + rng.uniform(-0.1, 0.1, 2)
        g = np.array([g_xy[0], g_xy[1], 0.0])
        g[2] = -(d + n[0] * g[0] + n[1] * g[1]) / n[2]
        h = rng.uniform(0.25, 0.45)
        tilt = np.deg2rad(rng.uniform(0.0, 25.0))
        phi = rng.uniform(0.0, 2 * np.pi)
        axis = np.cos(phi) * e1 + np.sin(phi) * e2
        v = -n * np.cos(tilt) + np.cross(axis, -n) * np.sin(tilt)   # view direction
        v /= np.linalg.norm(v)
        cam_pos = g - h * v
        roll = rng.uniform(0.0, 2 * np.pi)
        x0 = np.cross(v, n); x0 = e1 if np.linalg.norm(x0) < 1e-6 else x0 / np.linalg.norm(x0)
        y0 = np.cross(v, x0)
        xa = np.cos(roll) * x0 + np.sin(roll) * y0
        ya = np.cross(v, xa)
        B_cam = np.eye(4)
        B_cam[:3, 0], B_cam[:3, 1], B_cam[:3, 2], B_cam[:3, 3] = xa, ya, v, cam_pos
        mats.append(B_cam @ Finv)

        m = n_points
        if ragged:
            m = max(1, int(n_points * rng.uniform(0.7, 1.0)))
        side = int(np.ceil(np.sqrt(m)))
        idx = rng.permutation(side * side)[:m] if side * side != m else np.arange(m)
        iu = (idx % side).astype(np.float64)
        iv = (idx // side).astype(np.float64)
        u = ((iu + rng.uniform(0.0, 1.0, m)) / side - 0.5) * 0.7
        w = ((iv + rng.uniform(0.0, 1.0, m)) / side - 0.5) * 0.7
        dirs = np.stack([u, w, np.ones(m)], axis=1)
        n_c = B_cam[:3, :3].T @ n
        d_c = float(n @ cam_pos + d)
        t = -d_c / (dirs @ n_c)
        t = t + rng.normal(0.0, noise, m) if noise > 0 else t
        clouds.append((dirs * t[:, None]).astype(np.float32))

    plane = np.concatenate([n, [d]]).astype(np.float32)
    return SyntheticScene(
        clouds=clouds,
        base_T_flange=np.stack(mats),
        plane=plane,
        pose_true=POSE_TRUE.copy(),
        pose_init=POSE_TRUE + INIT_OFFSET,
        seed=seed,
        meta={"n_clouds": n_clouds, "n_points_per_cloud": n_points, "noise": noise, "ragged": ragged},
    )
