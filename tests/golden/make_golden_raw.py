"""Regenerates tests/golden/calib1_raw.npz: the four raw clouds of the reference's example/calib_1
(data files of the reference, cut at z <= 0.8 as src/camera_calibration_optimization.cpp:167-169 does,
float32 as stored) with what oracle/ccal_oracle.c computes from them -- the plane fitted to all 862 392
points carried to the base frame by the initial pose (:240-277), and the raw pass of printError
(:379-405: per-cloud mean and maximum distance) at the initial pose and at the 6-D descent's result.
The reference holds no expected output for any of this: parity unpinned, as for every fixture here.

Run from the repo root in the build container (needs /root/reference and tests/golden/calib1.npz):
    python tests/golden/make_golden_raw.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.golden.make_golden import REF, read_ply_xyz  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    small = np.load(os.path.join(OUT, "calib1.npz"))
    clouds = []
    for i in range(1, 5):
        a = read_ply_xyz(REF + "pointcloud/cloud_%d.ply" % i)
        clouds.append(np.ascontiguousarray(a[a[:, 2].astype(np.float64) <= 0.8]))
    B, pose, gd6 = small["base_T_flange"], small["pose_init"], small["gd6_x"]
    raw = O.OracleScene(clouds, B)
    plane = O.fit_plane(raw.combined_cloud(pose))
    assert np.array_equal(plane, small["plane_full"])
    for k, c in enumerate(clouds):                       # the committed downsampled clouds come from these
        assert np.array_equal(O.voxel_grid(c), small["ds_%d" % k])
    mean0, max0 = raw.per_cloud_error(pose, plane)
    mean1, max1 = raw.per_cloud_error(gd6, plane)
    np.savez_compressed(
        os.path.join(OUT, "calib1_raw.npz"),
        raw_0=clouds[0], raw_1=clouds[1], raw_2=clouds[2], raw_3=clouds[3],
        plane_full=plane,
        raw_mean_init=mean0, raw_max_init=max0, raw_mean_gd6=mean1, raw_max_gd6=max1,
        raw_cost_init=np.array(raw.get_error(pose, plane)), raw_cost_gd6=np.array(raw.get_error(gd6, plane)),
    )
    print("wrote calib1_raw.npz:", [len(c) for c in clouds], plane, mean0 * 1000, max0 * 1000)


if __name__ == "__main__":
    main()
