"""Latency of ccal_eval for runs of nearby poses (an optimiser driving the entry point itself) and for
unrelated poses, 20 x 300k points."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
s = make_scene(20, 300000)
rng = np.random.default_rng(1)
for split in (True, False):
    for n in (1, 13, 31):
        near = [s.pose_init + rng.normal(0, 1e-6, (n, 6)) for _ in range(60)]
        far = [s.pose_init + rng.normal(0, 1, (n, 6)) * np.array([0.02] * 3 + [0.1] * 3) for _ in range(60)]
        for name, lst in (("nearby", near), ("unrelated", far)):
            opt = cc.Optimizer(0)       # a fresh context: the back-off state of one series must not leak into the next
            opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
            opt.set_chain_split(split)
            for p in lst[:10]:
                opt.get_error(p)
            t = time.time()
            for p in lst[10:]:
                opt.get_error(p)
            opt.close()
            print(f"split {split}: {n:2d} poses per call, {name}: {(time.time() - t) / 50 * 1e3:.3f} ms per call", flush=True)
