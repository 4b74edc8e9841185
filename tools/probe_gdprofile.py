"""The benchmark's descent (20 x 300k points, from the winner of its 10^6 grid) with the library's
profiling counters: where an iteration's time goes (kernels, host before / inside / after the waits)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
s = make_scene(20, 300_000)
opt = cc.Optimizer(0)
opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
n = 10
grid = cc.Grid.make([-20] * 3 + [-5] * 3, [40.0 / (n - 1)] * 3 + [10.0 / (n - 1)] * 3, [n] * 6)
opt.set_search_mode(1)
idx, cost, pose, _ = opt.grid_search(s.pose_init, grid)
o = cc.GdOptions.default()
if len(sys.argv) > 1:
    o.max_iterations = int(sys.argv[1])
if os.environ.get("PROBE_PIPELINING"):
    opt.set_gd_pipelining(True)
for rep in range(2):
    t = time.perf_counter(); r = opt.gradient_descent(pose, o); dt = time.perf_counter() - t
    print(f"descent: {r.iterations} iterations, {r.n_evals} evals, fval {r.fval!r}, {dt:.4f} s = {dt / max(r.iterations, 1) * 1e6:.1f} us/iteration", flush=True)
opt.set_profiling(True); opt.profile(reset=True)
t = time.perf_counter(); r = opt.gradient_descent(pose, o); dt = time.perf_counter() - t
p = opt.profile()
it = max(r.iterations, 1)
print(f"profiled: {dt:.4f} s = {dt / it * 1e6:.1f} us/iteration; x = {r.xval.tolist()}")
for name, _ in p._fields_:
    v = getattr(p, name)
    if v:
        print(f"  {name:22s} {v:14.3f}" + (f"   {v / it * 1e3:8.1f} us/iteration" if name.endswith("_ms") else ""))
print(f"  poses per split request {p.split_pose_points / s.n_points / max(p.split_requests, 1):.2f}; per screen request {p.screen_pose_points / s.n_points / max(p.screen_requests, 1):.2f}")
