import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
if os.environ.get('PROBE_LIB'):
    import cameracalibration_amd.binding as _b
    _b.lib_path = lambda: os.path.abspath(os.environ['PROBE_LIB'])
from cameracalibration_amd.synthetic import make_scene
nc, npts = int(sys.argv[1]), int(sys.argv[2])
s = make_scene(nc, npts)
opt = cc.Optimizer(0)
opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
o = cc.GdOptions.default()
if os.environ.get("PROBE_MAXIT"):
    o.max_iterations = int(os.environ["PROBE_MAXIT"])
if os.environ.get('PROBE_NOSPLIT'):
    opt.set_chain_split(False)
if os.environ.get('PROBE_NOSCREEN'):
    opt.set_gd_screening(False)
t = time.time(); r = opt.gradient_descent(s.pose_init, o); print("gd", r.iterations, r.n_evals, r.fval, time.time() - t, flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "10":
    x0 = np.concatenate([r.xval, s.plane.astype(np.float64)])
    t = time.time(); r10 = opt.gradient_descent(x0, o); print("gd10", r10.iterations, r10.n_evals, r10.fval, time.time() - t, flush=True)
