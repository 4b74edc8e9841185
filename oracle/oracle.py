"""ctypes access to the CPU oracle (oracle/libccal_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and
by the ``cpu_baseline`` leg of bench.py, always as the checker.  The product
(cameracalibration_amd, libccal.so, the CLI) never imports or links this.
Parity status: unpinned (see ccal_oracle.h).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None

_D = C.POINTER(C.c_double)
_F = C.POINTER(C.c_float)


class Scene(C.Structure):
    _fields_ = [
        ("n_clouds", C.c_int),
        ("xyz", C.POINTER(_F)),
        ("n_pts", C.POINTER(C.c_size_t)),
        ("base_T_flange", _D),
    ]


class Grid(C.Structure):
    _fields_ = [("min", C.c_double * 6), ("step", C.c_double * 6), ("count", C.c_int * 6)]


class GdOpts(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("min_gradient_len", C.c_double),
        ("min_step_len", C.c_double),
        ("momentum", C.c_double),
        ("fd_eps", C.c_double),
        ("wolfe_decrease", C.c_double),
        ("wolfe_c1", C.c_double),
        ("wolfe_c2", C.c_double),
        ("wolfe_min_step", C.c_double),
        ("wolfe_max_step", C.c_double),
        ("wolfe_max_it", C.c_int),
    ]


class GdResult(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("fval", C.c_double),
        ("xval", C.c_double * 10),
        ("n_evals", C.c_int64),
    ]


def build() -> str:
    """Compiles the oracle with gcc if the shared object is missing or stale."""
    so = os.path.join(_HERE, "libccal_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ccal_oracle.c", "ccal_oracle_omp.c", "ccal_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libccal_oracle.so")
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        L.orc_degree_to_radian.restype = C.c_double
        L.orc_degree_to_radian.argtypes = [C.c_int]
        L.orc_pose_to_matrix.argtypes = [_D, _D]
        L.orc_matmul4.argtypes = [_D, _D, _D]
        L.orc_transform_point.argtypes = [_D, _F, _F]
        L.orc_distance_planef.restype = C.c_double
        L.orc_distance_planef.argtypes = [_F, _F]
        L.orc_distance_planed.restype = C.c_double
        L.orc_distance_planed.argtypes = [_D, _F]
        L.orc_get_error.restype = C.c_double
        L.orc_get_error.argtypes = [C.POINTER(Scene), _D, _F]
        L.orc_get_error_plane.restype = C.c_double
        L.orc_get_error_plane.argtypes = [C.POINTER(Scene), _D, _D]
        L.orc_per_cloud_error.argtypes = [C.POINTER(Scene), _D, _F, _D, _D]
        L.orc_grid_default.argtypes = [C.POINTER(Grid)]
        L.orc_grid_size.restype = C.c_int64
        L.orc_grid_size.argtypes = [C.POINTER(Grid)]
        L.orc_grid_pose.argtypes = [C.POINTER(Grid), _D, C.c_int64, _D]
        L.orc_grid_search.argtypes = [
            C.POINTER(Scene), _F, _D, C.POINTER(Grid), C.c_int64, C.c_int64,
            C.POINTER(C.c_int64), _D, _D, _D,
        ]
        L.orc_grid_costs_omp.argtypes = [
            C.POINTER(Scene), _F, _D, C.POINTER(Grid), C.c_int64, C.c_int64, _D,
            C.POINTER(C.c_int64), _D,
        ]
        L.orc_omp_threads.restype = C.c_int
        L.orc_fit_plane.argtypes = [_F, C.c_size_t, _F]
        L.orc_combined_cloud.restype = C.c_size_t
        L.orc_combined_cloud.argtypes = [C.POINTER(Scene), _D, _F]
        L.orc_voxel_grid.restype = C.c_size_t
        L.orc_voxel_grid.argtypes = [_F, C.c_size_t, C.c_float, _F]
        L.orc_gd_default_opts.argtypes = [C.POINTER(GdOpts)]
        L.orc_gd_minimize.argtypes = [
            C.POINTER(Scene), _F, C.c_int, _D, C.POINTER(GdOpts), C.POINTER(GdResult),
        ]
        L.orc_ransac_plane.argtypes = [C.POINTER(Scene), _D, C.c_int, C.c_float, C.c_uint64, _D,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int), _D, _D]
        L.orc_ransac_hypothesis_plane.restype = C.c_int
        L.orc_ransac_hypothesis_plane.argtypes = [C.POINTER(Scene), _D, C.c_uint64, C.c_int, _D]
        L.orc_ransac_counts.argtypes = [C.POINTER(Scene), _D, C.c_uint64, C.c_float, C.POINTER(C.c_int), C.c_int,
                                        C.POINTER(C.c_int64)]
        L.orc_point_errors.argtypes = [C.POINTER(Scene), _D, _D, C.c_int, _D, C.POINTER(C.c_ubyte)]
        L.orc_parse_pose_csv.restype = C.c_int
        L.orc_parse_pose_csv.argtypes = [C.c_char_p, C.c_double, C.c_int, _D]
        L.orc_file_id.restype = C.c_int
        L.orc_file_id.argtypes = [C.c_char_p]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(_D)


def _f(a):
    return a.ctypes.data_as(_F)


class OracleScene:
    """Keeps the numpy buffers alive behind an orc_scene."""

    def __init__(self, clouds: Sequence[np.ndarray], base_T_flange: np.ndarray):
        self.clouds = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1, 3) for c in clouds]
        n = len(self.clouds)
        self.B = np.ascontiguousarray(base_T_flange, dtype=np.float64).reshape(n, 16)
        self._ptrs = (_F * n)(*[_f(c) for c in self.clouds])
        self._sizes = (C.c_size_t * n)(*[c.shape[0] for c in self.clouds])
        self.c = Scene(n, self._ptrs, self._sizes, _d(self.B))
        self.n_clouds = n

    def get_error(self, pose, plane) -> float:
        p = np.ascontiguousarray(pose, dtype=np.float64)
        pl = np.ascontiguousarray(plane, dtype=np.float32)
        return float(lib().orc_get_error(C.byref(self.c), _d(p), _f(pl)))

    def get_error_plane(self, pose, plane) -> float:
        p = np.ascontiguousarray(pose, dtype=np.float64)
        pl = np.ascontiguousarray(plane, dtype=np.float64)
        return float(lib().orc_get_error_plane(C.byref(self.c), _d(p), _d(pl)))

    def per_cloud_error(self, pose, plane):
        p = np.ascontiguousarray(pose, dtype=np.float64)
        pl = np.ascontiguousarray(plane, dtype=np.float32)
        mean = np.empty(self.n_clouds)
        mx = np.empty(self.n_clouds)
        lib().orc_per_cloud_error(C.byref(self.c), _d(p), _f(pl), _d(mean), _d(mx))
        return mean, mx

    def grid_search(self, plane, base, grid: Grid, begin=0, end=None, want_costs=False, omp=False):
        pl = np.ascontiguousarray(plane, dtype=np.float32)
        b = np.ascontiguousarray(base, dtype=np.float64)
        if end is None:
            end = int(lib().orc_grid_size(C.byref(grid)))
        idx = C.c_int64(-1)
        cost = C.c_double(0)
        pose = np.empty(6)
        costs = np.empty(end - begin) if (want_costs or omp) else None
        if omp:
            lib().orc_grid_costs_omp(C.byref(self.c), _f(pl), _d(b), C.byref(grid), begin, end,
                                     _d(costs), C.byref(idx), C.byref(cost))
            grid_pose(grid, b, idx.value, pose)
        else:
            lib().orc_grid_search(C.byref(self.c), _f(pl), _d(b), C.byref(grid), begin, end,
                                  C.byref(idx), C.byref(cost), _d(pose),
                                  _d(costs) if costs is not None else None)
        return int(idx.value), float(cost.value), pose, costs

    def combined_cloud(self, pose) -> np.ndarray:
        p = np.ascontiguousarray(pose, dtype=np.float64)
        total = sum(c.shape[0] for c in self.clouds)
        out = np.empty((total, 3), dtype=np.float32)
        n = lib().orc_combined_cloud(C.byref(self.c), _d(p), _f(out))
        return out[:n]

    def point_errors(self, pose, plane, arithmetic: int):
        p = np.ascontiguousarray(pose, dtype=np.float64)
        pl = np.ascontiguousarray(plane, dtype=np.float64)
        n = sum(c.shape[0] for c in self.clouds)
        dist = np.empty(n)
        cls = np.empty(n, dtype=np.uint8)
        lib().orc_point_errors(C.byref(self.c), _d(p), _d(pl), int(arithmetic), _d(dist),
                               cls.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return dist, cls

    def ransac_plane(self, pose, n_hyp: int, threshold: float, seed: int):
        """(plane float32[4], inliers, winning hypothesis).  Sampling and counting in C; the refit on the
        winner's inliers here with numpy's LAPACK eigen-decomposition (PCL's optimizeModelCoefficients:
        centroid + eigenvector of the smallest covariance eigenvalue, d = -n.centroid)."""
        p = np.ascontiguousarray(pose, dtype=np.float64)
        hyp = np.empty(5)
        m = np.empty(10)
        org = np.empty(3)
        inl = C.c_int64(0)
        best = C.c_int(-1)
        lib().orc_ransac_plane(C.byref(self.c), _d(p), int(n_hyp), float(threshold), int(seed), _d(hyp),
                               C.byref(inl), C.byref(best), _d(m), _d(org))
        n, d = hyp[:3].copy(), hyp[3]
        if best.value >= 0 and m[0] >= 3:
            N = m[0]
            c = m[1:4] / N
            S = np.array([[m[4], m[5], m[7]], [m[5], m[6], m[8]], [m[7], m[8], m[9]]]) / N - np.outer(c, c)
            w, V = np.linalg.eigh(S)
            e = V[:, 0]                                    # eigenvalues ascending
            e = e / np.linalg.norm(e)
            if e[2] < 0:
                e = -e
            n, d = e, -float(e @ (c + org))
        return np.concatenate([n, [d]]).astype(np.float32), int(inl.value), int(best.value)

    def ransac_hypothesis_plane(self, pose, seed: int, h: int):
        """(a, b, c, d, |n|) of hypothesis h, or None for a collinear sample."""
        p = np.ascontiguousarray(pose, dtype=np.float64)
        pl = np.empty(5)
        ok = lib().orc_ransac_hypothesis_plane(C.byref(self.c), _d(p), int(seed), int(h), _d(pl))
        return pl if ok else None

    def ransac_counts(self, pose, seed: int, threshold: float, hyps):
        """Inlier counts of the listed hypotheses over every point (host threads over points)."""
        p = np.ascontiguousarray(pose, dtype=np.float64)
        h = np.ascontiguousarray(hyps, dtype=np.int32)
        out = np.empty(len(h), dtype=np.int64)
        lib().orc_ransac_counts(C.byref(self.c), _d(p), int(seed), float(threshold),
                                h.ctypes.data_as(C.POINTER(C.c_int)), len(h),
                                out.ctypes.data_as(C.POINTER(C.c_int64)))
        return out

    def gd_minimize(self, plane, x0, opts: Optional[GdOpts] = None) -> GdResult:
        pl = np.ascontiguousarray(plane, dtype=np.float32)
        x = np.ascontiguousarray(x0, dtype=np.float64)
        if opts is None:
            opts = gd_default_opts()
        r = GdResult()
        lib().orc_gd_minimize(C.byref(self.c), _f(pl), len(x), _d(x), C.byref(opts), C.byref(r))
        return r


def gd_default_opts() -> GdOpts:
    o = GdOpts()
    lib().orc_gd_default_opts(C.byref(o))
    return o


def grid_default() -> Grid:
    g = Grid()
    lib().orc_grid_default(C.byref(g))
    return g


def make_grid(mins, steps, counts) -> Grid:
    g = Grid()
    for k in range(6):
        g.min[k], g.step[k], g.count[k] = float(mins[k]), float(steps[k]), int(counts[k])
    return g


def grid_pose(grid: Grid, base, index: int, out=None) -> np.ndarray:
    b = np.ascontiguousarray(base, dtype=np.float64)
    if out is None:
        out = np.empty(6)
    lib().orc_grid_pose(C.byref(grid), _d(b), int(index), _d(out))
    return out


def pose_to_matrix(pose) -> np.ndarray:
    p = np.ascontiguousarray(pose, dtype=np.float64)
    T = np.empty(16)
    lib().orc_pose_to_matrix(_d(p), _d(T))
    return T.reshape(4, 4)


def fit_plane(xyz: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.empty(4, dtype=np.float32)
    lib().orc_fit_plane(_f(a), a.shape[0], _f(out))
    return out


def voxel_grid(xyz: np.ndarray, leaf: float = 0.02) -> np.ndarray:
    a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.empty_like(a)
    n = lib().orc_voxel_grid(_f(a), a.shape[0], np.float32(leaf), _f(out))
    return out[:n].copy()


def parse_pose_csv(line: str, scale: float = 1.0, degree: bool = False) -> np.ndarray:
    out = np.zeros(6)
    n = lib().orc_parse_pose_csv(line.encode(), float(scale), int(degree), _d(out))
    return out[:n]


def file_id(name: str) -> int:
    return int(lib().orc_file_id(name.encode()))


def omp_threads() -> int:
    return int(lib().orc_omp_threads())
