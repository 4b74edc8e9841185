"""ctypes binding of libccal.so, shaped after the reference's ``Optimizer`` class.

Reference seams mirrored here (src/camera_calibration_optimization.cpp):
  Optimizer::getInputs   :148-278  -> Optimizer.upload_clouds / set_plane / fit_plane
  Optimizer::getError    :280-347  -> Optimizer.get_error
  Optimizer::printError  :348-407  -> Optimizer.per_cloud_error
  discreteCombinatorialOptimization :526-558 -> Optimizer.grid_search
  gradientDescent[WithPlane]        :411-524 -> Optimizer.gradient_descent

There is no CPU implementation behind these calls.  ``load_library`` raises if
``libccal.so`` has not been built, and every compute call raises ``CcalError``
when the library reports a failure (for instance: no GPU).
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

CCAL_DOWNSAMPLED = 0
CCAL_RAW = 1

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path() -> str:
    return os.path.join(_HERE, "libccal.so")


class CcalError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"ccal status {status}: {message}")
        self.status = status


class Grid(C.Structure):
    """ccal_grid: axis k perturbs pose[k]; 0..2 in mm, 3..5 in degrees."""

    _fields_ = [("min", C.c_double * 6), ("step", C.c_double * 6), ("count", C.c_int * 6)]

    @classmethod
    def default(cls) -> "Grid":
        g = cls()
        load_library().ccal_grid_default(C.byref(g))
        return g

    @classmethod
    def make(cls, mins: Sequence[float], steps: Sequence[float], counts: Sequence[int]) -> "Grid":
        g = cls()
        for k in range(6):
            g.min[k] = float(mins[k])
            g.step[k] = float(steps[k])
            g.count[k] = int(counts[k])
        return g

    @property
    def size(self) -> int:
        n = 1
        for k in range(6):
            n *= self.count[k]
        return n


class GridStats(C.Structure):
    _fields_ = [
        ("n_candidates", C.c_int64),
        ("n_rescored", C.c_int64),
        ("residuals", C.c_double),
        ("kernel_ms", C.c_double),
        ("kernel_launches", C.c_int),
        ("total_ms", C.c_double),
        ("band_abs", C.c_double),
        ("band_rel", C.c_double),
    ]


class Profile(C.Structure):
    """ccal_profile: device-time counters of the exact and screening passes, plane fits and point report."""

    _fields_ = [
        ("split_chain_ms", C.c_double), ("split_finish_ms", C.c_double), ("split_pose_points", C.c_double),
        ("split_requests", C.c_int64),
        ("whole_chain_ms", C.c_double), ("whole_pose_points", C.c_double), ("whole_requests", C.c_int64),
        ("screen_ms", C.c_double), ("screen_pose_points", C.c_double), ("screen_requests", C.c_int64),
        ("exact_host_ms", C.c_double), ("exact_wait_ms", C.c_double),
        ("screen_host_ms", C.c_double), ("screen_wait_ms", C.c_double),
        ("ransac_score_ms", C.c_double), ("ransac_tests", C.c_double), ("ransac_calls", C.c_int64),
        ("ransac_recount_ms", C.c_double), ("ransac_recounted", C.c_int64),
        ("moments_ms", C.c_double), ("moments_points", C.c_double), ("moments_calls", C.c_int64),
        ("point_error_ms", C.c_double), ("point_error_points", C.c_double),
    ]


class GdOptions(C.Structure):
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

    @classmethod
    def default(cls) -> "GdOptions":
        o = cls()
        load_library().ccal_gd_default_opts(C.byref(o))
        return o


class _GdResultC(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("fval", C.c_double),
        ("xval", C.c_double * 10),
        ("n_evals", C.c_int64),
    ]


@dataclass
class GdResult:
    iterations: int
    converged: bool
    fval: float
    xval: np.ndarray
    n_evals: int


_LIB: Optional[C.CDLL] = None

# name -> (restype, argtypes); also the list the symbol-export test walks.
_P = C.POINTER
_D = _P(C.c_double)
_F = _P(C.c_float)
_CTX = C.c_void_p
SIGNATURES = {
    "ccal_create": (C.c_int, [C.c_int, _P(_CTX)]),
    "ccal_destroy": (C.c_int, [_CTX]),
    "ccal_last_error": (C.c_char_p, [_CTX]),
    "ccal_status_string": (C.c_char_p, [C.c_int]),
    "ccal_device_count": (C.c_int, []),
    "ccal_upload_clouds": (C.c_int, [_CTX, C.c_int, C.c_int, _P(_F), _P(C.c_size_t), _D]),
    "ccal_downsample_raw": (C.c_int, [_CTX, C.c_float, _P(C.c_size_t)]),
    "ccal_cloud_size": (C.c_int, [_CTX, C.c_int, C.c_int, _P(C.c_size_t)]),
    "ccal_download_cloud": (C.c_int, [_CTX, C.c_int, C.c_int, _F]),
    "ccal_set_plane": (C.c_int, [_CTX, _F]),
    "ccal_get_plane": (C.c_int, [_CTX, _F]),
    "ccal_eval": (C.c_int, [_CTX, C.c_int, C.c_int, _D, _D, _D, _D, _D]),
    "ccal_eval_fast": (C.c_int, [_CTX, C.c_int, C.c_int64, _D, _F]),
    "ccal_eval_bounded": (C.c_int, [_CTX, C.c_int, C.c_int, _D, _D, _D, _D]),
    "ccal_ordered_sum": (C.c_int, [_CTX, _D, C.c_size_t, _D]),
    "ccal_set_fast_kernel": (C.c_int, [_CTX, C.c_int]),
    "ccal_set_search_mode": (C.c_int, [_CTX, C.c_int]),
    "ccal_set_gd_screening": (C.c_int, [_CTX, C.c_int]),
    "ccal_set_chain_split": (C.c_int, [_CTX, C.c_int]),
    "ccal_set_gd_pipelining": (C.c_int, [_CTX, C.c_int]),
    "ccal_set_profiling": (C.c_int, [_CTX, C.c_int]),
    "ccal_get_profile": (C.c_int, [_CTX, _P(Profile), C.c_int]),
    "ccal_point_errors": (C.c_int, [_CTX, C.c_int, _D, _D, C.c_int, _D, _P(C.c_ubyte)]),
    "ccal_write_config": (C.c_int, [C.c_char_p, C.c_char_p, _D]),
    "ccal_grid_default": (None, [_P(Grid)]),
    "ccal_grid_size": (C.c_int64, [_P(Grid)]),
    "ccal_grid_pose": (None, [_P(Grid), _D, C.c_int64, _D]),
    "ccal_grid_search": (
        C.c_int,
        [_CTX, C.c_int, _D, _P(Grid), C.c_int64, C.c_int64, _P(C.c_int64), _D, _D, _P(GridStats)],
    ),
    "ccal_grid_costs": (C.c_int, [_CTX, C.c_int, _D, _P(Grid), C.c_int64, C.c_int64, _F]),
    "ccal_grid_error_bound": (C.c_int, [_CTX, C.c_int, _D, _P(Grid), _D, _D]),
    "ccal_argmin_merge": (C.c_int, [C.c_int, _D, _P(C.c_int64)]),
    "ccal_device_memory": (C.c_int, [_CTX, _P(C.c_size_t), _P(C.c_size_t)]),
    "ccal_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ccal_comm_create": (C.c_int, [_CTX, C.c_int, C.c_int, C.c_char_p, _P(C.c_void_p)]),
    "ccal_comm_destroy": (C.c_int, [C.c_void_p]),
    "ccal_comm_info": (C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int)]),
    "ccal_argmin_allgather": (C.c_int, [_CTX, C.c_void_p, C.c_double, C.c_int64, _D, _P(C.c_int64), _P(C.c_int)]),
    "ccal_fit_plane_lsq": (C.c_int, [_CTX, C.c_int, _D, _F]),
    "ccal_fit_plane_lsq_static": (C.c_int, [_CTX, C.c_int, _D, _F]),
    "ccal_ransac_plane": (
        C.c_int,
        [_CTX, C.c_int, _D, C.c_int, C.c_float, C.c_uint64, _F, _P(C.c_int64), _P(C.c_int)],
    ),
    "ccal_gd_default_opts": (None, [_P(GdOptions)]),
    "ccal_gd_minimize": (C.c_int, [_CTX, C.c_int, C.c_int, _D, _P(GdOptions), _P(_GdResultC)]),
    "ccal_gd_minimize_multi": (
        C.c_int,
        [_CTX, C.c_int, C.c_int, C.c_int, _D, _P(GdOptions), _P(_GdResultC)],
    ),
    "ccal_pose_to_matrix": (None, [_D, _D]),
    "ccal_read_point_cloud": (C.c_int, [C.c_char_p, C.c_char_p, _P(_F), _P(C.c_size_t)]),
    "ccal_voxel_grid": (C.c_int, [_F, C.c_size_t, C.c_float, _F, _P(C.c_size_t)]),
    "ccal_free": (None, [C.c_void_p]),
    "ccal_parse_pose": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, _D]),
    "ccal_filter_z": (C.c_int, [_F, C.c_size_t, C.c_double, _F, _P(C.c_size_t)]),
    "ccal_run_config": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
}


def load_library() -> C.CDLL:
    """Loads libccal.so (built in-tree by ``__graft_entry__.build()`` / make)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make -C cameracalibration_amd/csrc` "
            "(there is no CPU fallback)"
        )
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def device_count() -> int:
    return int(load_library().ccal_device_count())


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(_D)


def pose_to_matrix(pose: Sequence[float]) -> np.ndarray:
    p = np.ascontiguousarray(pose, dtype=np.float64)
    out = np.empty(16, dtype=np.float64)
    load_library().ccal_pose_to_matrix(_dptr(p), _dptr(out))
    return out.reshape(4, 4)


def grid_pose(grid: Grid, base_pose: Sequence[float], index: int) -> np.ndarray:
    b = np.ascontiguousarray(base_pose, dtype=np.float64)
    out = np.empty(6, dtype=np.float64)
    load_library().ccal_grid_pose(C.byref(grid), _dptr(b), int(index), _dptr(out))
    return out


class Optimizer:
    """One GPU context; the counterpart of the reference's global ``Optimizer opti``."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._ctx = _CTX()
        rc = self._lib.ccal_create(int(device), C.byref(self._ctx))
        if rc != 0:
            raise CcalError(rc, self._lib.ccal_status_string(rc).decode() + " (ccal_create)")
        self.n_clouds = {CCAL_DOWNSAMPLED: 0, CCAL_RAW: 0}
        self.n_points = {CCAL_DOWNSAMPLED: 0, CCAL_RAW: 0}

    # -- lifetime ----------------------------------------------------------
    def close(self) -> None:
        self.comm_destroy()
        if self._ctx:
            self._lib.ccal_destroy(self._ctx)
            self._ctx = _CTX()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise CcalError(rc, self._lib.ccal_last_error(self._ctx).decode())

    # -- inputs --------------------------------------------------------------
    def upload_clouds(self, which: int, clouds: Sequence[np.ndarray], base_T_flange: np.ndarray) -> None:
        """clouds[j]: (N_j,3) float32; base_T_flange: (n,4,4) float64, already mapped."""
        n = len(clouds)
        arrs = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1, 3) for c in clouds]
        ptrs = (_F * max(n, 1))(*[a.ctypes.data_as(_F) for a in arrs])
        sizes = (C.c_size_t * max(n, 1))(*[a.shape[0] for a in arrs])
        B = np.ascontiguousarray(base_T_flange, dtype=np.float64).reshape(-1)
        if B.size != 16 * n:
            raise ValueError("base_T_flange must hold one 4x4 matrix per cloud")
        self._check(self._lib.ccal_upload_clouds(self._ctx, which, n, ptrs, sizes, _dptr(B)))
        self.n_clouds[which] = n
        self.n_points[which] = int(sum(a.shape[0] for a in arrs))

    def downsample_raw(self, leaf: float = 0.02):
        """Device voxel grid: RAW set -> DOWNSAMPLED set; returns the per-cloud centroid counts."""
        n = self.n_clouds[CCAL_RAW]
        sizes = (C.c_size_t * max(n, 1))()
        self._check(self._lib.ccal_downsample_raw(self._ctx, float(leaf), sizes))
        self.n_clouds[CCAL_DOWNSAMPLED] = n
        self.n_points[CCAL_DOWNSAMPLED] = int(sum(sizes[:n]))
        return [int(v) for v in sizes[:n]]

    def download_cloud(self, which: int, cloud: int) -> np.ndarray:
        n = C.c_size_t(0)
        self._check(self._lib.ccal_cloud_size(self._ctx, which, cloud, C.byref(n)))
        out = np.empty((n.value, 3), dtype=np.float32)
        self._check(self._lib.ccal_download_cloud(self._ctx, which, cloud, out.ctypes.data_as(_F)))
        return out

    def set_fast_kernel(self, kind: int) -> None:
        """0 = bf16x3 split on the bf16 matrix pipe, 1 = FP32-input MFMA, 2 = bf16x3 as 32x32x16 pairs."""
        self._check(self._lib.ccal_set_fast_kernel(self._ctx, int(kind)))

    def set_search_mode(self, mode: int) -> None:
        """0 = dense (every residual on the matrix cores), 1 = separable (one histogram pass per rotation)."""
        self._check(self._lib.ccal_set_search_mode(self._ctx, int(mode)))

    def set_gd_screening(self, on: bool) -> None:
        """Line-search trials are first scored on the fast path and decided from that score when its
        error bound leaves no doubt (default on; results are identical either way)."""
        self._check(self._lib.ccal_set_gd_screening(self._ctx, 1 if on else 0))

    def set_chain_split(self, on: bool) -> None:
        """A descent's small exact requests cut every (pose, cloud) sum into concurrently folded parts
        (default on; results are identical either way)."""
        self._check(self._lib.ccal_set_chain_split(self._ctx, 1 if on else 0))

    def set_gd_pipelining(self, on: bool) -> None:
        """A lone descent issues its next screening pass ahead of the current trial's curvature test
        (default off: measured slower; same iterates and counts either way)."""
        self._check(self._lib.ccal_set_gd_pipelining(self._ctx, 1 if on else 0))

    def set_profiling(self, on: bool) -> None:
        """HIP-event timing of the exact and screening kernels (read with ``profile()``)."""
        self._check(self._lib.ccal_set_profiling(self._ctx, 1 if on else 0))

    def profile(self, reset: bool = True) -> Profile:
        p = Profile()
        self._check(self._lib.ccal_get_profile(self._ctx, C.byref(p), 1 if reset else 0))
        return p

    def set_plane(self, plane: Sequence[float]) -> None:
        p = np.ascontiguousarray(plane, dtype=np.float32)
        self._check(self._lib.ccal_set_plane(self._ctx, p.ctypes.data_as(_F)))

    def get_plane(self) -> np.ndarray:
        p = np.empty(4, dtype=np.float32)
        self._check(self._lib.ccal_get_plane(self._ctx, p.ctypes.data_as(_F)))
        return p

    # -- objective -----------------------------------------------------------
    def get_error(self, poses, planes=None, which: int = CCAL_DOWNSAMPLED) -> np.ndarray:
        """Reference-arithmetic objective for a batch of poses (n,6) [+ planes (n,4)]."""
        P = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
        n = P.shape[0]
        cost = np.empty(n, dtype=np.float64)
        pl = None
        if planes is not None:
            pl = np.ascontiguousarray(planes, dtype=np.float64).reshape(-1, 4)
            if pl.shape[0] != n:
                raise ValueError("one plane per pose")
        self._check(
            self._lib.ccal_eval(
                self._ctx, which, n, _dptr(P), _dptr(pl) if pl is not None else None,
                _dptr(cost), None, None,
            )
        )
        return cost

    def per_cloud_error(self, pose, which: int = CCAL_DOWNSAMPLED):
        """(cost, mean[n_clouds], max[n_clouds]) in metres for one pose."""
        P = np.ascontiguousarray(pose, dtype=np.float64).reshape(1, 6)
        nc = self.n_clouds[which]
        cost = np.empty(1, dtype=np.float64)
        mean = np.empty(nc, dtype=np.float64)
        mx = np.empty(nc, dtype=np.float64)
        self._check(
            self._lib.ccal_eval(self._ctx, which, 1, _dptr(P), None, _dptr(cost), _dptr(mean), _dptr(mx))
        )
        return float(cost[0]), mean, mx

    def ordered_sum(self, values) -> float:
        """((0 + v0) + v1) + ... with a rounding after every add, evaluated on the GPU."""
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        out = C.c_double(0.0)
        self._check(self._lib.ccal_ordered_sum(self._ctx, _dptr(v), v.size, C.byref(out)))
        return float(out.value)

    def point_errors(self, pose, plane=None, viewer_arithmetic: bool = False, which: int = CCAL_RAW):
        """(distances [m], classes 0..3) of every point of the set for one pose."""
        p = np.ascontiguousarray(pose, dtype=np.float64)
        pl = None if plane is None else np.ascontiguousarray(plane, dtype=np.float64)
        n = self.n_points[which]
        dist = np.empty(n, dtype=np.float64)
        cls = np.empty(n, dtype=np.uint8)
        self._check(
            self._lib.ccal_point_errors(self._ctx, which, _dptr(p), _dptr(pl) if pl is not None else None,
                                        1 if viewer_arithmetic else 0, _dptr(dist),
                                        cls.ctypes.data_as(_P(C.c_ubyte)))
        )
        return dist, cls

    def get_error_fast(self, poses, which: int = CCAL_DOWNSAMPLED) -> np.ndarray:
        P = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
        cost = np.empty(P.shape[0], dtype=np.float32)
        self._check(
            self._lib.ccal_eval_fast(self._ctx, which, P.shape[0], _dptr(P), cost.ctypes.data_as(_F))
        )
        return cost

    def get_error_bounded(self, poses, planes=None, which: int = CCAL_DOWNSAMPLED):
        """(cost, bound): the FP32 objective and a bound on its distance from get_error's value."""
        P = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
        pl = None if planes is None else np.ascontiguousarray(planes, dtype=np.float64).reshape(-1, 4)
        cost = np.empty(P.shape[0], dtype=np.float64)
        bound = np.empty(P.shape[0], dtype=np.float64)
        self._check(self._lib.ccal_eval_bounded(self._ctx, which, P.shape[0], _dptr(P),
                                                _dptr(pl) if pl is not None else None, _dptr(cost), _dptr(bound)))
        return cost, bound

    # -- search ----------------------------------------------------------------
    def grid_search(self, base_pose, grid: Optional[Grid] = None, begin: int = 0,
                    end: Optional[int] = None, which: int = CCAL_DOWNSAMPLED):
        """Returns (best_index, best_cost, best_pose, GridStats) for candidates [begin,end)."""
        g = grid if grid is not None else Grid.default()
        if end is None:
            end = g.size
        b = np.ascontiguousarray(base_pose, dtype=np.float64)
        idx = C.c_int64(-1)
        cost = C.c_double(0.0)
        pose = np.empty(6, dtype=np.float64)
        st = GridStats()
        self._check(
            self._lib.ccal_grid_search(
                self._ctx, which, _dptr(b), C.byref(g), int(begin), int(end),
                C.byref(idx), C.byref(cost), _dptr(pose), C.byref(st),
            )
        )
        return int(idx.value), float(cost.value), pose, st

    def grid_costs(self, base_pose, grid: Optional[Grid] = None, begin: int = 0,
                   end: Optional[int] = None, which: int = CCAL_DOWNSAMPLED) -> np.ndarray:
        """Fast-path cost of every candidate of [begin,end) in the current search mode."""
        g = grid if grid is not None else Grid.default()
        if end is None:
            end = g.size
        b = np.ascontiguousarray(base_pose, dtype=np.float64)
        out = np.empty(int(end) - int(begin), dtype=np.float32)
        self._check(self._lib.ccal_grid_costs(self._ctx, which, _dptr(b), C.byref(g), int(begin), int(end),
                                              out.ctypes.data_as(_F)))
        return out

    def grid_error_bound(self, base_pose, grid: Optional[Grid] = None, which: int = CCAL_DOWNSAMPLED):
        """(abs, rel): |grid_costs value - get_error value| <= abs + rel * grid_costs value on the whole grid."""
        g = grid if grid is not None else Grid.default()
        b = np.ascontiguousarray(base_pose, dtype=np.float64)
        a, r = C.c_double(0.0), C.c_double(0.0)
        self._check(self._lib.ccal_grid_error_bound(self._ctx, which, _dptr(b), C.byref(g), C.byref(a), C.byref(r)))
        return float(a.value), float(r.value)

    # -- the arg-min exchange over RCCL (one process per GPU) -----------------------
    def comm_create(self, world_size: int, rank: int, unique_id: bytes) -> None:
        """Collective: joins the RCCL communicator named by `unique_id` (comm_unique_id() of rank 0)."""
        if len(unique_id) != 128:
            raise ValueError("the communicator id is 128 bytes")
        comm = C.c_void_p()
        self._check(self._lib.ccal_comm_create(self._ctx, int(world_size), int(rank), unique_id, C.byref(comm)))
        self._comm = comm

    def comm_destroy(self) -> None:
        if getattr(self, "_comm", None):
            self._lib.ccal_comm_destroy(self._comm)
            self._comm = None

    def comm_info(self):
        """(world size, rank) of the communicator as RCCL itself reports them."""
        n, k = C.c_int(0), C.c_int(-1)
        self._check(self._lib.ccal_comm_info(self._comm, C.byref(n), C.byref(k)))
        return int(n.value), int(k.value)

    def argmin_allgather(self, cost: float, index: int):
        """(best_cost, best_index, winner_rank) over all ranks' (cost, index); index < 0 = empty shard."""
        bc, bi, w = C.c_double(0.0), C.c_int64(-1), C.c_int(-1)
        self._check(self._lib.ccal_argmin_allgather(self._ctx, self._comm, float(cost), int(index),
                                                    C.byref(bc), C.byref(bi), C.byref(w)))
        return float(bc.value), int(bi.value), int(w.value)

    def device_memory(self):
        """(free, total) bytes of the context's GPU."""
        f, t = C.c_size_t(0), C.c_size_t(0)
        self._check(self._lib.ccal_device_memory(self._ctx, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def fit_plane(self, pose, which: int = CCAL_RAW) -> np.ndarray:
        p = np.ascontiguousarray(pose, dtype=np.float64)
        out = np.empty(4, dtype=np.float32)
        self._check(self._lib.ccal_fit_plane_lsq(self._ctx, which, _dptr(p), out.ctypes.data_as(_F)))
        return out

    def fit_plane_static(self, pose, which: int = CCAL_RAW) -> np.ndarray:
        """fit_plane through the entry point that allocates nothing (static device scratch)."""
        p = np.ascontiguousarray(pose, dtype=np.float64)
        out = np.empty(4, dtype=np.float32)
        self._check(self._lib.ccal_fit_plane_lsq_static(self._ctx, which, _dptr(p), out.ctypes.data_as(_F)))
        return out

    def ransac_plane(self, pose, n_hyp: int, threshold: float, seed: int, which: int = CCAL_RAW):
        """(plane float32[4], inlier count, index of the winning hypothesis)."""
        p = np.ascontiguousarray(pose, dtype=np.float64)
        out = np.empty(4, dtype=np.float32)
        inl = C.c_int64(0)
        best = C.c_int(-1)
        self._check(
            self._lib.ccal_ransac_plane(
                self._ctx, which, _dptr(p), int(n_hyp), float(threshold), int(seed),
                out.ctypes.data_as(_F), C.byref(inl), C.byref(best),
            )
        )
        return out, int(inl.value), int(best.value)

    def gradient_descent(self, x0, opts: Optional[GdOptions] = None,
                         which: int = CCAL_DOWNSAMPLED):
        """x0: (6,) or (10,) for one descent, (n,6)/(n,10) for lock-step multi-start."""
        X = np.ascontiguousarray(x0, dtype=np.float64)
        single = X.ndim == 1
        X = X.reshape(1, -1) if single else X
        n, dim = X.shape
        res = (_GdResultC * n)()
        self._check(
            self._lib.ccal_gd_minimize_multi(
                self._ctx, which, dim, n, _dptr(X), C.byref(opts) if opts is not None else None, res
            )
        )
        out = [
            GdResult(r.iterations, bool(r.converged), r.fval, np.array(r.xval[:dim]), r.n_evals)
            for r in res
        ]
        return out[0] if single else out


def comm_unique_id() -> bytes:
    """128 bytes naming a new RCCL communicator (rank 0 makes it, every rank passes it to comm_create)."""
    buf = C.create_string_buffer(128)
    rc = load_library().ccal_comm_unique_id(buf)
    if rc != 0:
        raise CcalError(rc, "ccal_comm_unique_id (is librccl available?)")
    return buf.raw


def argmin_merge(costs: Sequence[float], indices: Sequence[int]) -> int:
    c = np.ascontiguousarray(costs, dtype=np.float64)
    i = np.ascontiguousarray(indices, dtype=np.int64)
    return int(load_library().ccal_argmin_merge(len(c), _dptr(c), i.ctypes.data_as(_P(C.c_int64))))


def read_point_cloud(filename: str, metric: str = "m") -> np.ndarray:
    """Cloud file (.ply / .pcd / .xyz) -> (N,3) float32, as readPointCloud (helpers.cpp:108-207)."""
    lib = load_library()
    ptr = _F()
    n = C.c_size_t(0)
    rc = lib.ccal_read_point_cloud(filename.encode(), metric.encode(), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise CcalError(rc, f"cannot read {filename}")
    try:
        out = np.ctypeslib.as_array(ptr, shape=(n.value * 3,)).copy() if n.value else np.zeros(0, np.float32)
    finally:
        lib.ccal_free(ptr)
    return out.reshape(-1, 3)


def voxel_grid(xyz: np.ndarray, leaf: float = 0.02) -> np.ndarray:
    a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.empty_like(a)
    n = C.c_size_t(0)
    rc = load_library().ccal_voxel_grid(a.ctypes.data_as(_F), a.shape[0], float(leaf),
                                        out.ctypes.data_as(_F), C.byref(n))
    if rc != 0:
        raise CcalError(rc, "voxel_grid")
    return out[: n.value].copy()


def parse_pose(value_csv: str, metric: str = "m", angle: str = "rad") -> np.ndarray:
    """getTransVector (camera_calibration_optimization.cpp:70-98): CSV -> (tx,ty,tz,rz,ry,rx) in m / rad."""
    out = np.zeros(6, dtype=np.float64)
    rc = load_library().ccal_parse_pose(value_csv.encode(), metric.encode(), angle.encode(), _dptr(out))
    if rc != 0:
        raise CcalError(rc, f"cannot parse pose '{value_csv}'")
    return out


def filter_z(xyz: np.ndarray, z_max: float = 0.8) -> np.ndarray:
    """The depth cut `pts.z<=0.8` of camera_calibration_optimization.cpp:167-169 (double comparison)."""
    a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.empty_like(a)
    n = C.c_size_t(0)
    rc = load_library().ccal_filter_z(a.ctypes.data_as(_F), a.shape[0], float(z_max),
                                      out.ctypes.data_as(_F), C.byref(n))
    if rc != 0:
        raise CcalError(rc, "filter_z")
    return out[: n.value].copy()


def run_config(config: str, results: str, errors: str, verbose: bool = False) -> None:
    """The whole optimiser run (the CLI's body) as a library call."""
    buf = C.create_string_buffer(1024)
    rc = load_library().ccal_run_config(config.encode(), results.encode(), errors.encode(),
                                        int(verbose), buf, len(buf))
    if rc != 0:
        raise CcalError(rc, buf.value.decode(errors="replace"))


def write_config(config_in: str, config_out: str, pose) -> None:
    p = np.ascontiguousarray(pose, dtype=np.float64)
    rc = load_library().ccal_write_config(config_in.encode(), config_out.encode(), _dptr(p))
    if rc != 0:
        raise CcalError(rc, f"cannot write {config_out}")
