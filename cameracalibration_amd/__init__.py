"""cameracalibration_amd -- MI355X-native hot path of REXJJ/CameraCalibration.

The product is ``libccal.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/ccal.h``) and the C++ ``camera_calibration_optimization`` CLI built on
it.  This package is the thin Python host layer used by the tests and the
benchmark: a ctypes binding that mirrors the reference's ``Optimizer`` member
functions (src/camera_calibration_optimization.cpp:42-61) and a synthetic scene
generator.  Nothing here computes a residual on the CPU: every objective value
comes from the HIP library, and importing the binding fails loudly when the
library is missing.
"""

from .binding import (  # noqa: F401
    CCAL_DOWNSAMPLED,
    CCAL_RAW,
    CcalError,
    GdOptions,
    GdResult,
    Grid,
    GridStats,
    Optimizer,
    comm_unique_id,
    device_count,
    filter_z,
    parse_pose,
    grid_pose,
    lib_path,
    load_library,
    pose_to_matrix,
    read_point_cloud,
    run_config,
    voxel_grid,
    write_config,
)
from .synthetic import SyntheticScene, make_scene  # noqa: F401
