"""Which kernel sources a profile record was measured on.

The benchmark quotes a few hardware counters (HBM bytes per launch, VALU instructions per
(pose, point)) from rocprofv3 PMC passes that are collected in runs of their own.  A record is
stamped with a hash of the source files its kernel is compiled from; the benchmark recomputes the
hash from the tree it runs in and nulls the quoted figure when the two differ, so a counter is never
reported for a kernel that has changed since it was profiled.
"""

from __future__ import annotations

import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")

# kernel key (tools/summarize_pmc.py) -> the headers that hold the kernel and everything it inlines
KERNEL_SOURCES = {
    "residual": ("kernels_residual.hpp", "bf16x3_split.hpp", "kernels_common.hpp", "pose_math.hpp"),
    "residual_f32": ("kernels_residual.hpp", "bf16x3_split.hpp", "kernels_common.hpp", "pose_math.hpp"),
    "separable": ("kernels_separable.hpp", "kernels_common.hpp"),
    "split_chain": ("kernels_exact.hpp", "kernels_common.hpp", "pose_math.hpp"),
    "split_finish": ("kernels_exact.hpp", "kernels_common.hpp", "pose_math.hpp"),
    "screen": ("kernels_exact.hpp", "kernels_common.hpp", "pose_math.hpp"),
    "point_error": ("kernels_exact.hpp", "kernels_common.hpp", "pose_math.hpp"),
    "moments": ("kernels_fit.hpp", "kernels_common.hpp", "pose_math.hpp"),
    "inlier": ("kernels_fit.hpp", "bf16x3_split.hpp", "kernels_common.hpp", "pose_math.hpp"),
    "exact_count": ("kernels_fit.hpp", "kernels_common.hpp", "pose_math.hpp"),
}


def kernel_source_sha(key: str) -> str:
    """sha256 (hex, first 16 digits) over the source files of kernel `key`, in the order listed."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[key]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]
