"""CPU tests of the C-ABI library: it loads, exports every declared symbol,
and its host-side logic (grid decoding, pose matrix, merge) matches the oracle.
No compute entry point is exercised here (that needs a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ccal.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ccal_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(ccal):
    lib = ccal.load_library()
    names = declared_symbols()
    assert "ccal_grid_search" in names and "ccal_eval" in names and len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ccal.h but not exported"
    # and the binding covers all of them
    from cameracalibration_amd.binding import SIGNATURES
    assert sorted(SIGNATURES) == names


def test_grid_pose_and_size_match_oracle(ccal, oracle):
    g = ccal.Grid.default()
    og = oracle.grid_default()
    assert g.size == 11 ** 6 == oracle.lib().orc_grid_size(og)
    base = np.array([0.0204504, 0.111258, 0.0505291, -3.15376, -0.0280023, 1.58043])
    rng = np.random.default_rng(0)
    for idx in [0, 1, 10, 11, 121, 11 ** 6 - 1] + list(rng.integers(0, 11 ** 6, 50)):
        np.testing.assert_array_equal(ccal.grid_pose(g, base, int(idx)),
                                      oracle.grid_pose(og, base, int(idx)))
    g2 = ccal.Grid.make([-20, -20, -24, -5, -5, -6], [2, 2, 2, 0.5, 0.5, 0.5], [20, 20, 25, 20, 20, 25])
    og2 = oracle.make_grid([-20, -20, -24, -5, -5, -6], [2, 2, 2, 0.5, 0.5, 0.5], [20, 20, 25, 20, 20, 25])
    assert g2.size == 10 ** 8
    for idx in rng.integers(0, 10 ** 8, 50):
        np.testing.assert_array_equal(ccal.grid_pose(g2, base, int(idx)),
                                      oracle.grid_pose(og2, base, int(idx)))


def test_pose_matrix_bit_equal_to_oracle(ccal, oracle):
    """Many poses: glibc's sincos() (what gcc makes of Eigen's sin and cos of one angle) differs from
    sin()/cos() in the last place for about one angle in a thousand, so a library that took the
    other route would agree with the oracle on a few hundred poses and differ on the next."""
    rng = np.random.default_rng(2)
    for _ in range(40000):
        pose = rng.normal(0, 1, 6) * np.array([0.5, 0.5, 0.5, 3.0, 1.5, 3.0])
        assert np.array_equal(ccal.pose_to_matrix(pose), oracle.pose_to_matrix(pose)), pose


def test_argmin_merge(ccal):
    from cameracalibration_amd.binding import argmin_merge
    assert argmin_merge([3.0, 1.0, 1.0, 2.0], [5, 9, 7, 1]) == 2      # tie -> lowest index
    assert argmin_merge([3.0, 1.0], [-1, -1]) == -1                   # empty shards
    assert argmin_merge([1e9, 0.5], [-1, 4]) == 1


def test_status_strings_and_no_device_behaviour(ccal):
    lib = ccal.load_library()
    assert lib.ccal_status_string(0) == b"ok"
    assert lib.ccal_status_string(-2) == b"device error"
    if ccal.device_count() == 0:
        # the product must fail loudly without a GPU, never fall back
        with pytest.raises(ccal.CcalError):
            ccal.Optimizer(0)


def test_gd_default_options(ccal):
    o = ccal.GdOptions.default()
    assert (o.max_iterations, o.min_gradient_len, o.min_step_len, o.momentum) == (10000, 1e-6, 1e-6, 0.4)
    assert o.fd_eps == np.sqrt(np.finfo(np.float64).eps)
    assert (o.wolfe_decrease, o.wolfe_c1, o.wolfe_c2) == (0.8, 1e-4, 0.9)


def test_header_is_plain_c_and_links(tmp_path, ccal):
    """include/ccal.h compiles as C99 and as C++11, and a C program links against libccal.so
    by the header's declarations alone (no GPU call is made)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_ccal.c"
    src.write_text(
        '#include "ccal.h"\n'
        "int main(void) {\n"
        "    ccal_grid g; ccal_gd_opts o; double pose[6] = {0}, T[16];\n"
        "    ccal_grid_default(&g); ccal_gd_default_opts(&o); ccal_pose_to_matrix(pose, T);\n"
        "    return !(ccal_grid_size(&g) == 1771561 && o.max_iterations == 10000 && T[0] == 1.0 &&\n"
        "             ccal_status_string(CCAL_OK) != 0);\n"
        "}\n")
    inc = os.path.join(root, "include")
    libdir = os.path.join(root, "cameracalibration_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])
    exe = tmp_path / "use_ccal"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lccal",
                           "-Wl,-rpath," + libdir])
    assert subprocess.run([str(exe)]).returncode == 0
