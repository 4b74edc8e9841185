import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def calib1():
    z = np.load(os.path.join(GOLDEN, "calib1.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def calib1_raw():
    """The four raw clouds of the reference's example/calib_1 (z <= 0.8) with the oracle's plane and
    raw error report (tests/golden/make_golden_raw.py)."""
    z = np.load(os.path.join(GOLDEN, "calib1_raw.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def synthetic_golden():
    return json.load(open(os.path.join(GOLDEN, "synthetic.json")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ccal():
    import cameracalibration_amd as cc
    cc.load_library()
    return cc


@pytest.fixture()
def gpu_opt(ccal):
    """A fresh context on GPU 0; fails (does not skip) when there is no GPU."""
    assert ccal.device_count() > 0, "a gpu-marked test was run without a GPU"
    o = ccal.Optimizer(0)
    yield o
    o.close()
