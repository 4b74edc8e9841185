"""The entry point that allocates nothing (ccal_fit_plane_lsq_static: static device scratch, kernel
launches and one stream wait only) against ccal_fit_plane_lsq, through the C ABI: bit-equal planes on two
calls in a row, the second on a smaller set (the scratch keeps the first call's rows), and the capacity
limit is reported.  Collected last (file name) so that a failure here hides no other test."""
import numpy as np
import pytest

from cameracalibration_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


def test_static_plane_fit_equals_plane_fit(gpu_opt, ccal):
    big = make_scene(9, 150_000, seed=41, ragged=True)
    small = make_scene(3, 5_000, seed=42, ragged=True)
    for s in (big, small):                                   # second call: fewer clouds, fewer blocks
        gpu_opt.upload_clouds(ccal.CCAL_RAW, s.clouds, s.base_T_flange)
        # the entry point comes first, on a context that has never fitted a plane: whatever it needs
        # must already be there, and the GPU's free memory must not move across its calls
        free0, _ = gpu_opt.device_memory()
        got_first = [gpu_opt.fit_plane_static(pose) for pose in (s.pose_init, s.pose_true)]
        assert gpu_opt.device_memory()[0] == free0, "ccal_fit_plane_lsq_static changed the free device memory"
        for pose, first in zip((s.pose_init, s.pose_true), got_first):
            want = gpu_opt.fit_plane(pose)
            got = gpu_opt.fit_plane_static(pose)
            assert got.tobytes() == want.tobytes() == first.tobytes()
            assert abs(np.linalg.norm(got[:3]) - 1.0) < 1e-6
    np.testing.assert_allclose(got, small.plane, atol=2e-3)
    # the other set of the same context, and a second context on the same GPU (same static array)
    gpu_opt.upload_clouds(ccal.CCAL_DOWNSAMPLED, big.clouds[:2], big.base_T_flange[:2])
    assert gpu_opt.fit_plane_static(big.pose_init, which=0).tobytes() == gpu_opt.fit_plane(big.pose_init, which=0).tobytes()
    with ccal.Optimizer(0) as other:
        other.upload_clouds(ccal.CCAL_RAW, small.clouds, small.base_T_flange)
        assert other.fit_plane_static(small.pose_init).tobytes() == gpu_opt.fit_plane(small.pose_init).tobytes()


def test_static_plane_fit_rejects_sets_beyond_its_scratch(gpu_opt, ccal):
    n = 1025                                                  # one cloud more than the scratch holds
    clouds = [np.float32([[0.1 * (j % 7), 0.2, 0.3 + 0.001 * j], [0.0, 0.1 * (j % 5), 0.31], [0.05, 0.0, 0.29]]) for j in range(n)]
    B = np.repeat(np.eye(4).reshape(1, 16), n, axis=0)
    gpu_opt.upload_clouds(ccal.CCAL_RAW, clouds, B)
    with pytest.raises(ccal.CcalError) as e:
        gpu_opt.fit_plane_static(np.zeros(6))
    assert e.value.status == -4 and "static scratch" in str(e.value)
    assert np.isfinite(gpu_opt.fit_plane(np.zeros(6))).all()   # the allocating entry point takes it
