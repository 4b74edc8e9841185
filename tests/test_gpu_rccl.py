"""The exchanges of the multi-GPU path over RCCL (backend "nccl") on device tensors.  A one-GPU box
can hold a world of one rank only; the world-size-2 logic is covered on CPU over gloo
(test_distributed_cpu.py).  Runs in a child process so the process group does not outlive it."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_exchanges_single_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "probe_rccl.py")], env=env,
                       capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.strip().endswith("ok")


@pytest.mark.gpu
def test_library_argmin_allgather_single_rank(gpu_opt, ccal):
    """ccal_argmin_allgather through the C ABI in a world of one rank: the library opens librccl itself,
    the 16-byte (cost, index) record goes through ncclAllGather on the context's stream and comes back as
    the winner; an empty shard (index < 0) yields the reference's initial 1e9 / no index (:531)."""
    uid = ccal.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    gpu_opt.comm_create(1, 0, uid)
    try:
        assert gpu_opt.argmin_allgather(0.00123456789, 987654321012) == (0.00123456789, 987654321012, 0)
        assert gpu_opt.argmin_allgather(float.fromhex("0x1.23456789abcdep-11"), 0) == (float.fromhex("0x1.23456789abcdep-11"), 0, 0)
        assert gpu_opt.argmin_allgather(5.0, -1) == (1e9, -1, -1)
        # after a search: the shard's own result is the job's result
        from cameracalibration_amd.synthetic import make_scene
        s = make_scene(3, 400, seed=2)
        gpu_opt.upload_clouds(0, s.clouds, s.base_T_flange)
        gpu_opt.set_plane(s.plane)
        idx, cost, _, _ = gpu_opt.grid_search(s.pose_init, ccal.Grid.make([-20] * 3 + [-5] * 3, [20] * 3 + [5] * 3, [3] * 6))
        assert gpu_opt.argmin_allgather(cost, idx) == (cost, idx, 0)
    finally:
        gpu_opt.comm_destroy()
    with pytest.raises(ccal.CcalError):
        gpu_opt.comm_create(2, 5, uid)          # rank outside the world
