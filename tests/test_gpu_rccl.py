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
