"""A bounded, fixed-seed slice of the randomised GPU-vs-oracle stress harness (tests/stress_gpu.py):
random scenes over decades of scale with special values, random poses, planes, grids and optimiser
options, checked to the bars of the parity tests (exact path and descents bit-equal to the oracle,
searches with the oracle's winner in both modes, ingest kernels).  One pass, the same rounds every run."""
import pytest

pytestmark = pytest.mark.gpu

ROUNDS = 1500


def test_stress_slice_seed_20251112(capsys):
    from tests import stress_gpu
    stats = stress_gpu.run_rounds(ROUNDS, seed=20251112)
    out = capsys.readouterr().out            # every mismatch prints a line that starts with MISMATCH
    assert stats["bad"] == 0, out[-4000:]
    assert "MISMATCH" not in out
    assert stats["exact"] + stats["search"] + stats["gd"] + stats["ingest"] == ROUNDS
    assert min(stats["exact"], stats["search"], stats["gd"], stats["ingest"]) > ROUNDS // 10
