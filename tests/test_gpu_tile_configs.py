"""Every GEMM tile configuration must give the same answers: the tile size is picked from the problem size
(a cost model in gemm.hip), so the parity cases -- which are small -- would otherwise never run the 256x256
configuration that the full-size batches use.  LSEG_GEMM_TILE is read once per process, hence subprocesses."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tile", ["1", "2", "6"])        # 64x64, 128x128, 256x256 (where a specialised epilogue exists)
def test_forced_tile_config_passes_the_parity_suite(tile):
    env = dict(os.environ, LSEG_GEMM_TILE=tile)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "--timeout", "300",
           os.path.join(ROOT, "tests", "test_gpu_ops.py"), os.path.join(ROOT, "tests", "test_gpu_forward.py"),
           "-k", "gemm or conv or tiny_forward_matches_oracle or vitb32 or zero_shot or head"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
