"""GPU: the layer-wise backward with the activation passes as kernels of their own (IRON_TRAIN_FUSE=0) -- the path every shape the
fused row-GEMM epilogues do not take still runs (a material net's skip layer, widths other than 217..256) -- stays at parity.  The
switch is read once per process, so the selected tests run in a child pytest (one process at a time)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_unfused_backward_still_matches():
    env = dict(os.environ, IRON_TRAIN_FUSE="0")
    sel = "sdf_get_all_backward_vs_autograd or additive_over_points or eikonal or render_network_backward or g14_training"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_train.py"), "-x", "-q", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
    print(r.stdout.strip().splitlines()[-1])
