"""GPU: the exact-fp32 MFMA core (csrc/mlp_core.h; IRON_MLP_CORE=f32) stays at parity with the reference.  The core is
chosen once per process, so the check runs in a child process (one at a time)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_f32_core_parity():
    env = dict(os.environ, IRON_MLP_CORE="f32")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_f32core_check.py")], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "F32CORE_CHECK OK" in r.stdout
    print(r.stdout.strip().splitlines()[-1])
