"""GPU: the experimental two-waves-per-SIMD "w16" core (csrc/w16.hip) gives the reference's SDF values.  It is not
the default core (it measured level with h2); the test keeps the prototype honest.  The core is chosen once per process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_w16_core_sdf_values():
    env = dict(os.environ, IRON_MLP_CORE="w16")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_w16_check.py")], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "W16_CHECK OK" in r.stdout
    print(r.stdout.strip().splitlines()[-1])
