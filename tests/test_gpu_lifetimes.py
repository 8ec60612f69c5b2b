"""Handles created, used and destroyed over and over leave neither device memory nor host memory behind
(tools/leak_check.py in a child process: contexts, renderers, panners with extent, gain interpolators, pinned
arrays)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_growth_over_create_use_destroy_cycles():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "leak_check.py"), "60"], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:]
    assert "growth after warm-up" in res.stdout
