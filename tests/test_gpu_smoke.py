"""The driver's entry point inside the tested surface: __graft_entry__.smoke() is what the round-end GPU record runs
before the bench (round 4 ended with it red on a stale kernel expectation that no test looked at)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_smoke_entry_point(capsys):
    import __graft_entry__ as g
    g.smoke()
    out = capsys.readouterr().out
    assert out.count("rel RMS vs CPU oracle") == 2, out


@pytest.mark.parametrize("force", ["1", "4", "5", "6"])
def test_smoke_entry_point_with_a_forced_kernel(force, monkeypatch):
    """EARHIP_MFMA forces a gain kernel: smoke() then only checks the numbers"""
    import __graft_entry__ as g
    monkeypatch.setenv("EARHIP_MFMA", force)
    g.smoke()
