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


def test_context_options_roundtrip_and_unknown_key():
    """earhip_ctx_set_option / _get_option (include/earhip.h): keys with or without the EARHIP_ prefix, any case; NULL
    returns an option to its default; an unknown key is an invalid argument"""
    import pytest as _pytest
    from libear_amd import capi
    c = capi.Context(0)
    try:
        assert c.get_option("H2_TILE") in (None, int(os.environ["EARHIP_H2_TILE"]) if "EARHIP_H2_TILE" in os.environ else None)
        c.set_option("h2_tile", 256)
        assert c.get_option("EARHIP_H2_TILE") == 256
        c.set_option("EARHIP_H2_TILE", None)
        assert c.get_option("H2_TILE") is None
        with _pytest.raises(capi.InvalidArgument):
            c.set_option("NO_SUCH_KNOB", 1)
    finally:
        c.close()


def test_the_environment_is_only_read_when_a_context_is_created(monkeypatch):
    """a knob exported AFTER the context exists does not reach it; the same knob as an option does"""
    import numpy as np
    import scenes
    from layouts import LAYOUTS
    from libear_amd import capi
    monkeypatch.delenv("EARHIP_MFMA", raising=False)
    c = capi.Context(0)
    try:
        n, m, block, nblocks = len(LAYOUTS["0+5+0"]), 64, 512, 2
        dec = capi.design_decorrelators(LAYOUTS["0+5+0"])
        x = scenes.audio(m, block * nblocks)
        kinds = []
        for how in ("env", "option"):
            if how == "env":
                monkeypatch.setenv("EARHIP_MFMA", "1")
            else:
                c.set_option("MFMA", 1)
            r = capi.Renderer(c, m, n, block, dec, 255, max_blocks=nblocks)
            for i, (t, d, f) in enumerate(scenes.dense_curves(m, n, block, nblocks)):
                r.set_object_points(i, t, d, f)
            r.process(x)
            kinds.append(r.gain_kernel())
            r.close()
        assert kinds == [3, 2], kinds  # (block-aligned curves: the option's exact-f32 kernel on the tile grid)
    finally:
        c.close()
