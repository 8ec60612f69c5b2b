"""Helpers for the GPU tests: one shared libearhip context."""
import pytest

_ctx = None


def ctx():
    global _ctx
    from libear_amd import capi
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


def set_renderer_curves(r, curves, two_bus=True):
    for m, (t, d, f) in enumerate(curves):
        r.set_object_points(m, t, d, f if two_bus else None)


def set_oracle_curves(o, curves, two_bus=True):
    import numpy as np
    for m, (t, d, f) in enumerate(curves):
        o.set_points(m, 0, t, d)
        o.set_points(m, 1, t, f if two_bus else np.zeros_like(d))
