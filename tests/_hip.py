"""Helpers for the GPU tests: one shared libearhip context."""
import pytest

_ctx = None


def ctx():
    global _ctx
    from libear_amd import capi
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


def with_options(env, fn):
    """run fn with the tuning knobs `env` ({"EARHIP_H2_TILE": "512", ...}; a value of None: the library's default) in force:
    as options of the shared test context (earhip_ctx_set_option: a context reads the environment once, when it is
    created) AND in the environment, for contexts that fn creates itself"""
    import os
    c = ctx()
    keep_env = {k: os.environ.get(k) for k in env}
    keep_opt = {k: c.get_option(k) for k in env}
    try:
        for k, v in env.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = str(v)
            c.set_option(k, v)
        return fn()
    finally:
        for k, v in keep_env.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        for k, v in keep_opt.items():
            c.set_option(k, v)


def set_renderer_curves(r, curves, two_bus=True):
    for m, (t, d, f) in enumerate(curves):
        r.set_object_points(m, t, d, f if two_bus else None)


def set_oracle_curves(o, curves, two_bus=True):
    import numpy as np
    for m, (t, d, f) in enumerate(curves):
        o.set_points(m, 0, t, d)
        o.set_points(m, 1, t, f if two_bus else np.zeros_like(d))
