"""libear_amd — MI355X-native ADM render DSP path (drop-in for libear's ear::dsp hot path).

The product is the C-ABI shared library ``libear_amd/lib/libearhip.so`` (HIP kernels for
gfx950 + the entry points declared in ``include/earhip.h``) and the C++14 classes in
``libear_amd/host/ear/dsp`` that mirror libear's interfaces over it.  This Python package is a
thin ctypes loader used by the tests, ``bench.py`` and ``__graft_entry__.py``; there is no CPU
fallback: importing :mod:`libear_amd.capi` fails loudly when the library has not been built.
"""
from .build import build, lib_path  # noqa: F401

__all__ = ["build", "lib_path"]
