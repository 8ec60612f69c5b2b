"""Generates tests/golden/kiss_fft_vectors.npz from the REFERENCE's own kissfft, compiled in
place into oracle/_ref/libref_kiss.so (oracle/Makefile target `ref`).  Run in the build
container, where /root/reference exists:  python tests/golden/make_golden.py

Fixture = inputs + the reference's outputs (data only).  Inputs come from numpy's PCG64 with
fixed seeds so the file can be regenerated identically.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle  # noqa: E402

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


def main():
    ref = _oracle.load_ref()
    assert ref is not None, "reference kissfft not built (no /root/reference?)"
    out = {}
    rng = np.random.default_rng(20261002)
    for n, inv in ((256, 0), (512, 0), (1024, 1), (2048, 1), (512, 1), (1024, 0)):
        x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
        y = np.empty_like(x)
        ref.ref_kiss_cfft_f32(C.c_size_t(n), inv, x.ctypes.data_as(f32p), y.ctypes.data_as(f32p))
        out[f"cfft32_n{n}_inv{inv}_in"] = x
        out[f"cfft32_n{n}_inv{inv}_out"] = y
    for nhalf in (256, 512, 1024):
        x = rng.uniform(-1, 1, 2 * nhalf).astype(np.float32)
        y = np.empty(nhalf, np.complex64)
        ref.ref_kiss_rfft_packed_f32(C.c_size_t(nhalf), x.ctypes.data_as(f32p), y.ctypes.data_as(f32p))
        out[f"rfft32_packed_nhalf{nhalf}_in"] = x
        out[f"rfft32_packed_nhalf{nhalf}_out"] = y
    n = 512
    x = np.exp(2j * np.pi * rng.uniform(0, 1, n)).astype(np.complex128)
    y = np.empty_like(x)
    ref.ref_kiss_cfft_f64(C.c_size_t(n), 1, x.ctypes.data_as(f64p), y.ctypes.data_as(f64p))
    out["cfft64_n512_inv1_in"] = x
    out["cfft64_n512_inv1_out"] = y
    path = os.path.join(HERE, "kiss_fft_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
