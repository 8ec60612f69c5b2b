"""HIP FFT plugin (include/earhip.h group B; libear include/ear/fft.hpp:27-50) vs the CPU oracle's
kissfft restatement.  Float FFTs of different factorizations differ in rounding; tolerance:
relative L2 <= 5e-7 against the oracle and against numpy's float64 transform."""
import numpy as np
import pytest

import _oracle
from _hip import ctx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_fft", [64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_forward_and_reverse(n_fft):
    from libear_amd import capi
    plan = capi.FFTPlan(ctx(), n_fft)
    rng = np.random.default_rng(n_fft)
    x = rng.uniform(-1, 1, n_fft).astype(np.float32)
    X = plan.forward(x)
    want = _oracle.rfft(x)
    ref = np.fft.rfft(x.astype(np.float64))
    assert np.linalg.norm(X - want) / np.linalg.norm(want) < 5e-7
    assert np.linalg.norm(X - ref) / np.linalg.norm(ref) < 5e-7
    y = plan.reverse(want)
    want_y = _oracle.irfft_unnorm(want, n_fft)
    assert np.linalg.norm(y - want_y) / np.linalg.norm(want_y) < 5e-7
    assert np.max(np.abs(y / n_fft - x)) < 3e-6  # un-normalised both ways
    plan.close()


@pytest.mark.parametrize("n_fft", [4, 6, 30, 90, 96, 882, 960, 1000, 1920, 2 * 1155, 3840, 6000, 2 * 97 * 31, 2 * 101, 2 * 1031, 2 * 4093, 4 * 499])
def test_sizes_that_are_not_powers_of_two(n_fft):
    """kissfft takes any even real length (src/fft_kiss.cpp:104-107): mixed-radix plans (4, 2, 3, 5 and the
    generic butterfly for every other prime) against the oracle's restatement of the same factorisations"""
    from libear_amd import capi
    plan = capi.FFTPlan(ctx(), n_fft)
    rng = np.random.default_rng(n_fft)
    x = rng.uniform(-1, 1, n_fft).astype(np.float32)
    X = plan.forward(x)
    want = _oracle.rfft(x)
    ref = np.fft.rfft(x.astype(np.float64))
    # (a prime factor p is a float32 sum of p terms per output, in kissfft as here: the error grows like sqrt(p))
    pmax = max(p for p in range(2, n_fft + 1) if n_fft % p == 0 and all(p % q for q in range(2, int(p ** 0.5) + 1)))
    tol = 6e-7 if pmax <= 97 else 6e-7 * (pmax / 97.0) ** 0.5
    assert np.linalg.norm(X - want) / np.linalg.norm(want) < tol
    assert np.linalg.norm(X - ref) / np.linalg.norm(ref) < tol
    y = plan.reverse(want)
    want_y = _oracle.irfft_unnorm(want, n_fft)
    assert np.linalg.norm(y - want_y) / np.linalg.norm(want_y) < tol
    assert np.max(np.abs(y / n_fft - x)) < 3e-6 * tol / 6e-7
    plan.close()


def test_golden_vectors():
    """the committed kissfft vectors (tests/golden) through the device transform"""
    import os
    from libear_amd import capi
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "kiss_fft_vectors.npz"))
    for nhalf in (256, 512, 1024):
        x = gold[f"rfft32_packed_nhalf{nhalf}_in"]
        packed = gold[f"rfft32_packed_nhalf{nhalf}_out"]
        want = np.empty(nhalf + 1, np.complex64)
        want[:nhalf] = packed
        want[nhalf] = packed[0].imag
        want[0] = packed[0].real
        plan = capi.FFTPlan(ctx(), 2 * nhalf)
        got = plan.forward(x)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 5e-7
        plan.close()


def test_bad_size_is_invalid_argument():
    from libear_amd import capi
    with pytest.raises(capi.InvalidArgument):
        capi.FFTPlan(ctx(), 1001)  # odd (kiss_fftr needs an even length)
    with pytest.raises(capi.InvalidArgument):
        capi.FFTPlan(ctx(), 16384)
