"""Pins the oracle's FFT restatement against the reference's own kissfft:
(1) live, bit-for-bit, against oracle/_ref/libref_kiss.so when it exists (build container, and
    on the GPU box as a prebuilt file), (2) against the committed golden vectors generated from
    it (tests/golden/make_golden.py), (3) against numpy's double-precision FFT for accuracy."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "kiss_fft_vectors.npz"))
f32p = C.POINTER(C.c_float)


def same_bits(a, b):
    # value equality: +0 == -0 allowed, everything else bit-identical
    return np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("n,inv", [(256, 0), (512, 0), (1024, 1), (2048, 1), (512, 1), (1024, 0)])
def test_cfft_matches_golden_bit_exact(n, inv):
    x = GOLD[f"cfft32_n{n}_inv{inv}_in"]
    want = GOLD[f"cfft32_n{n}_inv{inv}_out"]
    got = _oracle.cfft(x, inverse=bool(inv))
    assert same_bits(got, want)
    ref = np.fft.ifft(x.astype(np.complex128)) * n if inv else np.fft.fft(x.astype(np.complex128))
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 5e-7


@pytest.mark.parametrize("nhalf", [256, 512, 1024])
def test_rfft_matches_golden_bit_exact(nhalf):
    x = GOLD[f"rfft32_packed_nhalf{nhalf}_in"]
    packed = GOLD[f"rfft32_packed_nhalf{nhalf}_out"]
    got = _oracle.rfft(x)  # libear packing: n/2+1 bins (src/fft_kiss.cpp:61-71)
    want = np.empty(nhalf + 1, np.complex64)
    want[:nhalf] = packed
    want[nhalf] = packed[0].imag
    want[0] = packed[0].real
    assert same_bits(got, want)
    ref = np.fft.rfft(x.astype(np.float64))
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 5e-7


def test_cfft64_matches_golden_bit_exact():
    x = GOLD["cfft64_n512_inv1_in"]
    got = _oracle.cfft(x, inverse=True, double=True)
    assert same_bits(got, GOLD["cfft64_n512_inv1_out"])


@pytest.mark.parametrize("n,inv", [(64, 0), (128, 1), (480, 0), (960, 1), (1024, 0), (4096, 1),
                                   # radices beyond 2..5 (kissfft's generic butterfly): 7, 7*7, 11, 13*3, 97
                                   (14, 0), (441, 1), (2 * 11 * 5, 0), (39 * 4, 1), (97 * 8, 0), (1155, 1)])
def test_cfft_live_against_compiled_reference(n, inv):
    ref = _oracle.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(n + inv)
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    want = np.empty_like(x)
    ref.ref_kiss_cfft_f32(C.c_size_t(n), inv, x.ctypes.data_as(f32p), want.ctypes.data_as(f32p))
    assert same_bits(_oracle.cfft(x, inverse=bool(inv)), want)


def test_irfft_roundtrip_and_hermitian_extension():
    # reverse = full complex inverse of the Hermitian extension (src/fft_kiss.cpp:73-88)
    rng = np.random.default_rng(5)
    for n_fft in (1024, 2048):
        x = rng.uniform(-1, 1, n_fft).astype(np.float32)
        X = _oracle.rfft(x)
        y = _oracle.irfft_unnorm(X, n_fft)
        assert np.max(np.abs(y / n_fft - x)) < 2e-6
        full = np.concatenate([X, np.conj(X[1:-1][::-1])]).astype(np.complex64)
        assert same_bits(y, _oracle.cfft(full, inverse=True).real)
