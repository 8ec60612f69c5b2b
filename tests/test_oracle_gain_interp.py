"""CPU oracle GainInterpolator vs the reference tests' expectations
(reference tests/gain_interpolator_tests.cpp:58-257)."""
import numpy as np
import pytest

import _oracle
from refcases import chunks, expected_single, gain_interp_cases, is_approx, single_interp_expected


def test_linear_interp_single_apply_interp_closed_form():
    # :58-70  block_start=100, curve 50..250, 0.2 -> 0.8
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, 100).astype(np.float32)
    got = _oracle.gain_interp("single", [50, 250], np.float32([0.2, 0.8]).reshape(2, 1, 1), x, [100], t0=100)[0]
    want = single_interp_expected(x, 100, 50, 250, np.float64(np.float32(0.2)), np.float64(np.float32(0.8)))
    assert is_approx(got, want)
    # bit-level: same arithmetic in numpy float32
    t = np.arange(100, 200)
    p = ((t - 50).astype(np.float32) * (np.float32(1.0) / np.float32(200))).astype(np.float32)
    g = ((np.float32(1) - p) * np.float32(0.2) + p * np.float32(0.8)).astype(np.float32)
    assert np.array_equal(got, x * g)


def test_linear_interp_single_apply_constant():
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, 100).astype(np.float32)
    got = _oracle.gain_interp("single", [1000], np.float32([0.3]).reshape(1, 1, 1), x, [100])[0]
    assert np.array_equal(got, np.float32(0.3) * x)


@pytest.mark.parametrize("case", gain_interp_cases(), ids=lambda c: c[0])
def test_segmentation(case):
    name, pts, length, segs, block_sizes = case
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, length).astype(np.float32)
    want = expected_single(x, segs)
    times = [t for t, _ in pts]
    vals = np.float32([v for _, v in pts]).reshape(-1, 1, 1)
    outs = []
    for bs in block_sizes:
        got = _oracle.gain_interp("single", times, vals, x, chunks(length, bs))[0]
        assert is_approx(got, want), (name, bs)
        outs.append(got)
    for o in outs[1:]:  # chunking never changes a single bit
        assert np.array_equal(o, outs[0])


def test_vector_equals_sum_of_singles():
    # :187-219
    a, b = [0.0, 1.0], [1.0, 0.0]
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, 300).astype(np.float32)
    got = _oracle.gain_interp("vector", [100, 200], np.float32([a, b]).reshape(2, 1, 2), x, [300])
    for o in range(2):
        single = _oracle.gain_interp("single", [100, 200], np.float32([a[o], b[o]]).reshape(2, 1, 1), x, [300])[0]
        assert np.array_equal(got[o], single)


def test_matrix_equals_sum_of_singles():
    # :221-257, 3 -> 2
    a = [[0.0, 0.3], [0.5, 0.0], [0.4, 1.0]]
    b = [[0.6, 0.0], [0.0, 0.7], [1.0, 0.2]]
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (3, 300)).astype(np.float32)
    got = _oracle.gain_interp("matrix", [100, 200], np.float32([a, b]), x, [300])
    want = np.zeros((2, 300), np.float32)
    for i in range(3):
        for o in range(2):
            want[o] += _oracle.gain_interp("single", [100, 200],
                                           np.float32([a[i][o], b[i][o]]).reshape(2, 1, 1), x[i], [300])[0]
    assert np.array_equal(got, want)  # same accumulation order -> identical


def test_unsorted_points_follow_reference_search():
    # The reference's "interpolation points are not sorted" throw (gain_interpolator.hpp:123-124)
    # is unreachable: its cached linear search can never reverse direction.  Unsorted times are
    # therefore processed with whatever segment the search lands on; the restatement does the same.
    x = np.ones(400, np.float32)
    got = _oracle.gain_interp("single", [100, 300, 200], np.float32([0.0, 1.0, 0.5]).reshape(3, 1, 1), x, [400])[0]
    assert np.all(got[:100] == 0.0) and np.all(got[300:] == 0.5)
    assert np.allclose(got[100:300], np.arange(200) / 200.0, atol=1e-6)


def test_empty_points_are_an_error():
    x = np.zeros(10, np.float32)
    with pytest.raises(_oracle.OracleError):
        _oracle.gain_interp("single", [], np.zeros((0, 1, 1), np.float32), x, [10])


def test_config1_one_object_to_0_5_0_vector_closed_form():
    """BASELINE config 1 as written: 1 object -> 0+5+0 (6 channels incl. LFE1), block 512, a full-length ramp per
    block through GainInterpolator<LinearInterpVector> (gain_interpolator.hpp:53-87,214-241), against the closed
    form of the reference's own test (tests/gain_interpolator_tests.cpp:58-70: p = (float)(t - start) * (1.0f /
    (float)(end - start)), g = (1 - p) s + p e, out = in * g), bit for bit, for several chunkings of the stream."""
    from layouts import LAYOUTS
    n, block, nblocks = len(LAYOUTS["0+5+0"]), 512, 4
    assert n == 6
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, block * nblocks).astype(np.float32)
    times = [block * k for k in range(nblocks + 1)]
    vals = rng.uniform(0, 1, (nblocks + 1, 1, n)).astype(np.float32)
    vals[:, 0, 3] = 0.0  # LFE1: the gain calculators leave it at exactly zero (gain_calculator_objects.cpp:51-52)
    want = np.zeros((n, block * nblocks), np.float32)
    for k in range(nblocks):
        t = np.arange(block * k, block * (k + 1))
        p = ((t - block * k).astype(np.float32) * (np.float32(1.0) / np.float32(block))).astype(np.float32)
        for c in range(n):
            g = ((np.float32(1) - p) * vals[k, 0, c] + p * vals[k + 1, 0, c]).astype(np.float32)
            want[c, t] = x[t] * g
    for sizes in ([block] * nblocks, [block * nblocks], chunks(block * nblocks, 100)):
        got = _oracle.gain_interp("vector", times, vals, x, sizes)
        assert np.array_equal(got, want)
    assert not want[3].any()
