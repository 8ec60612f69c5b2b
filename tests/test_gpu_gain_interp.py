"""HIP interpolation policies and whole-curve GainInterpolator vs the CPU oracle
(reference include/ear/dsp/gain_interpolator.hpp; cases of reference
tests/gain_interpolator_tests.cpp:58-257).

Bars: 1->1 and 1->N (no accumulation) bit-exact; M->N bit-exact in strict mode, relative RMS
<= 1e-6 in the default (FMA / tree accumulation) mode."""
import numpy as np
import pytest

import _oracle
from _hip import ctx
from refcases import chunks, expected_single, gain_interp_cases, is_approx
from scenes import rel_rms

pytestmark = pytest.mark.gpu


def test_single_apply_interp_and_constant_bit_exact():
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (1, 100)).astype(np.float32)
    out = np.zeros((1, 100), np.float32)
    ctx().apply_interp(x, out, 0, 100, 100, 50, 250, [0.2], [0.8])  # reference test :58-70
    want = _oracle.gain_interp("single", [50, 250], np.float32([0.2, 0.8]).reshape(2, 1, 1), x, [100], t0=100)
    assert np.array_equal(out, want)
    out2 = np.zeros((1, 100), np.float32)
    ctx().apply_constant(x, out2, 0, 100, [0.3])  # :72-81
    assert np.array_equal(out2[0], np.float32(0.3) * x[0])


def test_policy_writes_only_the_range():
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (1, 64)).astype(np.float32)
    out = np.full((3, 64), 7.0, np.float32)
    ctx().apply_constant(x, out, 10, 33, [1.0, 2.0, 3.0])
    assert np.all(out[:, :10] == 7.0) and np.all(out[:, 33:] == 7.0)
    assert np.array_equal(out[:, 10:33], np.float32([[1.0], [2.0], [3.0]]) * x[:, 10:33])


def test_apply_interp_extrapolates_like_a_direct_call():
    # apply_interp called with samples outside [start, end) and equal points still ramps
    x = np.ones((1, 40), np.float32)
    out = np.zeros((2, 40), np.float32)
    ctx().apply_interp(x, out, 0, 40, 0, 10, 30, [0.0, 0.5], [1.0, 0.5])
    t = np.arange(40)
    p = ((t - 10).astype(np.float32) * (np.float32(1.0) / np.float32(20))).astype(np.float32)
    for o, (s, e) in enumerate(((0.0, 1.0), (0.5, 0.5))):
        g = ((np.float32(1) - p) * np.float32(s) + p * np.float32(e)).astype(np.float32)
        assert np.array_equal(out[o], g)


@pytest.mark.parametrize("case", gain_interp_cases(), ids=lambda c: c[0])
def test_segmentation_cases(case):
    from libear_amd import capi
    name, pts, length, segs, block_sizes = case
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, length).astype(np.float32)
    want = expected_single(x, segs)
    times = [t for t, _ in pts]
    vals = np.float32([v for _, v in pts]).reshape(-1, 1, 1)
    gi = capi.GainInterp(ctx(), 1, 1)
    gi.set_points(times, vals)
    for bs in block_sizes:
        got = np.zeros(length, np.float32)
        ofs = 0
        for n in chunks(length, bs):
            got[ofs:ofs + n] = gi.process(ofs, x[ofs:ofs + n])[0]
            ofs += n
        assert is_approx(got, want), (name, bs)
        assert np.array_equal(got, _oracle.gain_interp("single", times, vals, x, [length])[0]), (name, bs)
    gi.close()


def test_vector_bit_exact():
    from libear_amd import capi
    a, b = [0.0, 1.0], [1.0, 0.0]
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, 300).astype(np.float32)
    vals = np.float32([a, b]).reshape(2, 1, 2)
    gi = capi.GainInterp(ctx(), 1, 2)
    gi.set_points([100, 200], vals)
    got = gi.process(0, x)
    assert np.array_equal(got, _oracle.gain_interp("vector", [100, 200], vals, x, [300]))
    gi.close()


@pytest.mark.parametrize("strict", [True, False])
def test_matrix_3_to_2(strict):
    from libear_amd import capi
    a = [[0.0, 0.3], [0.5, 0.0], [0.4, 1.0]]
    b = [[0.6, 0.0], [0.0, 0.7], [1.0, 0.2]]
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (3, 300)).astype(np.float32)
    vals = np.float32([a, b])
    want = _oracle.gain_interp("matrix", [100, 200], vals, x, [300])
    ctx().set_strict(strict)
    try:
        gi = capi.GainInterp(ctx(), 3, 2)
        gi.set_points([100, 200], vals)
        got = gi.process(0, x)
        gi.close()
    finally:
        ctx().set_strict(False)
    if strict:
        assert np.array_equal(got, want)
    else:
        assert rel_rms(got, want) <= 1e-6


@pytest.mark.parametrize("m,n,length", [(64, 10, 2048), (32, 24, 1024), (256, 24, 1536), (7, 5, 333)])
@pytest.mark.parametrize("strict", [True, False])
def test_matrix_dense_ramps(m, n, length, strict):
    """LinearInterpMatrix at C2/C3-like shapes, a point every 100 samples (the reference's
    matrix_benchmark pattern, tests/gain_interpolator_tests.cpp:259-296)."""
    from libear_amd import capi
    rng = np.random.default_rng(m * 1000 + n)
    times = list(range(100, length, 100))
    vals = rng.uniform(0, 1, (len(times), m, n)).astype(np.float32)
    vals[2] = vals[1]  # one constant segment
    x = rng.uniform(-1, 1, (m, length)).astype(np.float32)
    want = _oracle.gain_interp("matrix", times, vals, x, [length])
    ctx().set_strict(strict)
    try:
        gi = capi.GainInterp(ctx(), m, n)
        gi.set_points(times, vals)
        got = np.concatenate([gi.process(0, x[:, :700]), gi.process(700, x[:, 700:])], axis=1)
        gi.close()
    finally:
        ctx().set_strict(False)
    if strict:
        assert np.array_equal(got, want)
    else:
        assert rel_rms(got, want) <= 1e-6


@pytest.mark.parametrize("seed", list(range(12)))
def test_matrix_random_curves_and_call_partitions(seed):
    """Whole-curve GainInterpolator<LinearInterpMatrix>: random point times (duplicates, negative, beyond
    the end), random start time and call sizes; strict mode bit-exact, default mode <= 1e-6."""
    from libear_amd import capi
    rng = np.random.default_rng(50 + seed)
    m = int(rng.choice([1, 2, 9, 31, 32, 40, 65]))
    n = int(rng.choice([1, 3, 10, 24]))
    length = int(rng.integers(50, 2500))
    t0 = int(rng.integers(-500, 500))
    npts = int(rng.integers(1, 12))
    times = np.sort(rng.integers(t0 - 200, t0 + length + 200, npts))
    if npts > 3 and seed % 2:
        times[2] = times[1]  # a step
    vals = rng.uniform(0, 1, (npts, m, n)).astype(np.float32)
    if npts > 2 and seed % 3 == 0:
        vals[-1] = vals[-2]  # a constant segment
    x = rng.uniform(-1, 1, (m, length)).astype(np.float32)
    calls, left = [], length
    while left > 0:
        c = int(rng.integers(1, left + 1))
        calls.append(c)
        left -= c
    want = _oracle.gain_interp("matrix", times, vals, x, calls, t0=t0)
    for strict in (True, False):
        ctx().set_strict(strict)
        try:
            gi = capi.GainInterp(ctx(), m, n)
            gi.set_points(times, vals)
            parts, at = [], 0
            for c in calls:
                parts.append(gi.process(t0 + at, x[:, at:at + c]))
                at += c
            got = np.concatenate(parts, axis=1)
            gi.close()
        finally:
            ctx().set_strict(False)
        if strict:
            assert np.array_equal(got, want), (m, n, length, t0, calls)
        else:
            assert rel_rms(got, want) <= 1e-6, (m, n, length, t0, calls)


def test_errors():
    from libear_amd import capi
    gi = capi.GainInterp(ctx(), 1, 1)
    with pytest.raises(capi.InvalidArgument):
        gi.set_points([], np.zeros((0, 1, 1), np.float32))  # reference: UB; defined as an error
    with pytest.raises(capi.InvalidArgument) as e:
        gi.set_points([200, 100], np.zeros((2, 1, 1), np.float32))
    assert "not sorted" in str(e.value)
    gi.close()
