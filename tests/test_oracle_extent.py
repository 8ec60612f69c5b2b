"""Pins the oracle's polar extent panner (oracle/extent_oracle.hpp) with the reference's own tests, restated:
tests/extent_tests.cpp — test_basis :16-55, test_weight_func :57-114, test_pv :116-138, same_as_reference
:140-169 (the library's float core against the double implementation the tests keep beside it, 1e-5) — and
the extent-free identities of tests/gain_calculator_objects_tests.cpp."""
import numpy as np
import pytest

import _oracle
from _oracle import cart
from layouts import LAYOUTS, without_lfe


def is_approx(a, b, prec):  # Eigen's isApprox: |a - b| <= prec * min(|a|, |b|)
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) <= prec * min(np.linalg.norm(a), np.linalg.norm(b))


def interp(x, xp, yp):
    return float(np.interp(x, xp, yp))


@pytest.fixture(scope="module")
def ext():
    return _oracle.PolarExtent("9+10+3")


def test_basis():
    eps = 1e-6
    cases = [
        ((0.0, 0.0), np.eye(3)),
        ((90.0, 0.0), [[0, 1, 0], [-1, 0, 0], [0, 0, 1]]),
        ((-90.0, 0.0), [[0, -1, 0], [1, 0, 0], [0, 0, 1]]),
        ((180.0, 0.0), [[-1, 0, 0], [0, -1, 0], [0, 0, 1]]),
        ((0.0, 90.0), [[1, 0, 0], [0, 0, 1], [0, -1, 0]]),
        ((0.0, -90.0), [[1, 0, 0], [0, 0, -1], [0, 1, 0]]),
        # slightly off the pole: as if pointing forwards
        ((90.0, 90.0 - 1e-6), [[1, 0, 0], [0, 0, 1], [0, -1, 0]]),
        ((90.0, -90.0 + 1e-6), [[1, 0, 0], [0, 0, -1], [0, 1, 0]]),
    ]
    for (az, el), want in cases:
        got = _oracle.extent_calc_basis(cart(az, el))
        assert is_approx(got, np.array(want, np.float64), eps), (az, el, got)


@pytest.mark.parametrize("which", [0, 1])
def test_weight_func(ext, which):
    """the weight falls from 1 to 0 over 10 degrees outside a width x height region (both forms: the
    reference only tests its test-side form and ties the library to it through same_as_reference)"""
    fade, height = 10.0, 10.0
    tol = 1e-9 if which == 1 else 2e-5  # (float core)
    centre = cart(0.0, 0.0)
    for width, azimuth in ((20.0, 0.0), (360.0, 0.0), (360.0, 180.0)):
        for elevation in np.linspace(-90.0, 90.0, 50):
            want = interp(elevation, [-(height / 2 + fade), -height / 2, height / 2, height / 2 + fade], [0, 1, 1, 0])
            assert ext.weight(centre, width, height, cart(azimuth, elevation), which) == pytest.approx(want, abs=tol)
            # swapped
            assert ext.weight(centre, height, width, cart(elevation, azimuth), which) == pytest.approx(want, abs=tol)
    width = 360.0  # (the reference's loop runs after the one above: width keeps its last value)
    for azimuth in np.linspace(-180.0, 180.0, 50):
        want = interp(azimuth, [-(width / 2 + fade), -width / 2, width / 2, width / 2 + fade], [0, 1, 1, 0])
        assert ext.weight(centre, width, height, cart(azimuth, 0.0), which) == pytest.approx(want, abs=tol)
        assert ext.weight(centre, height, width, cart(0.0, azimuth), which) == pytest.approx(want, abs=tol)


def test_pv(ext):
    g = _oracle.GainCalculatorObjects("9+10+3")
    for az, el in ((0.0, 0.0), (10.0, 20.0)):
        pos = cart(az, el)
        assert np.array_equal(ext.handle(pos, 0.0, 0.0)[0], g.psp(pos)[0][0])  # zero extent: the point source panner
    from libear_amd import capi  # channel positions: the native table (data)
    chans = [c for c in capi.layout_channels("9+10+3") if not c[3]]
    spk = cart([c[1] for c in chans], [c[2] for c in chans])
    for (az, el), tol in (((0.0, 0.0), 1e-5), ((30.0, 10.0), 1e-2)):
        pos = cart(az, el)
        pv = ext.handle(pos, 20.0, 10.0)[0]
        assert np.linalg.norm(pv) == pytest.approx(1.0, rel=1e-6)
        vv = pv @ spk
        vv /= np.linalg.norm(vv)
        assert is_approx(vv, pos, tol)


@pytest.mark.parametrize("layout", ["9+10+3", "0+5+0", "4+5+0", "0+2+0"])
def test_same_as_reference(layout):
    """1000 random positions and sizes: library form vs the tests' reference form, 1e-5 (extent_tests.cpp:140-169)"""
    e = _oracle.PolarExtent(layout)
    rng = np.random.default_rng(17)
    n = 1000 if layout == "9+10+3" else 250
    pos = rng.uniform(-1, 1, (n, 3))
    pos /= np.linalg.norm(pos, axis=1, keepdims=True)
    width, height = rng.uniform(0, 360, n), rng.uniform(0, 360, n)
    a = e.handle(pos, width, height, 0.0, which=0)
    b = e.handle(pos, width, height, 0.0, which=1)
    for i in range(n):
        assert is_approx(a[i], b[i], 1e-5), (i, pos[i], width[i], height[i], a[i], b[i])


def test_depth_and_distance(ext):
    """depth: rms of the near and far renderings; distance shrinks / grows the extent (polar_extent.cpp:62-70, :290-302)"""
    rng = np.random.default_rng(5)
    n = 200
    pos = rng.uniform(-1, 1, (n, 3))
    pos *= (rng.uniform(0.1, 2.0, n) / np.linalg.norm(pos, axis=1))[:, None]
    width, height, depth = rng.uniform(0, 200, n), rng.uniform(0, 100, n), rng.uniform(0, 1.5, n)
    a = ext.handle(pos, width, height, depth, which=0)
    b = ext.handle(pos, width, height, depth, which=1)
    for i in range(n):
        assert is_approx(a[i], b[i], 1e-5)
    assert _oracle.extent_mod(30.0, 1.0) == pytest.approx(30.0)
    assert _oracle.extent_mod(30.0, 0.0) == pytest.approx(360.0)
    assert _oracle.extent_mod(30.0, 2.0) < 30.0
    # at the origin everything is all around: every loudspeaker gets something
    assert np.all(ext.handle(np.zeros(3), 0.0, 0.0, 0.0)[0] > 0.0)


def test_grid():
    e = _oracle.PolarExtent("0+5+0")
    xyz = e.grid()
    assert xyz.shape[0] == e.num_points
    assert np.allclose(np.linalg.norm(xyz, axis=1), 1.0)
    el = np.degrees(np.arcsin(np.clip(xyz[:, 2], -1, 1)))
    rows = np.unique(np.round(el, 6))
    assert len(rows) == 37 and np.allclose(np.diff(rows), 5.0)
    assert np.sum(np.isclose(el, 0.0)) == 72 and np.sum(np.isclose(el, 90.0)) == 1


def test_gain_calculator_with_extent():
    """GainCalculatorObjects::calculate: extent pv x gain, LFE columns zero, diffuse split"""
    e = _oracle.PolarExtent("4+5+0")
    g = _oracle.GainCalculatorObjects("4+5+0")
    d0, f0 = g.calculate([0.0, 30.0, 45.0], [0.0, 0.0, 30.0])
    d1, f1 = e.calculate([0.0, 30.0, 45.0], [0.0, 0.0, 30.0])
    assert np.array_equal(d0, d1) and np.array_equal(f0, f1)  # zero extent
    d, f = e.calculate([10.0], [5.0], width=60.0, height=30.0, gain=0.5, diffuse=0.25)
    lfe = [i for i, c in enumerate(LAYOUTS["4+5+0"]) if c.startswith("LFE")]
    assert np.all(d[:, lfe] == 0) and np.all(f[:, lfe] == 0)
    power = np.sum(d.astype(np.float64) ** 2 + f.astype(np.float64) ** 2)
    assert power == pytest.approx(0.25, rel=1e-5)
    assert np.sum(f.astype(np.float64) ** 2) / power == pytest.approx(0.25, rel=1e-5)
    assert np.count_nonzero(d[0]) > 3  # spread over more loudspeakers than a point source
