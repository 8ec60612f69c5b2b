"""Pins the oracle's HOA decode-matrix design (hoa_oracle in oracle/panner_oracle.hpp: libear's
src/hoa/hoa.hpp:16-182, gain_calculator_hoa.cpp:8-72).  libear's own tests hold no values for it
(tests/hoa_tests.cpp checks the point set, tests/gain_calculator_hoa_tests.cpp warnings and exceptions), so
beyond those the spherical harmonics are checked against an independent implementation (scipy's associated
Legendre functions) and the design through the properties AllRAD is built on."""
import math

import numpy as np
import pytest
from scipy.special import lpmv

import _oracle
from layouts import LAYOUTS


def acn(order):
    return [(n, m) for n in range(order + 1) for m in range(-n, n + 1)]


def test_load_points():
    """tests/hoa_tests.cpp:7-16: 5200 points on the unit sphere"""
    p = _oracle.tdesign_points()
    assert p.shape == (5200, 3)
    assert np.allclose(np.linalg.norm(p, axis=1), 1.0, atol=1e-12)
    assert np.linalg.norm(p.mean(axis=0)) < 1e-6  # a spherical design: the points balance


def test_spherical_harmonics_against_scipy():
    """hoa.hpp:99-112 with BS.2076-1 normalisations; scipy's lpmv includes the Condon-Shortley phase"""
    rng = np.random.default_rng(0)
    fuma = {(0, 0): 1 / math.sqrt(2), (1, 0): 1, (1, 1): 1, (2, 0): 1, (2, 1): 2 / math.sqrt(3), (2, 2): 2 / math.sqrt(3),
            (3, 0): 1, (3, 1): math.sqrt(45 / 32), (3, 2): 3 / math.sqrt(5), (3, 3): math.sqrt(8 / 5)}
    for n, m in acn(6):
        for _ in range(5):
            az, el = rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi / 2, np.pi / 2)
            am = abs(m)
            leg = (-1.0) ** am * lpmv(am, n, np.sin(el))
            trig = math.sqrt(2) * math.cos(m * az) if m > 0 else (-math.sqrt(2) * math.sin(m * az) if m < 0 else 1.0)
            sn3d = math.sqrt(math.factorial(n - am) / math.factorial(n + am))
            want = {"SN3D": sn3d, "N3D": math.sqrt(2 * n + 1) * sn3d}
            if n <= 3:
                want["FuMa"] = fuma[(n, am)] * sn3d
            for norm, k in want.items():
                assert _oracle.sph_harm(n, m, az, el, norm) == pytest.approx(k * leg * trig, rel=1e-12, abs=1e-13)


def test_n3d_harmonics_are_orthonormal_over_the_design():
    p = _oracle.tdesign_points()
    az = -np.arctan2(p[:, 0], p[:, 1])
    el = np.arctan2(p[:, 2], np.hypot(p[:, 0], p[:, 1]))
    idx = acn(3)
    Y = np.array([[_oracle.sph_harm(n, m, a, e, "N3D") for a, e in zip(az[::4], el[::4])] for n, m in idx])
    gram = Y @ Y.T / Y.shape[1]
    assert np.allclose(gram, np.eye(len(idx)), atol=0.06)  # (a quarter of the points: a coarse quadrature)


@pytest.mark.parametrize("layout", ["0+5+0", "4+5+0", "9+10+3"])
def test_decode_matrix_properties(layout):
    idx = acn(3)
    orders, degrees = [n for n, _ in idx], [m for _, m in idx]
    D = _oracle.hoa_decode_matrix(layout, orders, degrees, "N3D")
    names = LAYOUTS[layout]
    assert D.shape == (len(names), 16)
    lfe = [i for i, nm in enumerate(names) if nm.startswith("LFE")]
    assert not D[lfe].any()
    # mean output power 1 for unit plane waves from the design's directions (normalize_decode_matrix)
    p = _oracle.tdesign_points()[::8]
    az = -np.arctan2(p[:, 0], p[:, 1])
    el = np.arctan2(p[:, 2], np.hypot(p[:, 0], p[:, 1]))
    Y = np.array([[_oracle.sph_harm(n, m, a, e, "N3D") for a, e in zip(az, el)] for n, m in idx])
    power = np.sum((D @ Y) ** 2, axis=0)
    assert np.mean(power) == pytest.approx(1.0, rel=0.05)
    # a plane wave is reproduced from its direction: the energy vector of the loudspeaker signals points there
    from libear_amd import capi
    spk = _oracle.cart([c[1] for c in capi.layout_channels(layout)], [c[2] for c in capi.layout_channels(layout)])
    for saz, sel in ((0.0, 0.0), (90.0, 0.0), (-30.0, 0.0), (110.0, 0.0)) + (((45.0, 30.0),) if layout != "0+5+0" else ()):
        y = np.array([_oracle.sph_harm(n, m, math.radians(saz), math.radians(sel), "N3D") for n, m in idx])
        e = ((D @ y) ** 2) @ spk
        e /= np.linalg.norm(e)
        assert float(e @ _oracle.cart(saz, sel)) > math.cos(math.radians(30.0)), (saz, sel)
    # normalisation conversion: decoding SN3D signals = decoding N3D signals scaled per coefficient
    Ds = _oracle.hoa_decode_matrix(layout, orders, degrees, "SN3D")
    conv = np.array([math.sqrt(2 * n + 1) for n in orders])
    assert np.allclose(Ds, D * conv[None, :], rtol=1e-12)
    # a subset of coefficients in another order gives the same columns up to the common power normalisation
    D1 = _oracle.hoa_decode_matrix(layout, [1, 0, 1], [1, 0, -1], "N3D")
    ratio = D1[:, 1] / np.where(D[:, 0] == 0, 1, D[:, 0])
    keep = D[:, 0] != 0
    assert np.allclose(ratio[keep], ratio[keep][0])


def test_exceptions():
    """tests/gain_calculator_hoa_tests.cpp:39-80 (+ the unknown normalisation of gain_calculator_hoa.cpp:39-42)"""
    with pytest.raises(_oracle.OracleError):
        _oracle.hoa_decode_matrix("0+5+0", [0, 1, 1, 1], [0, -1, 0])
    for orders, degrees in (([-1, 1, 1, 1], [0, -1, 0, 1]), ([0, 1, 1, 1], [0, -1, 0, 2]), ([0, 1, 1, 1], [0, -1, 0, -2])):
        with pytest.raises(_oracle.OracleError) as e:
            _oracle.hoa_decode_matrix("0+5+0", orders, degrees)
        assert e.value.code == 1
    with pytest.raises(_oracle.OracleError) as e:
        _oracle.hoa_decode_matrix("0+5+0", [0], [0], "foo")
    assert e.value.code == 1 and "unknown normalization" in str(e.value)


def _hoa_golden():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hoa_allrad_f64.npz")
    z = np.load(path)
    cases = sorted({k.rsplit("|", 1)[0] for k in z.files if "|" in k})
    return z, cases


def test_decode_matrix_against_the_independent_float64_fixture():
    """tests/golden/hoa_allrad_f64.npz (made by tests/golden/make_hoa_golden.py: scipy harmonics + numpy algebra over the
    panner's gains at the reference's own 5200 design directions): the oracle's restatement of hoa.hpp agrees to 1e-12.
    The reference's tests hold no values for this matrix, so this is "cannot drift", not "pinned by the reference"."""
    z, cases = _hoa_golden()
    assert len(cases) >= 6
    for key in cases:
        layout, order, kind = key.split("|")
        want = z[key + "|D"]
        got = _oracle.hoa_decode_matrix(layout, z[key + "|orders"].tolist(), z[key + "|degrees"].tolist(), kind)
        assert got.shape == want.shape, key
        assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.max(np.abs(want))), (key, np.max(np.abs(got - want)))
