"""Pins the oracle's Objects gain producer (oracle/panner_oracle.hpp) with the reference's own tests,
restated: tests/point_source_panner_tests.cpp (virtual n-gon :19-51, quad :53-78, stereo downmix :80-101,
extraPosVerticalNominal :115-283, every layout on a grid :333-417, loudspeaker positions :420-446) and
tests/gain_calculator_objects_tests.cpp:74-132 (centre / left / left-up, diffuse split, gain)."""
import numpy as np
import pytest

import _oracle
from _oracle import cart
from layouts import LAYOUTS, without_lfe


def approx(a, b, prec=1e-12):  # Eigen's isApprox: |a - b| <= prec * min(|a|, |b|)
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) <= max(prec, 1e-12) * min(np.linalg.norm(a), np.linalg.norm(b)) + 1e-15


def test_virtual_ngon():
    spk = cart([30.0, -30.0, 30.0, -30.0], [0.0, 0.0, 30.0, 30.0])
    dm = np.array([0.2, 0.2, 0.3, 0.3])
    vpos = dm @ spk
    pv = _oracle.region_handle("ngon", spk, vpos, centre=vpos, downmix=dm)
    assert pv is not None and approx(pv, dm / np.linalg.norm(dm))
    rng = np.random.default_rng(0)
    for _ in range(100):
        prop = rng.uniform(0, 1, 4)
        pos = spk.T @ prop
        pos /= np.linalg.norm(pos)
        pv = _oracle.region_handle("ngon", spk, pos, centre=vpos, downmix=dm)
        assert pv is not None
        back = pv @ spk
        back /= np.linalg.norm(back)
        assert approx(pos, back)


def test_quad():
    spk = cart([30.0, -30.0, 30.0, -30.0], [-15.0, -15.0, 15.0, 15.0])
    for i in range(4):
        want = np.zeros(4)
        want[i] = 1.0
        assert approx(_oracle.region_handle("quad", spk, spk[i]), want)
    assert approx(_oracle.region_handle("quad", spk, cart(0.0, 0.0)), [0.5] * 4)
    assert _oracle.region_handle("quad", spk, cart(180.0, 0.0)) is None


def test_stereo_downmix():
    for az, want in ((0.0, [np.sqrt(0.5)] * 2), (-30.0, [0.0, 1.0]), (-110.0, [0.0, np.sqrt(0.5)]),
                     (-180.0, [0.5, 0.5])):
        assert approx(_oracle.stereo_downmix_handle(cart(az, 0.0)), want)


EXTRA = {  # tests/point_source_panner_tests.cpp:119-283: positions of the extra channels and where they are mixed
    "0+5+0": ([(30, -30), (-30, -30), (0, -30), (110, -30), (-110, -30), (30, 30), (-30, 30), (0, 30), (110, 30), (-110, 30)],
              [0, 1, 2, 3, 4, 0, 1, 2, 3, 4]),
    "2+5+0": ([(30, -30), (-30, -30), (0, -30), (110, -30), (-110, -30), (110, 30), (-110, 30)], [0, 1, 2, 3, 4, 3, 4]),
    "4+5+0": ([(30, -30), (-30, -30), (0, -30), (110, -30), (-110, -30)], [0, 1, 2, 3, 4]),
    "4+5+1": ([(110, -30), (-110, -30)], [3, 4]),
    "3+7+0": ([(0, -30), (30, -30), (-30, -30), (90, -30), (-90, -30), (135, -30), (-135, -30)], [0, 1, 2, 5, 6, 7, 8]),
    "4+9+0": ([(30, -30), (-30, -30), (0, -30), (90, -30), (-90, -30), (135, -30), (-135, -30), (15, -30), (-15, -30)],
              [0, 1, 2, 3, 4, 5, 6, 11, 12]),
    "9+10+3": ([(135, -30), (-135, -30), (180, -30), (90, -30), (-90, -30)], [3, 4, 7, 8, 9]),
    "0+7+0": ([(30, -30), (-30, -30), (0, -30), (90, -30), (-90, -30), (135, -30), (-135, -30), (30, 30), (-30, 30), (0, 30),
               (90, 30), (-90, 30), (135, 30), (-135, 30)], [0, 1, 2, 3, 4, 5, 6, 0, 1, 2, 3, 4, 5, 6]),
    "4+7+0": ([(30, -30), (-30, -30), (0, -30), (90, -30), (-90, -30), (135, -30), (-135, -30)], [0, 1, 2, 3, 4, 5, 6]),
}


@pytest.mark.parametrize("layout", sorted(EXTRA))
def test_extra_pos_vertical_nominal(layout):
    pos, idx = _oracle.extra_pos_vertical_nominal(layout)
    assert [tuple(map(float, p)) for p in pos] == [tuple(map(float, p)) for p in EXTRA[layout][0]]
    assert list(idx) == EXTRA[layout][1]


def speaker_positions(layout):
    g = _oracle.GainCalculatorObjects(layout)
    return g


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_all_layouts_on_a_grid(layout):
    """non-negative, normalised gains everywhere; the velocity vector points at the source where the layout
    has no remapping; left/right mirror symmetry (point_source_panner_tests.cpp:333-417)"""
    from libear_amd import capi  # channel positions: the native table (data), checked against tests/layouts.py elsewhere
    chans = [c for c in capi.layout_channels(layout) if not c[3]]
    spk = cart([c[1] for c in chans], [c[2] for c in chans])
    flipped = spk * np.array([-1.0, 1.0, 1.0])
    flip = [int(np.argmin(np.linalg.norm(spk - f, axis=1))) for f in flipped]
    g = _oracle.GainCalculatorObjects(layout)
    for az in np.linspace(-180.0, 180.0, 21):
        for el in np.linspace(-90.0, 90.0, 11):
            pos = cart(az, el)
            pv, missed = g.psp(pos)
            assert missed == 0
            pv = pv[0]
            assert (pv >= 0.0).all()
            if layout == "0+2+0":
                if abs(az) <= 30.0 and el == 0.0:
                    assert abs(np.linalg.norm(pv) - 1.0) < 1e-9
                elif abs(az) >= 110.0 and el == 0.0:
                    assert abs(np.linalg.norm(pv) - np.sqrt(0.5)) < 1e-9
            else:
                assert abs(np.linalg.norm(pv) - 1.0) < 1e-9
            check = True
            if layout == "0+2+0":
                check = not (abs(az) >= 30.0 or el != 0.0)
            elif layout in ("0+5+0", "2+5+0", "0+7+0"):
                check = el == 0.0
            if layout == "9+10+3":
                check = check and not el < 0.0
            elif el != 0.0:
                check = False
            if check:
                vv = pv @ spk
                vv /= np.linalg.norm(vv)
                assert approx(vv, pos, 1e-9), (az, el)
            pvf, _ = g.psp(pos * np.array([-1.0, 1.0, 1.0]))
            assert approx(pv, pvf[0][flip], 1e-9), (az, el)


@pytest.mark.parametrize("layout", [l for l in sorted(LAYOUTS) if l != "0+2+0"])
def test_loudspeaker_positions_give_unit_gains(layout):
    """configure_full_polar_panner (point_source_panner_tests.cpp:420-446, :448-...)"""
    from libear_amd import capi
    chans = [c for c in capi.layout_channels(layout) if not c[3]]
    g = _oracle.GainCalculatorObjects(layout)
    for i, c in enumerate(chans):
        pv, missed = g.psp(cart(c[1], c[2]))
        want = np.zeros(len(chans))
        want[i] = 1.0
        assert missed == 0 and approx(pv[0], want, 1e-9)
    if layout == "0+5+0":
        for az, pair in ((15.0, (0, 2)), (-15.0, (1, 2))):
            want = np.zeros(5)
            want[list(pair)] = 1.0 / np.sqrt(2.0)
            assert approx(g.psp(cart(az, 0.0))[0][0], want, 1e-9)


def test_gain_calculator_objects_reference_cases():
    """tests/gain_calculator_objects_tests.cpp:74-132, layout 4+7+0 without LFE -> here with the LFE column,
    which stays exactly zero (gain_calculator_objects.cpp:50-52)"""
    names = LAYOUTS["4+7+0"]
    g = _oracle.GainCalculatorObjects("4+7+0")
    assert g.n_out == len(names)

    def nonzero(v):
        return {names[i]: float(x) for i, x in enumerate(v) if abs(x) >= 1e-6}

    for (az, el), ch in (((0.0, 0.0), "M+000"), ((30.0, 0.0), "M+030"), ((45.0, 30.0), "U+045")):
        d, f = g.calculate(az, el)
        assert nonzero(d[0]) == {ch: pytest.approx(1.0)} and nonzero(f[0]) == {}
    d, f = g.calculate(0.0, 0.0, diffuse=0.5)
    assert nonzero(d[0]) == {"M+000": pytest.approx(np.sqrt(0.5))} and nonzero(f[0]) == {"M+000": pytest.approx(np.sqrt(0.5))}
    d, f = g.calculate(0.0, 0.0, diffuse=1.0)
    assert nonzero(d[0]) == {} and nonzero(f[0]) == {"M+000": pytest.approx(1.0)}
    d, f = g.calculate(0.0, 0.0, gain=0.5)
    assert nonzero(d[0]) == {"M+000": pytest.approx(0.5)} and nonzero(f[0]) == {}
    lfe = names.index("LFE1")
    rng = np.random.default_rng(1)
    d, f = g.calculate(rng.uniform(-180, 180, 200), rng.uniform(-90, 90, 200), diffuse=rng.uniform(0, 1, 200))
    assert not d[:, lfe].any() and not f[:, lfe].any()
    assert np.allclose(np.linalg.norm(np.sqrt(d.astype(np.float64) ** 2 + f.astype(np.float64) ** 2), axis=1), 1.0, atol=1e-6)


def _full_positions(layout):
    from libear_amd import capi
    ch = capi.layout_channels(layout)
    return [c[0] for c in ch], np.array([c[1] for c in ch], np.float64), np.array([c[2] for c in ch], np.float64)


def test_screen_loudspeaker_positions():
    """tests/point_source_panner_tests.cpp:522-551: M+SC / M-SC wider than 25 degrees is not_implemented, outside
    5..25 and 35..60 invalid_argument"""
    names, az, el = _full_positions("4+9+0")
    for name, sign in (("M+SC", 1.0), ("M-SC", -1.0)):
        a = az.copy()
        a[names.index(name)] = sign * 40.0
        with pytest.raises(_oracle.OracleError) as e:
            _oracle.GainCalculatorObjects("4+9+0", (a, el))
        assert e.value.code == 4
        a[names.index(name)] = sign * 30.0
        with pytest.raises(_oracle.OracleError) as e:
            _oracle.GainCalculatorObjects("4+9+0", (a, el))
        assert e.value.code == 1
    _oracle.GainCalculatorObjects("4+9+0", (az, el))  # the nominal 15 degrees are fine


@pytest.mark.parametrize("layout", ["0+5+0", "4+5+0", "4+9+0", "9+10+3"])
def test_real_loudspeaker_positions(layout):
    """loudspeakers a few degrees off their nominal positions (Channel::polarPosition): a source AT a real position
    plays from that loudspeaker alone, gains stay non-negative and normalised, nominal positions given
    explicitly change nothing (point_source_panner.cpp:431-476: real positions for the geometry, nominal ones for
    the triangulation)"""
    names, az, el = _full_positions(layout)
    lfe = np.array([n.startswith("LFE") for n in names])
    g0 = _oracle.GainCalculatorObjects(layout)
    g1 = _oracle.GainCalculatorObjects(layout, (az, el))
    rng = np.random.default_rng(len(layout))
    taz, tel = rng.uniform(-180, 180, 200), rng.uniform(-30, 60, 200)
    assert np.array_equal(g0.calculate(taz, tel)[0], g1.calculate(taz, tel)[0])
    raz = az + rng.uniform(-4, 4, len(az))
    rel_ = np.clip(el + rng.uniform(-3, 3, len(el)), -90, 90)
    raz[np.abs(el) == 90] = az[np.abs(el) == 90]
    g = _oracle.GainCalculatorObjects(layout, (raz, rel_))
    d, _ = g.calculate(raz[~lfe], rel_[~lfe])
    assert np.allclose(d[:, ~lfe], np.eye(int((~lfe).sum())), atol=1e-6)
    for a, e in zip(taz[:60], tel[:60]):
        pv = g.calculate([a], [e])[0][0][~lfe].astype(np.float64)
        assert abs(np.linalg.norm(pv) - 1.0) < 1e-6 and (pv >= 0).all()
