"""Native decorrelator design and BS.2051 channel table (libearhip groups G and H, host code) vs the
reference's known answers (reference tests/decorrelate_tests.cpp:19-44), the layout data of
tests/layouts.py and the CPU oracle.  Host code: runs in the CPU suite AND in the GPU suite (the driver's
`-m gpu` run loads the same library on the GPU box)."""
import numpy as np
import pytest

import _oracle
from layouts import LAYOUTS, without_lfe

both = pytest.mark.parametrize("where", ["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])


@both
def test_known_answers(where):
    from libear_amd import capi
    dec = capi.design_decorrelator_basic(7, 512)
    kat = {0: -0.1124280906086625, 1: -0.00944671630601479, 255: 0.057714955000898516,
           256: -0.018996037984052125, 510: 0.08336121588594464, 511: -0.012216595581941523}
    for i, v in kat.items():
        assert abs(dec[i] - v) < 1e-12


@both
def test_matches_oracle_and_id_rule(where):
    from libear_amd import capi
    assert capi.compensation_delay() == 255
    for layout in ("0+5+0", "4+5+0", "9+10+3"):
        names = LAYOUTS[layout]
        got = capi.design_decorrelators(names)
        want = _oracle.design_decorrelators(names)
        assert got.shape == want.shape
        assert np.max(np.abs(got - want)) <= 2e-9  # float casts of doubles that agree to ~1e-16
    names = without_lfe(LAYOUTS["4+5+0"])
    f = capi.design_decorrelators(names)
    assert np.array_equal(f[names.index("M+030")], capi.design_decorrelator_basic(1, 512).astype(np.float32))


@both
def test_native_layout_table(where):
    """every BS.2051 layout: channel names, order and LFE flags as in the reference's table; the
    decorrelators "for 4+5+0 without LFE" give M+030 filter id 1 (reference
    tests/decorrelate_tests.cpp:35-44) without the caller knowing any channel name"""
    from libear_amd import capi
    assert sorted(capi.layout_names()) == sorted(LAYOUTS)
    for layout, names in LAYOUTS.items():
        chans = capi.layout_channels(layout)
        assert [c[0] for c in chans] == names
        assert [c[3] for c in chans] == [n.startswith("LFE") for n in names]
        for name, az, el, lfe in chans:
            if not lfe and name[1:].lstrip("H")[1:].isdigit():  # e.g. M+030, U-110, UH+180: azimuth in the name
                sign = -1.0 if name.lstrip("MUHTB")[0] == "-" else 1.0
                assert az == sign * float(name[-3:])
                assert el == {"M": 0.0, "U": 30.0, "B": -30.0, "T": 90.0}[name[0]] or name.startswith("UH")
    assert capi.layout_channels("9+10+3")[15] == ("T+000", 0.0, 90.0, False)
    with pytest.raises(capi.UnknownLayout):
        capi.layout_channels("1+2+3")
    f = capi.design_decorrelators_for_layout("4+5+0", without_lfe=True)
    names = without_lfe(LAYOUTS["4+5+0"])
    assert f.shape == (len(names), 512)
    assert np.array_equal(f[names.index("M+030")], capi.design_decorrelator_basic(1, 512).astype(np.float32))
    assert np.array_equal(capi.design_decorrelators_for_layout("9+10+3"), capi.design_decorrelators(LAYOUTS["9+10+3"]))


@both
def test_native_layout_ranges(where):
    """tests/bs2051_tests.cpp:27-56 on the native table: every nominal position lies inside its allowed ranges
    (all_positions_in_range) and, screen loudspeakers and LFE aside, no azimuth range spans more than 180 degrees
    (azimuth_ranges: catches inverted ranges); left / right partners mirror each other (test_symmetry)"""
    from libear_amd import capi

    def inside(x, start, end):  # src/common/geom.cpp:7-28
        while end - 360.0 > start:
            end -= 360.0
        while end < start:
            end += 360.0
        while x - 360.0 >= start:
            x -= 360.0
        while x < start:
            x += 360.0
        return x <= end
    for layout in capi.layout_names():
        chans, ranges = capi.layout_channels(layout), capi.layout_channel_ranges(layout)
        assert len(chans) == len(ranges)
        by_name = {c[0]: (c, r) for c, r in zip(chans, ranges)}
        for (name, az, el, lfe), ((a0, a1), (e0, e1)) in zip(chans, ranges):
            assert inside(az, a0, a1) and e0 <= el <= e1, (layout, name)
            if not lfe and "SC" not in name:
                end = a1
                while end < a0:
                    end += 360.0
                assert end - a0 <= 180.0, (layout, name)
            if "+" in name and name.replace("+", "-") in by_name and not name.endswith(("000", "180")):
                (_, paz, pel, _), ((p0, p1), pe) = by_name[name.replace("+", "-")]
                assert (paz, pel) == (-az, el) and (p0, p1) == (-a1, -a0) and pe == (e0, e1), (layout, name)
    assert capi.layout_channel_ranges("0+5+0")[4] == ((100.0, 120.0), (0.0, 15.0))  # M+110
    with pytest.raises(capi.InvalidArgument):
        capi.layout_channel_ranges("1+2+3")
