"""The device batch panner (libearhip group I, k_pan_objects) against the oracle's restatement of libear's
GainCalculatorObjects / point source panner (oracle/panner_oracle.hpp, pinned by the reference's own tests
in tests/test_oracle_panner.py): every BS.2051 layout, dense random directions and distances, a fine grid
around the loudspeakers and region borders, gain and diffuseness — float32 outputs equal to the last bit
(both sides compute in double and cast), plus the reference's known answers through the C ABI."""
import numpy as np
import pytest

import _oracle
from _hip import ctx
from layouts import LAYOUTS

pytestmark = pytest.mark.gpu

ULP = 1.2e-7  # one float32 ulp at 1.0: the double results may differ in their last bits before the cast


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_batch_panner_equals_oracle(layout):
    from libear_amd import capi
    rng = np.random.default_rng(len(layout) * 7 + sum(map(ord, layout)))
    n = 20000
    az = rng.uniform(-180.0, 180.0, n)
    el = np.degrees(np.arcsin(rng.uniform(-1.0, 1.0, n)))
    # the loudspeaker directions themselves and their neighbourhoods (region borders: the first region wins)
    chans = capi.layout_channels(layout)
    for i, (_, caz, cel, _) in enumerate(chans):
        k = 40 * i
        az[k:k + 40] = caz + np.r_[0.0, rng.normal(0, 0.5, 39)]
        el[k:k + 40] = np.clip(cel + np.r_[0.0, rng.normal(0, 0.5, 39)], -90, 90)
    az[-500:] = np.round(az[-500:] / 15.0) * 15.0  # directions on the 15 degree grid (exactly on many borders)
    el[-500:] = np.round(el[-500:] / 15.0) * 15.0
    dist = rng.uniform(0.1, 2.0, n)
    gain = rng.uniform(0.0, 2.0, n)
    diffuse = rng.choice([0.0, 0.25, 0.5, 1.0], n)
    p = capi.Panner(ctx(), layout)
    assert p.n_out == len(LAYOUTS[layout])
    d, f = p.calculate(az, el, dist, gain, diffuse)
    p.close()
    o = _oracle.GainCalculatorObjects(layout)
    wd, wf = o.calculate(az, el, dist, gain, diffuse)
    # at distances of 1 and more an object without extent is a point source: double on both sides, cast to float
    far = dist >= 1.0
    assert np.max(np.abs(d[far] - wd[far])) <= 2 * ULP and np.max(np.abs(f[far] - wf[far])) <= 2 * ULP
    assert np.mean(d[far] == wd[far]) > 0.999 and np.mean(f[far] == wf[far]) > 0.999  # (apart from rounding boundaries)
    # closer than 1 libear widens it (polar_extent.cpp:62-70, :290-302): the extent panner's float sums, held to
    # the 1e-5 of a gain vector's norm that libear's tests put between its own cores (tests/test_gpu_extent.py)
    near = ~far

    def rel(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        den = np.maximum(np.minimum(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)), 1e-30)
        return np.linalg.norm(a - b, axis=1) / den
    assert np.max(rel(d[near & (diffuse < 1.0)], wd[near & (diffuse < 1.0)])) <= 1e-5
    assert np.max(rel(f[near & (diffuse > 0.0)], wf[near & (diffuse > 0.0)])) <= 1e-5
    lfe = [i for i, nm in enumerate(LAYOUTS[layout]) if nm.startswith("LFE")]
    assert not d[:, lfe].any() and not f[:, lfe].any()
    # 3-sparse in practice: at most 4 loudspeakers carry a point source (quads), except under / above the
    # layout where a virtual n-gon spreads it
    nz = (d != 0).sum(axis=1) + 0
    assert np.median(nz[(diffuse < 1.0) & far]) <= 4


def test_reference_known_answers_through_the_c_abi():
    """tests/gain_calculator_objects_tests.cpp:74-132 (4+7+0) and point_source_panner_tests.cpp:420-446 (0+5+0)"""
    from libear_amd import capi
    names = LAYOUTS["4+7+0"]
    p = capi.Panner(ctx(), "4+7+0")

    def nonzero(v):
        return {names[i]: float(x) for i, x in enumerate(v) if abs(x) >= 1e-6}

    for (az, el), ch in (((0.0, 0.0), "M+000"), ((30.0, 0.0), "M+030"), ((45.0, 30.0), "U+045")):
        d, f = p.calculate(az, el)
        assert nonzero(d[0]) == {ch: pytest.approx(1.0)} and nonzero(f[0]) == {}
    d, f = p.calculate(0.0, 0.0, diffuse=0.5)
    assert nonzero(d[0]) == {"M+000": pytest.approx(np.sqrt(0.5))} and nonzero(f[0]) == {"M+000": pytest.approx(np.sqrt(0.5))}
    d, f = p.calculate(0.0, 0.0, diffuse=1.0)
    assert nonzero(d[0]) == {} and nonzero(f[0]) == {"M+000": pytest.approx(1.0)}
    d, f = p.calculate(0.0, 0.0, gain=0.5)
    assert nonzero(d[0]) == {"M+000": pytest.approx(0.5)}
    p.close()
    p = capi.Panner(ctx(), "0+5+0")
    d, _ = p.calculate([15.0, -15.0], [0.0, 0.0])
    n5 = LAYOUTS["0+5+0"]
    assert d[0][n5.index("M+030")] == pytest.approx(np.sqrt(0.5)) and d[0][n5.index("M+000")] == pytest.approx(np.sqrt(0.5))
    assert d[1][n5.index("M-030")] == pytest.approx(np.sqrt(0.5)) and d[1][n5.index("M+000")] == pytest.approx(np.sqrt(0.5))
    p.close()
    p = capi.Panner(ctx(), "0+2+0")  # stereo: 0 dB at the front to -3 dB at the back (point_source_panner_tests.cpp:80-101)
    d, _ = p.calculate([0.0, -30.0, -110.0, -180.0], [0.0] * 4)
    assert np.allclose(d, [[np.sqrt(0.5)] * 2, [0.0, 1.0], [0.0, np.sqrt(0.5)], [0.5, 0.5]], atol=1e-6)
    p.close()
    with pytest.raises(capi.UnknownLayout):  # (an InvalidArgument kind: ear::unknown_layout)
        capi.Panner(ctx(), "7+7+7")


def test_panned_scene_renders_within_tolerance():
    """moving point sources: gains from the device panner feed the renderer; the oracle renders the same
    scene from the ORACLE's gains — producer and render path checked end to end, per channel"""
    import scenes
    from libear_amd import capi
    layout, m, block, nblocks = "9+10+3", 96, 512, 6
    names = LAYOUTS[layout]
    n = len(names)
    total = block * nblocks
    dec = capi.design_decorrelators(names)
    az, el, diffuse, times = scenes.moving_sources(m, total, period=700, seed=5)
    p = capi.Panner(ctx(), layout)
    o = _oracle.GainCalculatorObjects(layout)
    curves, wcurves = [], []
    for i in range(m):
        d, f = p.calculate(az[i], el[i], None, None, diffuse[i])
        curves.append((times[i], d, f))
        wcurves.append((times[i],) + o.calculate(az[i], el[i], None, None, diffuse[i]))
    p.close()
    x = scenes.audio(m, total, seed=9)
    r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
    for i, (t, d, f) in enumerate(curves):
        r.set_object_points(i, t, d, f)
    got = r.process(x)
    r.close()
    w = _oracle.ObjectsRenderer(m, n, block, dec, 255)
    for i, (t, d, f) in enumerate(wcurves):
        w.set_points(i, 0, t, d)
        w.set_points(i, 1, t, f)
    want = w.process(x)
    assert scenes.rel_rms(got, want) <= 1e-6
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6


@pytest.mark.parametrize("layout,order,norm", [("0+5+0", 1, "SN3D"), ("9+10+3", 3, "SN3D"), ("4+5+0", 3, "N3D"),
                                               ("4+7+0", 3, "FuMa"), ("0+2+0", 2, "SN3D"), ("9+10+3", 5, "N3D")])
def test_hoa_decode_matrix_equals_oracle(layout, order, norm):
    """earhip_hoa_decode_matrix (AllRAD: the point source panner sampled on the device at the 5200 design
    directions) vs the oracle's restatement of GainCalculatorHOA: float32 values within 1e-6"""
    from libear_amd import capi
    idx = [(n, m) for n in range(order + 1) for m in range(-n, n + 1)]
    orders, degrees = [n for n, _ in idx], [m for _, m in idx]
    got = capi.hoa_decode_matrix(ctx(), layout, orders, degrees, norm)
    want = _oracle.hoa_decode_matrix(layout, orders, degrees, norm)
    assert got.shape == want.shape == (len(LAYOUTS[layout]), len(idx))
    assert np.max(np.abs(got - want)) <= 1e-6 * max(1.0, np.max(np.abs(want)))
    lfe = [i for i, nm in enumerate(LAYOUTS[layout]) if nm.startswith("LFE")]
    assert not got[lfe].any()


def test_hoa_decode_matrix_equals_the_float64_fixture():
    """the device's AllRAD design against tests/golden/hoa_allrad_f64.npz (independent float64 evaluation, see
    tests/test_oracle_hoa.py): float32 values within 1e-6 of the matrix's largest entry"""
    import os
    from libear_amd import capi
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hoa_allrad_f64.npz"))
    cases = sorted({k.rsplit("|", 1)[0] for k in z.files if "|" in k})
    assert len(cases) >= 6
    for key in cases:
        layout, order, kind = key.split("|")
        want = z[key + "|D"]
        got = capi.hoa_decode_matrix(ctx(), layout, z[key + "|orders"].tolist(), z[key + "|degrees"].tolist(), kind)
        assert got.shape == want.shape, key
        assert np.max(np.abs(got - want)) <= 1e-6 * max(1.0, np.max(np.abs(want))), (key, np.max(np.abs(got - want)))


def test_hoa_exceptions():
    """tests/gain_calculator_hoa_tests.cpp:39-80"""
    from libear_amd import capi
    for orders, degrees in (([0, 1, 1, 1], [0, -1, 0]), ([-1, 1, 1, 1], [0, -1, 0, 1]), ([0, 1, 1, 1], [0, -1, 0, 2]),
                            ([0, 1, 1, 1], [0, -1, 0, -2])):
        with pytest.raises(capi.InvalidArgument):
            capi.hoa_decode_matrix(ctx(), "0+5+0", orders, degrees)
    with pytest.raises(capi.AdmError) as e:  # (an InvalidArgument kind: ear::adm_error)
        capi.hoa_decode_matrix(ctx(), "0+5+0", [0], [0], "foo")
    assert "unknown normalization" in str(e.value) and isinstance(e.value, capi.InvalidArgument)


def test_hoa_bed_through_the_renderer_with_the_real_decode_matrix():
    """BASELINE config 5's shape with libear's actual decode matrix: an order-3 bed (16 channels, constant
    gains = the columns of the AllRAD matrix) + objects panned by the device producer, block 1024"""
    import scenes
    from libear_amd import capi
    layout, block, nblocks, n_obj = "9+10+3", 1024, 3, 48
    names = LAYOUTS[layout]
    n = len(names)
    idx = [(a, b) for a in range(4) for b in range(-a, a + 1)]
    D = capi.hoa_decode_matrix(ctx(), layout, [a for a, _ in idx], [b for _, b in idx], "SN3D")
    Dw = _oracle.hoa_decode_matrix(layout, [a for a, _ in idx], [b for _, b in idx], "SN3D").astype(np.float32)
    total = block * nblocks
    dec = capi.design_decorrelators(names)
    az, el, df, tms = scenes.moving_sources(n_obj, total, period=1024, seed=2, phase=0, ramp=1024)
    p = capi.Panner(ctx(), layout)
    o = _oracle.GainCalculatorObjects(layout)
    curves = [(np.zeros(1, np.int64), D[:, c][None, :].copy(), np.zeros((1, n), np.float32)) for c in range(16)]
    wcurves = [(np.zeros(1, np.int64), Dw[:, c][None, :].copy(), np.zeros((1, n), np.float32)) for c in range(16)]
    for i in range(n_obj):
        curves.append((tms[i],) + p.calculate(az[i], el[i], None, None, df[i]))
        wcurves.append((tms[i],) + o.calculate(az[i], el[i], None, None, df[i]))
    p.close()
    x = scenes.audio(16 + n_obj, total, seed=3)
    r = capi.Renderer(ctx(), 16 + n_obj, n, block, dec, 255, max_blocks=nblocks)
    for i, (t, d, f) in enumerate(curves):
        r.set_object_points(i, t, d, f)
    got = r.process(x)
    r.close()
    w = _oracle.ObjectsRenderer(16 + n_obj, n, block, dec, 255)
    for i, (t, d, f) in enumerate(wcurves):
        w.set_points(i, 0, t, d)
        w.set_points(i, 1, t, f)
    want = w.process(x)
    assert scenes.rel_rms(got, want) <= 1e-6 and scenes.rel_rms_per_channel(got, want) <= 1e-6


def _moved(layout, seed):
    """the layout's channels a few degrees off their nominal positions (full layout, LFE included)"""
    from libear_amd import capi
    ch = capi.layout_channels(layout)
    az = np.array([c[1] for c in ch], np.float64)
    el = np.array([c[2] for c in ch], np.float64)
    rng = np.random.default_rng(seed)
    raz = az + rng.uniform(-4, 4, len(az))
    rel = np.clip(el + rng.uniform(-3, 3, len(el)), -90, 90)
    raz[np.abs(el) == 90] = az[np.abs(el) == 90]
    return [c[0] for c in ch], az, el, raz, rel


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_real_loudspeaker_positions_equal_oracle(layout):
    """earhip_panner_create_positions: loudspeakers off their nominal positions (Channel::polarPosition) — point
    sources to the last bit of the oracle, objects with extent to 1e-5; the nominal positions given explicitly
    are the plain constructor"""
    from libear_amd import capi
    names, az, el, raz, rel = _moved(layout, sum(map(ord, layout)))
    rng = np.random.default_rng(5)
    n = 4000
    taz = rng.uniform(-180, 180, n)
    tel = np.degrees(np.arcsin(rng.uniform(-1, 1, n)))
    k = len(names)
    taz[:k], tel[:k] = raz, rel  # the real positions themselves
    p0 = capi.Panner(ctx(), layout)
    p1 = capi.Panner(ctx(), layout, (az, el))
    a, b = p0.calculate(taz, tel), p1.calculate(taz, tel)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    p0.close()
    p1.close()
    p = capi.Panner(ctx(), layout, (raz, rel))
    o = _oracle.PolarExtent(layout, (raz, rel))
    d, f = p.calculate(taz, tel, None, None, 0.25)
    wd, wf = o.calculate(taz, tel, diffuse=0.25)
    assert np.max(np.abs(d - wd)) <= 2 * ULP and np.max(np.abs(f - wf)) <= 2 * ULP
    assert np.mean(d == wd) > 0.999
    if layout != "0+2+0":  # a source at a real loudspeaker position plays from it alone
        lfe = np.array([nm.startswith("LFE") for nm in names])
        got = d[:k][~lfe][:, ~lfe].astype(np.float64) / np.sqrt(0.75)
        assert np.allclose(got, np.eye(int((~lfe).sum())), atol=1e-6)
    width, height = rng.uniform(0, 180, 600), rng.uniform(0, 90, 600)
    d, _ = p.calculate(taz[:600], tel[:600], None, None, None, width, height, None)
    wd, _ = o.calculate(taz[:600], tel[:600], width=width, height=height)
    den = np.minimum(np.linalg.norm(d, axis=1), np.linalg.norm(wd, axis=1))
    assert np.max(np.linalg.norm(d.astype(np.float64) - wd, axis=1) / den) <= 1e-5
    p.close()


def test_screen_loudspeakers_and_position_errors():
    """tests/point_source_panner_tests.cpp:522-551 through the C ABI; shape errors"""
    from libear_amd import capi
    names, az, el, _, _ = _moved("4+9+0", 1)
    for name, sign in (("M+SC", 1.0), ("M-SC", -1.0)):
        a = az.copy()
        a[names.index(name)] = sign * 40.0
        with pytest.raises(capi.NotImplementedInLibear):
            capi.Panner(ctx(), "4+9+0", (a, el))
        a[names.index(name)] = sign * 30.0
        with pytest.raises(capi.InvalidArgument):
            capi.Panner(ctx(), "4+9+0", (a, el))
    with pytest.raises(capi.InvalidArgument):
        capi.Panner(ctx(), "4+9+0", (az[:-1], el[:-1]))  # one position per channel of the full layout
    bad = az.copy()
    bad[0] = np.nan
    with pytest.raises(capi.InvalidArgument):
        capi.Panner(ctx(), "4+9+0", (bad, el))


def test_hoa_decode_matrix_with_real_positions():
    from libear_amd import capi
    layout = "4+5+0"
    _, az, el, raz, rel = _moved(layout, 9)
    idx = [(n, m) for n in range(3) for m in range(-n, n + 1)]
    orders, degrees = [n for n, _ in idx], [m for _, m in idx]
    got = capi.hoa_decode_matrix(ctx(), layout, orders, degrees, "SN3D", (raz, rel))
    want = _oracle.hoa_decode_matrix(layout, orders, degrees, "SN3D", (raz, rel))
    assert np.max(np.abs(got - want)) <= 1e-6 * max(1.0, np.max(np.abs(want)))
    nominal = capi.hoa_decode_matrix(ctx(), layout, orders, degrees, "SN3D")
    assert np.array_equal(nominal, capi.hoa_decode_matrix(ctx(), layout, orders, degrees, "SN3D", (az, el)))
    assert np.max(np.abs(got - nominal)) > 1e-3  # (moving the loudspeakers does change the decoder)
