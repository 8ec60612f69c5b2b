"""The device extent panner (libearhip group I, k_pan_objects_extent: one wave per object and metadata block)
against the oracle's restatement of libear's PolarExtent with its scalar core (oracle/extent_oracle.hpp, pinned
by the reference's own tests in tests/test_oracle_extent.py).

Tolerance: libear sums the weighted point gains in float, and its own cores (scalar, SIMD batches) add them in
different orders; its tests hold them to 1e-5 of the gain vector's norm (Eigen isApprox,
tests/extent_tests.cpp:140-169).  The device adds them in yet another order (a lane per 64th point, then a
butterfly): the same 1e-5, measured here at <= 2e-6."""
import numpy as np
import pytest

import _oracle
from _hip import ctx
from _oracle import cart
from layouts import LAYOUTS

pytestmark = pytest.mark.gpu

TOL = 1e-5


def rel(a, b):
    """Eigen's isApprox measure per row: |a - b| / min(|a|, |b|)"""
    a, b = a.astype(np.float64), b.astype(np.float64)
    den = np.minimum(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1))
    return np.linalg.norm(a - b, axis=1) / np.maximum(den, 1e-30)


def scene(rng, n, layout=None):
    az = rng.uniform(-180.0, 180.0, n)
    el = np.degrees(np.arcsin(rng.uniform(-1.0, 1.0, n)))
    dist = rng.uniform(0.05, 2.0, n)
    width = rng.uniform(0.0, 360.0, n)
    height = rng.uniform(0.0, 360.0, n)
    # small extents (the blend with the point source below 10 degrees), tall ones, round ones, none
    width[: n // 8] = rng.uniform(0.0, 12.0, n // 8)
    height[: n // 8] = rng.uniform(0.0, 12.0, n // 8)
    height[n // 8: n // 4] = width[n // 8: n // 4]
    width[n // 4: n // 4 + n // 16] = 0.0
    height[n // 4: n // 4 + n // 16] = 0.0
    depth = np.where(rng.uniform(0, 1, n) < 0.4, rng.uniform(0.0, 1.5, n), 0.0)
    gain = rng.uniform(0.1, 2.0, n)
    diffuse = rng.choice([0.0, 0.25, 0.5, 1.0], n)
    return az, el, dist, width, height, depth, gain, diffuse


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_extent_panner_equals_oracle(layout):
    from libear_amd import capi
    rng = np.random.default_rng(sum(map(ord, layout)))
    n = 1500
    az, el, dist, width, height, depth, gain, diffuse = scene(rng, n)
    # the poles, the origin, the loudspeakers themselves
    el[:4] = [90.0, -90.0, 90.0 - 1e-6, -90.0 + 1e-6]
    dist[4:8] = [0.0, 1e-12, 1.0, 1.0]
    depth[4:6] = [0.0, 0.5]
    chans = capi.layout_channels(layout)
    for i, (_, caz, cel, _) in enumerate(chans):
        az[10 + i], el[10 + i] = caz, cel
    p = capi.Panner(ctx(), layout)
    d, f = p.calculate(az, el, dist, gain, diffuse, width, height, depth)
    p.close()
    o = _oracle.PolarExtent(layout)
    wd, wf = o.calculate(az, el, dist, width, height, depth, gain, diffuse)
    has_d, has_f = diffuse < 1.0, diffuse > 0.0
    assert np.max(rel(d[has_d], wd[has_d])) <= TOL, np.argmax(rel(d, wd))
    assert np.max(rel(f[has_f], wf[has_f])) <= TOL
    assert not d[~has_d].any() and not f[~has_f].any()
    lfe = [i for i, nm in enumerate(LAYOUTS[layout]) if nm.startswith("LFE")]
    assert not d[:, lfe].any() and not f[:, lfe].any()


def test_zero_extent_is_the_point_source_panner():
    """width = height = depth = 0 through the extent entry point: bit-identical to the plain one
    (tests/extent_tests.cpp:122-125)"""
    from libear_amd import capi
    rng = np.random.default_rng(3)
    n = 3000
    az, el, dist, _, _, _, gain, diffuse = scene(rng, n)
    p = capi.Panner(ctx(), "9+10+3")
    d0, f0 = p.calculate(az, el, dist, gain, diffuse)
    d1, f1 = p.calculate(az, el, dist, gain, diffuse, 0.0, 0.0, 0.0)
    p.close()
    assert np.array_equal(d0, d1) and np.array_equal(f0, f1)


def test_reference_properties_through_the_c_abi():
    """tests/extent_tests.cpp:116-138 (test_pv): unit norm, the energy vector points at the object"""
    from libear_amd import capi
    chans = [c for c in capi.layout_channels("9+10+3")]
    spk = cart([c[1] for c in chans], [c[2] for c in chans])
    spk[[i for i, c in enumerate(chans) if c[3]]] = 0.0
    p = capi.Panner(ctx(), "9+10+3")
    for (az, el), tol in (((0.0, 0.0), 1e-5), ((30.0, 10.0), 1e-2)):
        d, _ = p.calculate(az, el, width=20.0, height=10.0)
        pv = d[0].astype(np.float64)
        assert np.linalg.norm(pv) == pytest.approx(1.0, rel=1e-6)
        vv = pv @ spk
        vv /= np.linalg.norm(vv)
        assert np.linalg.norm(vv - cart(az, el)) <= tol
    # a full-sphere extent feeds every loudspeaker; the weight shape is symmetric left / right
    d, _ = p.calculate(0.0, 0.0, width=360.0, height=360.0)
    real = [i for i, c in enumerate(chans) if not c[3]]
    assert np.all(d[0][real] > 0.01)
    names = [c[0] for c in chans]
    d, _ = p.calculate(0.0, 10.0, width=90.0, height=30.0)
    for i, nm in enumerate(names):
        if "+" in nm and nm.replace("+", "-") in names and not nm.endswith("000") and not nm.endswith("180"):
            assert d[0][i] == pytest.approx(d[0][names.index(nm.replace("+", "-"))], abs=2e-6)
    p.close()


def test_large_batch_and_device_pointers():
    """one launch for 2^16 (object, block) pairs equals the oracle on a sample and itself in pieces"""
    from libear_amd import capi
    rng = np.random.default_rng(11)
    n = 1 << 16
    az, el, dist, width, height, depth, gain, diffuse = scene(rng, n)
    p = capi.Panner(ctx(), "4+5+0")
    d, f = p.calculate(az, el, dist, gain, diffuse, width, height, depth)
    k = 4096
    d2, f2 = p.calculate(az[:k], el[:k], dist[:k], gain[:k], diffuse[:k], width[:k], height[:k], depth[:k])
    # the same through device pointers (what a renderer feeds its curves from): same bits, and the counter of
    # positions no region took — which the host-pointer form turns into an error itself — reads zero for this call
    import torch
    dev = {q: torch.from_numpy(np.ascontiguousarray(v[:k], np.float64)).cuda()
           for q, v in (("az", az), ("el", el), ("dist", dist), ("gain", gain), ("diffuse", diffuse), ("width", width),
                        ("height", height), ("depth", depth))}
    dd = torch.empty((k, p.n_out), dtype=torch.float32, device="cuda")
    df = torch.empty_like(dd)
    torch.cuda.synchronize()
    p.calculate_device(k, dev["az"].data_ptr(), dev["el"].data_ptr(), dev["dist"].data_ptr(), dev["gain"].data_ptr(),
                       dev["diffuse"].data_ptr(), dd.data_ptr(), df.data_ptr(), dev["width"].data_ptr(),
                       dev["height"].data_ptr(), dev["depth"].data_ptr())
    assert p.missed() == 0
    assert np.array_equal(dd.cpu().numpy(), d2) and np.array_equal(df.cpu().numpy(), f2)
    p.close()
    assert np.array_equal(d[:k], d2) and np.array_equal(f[:k], f2)
    o = _oracle.PolarExtent("4+5+0")
    idx = rng.choice(n, 500, replace=False)
    wd, wf = o.calculate(az[idx], el[idx], dist[idx], width[idx], height[idx], depth[idx], gain[idx], diffuse[idx])
    m = diffuse[idx] < 1.0
    assert np.max(rel(d[idx][m], wd[m])) <= TOL


def test_objects_with_extent_through_the_renderer():
    """metadata -> gains -> loudspeakers on the device: moving objects with width / height / depth panned by the
    device producer feed the fused render; the render is checked against the CPU render of the same gain curves
    (1e-6 per channel) and the curves themselves against the oracle's producer (1e-5 of each vector's norm)"""
    import scenes
    from libear_amd import capi
    layout, m, block, nblocks = "9+10+3", 64, 512, 6
    names = LAYOUTS[layout]
    n = len(names)
    total = block * nblocks
    dec = capi.design_decorrelators(names)
    az, el, diffuse, times = scenes.moving_sources(m, total, period=700, seed=15)
    p = capi.Panner(ctx(), layout)
    o = _oracle.PolarExtent(layout)
    r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
    w = _oracle.ObjectsRenderer(m, n, block, dec, 255)
    worst = 0.0
    for i in range(m):
        # (extents tied to the position: the held stretches of a trajectory keep equal gain vectors)
        width, height = 45.0 + 45.0 * np.sin(np.radians(az[i])), 22.0 + 22.0 * np.cos(np.radians(3 * el[i]))
        depth = np.where(np.sin(np.radians(7 * az[i])) > 0.4, 0.4, 0.0)
        dist = 1.0 + 0.4 * np.sin(np.radians(5 * el[i] + az[i]))
        d, f = p.calculate(az[i], el[i], dist, None, diffuse[i], width, height, depth)
        wd, wf = o.calculate(az[i], el[i], dist, width, height, depth, None, diffuse[i])
        both = np.concatenate([d, f], axis=1).astype(np.float64), np.concatenate([wd, wf], axis=1).astype(np.float64)
        worst = max(worst, float(np.max(rel(*both))))
        r.set_object_points(i, times[i], d, f)
        w.set_points(i, 0, times[i], d)
        w.set_points(i, 1, times[i], f)
    p.close()
    assert worst <= TOL
    x = scenes.audio(m, total, seed=16)
    got = r.process(x)
    r.close()
    want = w.process(x)
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6
