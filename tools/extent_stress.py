import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0]); sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
import torch
import _oracle
from libear_amd import capi
from layouts import LAYOUTS
ctx = capi.Context(0)
worst_all = 0
for layout in sorted(LAYOUTS):
    rng = np.random.default_rng(abs(hash(layout)) % 1000)
    n = 20000
    az = rng.uniform(-180, 180, n); el = np.degrees(np.arcsin(rng.uniform(-1, 1, n)))
    dist = rng.choice([0.0, 0.3, 0.9, 1.0, 1.1, 2.0, 5.0], n) * rng.uniform(0.9, 1.1, n)
    width = rng.choice([0, 1, 5, 9.99, 10, 10.01, 20, 45, 90, 179, 180, 181, 270, 359, 360, 400], n) * rng.uniform(0.999, 1.001, n)
    height = rng.choice([0, 1, 5, 9.99, 10, 10.01, 20, 45, 90, 179, 180, 181, 270, 359, 360], n) * rng.uniform(0.999, 1.001, n)
    depth = rng.choice([0, 0, 0.1, 1.0, 3.0], n)
    p = capi.Panner(ctx, layout)
    d, f = p.calculate(az, el, dist, None, None, width, height, depth)
    p.close()
    o = _oracle.PolarExtent(layout)
    wd, wf = o.calculate(az, el, dist, width, height, depth)
    a, b = d.astype(np.float64), wd.astype(np.float64)
    # an object AT the origin whose far rendering still needs the point source panner is 0 / 0 in libear (NaN gains)
    nan_a, nan_b = np.isnan(a).any(axis=1), np.isnan(b).any(axis=1)
    print(layout, "NaN rows: device", int(nan_a.sum()), "oracle", int(nan_b.sum()), "same rows", bool((nan_a == nan_b).all()))
    a, b = a[~(nan_a | nan_b)], b[~(nan_a | nan_b)]
    az, el, dist, width, height, depth = (v[~(nan_a | nan_b)] for v in (az, el, dist, width, height, depth))
    den = np.maximum(np.minimum(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)), 1e-30)
    r = np.linalg.norm(a - b, axis=1) / den
    i = int(np.argmax(r))
    print(layout, "max rel", r.max(), "at", az[i], el[i], dist[i], width[i], height[i], depth[i], "mean", r.mean())
    worst_all = max(worst_all, r.max())
print("WORST", worst_all)
