"""Cost of updating gain curves between blocks (real-time use): set_object_points for all / a few
objects + commit + one block-mode call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from layouts import LAYOUTS
from libear_amd import capi

M, B = 1024, 512
names = LAYOUTS["9+10+3"]; N = len(names)
dec = capi.design_decorrelators(names)
rng = np.random.default_rng(0)
ctx = capi.Context(0, None)
r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=1)
x = rng.uniform(-1, 1, (M, B)).astype(np.float32)
def pts(t0):
    t = np.array([t0, t0 + B, t0 + 2 * B, t0 + 3 * B], np.int64)
    return t, rng.uniform(0, 1, (4, N)).astype(np.float32), rng.uniform(0, 1, (4, N)).astype(np.float32)
for m in range(M):
    r.set_object_points(m, *pts(0))
r.commit(); r.process(x)
for nupd in (0, 16, 1024):
    ts = []
    for it in range(30):
        t0 = time.perf_counter()
        for m in range(nupd):
            r.set_object_points(m, *pts((it + 1) * B))
        t1 = time.perf_counter()
        r.commit()
        t2 = time.perf_counter()
        r.process(x)
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    a = np.median(np.array(ts), axis=0) * 1e3
    print(f"{nupd:5d} objects updated per block: set_points {a[0]:.3f} ms (python side), commit {a[1]:.3f} ms, process {a[2]:.3f} ms")
