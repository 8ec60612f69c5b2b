"""Cost of updating gain curves between blocks (real-time use): 1024 objects, each a sliding window of 64 curve points
(one per block); every block `nupd` objects get their window replaced (set_object_points: the oldest point dropped, a new
one appended), then commit + one block-mode call.  commit() should cost what CHANGED, not what exists
(libear: interp_points is a public per-object vector the caller mutates freely, gain_interpolator.hpp:42-43)."""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from layouts import LAYOUTS
from libear_amd import capi

M, B, W = 1024, 512, 64
names = LAYOUTS["9+10+3"]; N = len(names)
dec = capi.design_decorrelators(names)
rng = np.random.default_rng(0)
ctx = capi.Context(0, None)
r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=1)
x = rng.uniform(-1, 1, (M, B)).astype(np.float32)
lib = capi.load()


def window(first_block):
    t = (first_block + np.arange(W, dtype=np.int64)) * B
    return t, rng.uniform(0, 1, (W, N)).astype(np.float32), rng.uniform(0, 1, (W, N)).astype(np.float32)


for m in range(M):
    r.set_object_points(m, *window(0))
r.commit(); r.process(x)
out = np.zeros((N, B), np.float32)
ip, op = capi._chan_ptrs(x), capi._chan_ptrs(out)  # built once: not part of a call


def process():
    capi.check(lib.earhip_render_process(r.h, C.c_size_t(1), ip, op))
# the windows of the objects that change, prepared ahead (what is timed is the library, not numpy)
for nupd in (0, 16, 64, 1024):
    ts = []
    for it in range(40):
        wins = [window(it + 1) for _ in range(nupd)]
        t0 = time.perf_counter()
        for m in range(nupd):
            r.set_object_points((it * 16 + m) % M, *wins[m])
        t1 = time.perf_counter()
        r.commit()
        t2 = time.perf_counter()
        process()
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    a = np.median(np.array(ts[5:]), axis=0) * 1e3
    print(f"{nupd:5d} of {M} objects replaced per block ({W}-point windows): set_object_points {a[0]:.3f} ms (incl. the Python side), "
          f"commit {a[1] * 1e3:.1f} us, block-mode process {a[2]:.3f} ms")
