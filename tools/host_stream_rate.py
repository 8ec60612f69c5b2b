"""PCIe-inclusive rate of the host-pointer entry point (earhip_render_process): T blocks per call
from pageable host arrays, incl. the staging copy, H2D, K0/K1/K2, D2H and the sync."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from layouts import LAYOUTS
from libear_amd import capi

M, B, T = 1024, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 64
names = LAYOUTS["9+10+3"]; N = len(names)
dec = capi.design_decorrelators(names)
curves = scenes.dense_curves(M, N, B, T, seed=7)
x = np.random.default_rng(0).uniform(-1, 1, (M, B * T)).astype(np.float32)
ctx = capi.Context(0, None)
r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=T)
for m, (t, d, f) in enumerate(curves):
    r.set_object_points(m, t, d, f)
r.commit()
for _ in range(2):
    r.reset(0); r.process(x)
ts = []
for _ in range(5):
    r.reset(0)
    t0 = time.perf_counter(); r.process(x); ts.append(time.perf_counter() - t0)
dt = sorted(ts)[len(ts) // 2]
print(f"host stream T={T}: {dt*1e3:.2f} ms per call, {M*B*T/dt/1e9:.2f} Gsamples/s, "
      f"{x.nbytes/dt/1e9:.1f} GB/s of input over PCIe incl. staging")
