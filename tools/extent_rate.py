"""Rate of the Objects gain producer with extent (earhip_panner_calculate_extent): positions per second through
the host-pointer entry point; under rocprofv3 --kernel-trace --stats the kernels' own times."""
import sys
import time

import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from libear_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
layout = sys.argv[2] if len(sys.argv) > 2 else "9+10+3"
rng = np.random.default_rng(0)
az = rng.uniform(-180, 180, n)
el = np.degrees(np.arcsin(rng.uniform(-1, 1, n)))
ctx = capi.Context(0)
p = capi.Panner(ctx, layout)
cases = {
    "point sources (distance 1)": dict(),
    "extent 0-360 x 0-360": dict(width=rng.uniform(0, 360, n), height=rng.uniform(0, 360, n)),
    "extent 20 x 10": dict(width=20.0, height=10.0),
    "extent 90 x 30 with depth 0.5": dict(width=90.0, height=30.0, depth=0.5),
    "no extent, distance 0.2-2": dict(dist=rng.uniform(0.2, 2.0, n)),
}
for name, kw in cases.items():
    p.calculate(az[:1024], el[:1024], **{k: (v[:1024] if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
    t0 = time.perf_counter()
    d, f = p.calculate(az, el, **kw)
    dt = time.perf_counter() - t0
    print(f"{name}: {n} positions in {dt * 1e3:.2f} ms = {n / dt / 1e6:.2f} Mpositions/s (host pointers, incl. transfers)")
p.close()
