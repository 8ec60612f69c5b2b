"""debug aid: gain stage alone (one bus) through the piece-list kernel vs the oracle, error per tile"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes, _oracle
from libear_amd import capi

os.environ["EARHIP_MFMA"] = sys.argv[1] if len(sys.argv) > 1 else "5"
if len(sys.argv) > 2:
    os.environ["EARHIP_P2_TILE"] = sys.argv[2]
kind = sys.argv[3] if len(sys.argv) > 3 else "adm"
m, n, block, nblocks = int(os.environ.get("M", "64")), int(os.environ.get("N", "24")), 512, int(os.environ.get("NB", "4"))
total = block * nblocks
if kind == "adm":
    curves = scenes.adm_curves(m, n, total, period=700, ramp=150, seed=m)
elif kind == "dense":
    curves = scenes.dense_curves(m, n, block, nblocks, seed=m)
else:
    curves = scenes.constant_curves(m, n, seed=m)
x = scenes.audio(m, total, seed=m)
o = _oracle.ObjectsRenderer(m, n, block, np.zeros((n, 1), np.float32), 0)
for i, (t, d, f) in enumerate(curves):
    o.set_points(i, 0, t, d)
    o.set_points(i, 1, t, np.zeros_like(d))
want = o.process(x)
c = capi.Context(0)
r = capi.Renderer(c, m, n, block, None, 0, max_blocks=nblocks)
for i, (t, d, f) in enumerate(curves):
    r.set_object_points(i, t, d, None)
got = r.process(x)
print("plan", r.last_plan(), "rel rms", scenes.rel_rms(got, want))
T = r.last_plan()["tile"]
for t in range(0, total, T):
    e = scenes.rel_rms(got[:, t:t + T], want[:, t:t + T])
    sub = [f"{scenes.rel_rms(got[:, t + 64 * w:t + 64 * w + 64], want[:, t + 64 * w:t + 64 * w + 64]):.1e}" for w in range(T // 64)]
    print(f"tile {t // T:3d}: {e:.2e}  waves {sub}  cols {[f'{scenes.rel_rms(got[cc, t:t+T], want[cc, t:t+T]):.0e}' for cc in range(0, n, 5)]}")
