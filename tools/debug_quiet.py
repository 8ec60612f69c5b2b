import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import scenes
import test_gpu_render_full as T
from layouts import LAYOUTS
layout, block = "9+10+3", 512
m, nblocks = int(os.environ.get("M", "1024")), int(os.environ.get("NB", "512"))
names = LAYOUTS[layout]; n = len(names)
dec = T.decorrelators(layout)
lfe = [i for i, nm in enumerate(names) if nm.startswith("LFE")]
curves, levels = scenes.mixed_level_sparse(m, n, block, nblocks, lfe, seed=m)
x = T.device_audio(m, block * nblocks, 5, scale=levels)
out, plan = T.render_device(curves, x, n, block, dec, 255, [nblocks])
print(plan, "quiet objects:", int((levels < 2.0 ** -11).sum()))
xw = x[:, :3 * block].cpu().numpy()
want = T.oracle_window(curves, xw, n, block, dec, 255, 0)
got = out[:, :3 * block].cpu().numpy()
for c in range(n):
    print(c, f"{scenes.rel_rms(got[c:c+1], want[c:c+1]):.2e}", f"{np.linalg.norm(want[c]):.3e}")
