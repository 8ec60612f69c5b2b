import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from libear_amd import capi
os.environ["EARHIP_MFMA"] = "5"
os.environ["EARHIP_P2_TILE"] = "256"
m, n, block, nblocks = 64, 24, 512, 2
total = block * nblocks
c = capi.Context(0)
r = capi.Renderer(c, m, n, block, None, 0, max_blocks=nblocks)
for i in range(m):
    g = np.zeros((1, n), np.float32); g[0, :] = 1.0 + i / 64.0
    r.set_object_points(i, [0], g, None)
x = np.zeros((m, total), np.float32)
x[:, :] = (1.0 + np.arange(total) / 1024.0)[None, :]
got = r.process(x)
np.set_printoptions(linewidth=250, precision=4, suppress=True)
want = x.sum(0) * 0 + (x * (1.0 + np.arange(m) / 64.0)[:, None]).sum(0)
for t in range(0, total, 64):
    print(t, got[0, t:t + 6], got[23, t:t + 3], "want", want[t:t + 3])
