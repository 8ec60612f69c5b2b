"""Create / use / destroy every kind of handle many times and watch the device's free memory and the process's RSS."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
import scenes  # noqa: E402
from layouts import LAYOUTS  # noqa: E402
from libear_amd import capi  # noqa: E402


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0
    return 0.0


names = LAYOUTS["9+10+3"]
n = len(names)
dec = capi.design_decorrelators(names)
x = scenes.audio(64, 512 * 4, seed=1)
curves = scenes.adm_curves(64, n, 512 * 4, seed=2)
marks = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    ctx = capi.Context(0)
    r = capi.Renderer(ctx, 64, n, 512, dec, 255, max_blocks=4)
    for i, (t, d, f) in enumerate(curves):
        r.set_object_points(i, t, d, f)
    r.process(x)
    r.close()
    p = capi.Panner(ctx, "9+10+3")
    p.calculate(np.linspace(-180, 180, 256), np.zeros(256), None, None, None, 30.0, 10.0, 0.3)
    p.close()
    cc = capi.ConvContext(ctx, 512) if hasattr(capi, "ConvContext") else None
    gi = capi.GainInterp(ctx, 4, n)
    gi.set_points([0, 512], np.random.default_rng(0).uniform(0, 1, (2, 4, n)).astype(np.float32))
    gi.process(0, x[:4, :512])
    gi.close()
    a = ctx.pinned_array((8, 512))
    ctx.release(a)
    ctx.close()
    if it % 25 == 0 or it < 3:
        free, total = torch.cuda.mem_get_info(0)
        marks.append((it, (total - free) / 2**20, rss_mb()))
        print(f"iteration {it}: device memory in use {marks[-1][1]:.1f} MiB, host RSS {marks[-1][2]:.1f} MiB", flush=True)
grow_dev = marks[-1][1] - marks[2][1]
grow_rss = marks[-1][2] - marks[2][2]
print(f"growth after warm-up: device {grow_dev:.1f} MiB, host {grow_rss:.1f} MiB")
sys.exit(0 if grow_dev < 64 and grow_rss < 64 else 1)
