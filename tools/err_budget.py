"""tools/err_budget.py — where the distance to the CPU path comes from: GPU and CPU render of one scene against a
float64 evaluation (tests/scenes.render_f64), worst channel.  usage: python tools/err_budget.py [objects] [blocks]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import scenes  # noqa: E402
import test_gpu_render as T  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
layout, block = "9+10+3", 512
n = len(T.LAYOUTS[layout])
dec = T.decorrelators(layout)
total = block * nblocks
for name, curves in (("dense", scenes.dense_curves(m, n, block, nblocks, seed=1)),
                     ("adm", scenes.adm_curves(m, n, total, seed=2)),
                     ("moving", scenes.adm_curves(m, n, total, period=240, ramp=240, seed=3))):
    x = scenes.audio(m, total, seed=5)
    cpu = T.run_oracle(curves, x, n, block, dec, 255)
    gpu = T.run_hip(curves, x, n, block, dec, 255, [nblocks])
    f64 = scenes.render_f64(curves, x, n, dec, 255)
    print(f"{name:7s} worst channel: gpu-cpu {scenes.rel_rms_per_channel(gpu, cpu):.3e}  gpu-f64 {scenes.rel_rms_per_channel(gpu, f64):.3e}"
          f"  cpu-f64 {scenes.rel_rms_per_channel(cpu, f64):.3e}")
