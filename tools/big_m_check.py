"""Object counts beyond BASELINE's: the fused render against the oracle and against a float64 evaluation."""
import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0]); sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
import torch
import _oracle, scenes
from libear_amd import capi
from layouts import LAYOUTS
ctx = capi.Context(0)
names = LAYOUTS["9+10+3"]; n = len(names)
dec = capi.design_decorrelators(names)
for m, block, nblocks, kind in ((4096, 512, 6, "dense"), (3000, 512, 5, "adm"), (8192, 256, 4, "dense"), (2048, 1024, 3, "adm")):
    total = block * nblocks
    curves = scenes.dense_curves(m, n, block, nblocks, seed=1) if kind == "dense" else scenes.adm_curves(m, n, total, seed=2)
    x = scenes.audio(m, total, seed=3)
    r = capi.Renderer(ctx, m, n, block, dec, 255, max_blocks=nblocks)
    w = _oracle.ObjectsRenderer(m, n, block, dec, 255)
    for i, (t, d, f) in enumerate(curves):
        r.set_object_points(i, t, d, f)
        w.set_points(i, 0, t, d); w.set_points(i, 1, t, f)
    got = r.process(x); plan = r.last_plan(); r.close()
    want = w.process(x)
    truth = scenes.render_f64(curves, x, n, dec, 255)
    print(m, block, nblocks, kind, "kernel", plan["kernel"], "device vs f32 oracle", scenes.rel_rms(got, want),
          "| vs float64: device", scenes.rel_rms(got, truth), "oracle (libear's object-order f32 sum)", scenes.rel_rms(want, truth))
