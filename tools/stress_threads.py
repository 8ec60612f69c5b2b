"""Repeat tests/test_gpu_render.py::test_contexts_on_concurrent_threads' body and say WHERE a threaded result differs from the
serial one (which thread, which call, which channels / samples, by how much): python tools/stress_threads.py [repeats]"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from layouts import LAYOUTS
from libear_amd import capi

layout, block, nblocks = "4+5+0", 512, 8
names = LAYOUTS[layout]; n = len(names)
dec = capi.design_decorrelators(names)
total = block * nblocks

def work(seed, context, rounds, policy=True):
    m = 24 + 8 * seed
    curves = scenes.adm_curves(m, n, total, seed=seed) if seed % 2 else scenes.dense_curves(m, n, block, nblocks, seed=seed)
    x = scenes.audio(m, total, seed=10 + seed)
    r = capi.Renderer(context, m, n, block, dec, 255, max_blocks=nblocks)
    for i, (t, d, f) in enumerate(curves):
        r.set_object_points(i, t, d, f)
    outs, plans = [], []
    rng = np.random.default_rng(seed)
    sp, ep = rng.uniform(0, 1, (3, n)).astype(np.float32), rng.uniform(0, 1, (3, n)).astype(np.float32)
    xi = scenes.audio(3, 700, seed=20 + seed)
    for _ in range(rounds):
        r.reset(0)
        outs.append(r.process(x))
        plans.append(r.last_plan())
        o = np.zeros((n, 700), np.float32)
        if policy:
            context.apply_interp(xi, o, 10, 650, 1000, 990, 1800, sp, ep)
        outs.append(o)
    r.close()
    return outs, plans

c0 = capi.Context(0)
serial = [work(s, c0, 1) for s in range(4)]
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    results = [None] * 4
    def run(s):
        c = capi.Context(0)
        results[s] = work(s, c, 6)
        c.close()
    ths = [threading.Thread(target=run, args=(s,)) for s in range(4)]
    for t in ths: t.start()
    for t in ths: t.join()
    for s in range(4):
        outs, plans = results[s]
        for k, o in enumerate(outs):
            ref = serial[s][0][k % 2]
            if not np.array_equal(o, ref):
                bad += 1
                d = np.abs(o - ref)
                ch = np.nonzero(d.max(axis=1))[0]
                sm = np.nonzero(d.max(axis=0))[0]
                print(f"rep {rep} thread {s} output {k} ({'renderer' if k % 2 == 0 else 'policy'}): max |diff| {d.max():.3e} (max |ref| {np.abs(ref).max():.2f}), "
                      f"{len(ch)} channels {ch[:12]}, samples {sm.min()}..{sm.max()} ({len(sm)}), plan {plans[k // 2]}", flush=True)
print("mismatches:", bad)
