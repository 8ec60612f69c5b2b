"""Random configurations of the fused render against the CPU oracle (on the GPU box): object counts, layouts (1-3 column
tiles), block sizes, call splits, curve families with jittered periods — each under the kernels it can be forced onto.
    python tools/stress_random.py [cases] [seed]
Prints every case; exits non-zero when one is beyond 1e-6 per channel or not finite."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle, scenes
from _hip import set_oracle_curves, set_renderer_curves
from layouts import LAYOUTS
from libear_amd import capi


def jittered(n_obj, n_out, total, rng, lo, hi, hold):
    """a new target every lo..hi samples (per interval), reached over the whole interval or (hold) a part of it"""
    curves = []
    for m in range(n_obj):
        t = [int(rng.integers(-hi, 0))]
        while t[-1] < total + hi:
            t.append(t[-1] + int(rng.integers(lo, hi + 1)))
        t = np.array(t, np.int64)
        if hold:
            tt, k = [], 0
            for a, b in zip(t[:-1], t[1:]):
                r = int(rng.integers(max(1, (b - a) // 4), b - a + 1))
                tt += [a, a + r] if a + r < b else [a]
            t = np.array(sorted(set(tt + [int(t[-1])])), np.int64)
        d = rng.uniform(0, 1, (len(t), n_out)).astype(np.float32)
        f = rng.uniform(0, 1, (len(t), n_out)).astype(np.float32)
        if hold:  # constant stretches: repeat the previous point's gains on every other point
            for k in range(2, len(t), 2):
                d[k], f[k] = d[k - 1], f[k - 1]
        curves.append((t, d, f))
    return curves


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    layouts = ["0+5+0", "4+5+0", "9+10+3", "0+2+0"]
    bad = 0
    for case in range(ncases):
        layout = layouts[int(rng.integers(0, len(layouts)))]
        n = len(LAYOUTS[layout])
        block = int(rng.choice([64, 100, 256, 512, 512, 1024]))
        m = int(rng.choice([33, 64, 96, 130, 257, 600, 1100]))
        nblocks = int(rng.integers(3, 40 if m < 300 else 14))
        total = block * nblocks
        fam = str(rng.choice(["moving", "moving", "hold", "dense", "short", "mixed"]))
        if fam == "moving":
            lo = int(rng.choice([128, 130, 200, 240, 300, 700]))
            curves = jittered(m, n, total, rng, lo, lo + int(rng.integers(0, 200)), False)
        elif fam == "hold":
            curves = jittered(m, n, total, rng, 300, 1500, True)
        elif fam == "dense":
            curves = scenes.dense_curves(m, n, block, nblocks, seed=case)
        elif fam == "short":
            curves = jittered(m, n, total, rng, 5, 90, False)
        else:
            curves = jittered(m, n, total, rng, 240, 260, False)
            odd = jittered(m, n, total, rng, 7, 400, True)
            for i in range(0, m, 5):
                curves[i] = odd[i]
        calls, left = [], nblocks
        while left:
            c = int(rng.integers(1, left + 1))
            calls.append(c); left -= c
        x = (scenes.audio(m, total, seed=case) * np.float32(10.0 ** float(rng.uniform(-3, 0.3)))).astype(np.float32)
        dec = capi.design_decorrelators(LAYOUTS[layout])
        o = _oracle.ObjectsRenderer(m, n, block, dec, 255)
        set_oracle_curves(o, curves)
        want = o.process(x)
        # (exact f32 on block-aligned curves: the tile-grid kernel where the call is whole 512-sample tiles, else the slot kernel)
        forces = [None, "6", "5", "4" if fam == "dense" else "1"] + (["1"] if fam == "dense" else [])
        for force in forces:
            env = {"EARHIP_MFMA": force, "EARHIP_HG_TILE": str(rng.choice(["256", "512"])), "EARHIP_H2_WGS": str(rng.choice(["8", "256"])),
                   "EARHIP_P2_WGS": str(rng.choice(["1", "3", "8", "256"]))}  # (few workgroups: every one crosses list boundaries)
            keep = {k: os.environ.get(k) for k in env}
            for k, v in env.items():
                os.environ.pop(k, None)
                if v is not None: os.environ[k] = v
            try:
                c = capi.Context(0)
                r = capi.Renderer(c, m, n, block, dec, 255, max_blocks=max(calls))
                set_renderer_curves(r, curves, True)
                got = np.zeros((n, total), np.float32)
                ofs = 0
                for nb in calls:
                    got[:, ofs:ofs + nb * block] = r.process(x[:, ofs:ofs + nb * block]); ofs += nb * block
                plan = r.last_plan()
                r.close(); c.close()
            finally:
                for k, v in keep.items():
                    os.environ.pop(k, None)
                    if v is not None: os.environ[k] = v
            e = scenes.rel_rms_per_channel(got, want) if np.isfinite(got).all() else float("inf")
            # (the hinge kernel forced onto curves it can only send through its exact path — short ramps, a point per
            # 64 samples — sums a thousand objects on one accumulator: plan_mix never picks it there)
            ok = e <= (1.5e-6 if force == "6" and fam in ("short", "dense", "mixed", "hold") else 1e-6)
            bad += 0 if ok else 1
            print(f"case {case:3d} {fam:7s} {layout:7s} m={m:5d} B={block:5d} nb={nblocks:3d} calls={len(calls):2d} force={force} "
                  f"hg={env['EARHIP_HG_TILE']} wgs={env['EARHIP_H2_WGS']}/{env['EARHIP_P2_WGS']} -> kernel {plan['kernel']} tile {plan['tile']}: {e:.3e} {'ok' if ok else 'FAIL'}", flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
