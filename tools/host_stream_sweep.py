"""PCIe-inclusive rate of earhip_render_process from host channel pointers, swept over the pipeline's chunk size and the
staging threads: `python tools/host_stream_sweep.py [objects]` prints one line per (source, blocks per call, chunk MB, threads).
Channel-pointer arrays are built once (a C caller has them): what is timed is the C entry point."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from layouts import LAYOUTS
from libear_amd import capi

M, B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 512
names = LAYOUTS["9+10+3"]; N = len(names)
dec = capi.design_decorrelators(names)
lib = capi.load()
TMAX = 256
x_all = np.random.default_rng(0).uniform(-1, 1, (M, B * TMAX)).astype(np.float32)
curves = scenes.dense_curves(M, N, B, TMAX, seed=7)
combos = [(mb, th) for mb in (8, 16, 32, 64) for th in (16,)] + [(16, 8), (16, 32), (32, 32)]
for mb, th in combos:
    ctx = capi.Context(0, None)
    ctx.set_option("HOST_CHUNK_MB", str(mb)); ctx.set_option("HOST_THREADS", str(th))
    r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=TMAX)
    for m, (t, d, f) in enumerate(curves):
        r.set_object_points(m, t, d, f)
    r.commit()
    xpin = ctx.pinned_array((M, B * TMAX)); xpin[...] = x_all
    h2d, d2h = ctx.copy_bandwidth(xpin, reps=2)
    ref = xpin.nbytes / h2d / 1e6
    for T in (64, 256):
        xh = np.ascontiguousarray(x_all[:, :B * T]); yh = np.zeros((N, B * T), np.float32)
        xp = ctx.pinned_array((M, B * T)); xp[...] = xh; yp = ctx.pinned_array((N, B * T))
        for src, (xa, ya) in (("pageable", (xh, yh)), ("pinned", (xp, yp))):
            ip, op = capi._chan_ptrs(xa), capi._chan_ptrs(ya)
            ts = []
            for i in range(7):
                r.reset(0)
                t0 = time.perf_counter()
                capi.check(lib.earhip_render_process(r.h, ctypes.c_size_t(T), ip, op))
                ts.append(time.perf_counter() - t0)
            dt = sorted(ts[2:])[len(ts[2:]) // 2]
            print(f"chunk {mb:3d} MB threads {th:2d} {src:8s} T={T:3d}: {dt*1e3:7.3f} ms  {xa.nbytes/dt/1e9:5.1f} GB/s  frac of H2D ({ref:.1f}) {xa.nbytes/dt/1e9/ref:.3f}", flush=True)
        ctx.release(xp); ctx.release(yp)
    ctx.release(xpin)
    r.close(); ctx.close()
