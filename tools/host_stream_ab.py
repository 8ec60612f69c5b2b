"""Same-process A/B of the gather's knobs (HOST_NT: streaming stores into the staging buffer, HOST_BIND: staging threads on the rows'
node, HOST_THREADS) on long calls from pageable channel pointers: python tools/host_stream_ab.py [rounds]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from layouts import LAYOUTS
from libear_amd import capi
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M, B, N, TMAX = 1024, 512, 24, 256
dec = capi.design_decorrelators(LAYOUTS["9+10+3"])
lib = capi.load()
x_all = np.random.default_rng(0).uniform(-1, 1, (M, B * TMAX)).astype(np.float32)
curves = scenes.dense_curves(M, N, B, TMAX, seed=7)
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), " affinity:", len(os.sched_getaffinity(0)), " loadavg:", open("/proc/loadavg").read().strip())
except OSError:
    pass
variants = [("defaults", {}), ("first chunk whole", {"HOST_FIRST": "0"}), ("nt0", {"HOST_NT": "0"}), ("bind1", {"HOST_BIND": "1"}),
            ("th16", {"HOST_THREADS": "16"}), ("th4", {"HOST_THREADS": "4"}), ("chunk 8 MB", {"HOST_CHUNK_MB": "8"}), ("chunk 32 MB", {"HOST_CHUNK_MB": "32"})]
rs = []
ctxs = []
for name, opts in variants:
    ctx = capi.Context(0, None)
    for k, v in opts.items(): ctx.set_option(k, v)
    r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=TMAX)
    for m, (t, d, f) in enumerate(curves): r.set_object_points(m, t, d, f)
    r.commit()
    rs.append(r); ctxs.append(ctx)
h2d, _ = ctxs[0].copy_bandwidth(x_all, reps=2)
ref = x_all.nbytes / h2d / 1e6
outs = {}
for T in (64, 256):
    xh = np.ascontiguousarray(x_all[:, :B * T]); yh = np.zeros((N, B * T), np.float32)
    ip, op = capi._chan_ptrs(xh), capi._chan_ptrs(yh)
    for rd in range(rounds):
        for (name, _), r in zip(variants, rs):
            ts = []
            for i in range(5):
                r.reset(0)
                t0 = time.perf_counter()
                capi.check(lib.earhip_render_process(r.h, ctypes.c_size_t(T), ip, op))
                ts.append(time.perf_counter() - t0)
            dt = sorted(ts[1:])[len(ts[1:]) // 2]
            if rd == 0: outs[(T, name)] = yh.copy()
            print(f"T={T:3d} round {rd} {name:16s}: {dt*1e3:7.3f} ms  {xh.nbytes/dt/1e9:5.1f} GB/s  frac {xh.nbytes/dt/1e9/ref:.3f}  (min {min(ts)*1e3:.3f} max {max(ts)*1e3:.3f})", flush=True)
    base = outs[(T, variants[0][0])]
    print("outputs identical across variants:", all(np.array_equal(base, outs[(T, n)]) for n, _ in variants))
