"""Where the caller's pages and the library's staging threads sit (NUMA nodes) against the PCIe-inclusive rate of
earhip_render_process from pageable channel pointers: `python tools/host_stream_numa.py`.
The main thread's affinity decides first touch (the pages' node); staging threads inherit the affinity of the thread that
makes the first long call (they are created there)."""
import ctypes, glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def cpulist(path):
    out = []
    for part in open(path).read().strip().split(","):
        if not part: continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out
nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    c = cpulist(d + "/cpulist")
    if c: nodes[int(d.rsplit("node", 1)[1])] = c
allcpus = sorted(os.sched_getaffinity(0))
print("nodes:", {k: f"{len(v)} cpus ({v[0]}..{v[-1]})" for k, v in nodes.items()}, "allowed:", len(allcpus))
for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(f, open(f).read().strip())
import scenes
from layouts import LAYOUTS
from libear_amd import capi
M, B, N = 1024, 512, 24
names = LAYOUTS["9+10+3"]
dec = capi.design_decorrelators(names)
lib = capi.load()
curves = scenes.dense_curves(M, N, B, 64, seed=7)
usable = {k: [c for c in v if c in allcpus] for k, v in nodes.items()}
usable = {k: v for k, v in usable.items() if v}
cases = [(pn, tn, bind) for pn in usable for tn in [None] + list(usable) for bind in (0, 1)]
for pages_node, thr_node, bind in cases:
    os.sched_setaffinity(0, usable[pages_node])
    T = 64
    xh = np.random.default_rng(0).uniform(-1, 1, (M, B * T)).astype(np.float32)   # first touch here
    yh = np.zeros((N, B * T), np.float32)
    os.sched_setaffinity(0, allcpus)
    ctx = capi.Context(0, None)
    ctx.set_option("HOST_BIND", str(bind))
    r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=T)
    for m, (t, d, f) in enumerate(curves):
        r.set_object_points(m, t, d, f)
    r.commit()
    h2d, _ = ctx.copy_bandwidth(xh, reps=2)
    ref = xh.nbytes / h2d / 1e6
    ip, op = capi._chan_ptrs(xh), capi._chan_ptrs(yh)
    if thr_node is not None: os.sched_setaffinity(0, usable[thr_node])
    ts = []
    for i in range(9):
        r.reset(0)
        t0 = time.perf_counter()
        capi.check(lib.earhip_render_process(r.h, ctypes.c_size_t(T), ip, op))
        ts.append(time.perf_counter() - t0)
        if i == 0: os.sched_setaffinity(0, allcpus)   # (the threads exist now; the caller itself is free again)
    dt = sorted(ts[2:])[len(ts[2:]) // 2]
    print(f"pages on node {pages_node}, staging threads {'free' if thr_node is None else 'started on node %d' % thr_node}, HOST_BIND={bind}: {dt*1e3:7.3f} ms  {xh.nbytes/dt/1e9:5.1f} GB/s  frac of plain copy ({ref:.1f}) {xh.nbytes/dt/1e9/ref:.3f}", flush=True)
    r.close(); ctx.close()
