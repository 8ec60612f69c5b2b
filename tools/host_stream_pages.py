"""Does it matter who first touched the caller's rows?  The same 256-block call from rows made by numpy in the calling thread and from
rows that torch's `.cpu()` filled (bench.py's 256-block case), with the NUMA node of the rows and of the calling thread printed."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
from layouts import LAYOUTS
from libear_amd import capi
libc = ctypes.CDLL(None, use_errno=True)
def node_of(addr):
    pg = ctypes.c_void_p(addr & ~4095); st = ctypes.c_int(-1)
    rc = libc.syscall(279, 0, ctypes.c_ulong(1), ctypes.byref(pg), None, ctypes.byref(st), 0)  # SYS_move_pages (x86-64)
    return st.value if rc == 0 else -9
def cpu_node():
    cpu = libc.sched_getcpu()
    for d in os.listdir("/sys/devices/system/node"):
        if d.startswith("node") and d[4:].isdigit():
            for part in open(f"/sys/devices/system/node/{d}/cpulist").read().strip().split(","):
                a, _, b = part.partition("-")
                if int(a) <= cpu <= int(b or a): return cpu, int(d[4:])
    return cpu, -1
M, B, N, T = 1024, 512, 24, 256
dec = capi.design_decorrelators(LAYOUTS["9+10+3"])
lib = capi.load()
curves = scenes.dense_curves(M, N, B, T, seed=7)
ctx = capi.Context(0, None)
r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=T)
for m, (t, d, f) in enumerate(curves): r.set_object_points(m, t, d, f)
r.commit()
xd = torch.rand((M, B * T), device="cuda") * 2 - 1
x_torch = np.ascontiguousarray(xd.cpu().numpy())
x_numpy = np.array(x_torch, copy=True)
h2d, _ = ctx.copy_bandwidth(x_numpy, reps=2)
ref = x_numpy.nbytes / h2d / 1e6
yh = np.zeros((N, B * T), np.float32)
op = capi._chan_ptrs(yh)
def throttled():
    try:
        d = dict(ln.split() for ln in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)), int(d.get("usage_usec", 0))
    except OSError:
        return 0, 0, 0
for rd in range(4):
    for name, xa in (("numpy-touched", x_numpy), ("torch .cpu()", x_torch)):
        ip = capi._chan_ptrs(xa)
        nodes = [node_of(xa.ctypes.data + i * xa.strides[0]) for i in (0, M // 2, M - 1)]
        th0 = throttled(); w0 = time.perf_counter()
        ts = []
        for i in range(6):
            r.reset(0)
            t0 = time.perf_counter()
            capi.check(lib.earhip_render_process(r.h, ctypes.c_size_t(T), ip, op))
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts[1:])[len(ts[1:]) // 2]
        print(f"round {rd} {name:14s} rows on nodes {nodes}, caller on cpu/node {cpu_node()}: {dt*1e3:7.3f} ms  frac {xa.nbytes/dt/1e9/ref:.3f}  (min {min(ts)*1e3:.2f} max {max(ts)*1e3:.2f})  throttled +{throttled()[0]-th0[0]} periods, {throttled()[1]-th0[1]} us; cpu {(throttled()[2]-th0[2])/1e3:.0f} ms in {(time.perf_counter()-w0)*1e3:.0f} ms wall", flush=True)
