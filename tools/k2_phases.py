"""Phase times inside k_decorrelate_wave (a library built with -DEARHIP_K2_PROF, EARHIP_LIB pointing at it): one headline-shaped
call (or --config C3 shape with argv[1] = 256), then the stamps of wave 0 of the grid's first and a middle workgroup."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
from layouts import LAYOUTS
from libear_amd import capi

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = 512
names = LAYOUTS["9+10+3"]; N = len(names)
dec = capi.design_decorrelators(names)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=T)
for m, (t, d, f) in enumerate(scenes.dense_curves(M, N, B, T, seed=7)):
    r.set_object_points(m, t, d, f)
r.commit()
x = torch.rand((M, B * T), device="cuda") * 2 - 1
out = torch.zeros((N, B * T), device="cuda")
for _ in range(30):
    r.reset(0); r.process_device(T, x.data_ptr(), B * T, out.data_ptr(), B * T)
ctx.synchronize()
buf = (ctypes.c_ulonglong * 64)()
capi.check(capi.load().earhip_debug_k2_prof(ctx.h, buf))
st = np.array(buf[:], dtype=np.uint64).reshape(2, 32).astype(np.int64)
for wg in range(2):
    s = st[wg]
    print(f"workgroup {'first' if wg == 0 else 'middle'}: table set-up {s[1]-s[0]} cycles")
    i = 2
    prev_end = s[1]
    pair = 0
    while i + 4 < 32 and s[i] > 0 and s[i + 4] >= s[i]:
        print(f"  pair {pair}: since previous phase {s[i]-prev_end:6d} | wait for loads {s[i+1]-s[i]:6d} | forward {s[i+2]-s[i+1]:6d} | x H {s[i+3]-s[i+2]:6d} | inverse {s[i+4]-s[i+3]:6d}")
        prev_end = s[i + 4]; i += 5; pair += 1
    if i < 32 and s[i] > 0: print(f"  after the last pair's stores: {s[i]-prev_end} cycles; whole wave {s[i]-s[0]} cycles")
