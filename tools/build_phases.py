"""Phase times inside the list builders (a library built with -DEARHIP_BUILD_PROF, EARHIP_LIB pointing at it):
python tools/build_phases.py adm|moving  -> the stamps of thread 0 of the grid's first and a middle workgroup."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
from layouts import LAYOUTS
from libear_amd import capi

kind = sys.argv[1] if len(sys.argv) > 1 else "adm"
M, T, B = 1024, 1024, 512
names = LAYOUTS["9+10+3"]; N = len(names)
dec = capi.design_decorrelators(names)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=T)
curves = scenes.adm_curves(M, N, B * T, seed=11) if kind == "adm" else scenes.adm_curves(M, N, B * T, period=240, ramp=240, seed=12)
for m, (t, d, f) in enumerate(curves):
    r.set_object_points(m, t, d, f)
r.commit()
x = torch.rand((M, B * T), device="cuda") * 2 - 1
out = torch.zeros((N, B * T), device="cuda")
for _ in range(20):
    r.reset(0); r.process_device(T, x.data_ptr(), B * T, out.data_ptr(), B * T)
ctx.synchronize()
print(kind, "plan", r.last_plan())
buf = (ctypes.c_ulonglong * 64)()
capi.check(capi.load().earhip_debug_build_prof(ctx.h, buf))
st = np.array(buf[:], dtype=np.uint64).reshape(2, 32).astype(np.int64)
for wg in range(2):
    s = st[wg]
    idx = [i for i in range(32) if s[i] > 0]
    print(f"workgroup {'first' if wg == 0 else 'middle'}: marks (index: cycles since the one before)")
    prev = s[idx[0]]
    print("  " + "  ".join(f"{i}:{s[i]-prev if k == 0 else s[i]-s[idx[k-1]]}" for k, i in enumerate(idx)))
    print(f"  whole: {s[idx[-1]] - s[idx[0]]} cycles")
