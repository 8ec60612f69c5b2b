"""Where the waves of the hinge kernel spend their chunk loop (a library built with -DEARHIP_HG_PROF, EARHIP_LIB pointing at it):
python tools/hg_phases.py [moving|bursty-moving]  -> per wave of the grid's first and a middle workgroup the cycles summed per phase."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
from layouts import LAYOUTS
from libear_amd import capi

kind = sys.argv[1] if len(sys.argv) > 1 else "moving"
M, T, B = 1024, 1024, 512
names = LAYOUTS["9+10+3"]; N = len(names)
dec = capi.design_decorrelators(names)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=T)
curves = scenes.adm_curves(M, N, B * T, period=240, ramp=240, seed=12)
for m, (t, d, f) in enumerate(curves):
    r.set_object_points(m, t, d, f)
r.commit()
x = torch.rand((M, B * T), device="cuda") * 2 - 1
if kind == "bursty-moving":
    lv = torch.as_tensor(scenes.bursty_levels(M, T, seed=5), device="cuda", dtype=torch.float32)
    x = (x.view(M, T, B) * lv[:, :, None]).reshape(M, B * T).contiguous()
out = torch.zeros((N, B * T), device="cuda")
for _ in range(20):
    r.reset(0); r.process_device(T, x.data_ptr(), B * T, out.data_ptr(), B * T)
ctx.synchronize()
print(kind, "plan", r.last_plan(), "robust", r.hinge_robust())
buf = (ctypes.c_ulonglong * 128)()
capi.check(capi.load().earhip_debug_hg_prof(ctx.h, buf))
st = np.array(buf[:], dtype=np.uint64).reshape(2, 8, 8).astype(np.int64)
names_ = ["barrier", "rows+split", "line", "stage+ring", "kinks"]
for wg in range(2):
    print(f"workgroup {'first' if wg == 0 else 'middle'}: cycles per chunk and wave (s_memtime counts at 100 MHz x ? — compare the columns)")
    print("  wave  chunks  kink sets/chunk  " + "  ".join(f"{n:>11s}" for n in names_) + "        sum   whole kernel")
    for w in range(8):
        s = st[wg, w]
        nc = max(int(s[5]), 1)
        print(f"  {w:4d}  {int(s[5]):6d}  {s[6] / nc:15.2f}  " + "  ".join(f"{s[i] / nc:11.1f}" for i in range(5)) + f"  {s[:5].sum() / nc:9.1f}  {int(s[7]):12d}")
