"""Average shader clock while the bench workload runs back to back (DVFS check)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, scenes
from layouts import LAYOUTS
from libear_amd import capi
names = LAYOUTS["9+10+3"]; M, N, B, T = 1024, 24, 512, 1024
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
r = capi.Renderer(ctx, M, N, B, capi.design_decorrelators(names), 255, max_blocks=T)
for m, (t, d, f) in enumerate(scenes.dense_curves(M, N, B, T)): r.set_object_points(m, t, d, f)
x = torch.rand((M, B * T), device=dev) * 2 - 1; out = torch.zeros((N, B * T), device=dev)
probes = torch.zeros((2, 2), dtype=torch.int64, device=dev)
lib = capi.load()
def run(n):
    for _ in range(n):
        r.reset(0); r.process_device(T, x.data_ptr(), B * T, out.data_ptr(), B * T)
run(5); torch.cuda.synchronize()
for steps in (1, 10, 100, 300):
    lib.earhip_debug_clock_probe(ctx.h, C.c_void_p(probes[0].data_ptr()))
    run(steps)
    lib.earhip_debug_clock_probe(ctx.h, C.c_void_p(probes[1].data_ptr()))
    torch.cuda.synchronize()
    p = probes.cpu().numpy()
    dcyc, dwall = int(p[1, 0] - p[0, 0]), int(p[1, 1] - p[0, 1])
    print(f"steps={steps}: shader cycles {dcyc}, wall ticks {dwall}, ratio {dcyc / dwall:.3f}, "
          f"ms/step at 100 MHz wall {dwall / 1e5 / steps:.4f}")
