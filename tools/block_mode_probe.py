"""debug aid: phases of one block-mode call (EARHIP_DEBUG_TIMING) at the headline shape"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes
from layouts import LAYOUTS
from libear_amd import capi
names = LAYOUTS["9+10+3"]
M, N, B = int(os.environ.get("M", "1024")), len(names), 512
c = capi.Context(0)
dec = capi.design_decorrelators(names)
r = capi.Renderer(c, M, N, B, dec, 255, max_blocks=1)
for m, (t, d, f) in enumerate(scenes.dense_curves(M, N, B, 32)):
    r.set_object_points(m, t, d, f)
x = scenes.audio(M, B)
ob = np.zeros((N, B), np.float32)
ip, op = capi._chan_ptrs(x), capi._chan_ptrs(ob)
lib = capi.load()
for i in range(30):
    if i == 24:
        c.set_option("DEBUG_TIMING", 1)
    capi.check(lib.earhip_render_process(r.h, ctypes.c_size_t(1), ip, op))
