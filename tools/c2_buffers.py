"""Config 2's two modes (K1 0.226 / 0.246 ms per process): is the mode a property of the PROCESS or of the BUFFER?  One
process, several input and output buffers allocated one after the other, K1 timed on every (input, output) pair."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import scenes  # noqa: E402
from layouts import LAYOUTS  # noqa: E402
from libear_amd import capi  # noqa: E402

names = LAYOUTS["4+5+0"]
M, N, B, T = 64, len(names), 512, 8192
total = B * T
pad = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
r = capi.Renderer(ctx, M, N, B, None, 0, max_blocks=T)
for m, (t, d, f) in enumerate(scenes.dense_curves(M, N, B, T, seed=7)):
    r.set_object_points(m, t, d, None)
r.commit()
stride = total + pad
ins = []
for i in range(nbuf):
    x = torch.rand((M, stride), device=dev, dtype=torch.float32) * 2 - 1
    ins.append(x)
    if i % 2 == 0:  # (something of another size in between: the next buffer lands elsewhere)
        ins.append(None)
        _spacer = torch.empty((37 * (i + 1), 1 << 20), device=dev, dtype=torch.uint8)
outs = [torch.zeros((N, total), device=dev, dtype=torch.float32) for _ in range(2)]
ins = [x for x in ins if x is not None]
for xi, x in enumerate(ins):
    for oi, o in enumerate(outs):
        for _ in range(20):
            r.reset(0)
            r.process_device(T, x.data_ptr(), stride, o.data_ptr(), total)
        r.enable_timing(1)
        for _ in range(40):
            r.reset(0)
            r.process_device(T, x.data_ptr(), stride, o.data_ptr(), total)
        tm = r.get_timing()
        r.enable_timing(False)
        print(f"input {xi} {hex(x.data_ptr())} output {oi} {hex(o.data_ptr())}: K1 {tm['gain_mix_ms'] / tm['gain_mix_launches']:.4f} ms")
