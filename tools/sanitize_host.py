import ctypes as C, numpy as np, sys, os
lib = C.CDLL(os.path.join(os.environ["EARHIP_SANITIZE_DIR"], "libhost.so"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import layouts
# decorrelator design on every layout
lib.earhip_decorrelator_size.restype = C.c_int
n = lib.earhip_decorrelator_size()
for name in ("0+5+0", "4+9+0", "9+10+3"):
    ch = layouts.without_lfe(layouts.LAYOUTS[name])
    names = (C.c_char_p * len(ch))(*[c.encode() for c in ch])
    out = np.zeros((len(ch), n), np.float32)
    rc = lib.earhip_design_decorrelators(C.c_int(len(ch)), names, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0, rc
    print(name, out.shape, float(np.abs(out).sum()))
# VBS adapter with ragged call sizes
CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)), C.c_void_p)
B, nin, nout = 64, 3, 2
def cb(ins, outs, user):
    for o in range(nout):
        for i in range(B):
            outs[o][i] = ins[o][i] + ins[o + 1][i]
    return 0
cbk = CB(cb)
h = C.c_void_p()
rc = lib.earhip_vbs_create(C.c_size_t(B), C.c_size_t(nin), C.c_size_t(nout), cbk, None, C.byref(h))
assert rc == 0, rc
rng = np.random.default_rng(0)
total = 1000
x = rng.standard_normal((nin, total)).astype(np.float32)
y = np.zeros((nout, total), np.float32)
pos = 0
fp = C.POINTER(C.c_float)
while pos < total:
    n = min(int(rng.integers(0, 150)), total - pos)
    ins = (fp * nin)(*[C.cast(x[i, pos:].ctypes.data, fp) for i in range(nin)])
    outs = (fp * nout)(*[C.cast(y[i, pos:].ctypes.data, fp) for i in range(nout)])
    rc = lib.earhip_vbs_process(h, C.c_size_t(n), ins, outs)
    assert rc == 0
    pos += n
ref = np.zeros_like(y); ref[:, B:] = (x[:2] + x[1:3])[:, :total - B]
print("vbs max err", np.abs(y - ref).max())
lib.earhip_vbs_destroy(h)
