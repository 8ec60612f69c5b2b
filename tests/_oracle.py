"""ctypes bindings for the CPU oracle (oracle/ear_oracle.hpp through oracle/oracle_capi.cpp).

Test infrastructure only.  The oracle is the checker; nothing under libear_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i64p = C.POINTER(C.c_int64)
szp = C.POINTER(C.c_size_t)


def _build():
    subprocess.run(["make", "-s", "-C", ODIR], check=True, stdout=subprocess.DEVNULL)


def load(native=False):
    name = "liboracle_native.so" if native else "liboracle.so"
    path = os.path.join(ODIR, "_build", name)
    src_newer = (not os.path.exists(path)) or any(
        os.path.getmtime(os.path.join(ODIR, f)) > os.path.getmtime(path)
        for f in ("ear_oracle.hpp", "oracle_capi.cpp", "panner_oracle.hpp", "extent_oracle.hpp", "bs2051_data.h"))
    if src_newer:
        _build()
    lib = C.CDLL(path)
    lib.oracle_last_error.restype = C.c_char_p
    for fn in ("oracle_conv_ctx_create", "oracle_conv_filter_create", "oracle_delay_create",
               "oracle_vbs_create", "oracle_render_create"):
        getattr(lib, fn).restype = C.c_void_p
    lib.oracle_conv_filter_num_blocks.restype = C.c_size_t
    return lib


def load_ref():
    """The reference's own kissfft compiled in place (oracle/_ref); None when absent."""
    path = os.path.join(ODIR, "_ref", "libref_kiss.so")
    if not os.path.exists(path):
        if os.path.exists("/root/reference/submodules/kissfft/kissfft.hh"):
            _build()
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


class OracleError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def _chk(rc):
    if rc != 0:
        raise OracleError(rc, lib().oracle_last_error().decode())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a, t=f32p):
    return a.ctypes.data_as(t)


KIND = {"single": 0, "vector": 1, "matrix": 2}


def gain_interp(kind, times, values, x, call_sizes, t0=0):
    """values [npoints][n_in][n_out]; x [n_in][total] -> [n_out][total]."""
    values = _f32(values)
    npts, n_in, n_out = values.shape
    x = _f32(x).reshape(n_in, -1)
    total = x.shape[1]
    assert sum(call_sizes) == total
    out = np.zeros((n_out, total), np.float32)
    t = np.ascontiguousarray(times, dtype=np.int64)
    cs = np.ascontiguousarray(call_sizes, dtype=np.uintp)
    _chk(lib().oracle_gain_interp(KIND[kind], n_in, n_out, npts, ptr(t, i64p), ptr(values),
                                  C.c_int64(t0), ptr(cs, szp), len(cs), ptr(x), ptr(out)))
    return out


def cfft(x, inverse=False, double=False):
    dt = np.complex128 if double else np.complex64
    x = np.ascontiguousarray(x, dtype=dt)
    out = np.empty_like(x)
    fn = lib().oracle_cfft_f64 if double else lib().oracle_cfft_f32
    p = f64p if double else f32p
    _chk(fn(C.c_size_t(x.size), int(inverse), ptr(x, p), ptr(out, p)))
    return out


def rfft(x):
    x = _f32(x)
    out = np.empty(x.size // 2 + 1, np.complex64)
    _chk(lib().oracle_rfft_forward(C.c_size_t(x.size), ptr(x), ptr(out)))
    return out


def irfft_unnorm(X, n_fft):
    X = np.ascontiguousarray(X, dtype=np.complex64)
    out = np.empty(n_fft, np.float32)
    _chk(lib().oracle_rfft_reverse(C.c_size_t(n_fft), ptr(X), ptr(out)))
    return out


class ConvCtx:
    def __init__(self, block_size):
        self.block_size = block_size
        self.h = C.c_void_p(lib().oracle_conv_ctx_create(C.c_size_t(block_size)))


class ConvFilter:
    def __init__(self, ctx, taps):
        taps = _f32(taps)
        self.ctx = ctx
        self.h = C.c_void_p(lib().oracle_conv_filter_create(ctx.h, C.c_size_t(taps.size), ptr(taps)))

    def num_blocks(self):
        return lib().oracle_conv_filter_num_blocks(self.h)

    def spectrum(self, block):
        out = np.empty(self.ctx.block_size + 1, np.complex64)
        lib().oracle_conv_filter_spectrum(self.h, C.c_size_t(block), ptr(out))
        return out


class BlockConvolver:
    def __init__(self, ctx, filt=None, num_blocks=0):
        self.ctx = ctx
        h = C.c_void_p()
        _chk(lib().oracle_conv_create(ctx.h, filt.h if filt else None, C.c_size_t(num_blocks), C.byref(h)))
        self.h = h

    def set_filter(self, f):
        _chk(lib().oracle_conv_set_filter(self.h, f.h if f else None))

    def crossfade_filter(self, f):
        _chk(lib().oracle_conv_crossfade_filter(self.h, f.h if f else None))

    def fade_down(self):
        self.crossfade_filter(None)

    def unset_filter(self):
        self.set_filter(None)

    def process(self, x):
        out = np.empty(self.ctx.block_size, np.float32)
        if x is None:
            _chk(lib().oracle_conv_process(self.h, None, ptr(out)))
        else:
            x = _f32(x)
            assert x.size == self.ctx.block_size
            _chk(lib().oracle_conv_process(self.h, ptr(x), ptr(out)))
        return out


class DelayBuffer:
    def __init__(self, nch, delay):
        self.nch = nch
        self.h = C.c_void_p(lib().oracle_delay_create(C.c_size_t(nch), C.c_size_t(delay)))

    def process(self, x):
        x = _f32(x).reshape(self.nch, -1)
        out = np.empty_like(x)
        n = x.shape[1]
        _chk(lib().oracle_delay_process(self.h, C.c_size_t(self.nch), C.c_size_t(n), ptr(x), ptr(out),
                                        C.c_size_t(n), C.c_size_t(0)))
        return out


PROCESS_CB = C.CFUNCTYPE(None, C.POINTER(f32p), C.POINTER(f32p), C.c_void_p)


class VariableBlockSizeAdapter:
    def __init__(self, block_size, n_in, n_out, fn):
        """fn(in [n_in][B] ndarray) -> out [n_out][B] ndarray."""
        self.B, self.n_in, self.n_out = block_size, n_in, n_out

        def cb(inp, outp, _user):
            x = np.stack([np.ctypeslib.as_array(inp[c], (block_size,)) for c in range(n_in)])
            y = _f32(fn(x))
            for c in range(n_out):
                np.ctypeslib.as_array(outp[c], (block_size,))[:] = y[c]

        self._cb = PROCESS_CB(cb)
        self.h = C.c_void_p(lib().oracle_vbs_create(C.c_size_t(block_size), C.c_size_t(n_in),
                                                    C.c_size_t(n_out), self._cb, None))

    def get_delay(self):
        return lib().oracle_vbs_get_delay(self.h)

    def process(self, x):
        x = _f32(x).reshape(self.n_in, -1)
        n = x.shape[1]
        out = np.empty((self.n_out, n), np.float32)
        _chk(lib().oracle_vbs_process(self.h, C.c_size_t(self.n_in), C.c_size_t(self.n_out), C.c_size_t(n),
                                      ptr(x), ptr(out), C.c_size_t(max(n, 1)), C.c_size_t(0)))
        return out


def design_decorrelator_basic(dec_id, size=512):
    out = np.empty(size, np.float64)
    _chk(lib().oracle_design_decorrelator_basic(dec_id, size, ptr(out, f64p)))
    return out


def design_decorrelators(names):
    out = np.empty((len(names), 512), np.float32)
    _chk(lib().oracle_design_decorrelators("\n".join(names).encode(), ptr(out)))
    return out


def compensation_delay():
    return lib().oracle_decorrelator_compensation_delay()


class ObjectsRenderer:
    """Composed Objects render block (docs/dsp.rst:40-71), CPU oracle."""

    def __init__(self, n_obj, n_out, block, filters, delay, native=False):
        self.lib = load(native) if native else lib()
        filters = _f32(filters)
        self.M, self.N, self.B = n_obj, n_out, block
        self.lib.oracle_render_create.restype = C.c_void_p
        self.h = C.c_void_p(self.lib.oracle_render_create(
            C.c_size_t(n_obj), C.c_size_t(n_out), C.c_size_t(block), ptr(filters),
            C.c_size_t(filters.shape[1]), C.c_size_t(delay)))

    def set_points(self, obj, bus, times, gains):
        t = np.ascontiguousarray(times, dtype=np.int64)
        g = _f32(gains).reshape(len(t), self.N)
        rc = self.lib.oracle_render_set_points(self.h, C.c_size_t(obj), bus, len(t), ptr(t, i64p), ptr(g),
                                               C.c_size_t(self.N))
        assert rc == 0

    def process(self, x):
        x = _f32(x).reshape(self.M, -1)
        nblocks = x.shape[1] // self.B
        assert nblocks * self.B == x.shape[1]
        out = np.empty((self.N, x.shape[1]), np.float32)
        rc = self.lib.oracle_render_process(self.h, C.c_size_t(self.M), C.c_size_t(self.N), C.c_size_t(self.B),
                                            C.c_size_t(nblocks), ptr(x), ptr(out))
        assert rc == 0
        return out

    def __del__(self):
        try:
            self.lib.oracle_render_destroy(self.h)
        except Exception:
            pass


def cart(az, el, dist=1.0):
    """libear's polar -> Cartesian convention (src/common/geom.cpp:82-87)"""
    az, el = np.radians(-np.asarray(az, np.float64)), np.radians(np.asarray(el, np.float64))
    return np.stack([np.sin(az) * np.cos(el) * dist, np.cos(az) * np.cos(el) * dist, np.sin(el) * dist + 0 * az], axis=-1)


def _panner_handle(layout, positions):
    """oracle_panner_create[_positions]: positions = (azimuths, elevations) of every channel of the full layout"""
    L = lib()
    L.oracle_panner_create.restype = C.c_void_p
    L.oracle_panner_create_positions.restype = C.c_void_p
    if positions is None:
        h = C.c_void_p(L.oracle_panner_create(layout.encode()))
        if not h:
            raise OracleError(1, L.oracle_last_error().decode())
        return h
    az = np.ascontiguousarray(positions[0], np.float64)
    el = np.ascontiguousarray(positions[1], np.float64)
    st = C.c_int(0)
    h = C.c_void_p(L.oracle_panner_create_positions(layout.encode(), ptr(az, f64p), ptr(el, f64p), C.byref(st)))
    if st.value:
        raise OracleError(st.value, L.oracle_last_error().decode())
    return h


class GainCalculatorObjects:
    """Objects gain producer without extent parameters (oracle/panner_oracle.hpp + extent_oracle.hpp): point-source
    pan (widened at distances under 1, as libear's PolarExtent does), LFE mask, diffuse split."""

    def __init__(self, layout, positions=None):
        self.h = _panner_handle(layout, positions)
        self.n_out = lib().oracle_panner_n_out(self.h)
        self.n_psp = lib().oracle_psp_n_out(self.h)

    def calculate(self, az, el, dist=None, gain=None, diffuse=None):
        """arrays [n] -> (direct, diffuse) float32 [n][n_out]"""
        az = np.ascontiguousarray(np.atleast_1d(az), np.float64)
        n = az.size
        el = np.ascontiguousarray(np.broadcast_to(np.asarray(el, np.float64), (n,)))
        dist = np.ascontiguousarray(np.broadcast_to(np.asarray(1.0 if dist is None else dist, np.float64), (n,)))
        gain = np.ascontiguousarray(np.broadcast_to(np.asarray(1.0 if gain is None else gain, np.float64), (n,)))
        diffuse = np.ascontiguousarray(np.broadcast_to(np.asarray(0.0 if diffuse is None else diffuse, np.float64), (n,)))
        d = np.zeros((n, self.n_out), np.float32)
        f = np.zeros((n, self.n_out), np.float32)
        missed = lib().oracle_panner_calculate(self.h, C.c_size_t(n), ptr(az, f64p), ptr(el, f64p), ptr(dist, f64p),
                                               ptr(gain, f64p), ptr(diffuse, f64p), ptr(d), ptr(f))
        assert missed == 0, f"{missed} positions not handled by any region"
        return d, f

    def psp(self, xyz):
        """the inner point source panner on Cartesian positions [n][3] -> (pv [n][channels without LFE], missed)"""
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        pv = np.zeros((xyz.shape[0], self.n_psp), np.float64)
        missed = lib().oracle_psp_handle(self.h, C.c_size_t(xyz.shape[0]), ptr(xyz, f64p), ptr(pv, f64p))
        return pv, missed

    def __del__(self):
        try:
            lib().oracle_panner_destroy(self.h)
        except Exception:
            pass


def _f64(x, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.float64), (n,)))


class PolarExtent:
    """libear's polar extent panner behind GainCalculatorObjects (oracle/extent_oracle.hpp): the library's
    form (float core, `which=0`) and the form libear's tests keep beside it (double, `which=1`)."""

    def __init__(self, layout, positions=None):
        L = lib()
        L.oracle_extent_weight.restype = C.c_double
        L.oracle_extent_mod.restype = C.c_double
        self.h = _panner_handle(layout, positions)
        self.n_out = L.oracle_panner_n_out(self.h)
        self.n_psp = L.oracle_psp_n_out(self.h)
        self.num_points = L.oracle_extent_num_points(self.h)

    def grid(self):
        xyz = np.zeros((self.num_points, 3), np.float64)
        lib().oracle_extent_grid(ptr(xyz, f64p))
        return xyz

    def handle(self, xyz, width, height, depth=0.0, which=0):
        """Cartesian positions [n][3], extents in degrees -> pv [n][channels without LFE] float64"""
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        n = xyz.shape[0]
        w, h, d = _f64(width, n), _f64(height, n), _f64(depth, n)
        pv = np.zeros((n, self.n_psp), np.float64)
        missed = lib().oracle_extent_handle(self.h, which, C.c_size_t(n), ptr(xyz, f64p), ptr(w, f64p), ptr(h, f64p),
                                            ptr(d, f64p), ptr(pv, f64p))
        assert missed == 0, f"{missed} positions not handled by any region"
        return pv

    def weight(self, centre, width, height, point, which=0):
        centre = np.ascontiguousarray(centre, np.float64)
        point = np.ascontiguousarray(point, np.float64)
        return lib().oracle_extent_weight(self.h, which, ptr(centre, f64p), C.c_double(width), C.c_double(height),
                                          ptr(point, f64p))

    def calculate(self, az, el, dist=None, width=0.0, height=0.0, depth=0.0, gain=None, diffuse=None):
        """GainCalculatorObjects::calculate with extent: arrays [n] -> (direct, diffuse) float32 [n][n_out]"""
        az = np.ascontiguousarray(np.atleast_1d(az), np.float64)
        n = az.size
        a = [_f64(v, n) for v in (el, 1.0 if dist is None else dist, width, height, depth,
                                  1.0 if gain is None else gain, 0.0 if diffuse is None else diffuse)]
        d = np.zeros((n, self.n_out), np.float32)
        f = np.zeros((n, self.n_out), np.float32)
        missed = lib().oracle_extent_calculate(self.h, C.c_size_t(n), ptr(az, f64p), *[ptr(v, f64p) for v in a], ptr(d), ptr(f))
        assert missed == 0, f"{missed} positions not handled by any region"
        return d, f

    def __del__(self):
        try:
            lib().oracle_panner_destroy(self.h)
        except Exception:
            pass


def extent_calc_basis(xyz):
    xyz = np.ascontiguousarray(xyz, np.float64)
    m = np.zeros((3, 3), np.float64)
    lib().oracle_extent_calc_basis(ptr(xyz, f64p), ptr(m, f64p))
    return m


def extent_mod(extent, distance):
    lib().oracle_extent_mod.restype = C.c_double
    return lib().oracle_extent_mod(C.c_double(extent), C.c_double(distance))


def extra_pos_vertical_nominal(layout):
    az, el = np.zeros(32), np.zeros(32)
    idx = np.zeros(32, np.int32)
    n = lib().oracle_extra_pos_vertical_nominal(layout.encode(), ptr(az, f64p), ptr(el, f64p), idx.ctypes.data_as(C.POINTER(C.c_int)))
    assert n >= 0, lib().oracle_last_error().decode()
    return list(zip(az[:n], el[:n])), list(idx[:n])


def region_handle(kind, positions, xyz, centre=None, downmix=None):
    """kind: 'triplet' | 'quad' | 'ngon'; returns pv [n] or None"""
    positions = np.ascontiguousarray(positions, np.float64)
    n = positions.shape[0]
    xyz = np.ascontiguousarray(xyz, np.float64)
    centre = np.ascontiguousarray(np.zeros(3) if centre is None else centre, np.float64)
    downmix = np.ascontiguousarray(np.zeros(n) if downmix is None else downmix, np.float64)
    pv = np.zeros(n, np.float64)
    ok = lib().oracle_region_handle({"triplet": 0, "quad": 1, "ngon": 2}[kind], n, ptr(positions, f64p), ptr(centre, f64p),
                                    ptr(downmix, f64p), ptr(xyz, f64p), ptr(pv, f64p))
    return pv if ok else None


def stereo_downmix_handle(xyz):
    xyz = np.ascontiguousarray(xyz, np.float64)
    pv = np.zeros(2, np.float64)
    return pv if lib().oracle_stereo_downmix_handle(ptr(xyz, f64p), ptr(pv, f64p)) else None


def hoa_decode_matrix(layout, orders, degrees, normalization="SN3D", positions=None):
    """AllRAD decode matrix of libear's GainCalculatorHOA (oracle/panner_oracle.hpp, hoa_oracle):
    [n_channels][n_coef] float64, LFE rows zero"""
    o = np.ascontiguousarray(orders, np.int32)
    d = np.ascontiguousarray(degrees, np.int32)
    if len(o) != len(d):
        raise OracleError(1, "orders and degrees must be the same size")
    out = np.zeros((64, max(len(o), 1)), np.float64)
    nch = C.c_int(0)
    ip = C.POINTER(C.c_int)
    az = el = None
    if positions is not None:
        az, el = (np.ascontiguousarray(v, np.float64) for v in positions)
    rc = lib().oracle_hoa_decode_matrix_positions(layout.encode(), None if az is None else ptr(az, f64p),
                                                  None if el is None else ptr(el, f64p), len(o), o.ctypes.data_as(ip),
                                                  d.ctypes.data_as(ip), normalization.encode(), ptr(out, f64p), C.byref(nch))
    if rc:
        raise OracleError(rc, lib().oracle_last_error().decode())
    return out.reshape(-1)[:nch.value * len(o)].reshape(nch.value, len(o)).copy()


def sph_harm(n, m, az, el, norm="N3D"):
    lib().oracle_sph_harm.restype = C.c_double
    return lib().oracle_sph_harm(int(n), int(m), C.c_double(az), C.c_double(el), norm.encode())


def tdesign_points():
    xyz = np.zeros((5200, 3), np.float64)
    lib().oracle_tdesign_points(ptr(xyz, f64p))
    return xyz
