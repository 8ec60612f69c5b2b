"""ctypes bindings of the libearhip C ABI (include/earhip.h).

Thin and explicit on purpose: every call goes through the C ABI exactly as a foreign-language
binding would, so the GPU parity tests exercise the drop-in boundary itself.  Arrays are numpy
float32, planar ``[channels][samples]``.
"""
import ctypes as C
import os

import numpy as np

from .build import lib_path

f32p = C.POINTER(C.c_float)
i64p = C.POINTER(C.c_int64)
pp_f32 = C.POINTER(f32p)

OK, INVALID_ARGUMENT, INTERNAL_ERROR, DEVICE_ERROR, NOT_IMPLEMENTED, UNKNOWN_LAYOUT, ADM_ERROR = 0, 1, 2, 3, 4, 5, 6


class EarHipError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class InvalidArgument(EarHipError, ValueError):
    """maps to ear::invalid_argument"""


class InternalError(EarHipError, RuntimeError):
    """maps to ear::internal_error"""


class UnknownLayout(InvalidArgument):
    """maps to ear::unknown_layout"""


class AdmError(InvalidArgument):
    """maps to ear::adm_error"""


class NotImplementedInLibear(EarHipError, RuntimeError):
    """maps to ear::not_implemented: a case libear itself refuses"""


class RenderConfig(C.Structure):
    _fields_ = [("n_objects", C.c_int), ("n_out", C.c_int), ("block_size", C.c_int),
                ("n_buses", C.c_int), ("decorrelators", f32p), ("n_taps", C.c_int),
                ("delay", C.c_int), ("max_blocks", C.c_int)]


PROCESS_FUNC = C.CFUNCTYPE(C.c_int, pp_f32, pp_f32, C.c_void_p)

_lib = None


def load():
    """Loads libearhip.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(libear_amd has no CPU fallback)")
    lib = C.CDLL(path)
    lib.earhip_last_error.restype = C.c_char_p
    lib.earhip_conv_filter_num_blocks.restype = C.c_size_t
    lib.earhip_conv_filter_num_blocks.argtypes = [C.c_void_p]
    lib.earhip_delay_get_delay.argtypes = [C.c_void_p]
    lib.earhip_vbs_get_delay.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def check(rc):
    if rc == OK:
        return
    msg = load().earhip_last_error().decode()
    if rc == INVALID_ARGUMENT:
        raise InvalidArgument(rc, msg)
    if rc == UNKNOWN_LAYOUT:
        raise UnknownLayout(rc, msg)
    if rc == ADM_ERROR:
        raise AdmError(rc, msg)
    if rc == NOT_IMPLEMENTED:
        raise NotImplementedInLibear(rc, msg)
    raise InternalError(rc, msg)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t=f32p):
    return a.ctypes.data_as(t)


def _chan_ptrs(a, offset=0):
    """float*[] over the rows of a C-contiguous [channels][samples] float32 array."""
    n = a.shape[0]
    arr = (f32p * max(n, 1))()
    base = a.ctypes.data
    for c in range(n):
        arr[c] = C.cast(base + (c * a.shape[1] + offset) * 4, f32p)
    return arr


def device_count():
    n = C.c_int(0)
    check(load().earhip_device_count(C.byref(n)))
    return n.value


class Context:
    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        check(load().earhip_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self.h)))

    def set_strict(self, strict):
        check(load().earhip_ctx_set_strict(self.h, int(bool(strict))))

    def synchronize(self):
        check(load().earhip_ctx_synchronize(self.h))

    def set_option(self, key, value=None):
        """a tuning knob of this context (include/earhip.h: earhip_ctx_set_option); value None: back to the default"""
        check(load().earhip_ctx_set_option(self.h, key.encode(), None if value is None else str(value).encode()))

    def get_option(self, key):
        """None when the option is at its default, else its integer value"""
        is_set, v = C.c_int(0), C.c_int(0)
        check(load().earhip_ctx_get_option(self.h, key.encode(), C.byref(is_set), C.byref(v)))
        return v.value if is_set.value else None

    def read_bandwidth(self, dev_ptr, rows, stride, nsamples, reps=5):
        """(ms linear stream over rows*stride floats, ms gain-stage row pattern over rows*nsamples floats)"""
        ms = (C.c_double * 2)()
        check(load().earhip_debug_read_bandwidth(self.h, C.c_void_p(dev_ptr), C.c_size_t(rows), C.c_size_t(stride),
                                                 C.c_size_t(nsamples), int(reps), ms))
        return ms[0], ms[1]

    def copy_bandwidth(self, host_array, reps=3):
        """(ms host -> device, ms device -> host) of a plain copy of the array's bytes (it is overwritten with its own
        contents by the second direction)"""
        ms = (C.c_double * 2)()
        check(load().earhip_debug_copy_bandwidth(self.h, C.c_void_p(host_array.ctypes.data), C.c_size_t(host_array.nbytes), int(reps), ms))
        return ms[0], ms[1]

    # --- host memory the device reaches directly (earhip_host_alloc / _register / _release)
    def pinned_array(self, shape):
        """float32 array in pinned, device-reachable host memory (C-contiguous: the rows of a [channels][samples]
        array are evenly spaced channel buffers).  Valid until close() or release(array)."""
        shape = tuple(int(v) for v in np.atleast_1d(shape))
        count = int(np.prod(shape))
        p = C.c_void_p()
        check(load().earhip_host_alloc(self.h, C.c_size_t(4 * max(count, 1)), C.byref(p)))
        buf = (C.c_float * max(count, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=np.float32, count=count).reshape(shape)
        a[...] = 0.0
        return a

    def register(self, a):
        """make the memory of a C-contiguous float32 array reachable by the device (hipHostRegister)"""
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        check(load().earhip_host_register(self.h, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)))

    def release(self, a):
        check(load().earhip_host_release(self.h, C.c_void_p(a.ctypes.data)))

    def close(self):
        if self.h:
            load().earhip_ctx_destroy(self.h)
            self.h = C.c_void_p()

    # --- (A) interpolation policies -------------------------------------------------
    def apply_interp(self, x, out, range_start, range_end, block_start, start, end, start_point, end_point):
        """x [n_in][n], out [n_out][n] (written in place on [range_start, range_end))."""
        sp, ep = _f32(start_point), _f32(end_point)
        n_in, n_out = x.shape[0], out.shape[0]
        assert sp.size == n_in * n_out and ep.size == n_in * n_out
        check(load().earhip_interp_apply_interp(
            self.h, n_in, n_out, _chan_ptrs(x), _chan_ptrs(out), C.c_int64(range_start), C.c_int64(range_end),
            C.c_int64(block_start), C.c_int64(start), C.c_int64(end), _ptr(sp), _ptr(ep)))

    def apply_constant(self, x, out, range_start, range_end, point):
        pt = _f32(point)
        n_in, n_out = x.shape[0], out.shape[0]
        assert pt.size == n_in * n_out
        check(load().earhip_interp_apply_constant(
            self.h, n_in, n_out, _chan_ptrs(x), _chan_ptrs(out), C.c_int64(range_start), C.c_int64(range_end),
            _ptr(pt)))


class GainInterp:
    """(A') whole-curve GainInterpolator with device-resident points."""

    def __init__(self, ctx, n_in, n_out):
        self.ctx, self.n_in, self.n_out = ctx, n_in, n_out
        self.h = C.c_void_p()
        check(load().earhip_gain_interp_create(ctx.h, n_in, n_out, C.byref(self.h)))

    def set_points(self, times, values):
        t = np.ascontiguousarray(times, dtype=np.int64)
        v = _f32(values).reshape(len(t), self.n_in, self.n_out) if len(t) else _f32(values)
        check(load().earhip_gain_interp_set_points(self.h, len(t), _ptr(t, i64p), _ptr(v)))

    def process(self, block_start, x):
        x = _f32(x).reshape(self.n_in, -1)
        out = np.zeros((self.n_out, x.shape[1]), np.float32)
        check(load().earhip_gain_interp_process(self.h, C.c_int64(block_start), C.c_size_t(x.shape[1]),
                                                _chan_ptrs(x), _chan_ptrs(out)))
        return out

    def process_device(self, block_start, nsamples, in_ptr, in_stride, out_ptr, out_stride):
        """device-resident planar buffers (16-byte aligned, strides multiples of 4); enqueues, does not synchronise"""
        check(load().earhip_gain_interp_process_device(self.h, C.c_int64(block_start), C.c_size_t(nsamples),
                                                       C.c_void_p(in_ptr), C.c_size_t(in_stride), C.c_void_p(out_ptr),
                                                       C.c_size_t(out_stride)))

    def close(self):
        if self.h:
            load().earhip_gain_interp_destroy(self.h)
            self.h = C.c_void_p()


class FFTPlan:
    def __init__(self, ctx, n_fft):
        self.n = n_fft
        self.h = C.c_void_p()
        check(load().earhip_fft_plan_create(ctx.h, C.c_size_t(n_fft), C.byref(self.h)))

    def forward(self, x):
        x = _f32(x)
        assert x.size == self.n
        out = np.empty(self.n // 2 + 1, np.complex64)
        check(load().earhip_fft_forward(self.h, _ptr(x), _ptr(out)))
        return out

    def reverse(self, X):
        X = np.ascontiguousarray(X, dtype=np.complex64)
        assert X.size == self.n // 2 + 1
        out = np.empty(self.n, np.float32)
        check(load().earhip_fft_reverse(self.h, _ptr(X), _ptr(out)))
        return out

    def close(self):
        if self.h:
            load().earhip_fft_plan_destroy(self.h)
            self.h = C.c_void_p()


class ConvCtx:
    def __init__(self, ctx, block_size):
        self.block_size = block_size
        self.h = C.c_void_p()
        check(load().earhip_conv_ctx_create(ctx.h, C.c_size_t(block_size), C.byref(self.h)))


class ConvFilter:
    def __init__(self, cctx, taps):
        taps = _f32(taps)
        self.h = C.c_void_p()
        check(load().earhip_conv_filter_create(cctx.h, C.c_size_t(taps.size), _ptr(taps), C.byref(self.h)))

    def num_blocks(self):
        return load().earhip_conv_filter_num_blocks(self.h)


class BlockConvolver:
    def __init__(self, cctx, filt=None, num_blocks=0):
        self.B = cctx.block_size
        self.h = C.c_void_p()
        check(load().earhip_conv_create(cctx.h, filt.h if filt else None, C.c_size_t(num_blocks),
                                        C.byref(self.h)))

    def set_filter(self, f):
        check(load().earhip_conv_set_filter(self.h, f.h if f else None))

    def crossfade_filter(self, f):
        check(load().earhip_conv_crossfade_filter(self.h, f.h if f else None))

    def fade_down(self):
        self.crossfade_filter(None)

    def unset_filter(self):
        self.set_filter(None)

    def process(self, x):
        out = np.empty(self.B, np.float32)
        if x is None:
            check(load().earhip_conv_process(self.h, None, _ptr(out)))
        else:
            x = _f32(x)
            assert x.size == self.B
            check(load().earhip_conv_process(self.h, _ptr(x), _ptr(out)))
        return out


class DelayBuffer:
    def __init__(self, ctx, nch, delay):
        self.nch = nch
        self.h = C.c_void_p()
        check(load().earhip_delay_create(ctx.h, C.c_size_t(nch), C.c_size_t(delay), C.byref(self.h)))

    def get_delay(self):
        return load().earhip_delay_get_delay(self.h)

    def process(self, x):
        x = _f32(x).reshape(self.nch, -1)
        out = np.empty_like(x)
        check(load().earhip_delay_process(self.h, C.c_size_t(x.shape[1]), _chan_ptrs(x), _chan_ptrs(out)))
        return out


class VariableBlockSizeAdapter:
    """Host-side adapter; fn(in [n_in][B]) -> out [n_out][B]."""

    def __init__(self, block_size, n_in, n_out, fn, ctx=None, raw=False):
        """ctx: keep the FIFO buffers in device-reachable host memory of that context (earhip_vbs_create_pinned);
        raw: fn(in_ptrs, out_ptrs) gets the adapter's own channel pointer arrays (to hand on to a C entry point)"""
        self.B, self.n_in, self.n_out = block_size, n_in, n_out
        self.error = None

        def cb(inp, outp, _user):
            try:
                if raw:
                    fn(inp, outp)
                    return OK
                x = np.stack([np.ctypeslib.as_array(inp[c], (block_size,)) for c in range(n_in)])
                y = _f32(fn(x))
                for c in range(n_out):
                    np.ctypeslib.as_array(outp[c], (block_size,))[:] = y[c]
                return OK
            except Exception as e:  # surfaced by process()
                self.error = e
                return INTERNAL_ERROR

        self._cb = PROCESS_FUNC(cb)
        self.h = C.c_void_p()
        if ctx is not None:
            check(load().earhip_vbs_create_pinned(ctx.h, C.c_size_t(block_size), C.c_size_t(n_in), C.c_size_t(n_out),
                                                  self._cb, None, C.byref(self.h)))
        else:
            check(load().earhip_vbs_create(C.c_size_t(block_size), C.c_size_t(n_in), C.c_size_t(n_out), self._cb,
                                           None, C.byref(self.h)))

    def close(self):
        if self.h:
            load().earhip_vbs_destroy(self.h)
            self.h = C.c_void_p()

    def get_delay(self):
        return load().earhip_vbs_get_delay(self.h)

    def process(self, x):
        x = _f32(x).reshape(self.n_in, -1)
        n = x.shape[1]
        out = np.empty((self.n_out, n), np.float32)
        xs = x if n else np.zeros((self.n_in, 1), np.float32)
        os_ = out if n else np.zeros((self.n_out, 1), np.float32)
        rc = load().earhip_vbs_process(self.h, C.c_size_t(n), _chan_ptrs(xs), _chan_ptrs(os_))
        if self.error is not None:
            e, self.error = self.error, None
            raise e
        check(rc)
        return out


def design_decorrelators(names):
    """(G) 512-tap decorrelator FIR per channel, [n][512] float32."""
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    out = np.empty((len(names), load().earhip_decorrelator_size()), np.float32)
    check(load().earhip_design_decorrelators(len(names), arr, _ptr(out)))
    return out


def layout_names():
    """(H) names of the BS.2051 layouts the library knows"""
    lib = load()
    lib.earhip_layout_name.restype = C.c_char_p
    return [lib.earhip_layout_name(i).decode() for i in range(lib.earhip_layout_count())]


def layout_channels(layout):
    """(H) [(name, azimuth, elevation, is_lfe)] of a BS.2051 layout, in layout order"""
    n = C.c_int(0)
    check(load().earhip_layout_num_channels(layout.encode(), C.byref(n)))
    out = []
    for i in range(n.value):
        name, az, el, lfe = C.c_char_p(), C.c_double(), C.c_double(), C.c_int()
        check(load().earhip_layout_channel(layout.encode(), i, C.byref(name), C.byref(az), C.byref(el), C.byref(lfe)))
        out.append((name.value.decode(), az.value, el.value, bool(lfe.value)))
    return out


def layout_channel_ranges(layout):
    """(H) [((az_lo, az_hi), (el_lo, el_hi))] per channel: where BS.2051 allows the real loudspeaker to stand"""
    n = C.c_int(0)
    check(load().earhip_layout_num_channels(layout.encode(), C.byref(n)))
    out = []
    for i in range(n.value):
        az, el = (C.c_double * 2)(), (C.c_double * 2)()
        check(load().earhip_layout_channel_ranges(layout.encode(), i, az, el))
        out.append(((az[0], az[1]), (el[0], el[1])))
    return out


def design_decorrelators_for_layout(layout, without_lfe=False):
    chans = [c for c in layout_channels(layout) if not (without_lfe and c[3])]
    out = np.empty((len(chans), load().earhip_decorrelator_size()), np.float32)
    check(load().earhip_design_decorrelators_for_layout(layout.encode(), int(without_lfe), _ptr(out)))
    return out


def design_decorrelator_basic(dec_id, size=512):
    out = np.empty(size, np.float64)
    check(load().earhip_design_decorrelator_basic(dec_id, size, _ptr(out, C.POINTER(C.c_double))))
    return out


def compensation_delay():
    return load().earhip_decorrelator_compensation_delay()


class Panner:
    """(I) gain-vector producer for Objects content (point-source pan, LFE mask, diffuse split), batched."""

    def __init__(self, ctx, layout, positions=None):
        """positions: (azimuths, elevations) in degrees of every channel of the full layout (LFE included): the
        loudspeakers' real positions; None: nominal"""
        self.h = C.c_void_p()
        if positions is None:
            check(load().earhip_panner_create(ctx.h, layout.encode(), C.byref(self.h)))
        else:
            f64 = C.POINTER(C.c_double)
            az, el = (np.ascontiguousarray(v, np.float64) for v in positions)
            assert az.size == el.size
            check(load().earhip_panner_create_positions(ctx.h, layout.encode(), int(az.size), _ptr(az, f64), _ptr(el, f64),
                                                        C.byref(self.h)))
        n = C.c_int(0)
        check(load().earhip_panner_num_channels(self.h, C.byref(n)))
        self.n_out = n.value

    def calculate(self, az, el, dist=None, gain=None, diffuse=None, width=None, height=None, depth=None):
        """arrays [n] (degrees) -> (direct, diffuse) float32 [n][n_out]; width / height / depth: the extent
        panner (all None: point sources)"""
        f64 = C.POINTER(C.c_double)
        az = np.ascontiguousarray(np.atleast_1d(az), np.float64)
        n = az.size

        def arr(v):
            return None if v is None else np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float64), (n,)))

        def opt(v):
            return None if v is None else _ptr(v, f64)
        el, dist, gain, diffuse = arr(el), arr(dist), arr(gain), arr(diffuse)
        width, height, depth = arr(width), arr(height), arr(depth)
        d = np.empty((n, self.n_out), np.float32)
        f = np.empty((n, self.n_out), np.float32)
        if width is None and height is None and depth is None:
            check(load().earhip_panner_calculate(self.h, C.c_size_t(n), _ptr(az, f64), _ptr(el, f64), opt(dist), opt(gain),
                                                 opt(diffuse), _ptr(d), _ptr(f)))
        else:
            check(load().earhip_panner_calculate_extent(self.h, C.c_size_t(n), _ptr(az, f64), _ptr(el, f64), opt(dist),
                                                        opt(width), opt(height), opt(depth), opt(gain), opt(diffuse),
                                                        _ptr(d), _ptr(f)))
        return d, f

    def calculate_device(self, n, az, el, dist, gain, diffuse, direct, diffuse_out, width=None, height=None, depth=None):
        """device addresses (ints; None = absent): enqueues on the context's stream, does not synchronise"""
        def vp(a):
            return None if a is None else C.c_void_p(int(a))
        check(load().earhip_panner_calculate_extent_device(self.h, C.c_size_t(n), vp(az), vp(el), vp(dist), vp(width), vp(height),
                                                           vp(depth), vp(gain), vp(diffuse), vp(direct), vp(diffuse_out)))

    def missed(self):
        """positions of the last calculate_device call that no region of the layout took (synchronises)"""
        n = C.c_uint(0)
        check(load().earhip_panner_missed(self.h, C.byref(n)))
        return n.value

    def close(self):
        if self.h:
            load().earhip_panner_destroy(self.h)
            self.h = C.c_void_p()


class Comm:
    """(J) RCCL communicator of the multi-GPU exchange: one per rank, made from rank 0's 128-byte id"""

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        check(load().earhip_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def channel_range(n_out, rank, world):
        pad, lo, hi = C.c_int(), C.c_int(), C.c_int()
        check(load().earhip_comm_channel_range(n_out, rank, world, C.byref(pad), C.byref(lo), C.byref(hi)))
        return pad.value, lo.value, hi.value

    def __init__(self, ctx, rank, world, uid):
        assert len(uid) == 128
        self.rank, self.world = rank, world
        self.h = C.c_void_p()
        check(load().earhip_comm_create(ctx.h, rank, world, uid, C.byref(self.h)))

    def exchange_device(self, slot, partial_ptr, owned_ptr, rows_per_rank, row_stride):
        check(load().earhip_render_exchange_device(self.h, int(slot), C.c_void_p(partial_ptr), C.c_void_p(owned_ptr),
                                                   C.c_size_t(rows_per_rank), C.c_size_t(row_stride)))

    def gather_device(self, slot, owned_ptr, full_ptr, rows_per_rank, row_stride, root=-1):
        """the owned slices of all ranks -> full [world * rows_per_rank][row_stride] on every rank (root < 0) or on
        `root` only (full_ptr may be None elsewhere)"""
        check(load().earhip_comm_gather_device(self.h, int(slot), C.c_void_p(owned_ptr),
                                               C.c_void_p(full_ptr) if full_ptr else None, C.c_size_t(rows_per_rank),
                                               C.c_size_t(row_stride), int(root)))

    def wait(self, slot):
        """orders the context's stream behind the last exchange / gather issued with this slot"""
        check(load().earhip_comm_wait(self.h, int(slot)))

    def last_exchange_ms(self, slot):
        ms = C.c_double(0.0)
        check(load().earhip_comm_last_exchange_ms(self.h, int(slot), C.byref(ms)))
        return ms.value

    def info(self):
        """what RCCL says about the communicator: ranks (ncclCommCount), this rank, its HIP device, the RCCL version"""
        v = (C.c_int * 4)()
        check(load().earhip_comm_info(self.h, v))
        return {"ranks": v[0], "rank": v[1], "device": v[2], "version": v[3]}

    def link_probe(self, nbytes=50 << 20, shift=1, reps=5):
        """GB/s a rank sends to rank + shift while it receives from rank - shift (collective; 0 with one rank)"""
        g = C.c_double(0.0)
        check(load().earhip_comm_link_probe(self.h, C.c_size_t(nbytes), int(shift), int(reps), C.byref(g)))
        return g.value

    def close(self):
        if self.h:
            load().earhip_comm_destroy(self.h)
            self.h = C.c_void_p()


def hoa_decode_matrix(ctx, layout, orders, degrees, normalization="SN3D", positions=None):
    """(I, HOA) AllRAD decode matrix [n_channels][n_coef] float32; positions: real loudspeaker positions as for Panner"""
    o = np.ascontiguousarray(orders, np.int32)
    d = np.ascontiguousarray(degrees, np.int32)
    if len(o) != len(d):
        raise InvalidArgument(INVALID_ARGUMENT, "orders and degrees must be the same size")
    n = len(layout_channels(layout))
    out = np.zeros((n, max(len(o), 1)), np.float32)
    ip = C.POINTER(C.c_int)
    if positions is None:
        check(load().earhip_hoa_decode_matrix(ctx.h, layout.encode(), len(o), o.ctypes.data_as(ip), d.ctypes.data_as(ip),
                                              normalization.encode(), _ptr(out)))
    else:
        f64 = C.POINTER(C.c_double)
        az, el = (np.ascontiguousarray(v, np.float64) for v in positions)
        check(load().earhip_hoa_decode_matrix_positions(ctx.h, layout.encode(), int(az.size), _ptr(az, f64), _ptr(el, f64),
                                                        len(o), o.ctypes.data_as(ip), d.ctypes.data_as(ip),
                                                        normalization.encode(), _ptr(out)))
    return out[:, :len(o)]


class Renderer:
    """(F) composed Objects render block."""

    def __init__(self, ctx, n_objects, n_out, block_size, decorrelators=None, delay=0, max_blocks=1):
        self.ctx = ctx
        self.M, self.N, self.B = n_objects, n_out, block_size
        self.K = 2 if decorrelators is not None else 1
        cfg = RenderConfig()
        cfg.n_objects, cfg.n_out, cfg.block_size, cfg.n_buses = n_objects, n_out, block_size, self.K
        if decorrelators is not None:
            self._dec = _f32(decorrelators).reshape(n_out, -1)
            cfg.decorrelators = _ptr(self._dec)
            cfg.n_taps = self._dec.shape[1]
        else:
            cfg.decorrelators = None
            cfg.n_taps = 0
        cfg.delay = delay
        cfg.max_blocks = max_blocks
        self.h = C.c_void_p()
        check(load().earhip_render_create(ctx.h, C.byref(cfg), C.byref(self.h)))

    def set_object_points(self, obj, times, direct, diffuse=None):
        t = np.ascontiguousarray(times, dtype=np.int64)
        d = _f32(direct).reshape(len(t), self.N)
        f = _f32(diffuse).reshape(len(t), self.N) if diffuse is not None else None
        check(load().earhip_render_set_object_points(self.h, obj, len(t), _ptr(t, i64p), _ptr(d),
                                                     _ptr(f) if f is not None else None))

    def commit(self):
        check(load().earhip_render_commit(self.h))

    def reset(self, sample_time=0):
        check(load().earhip_render_reset(self.h, C.c_int64(sample_time)))

    def process(self, x):
        """x [M][nblocks*B] host array -> [N][nblocks*B]."""
        x = _f32(x).reshape(self.M, -1)
        nblocks = x.shape[1] // self.B
        assert nblocks * self.B == x.shape[1]
        out = np.empty((self.N, x.shape[1]), np.float32)
        check(load().earhip_render_process(self.h, C.c_size_t(nblocks), _chan_ptrs(x), _chan_ptrs(out)))
        return out

    def process_into(self, x, out):
        """the same with the caller's own arrays, used as they are (e.g. Context.pinned_array: no staging copies)"""
        assert x.dtype == np.float32 and out.dtype == np.float32 and x.flags["C_CONTIGUOUS"] and out.flags["C_CONTIGUOUS"]
        nblocks = x.shape[1] // self.B
        assert x.shape == (self.M, nblocks * self.B) and out.shape == (self.N, nblocks * self.B)
        check(load().earhip_render_process(self.h, C.c_size_t(nblocks), _chan_ptrs(x), _chan_ptrs(out)))
        return out

    def process_device(self, nblocks, in_ptr, in_stride, out_ptr, out_stride):
        check(load().earhip_render_process_device(self.h, C.c_size_t(nblocks), C.c_void_p(in_ptr),
                                                  C.c_size_t(in_stride), C.c_void_p(out_ptr),
                                                  C.c_size_t(out_stride)))

    def enable_timing(self, on=True):
        """True / 1: time the kernels of every process call; n > 1: of every n-th call; False: stop"""
        check(load().earhip_render_enable_timing(self.h, int(on)))

    def get_timing(self):
        out = (C.c_double * 6)()
        check(load().earhip_render_get_timing(self.h, out))
        return {"gain_mix_ms": out[0], "gain_mix_launches": out[1], "decor_ms": out[2],
                "decor_launches": out[3], "prep_ms": out[4], "prep_launches": out[5]}

    def gain_kernel(self):
        """0 strict VALU, 1 f32 MFMA (slot lists), 2 f32 MFMA on the tile grid, 3 f16x2 MFMA, 4 f16x2 MFMA over piece lists, 5 f16x2 MFMA
        with hinges: the gain kernel of
        the last call"""
        kind = C.c_int(-1)
        check(load().earhip_render_gain_kernel(self.h, C.byref(kind)))
        return kind.value

    def hinge_standby(self):
        """True when the last call was planned for the hinge kernel (5) and the device handed it to the piece lists standing
        by (its inputs' levels spread further than the hinge kernel's span); synchronises the stream"""
        flag = C.c_int(0)
        check(load().earhip_render_hinge_standby(self.h, C.byref(flag)))
        return bool(flag.value)

    def hinge_robust(self):
        """True when the last call was the hinge kernel's (5) and ran its robust form: kink products in f32, because the levels of
        the call's inputs spread beyond the span of the packed-f16 products (earhip_render_hinge_robust); synchronises the stream"""
        flag = C.c_int(0)
        check(load().earhip_render_hinge_robust(self.h, C.byref(flag)))
        return bool(flag.value)

    def wide_form(self):
        """True / False: the form (wide / plain low pieces of the inputs) the split-operand kernel of the last call ran — picked on
        the device for long calls, wide for short ones; None when the kernel has no split operands; synchronises the stream"""
        flag = C.c_int(1)
        check(load().earhip_render_wide_form(self.h, C.byref(flag)))
        return None if flag.value < 0 else bool(flag.value)

    def scratch_bytes(self):
        """device scratch (descriptors, lists) the last call needed"""
        v = C.c_size_t(0)
        check(load().earhip_render_scratch_bytes(self.h, C.byref(v)))
        return v.value

    def scratch_regrows(self):
        """process calls that had to grow the list scratch themselves (0 on committed curves)"""
        v = C.c_long(0)
        check(load().earhip_render_scratch_regrows(self.h, C.byref(v)))
        return v.value

    def last_tail_blocks(self):
        """blocks of the last call that ran as a tail of their own behind the whole rounds of tiles (0: the call was not cut)"""
        v = C.c_int(0)
        check(load().earhip_render_last_tail_blocks(self.h, C.byref(v)))
        return v.value

    def last_host_chunks(self):
        """time chunks the last call from host channel pointers ran as (0: one piece)"""
        v = C.c_int(0)
        check(load().earhip_render_last_host_chunks(self.h, C.byref(v)))
        return v.value

    def last_list_layout(self):
        """layout of the last call's piece lists: True paired, False packed, None: none built"""
        v = C.c_int(0)
        check(load().earhip_render_last_list_layout(self.h, C.byref(v)))
        return None if v.value < 0 else bool(v.value)

    def last_plan(self):
        """launch plan of the last call: gain kernel, samples per workgroup tile, tiles, object splits"""
        out = (C.c_int * 4)()
        check(load().earhip_render_last_plan(self.h, out))
        return {"kernel": out[0], "tile": out[1], "ntiles": out[2], "gsplit": out[3]}

    def close(self):
        if self.h:
            load().earhip_render_destroy(self.h)
            self.h = C.c_void_p()
