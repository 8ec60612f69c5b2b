"""HIP DelayBuffer and the VariableBlockSizeAdapter: exact-shift properties, `==`
(reference tests/delay_buffer_tests.cpp:11-65, tests/variable_block_size_tests.cpp:10-66)."""
import numpy as np
import pytest

from _hip import ctx

pytestmark = pytest.mark.gpu


def test_delay_buffer_five_channels():
    from libear_amd import capi
    delay, sizes = 128, [64, 128, 256]
    total = sum(sizes)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (5, total)).astype(np.float32)
    db = capi.DelayBuffer(ctx(), 5, delay)
    assert db.get_delay() == delay
    out = np.zeros_like(x)
    ofs = 0
    for n in sizes:
        out[:, ofs:ofs + n] = db.process(x[:, ofs:ofs + n])
        ofs += n
    want = np.zeros_like(x)
    want[:, delay:] = x[:, :total - delay]
    assert np.array_equal(out, want)


def test_delay_buffer_single_channel_and_ragged_calls():
    from libear_amd import capi
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (1, 512)).astype(np.float32)
    out = capi.DelayBuffer(ctx(), 1, 128).process(x)
    want = np.zeros_like(x)
    want[:, 128:] = x[:, :384]
    assert np.array_equal(out, want)
    # calls shorter than the delay, delay 255 (decorrelatorCompensationDelay)
    x = rng.uniform(-1, 1, (3, 1000)).astype(np.float32)
    db = capi.DelayBuffer(ctx(), 3, 255)
    out = np.zeros_like(x)
    ofs = 0
    for n in [1, 7, 100, 254, 255, 256, 127]:
        out[:, ofs:ofs + n] = db.process(x[:, ofs:ofs + n])
        ofs += n
    want = np.zeros_like(x)
    want[:, 255:] = x[:, :745]
    assert np.array_equal(out[:, :ofs], want[:, :ofs])


def toy_process(x):
    return np.stack([x[0] * 2.0, x[1] * 3.0, x[0] * 4.0, x[1] * 5.0]).astype(np.float32)


def test_variable_block_size_adapter_around_device_gain_stage():
    """the adapter (host FIFO) wrapped around a device process function"""
    from libear_amd import capi
    B = 512
    sizes = [0, 512, 1024, 300, 500]
    total = sum(sizes)
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (2, total)).astype(np.float32)

    gains = np.float32([[2.0, 0.0, 4.0, 0.0], [0.0, 3.0, 0.0, 5.0]])  # [in][out]

    def device_process(blk):
        out = np.zeros((4, B), np.float32)
        ctx().set_strict(True)
        ctx().apply_constant(np.ascontiguousarray(blk), out, 0, B, gains)
        ctx().set_strict(False)
        return out

    want = np.zeros((4, total), np.float32)
    want[:, B:] = toy_process(x[:, :total - B])
    ad = capi.VariableBlockSizeAdapter(B, 2, 4, device_process)
    assert ad.get_delay() == B
    out = np.full((4, total), np.nan, np.float32)
    ofs = 0
    for n in sizes:
        out[:, ofs:ofs + n] = ad.process(x[:, ofs:ofs + n])
        ofs += n
    assert np.array_equal(out, want)


def test_adapter_with_pinned_fifo_around_the_renderer():
    """earhip_vbs_create_pinned: the adapter's FIFO rows are evenly spaced channel buffers in device-reachable host
    memory, so the renderer called from the callback (docs/dsp.rst:65-71: wrap the whole chain) takes its
    no-staging path — same bits as the adapter with ordinary buffers, calls of any size."""
    import ctypes
    import scenes
    from layouts import LAYOUTS
    from libear_amd import capi
    names = LAYOUTS["0+5+0"]
    m, n, B = 40, len(names), 512
    dec = capi.design_decorrelators(names)
    sizes = [100, 512, 700, 3, 1200, 557]
    total = sum(sizes)
    curves = scenes.adm_curves(m, n, total + B, period=500, ramp=120, seed=4)
    x = scenes.audio(m, total, seed=9)
    lib = capi.load()
    outs = []
    c = capi.Context(0)
    try:
        for pinned in (False, True):
            r = capi.Renderer(c, m, n, B, dec, 255, max_blocks=1)
            for i, (t, d, f) in enumerate(curves):
                r.set_object_points(i, t, d, f)
            ad = capi.VariableBlockSizeAdapter(
                B, m, n, lambda ip, op: capi.check(lib.earhip_render_process(r.h, ctypes.c_size_t(1), ip, op)),
                ctx=c if pinned else None, raw=True)
            out = np.zeros((n, total), np.float32)
            ofs = 0
            for k in sizes:
                out[:, ofs:ofs + k] = ad.process(x[:, ofs:ofs + k])
                ofs += k
            outs.append(out)
            ad.close()
            r.close()
    finally:
        c.close()
    assert np.all(outs[0][:, :B] == 0.0) and np.abs(outs[0][:, B:]).max() > 0
    assert np.array_equal(outs[0], outs[1])
