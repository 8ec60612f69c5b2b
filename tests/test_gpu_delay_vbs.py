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
