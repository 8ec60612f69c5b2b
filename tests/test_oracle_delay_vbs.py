"""CPU oracle DelayBuffer and VariableBlockSizeAdapter: exact-shift properties
(reference tests/delay_buffer_tests.cpp:11-65, tests/variable_block_size_tests.cpp:10-66)."""
import numpy as np

import _oracle


def test_delay_buffer_five_channels():
    delay, sizes = 128, [64, 128, 256]
    total = sum(sizes)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (5, total)).astype(np.float32)
    db = _oracle.DelayBuffer(5, delay)
    out = np.zeros_like(x)
    ofs = 0
    for n in sizes:
        out[:, ofs:ofs + n] = db.process(x[:, ofs:ofs + n])
        ofs += n
    want = np.zeros_like(x)
    want[:, delay:] = x[:, :total - delay]
    assert np.array_equal(out, want)


def test_delay_buffer_single_channel():
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (1, 512)).astype(np.float32)
    out = _oracle.DelayBuffer(1, 128).process(x)
    want = np.zeros_like(x)
    want[:, 128:] = x[:, :384]
    assert np.array_equal(out, want)


def toy_process(x):
    return np.stack([x[0] * 2.0, x[1] * 3.0, x[0] * 4.0, x[1] * 5.0]).astype(np.float32)


def test_variable_block_size_adapter():
    B = 512
    sizes = [0, 512, 1024, 300, 500]
    total = sum(sizes)
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (2, total)).astype(np.float32)
    want = np.zeros((4, total), np.float32)
    want[:, B:] = toy_process(x[:, :total - B])
    ad = _oracle.VariableBlockSizeAdapter(B, 2, 4, toy_process)
    assert ad.get_delay() == B
    out = np.full((4, total), np.nan, np.float32)
    ofs = 0
    for n in sizes:
        out[:, ofs:ofs + n] = ad.process(x[:, ofs:ofs + n])
        ofs += n
    assert np.array_equal(out, want)
