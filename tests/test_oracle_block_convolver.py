"""CPU oracle BlockConvolver through the reference's 15 scenarios, checked against the
brute-force time-domain convolution at the reference's tolerance, abs 1e-6 per sample
(reference tests/block_convolver_tests.cpp:77,191-193,197-356)."""
import numpy as np
import pytest

import _oracle
from refcases import conv_scenarios, generate_random


def test_filter_correct_num_blocks():
    # :197-206
    ctx = _oracle.ConvCtx(512)
    coeff = generate_random(2000, 100, 0)
    for n, want in ((1, 1), (511, 1), (512, 1), (513, 2)):
        assert _oracle.ConvFilter(ctx, coeff[:n]).num_blocks() == want


@pytest.mark.parametrize("sc", conv_scenarios(), ids=lambda s: s.name)
def test_scenario(sc):
    ctx = _oracle.ConvCtx(sc.block_size)
    got = sc.run(ctx, _oracle.ConvFilter, _oracle.BlockConvolver)
    want = sc.expected()
    assert np.max(np.abs(got - want)) < 1e-6


def test_wrong_context_and_too_many_blocks_raise():
    # block_convolver_impl.cpp:85-98
    c512, c256 = _oracle.ConvCtx(512), _oracle.ConvCtx(256)
    f256 = _oracle.ConvFilter(c256, np.ones(10, np.float32))
    conv = _oracle.BlockConvolver(c512, None, 1)
    with pytest.raises(_oracle.OracleError) as e:
        conv.set_filter(f256)
    assert e.value.code == 1
    f_long = _oracle.ConvFilter(c512, np.ones(513, np.float32))
    with pytest.raises(_oracle.OracleError) as e:
        conv.crossfade_filter(f_long)
    assert "too many blocks" in str(e.value)
