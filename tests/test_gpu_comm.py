"""The native exchange step (libearhip group J: RCCL reduce-scatter on the communicator's stream, ordered
against the context's stream by events).  One GPU per box: a one-rank communicator (the reduce-scatter is
then a copy) checks creation, stream ordering and the double-buffer protocol; the multi-rank arithmetic
(sharding, ragged ownership, sum) is covered by tests/test_distributed_gloo.py and test_gpu_bench.py."""
import numpy as np
import pytest

from _hip import ctx

pytestmark = pytest.mark.gpu


def test_single_rank_exchange_orders_behind_the_render_and_overlaps():
    import torch
    import scenes
    from layouts import LAYOUTS
    from libear_amd import capi
    names = LAYOUTS["4+5+0"]
    n, m, block, nblocks = len(names), 64, 512, 8
    total = block * nblocks
    uid = capi.Comm.unique_id()
    assert len(uid) == 128
    comm = capi.Comm(ctx(), 0, 1, uid)
    pad, lo, hi = capi.Comm.channel_range(n, 0, 1)
    assert (pad, lo, hi) == (n, 0, n)
    dec = capi.design_decorrelators(names)
    r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
    curves = scenes.dense_curves(m, n, block, nblocks)
    for i, (t, d, f) in enumerate(curves):
        r.set_object_points(i, t, d, f)
    x = torch.rand((m, total), device="cuda") * 2 - 1
    part = [torch.zeros((pad, total), device="cuda") for _ in range(2)]
    owned = [torch.full((pad, total), -7.0, device="cuda") for _ in range(2)]
    full = [torch.full((pad, total), -9.0, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for step in range(6):  # double-buffered: render into slot s, exchange it, meanwhile render the other slot
        s = step % 2
        comm.wait(s)
        r.reset(0)
        r.process_device(nblocks, x.data_ptr(), total, part[s].data_ptr(), total)
        comm.exchange_device(s, part[s].data_ptr(), owned[s].data_ptr(), pad, total)
        # the shared bus in one place: gather to a root (slot 0) / all-gather (slot 1), behind the exchange
        comm.gather_device(s, owned[s].data_ptr(), full[s].data_ptr(), pad, total, root=0 if s == 0 else -1)
    comm.wait(0)
    comm.wait(1)
    ctx().synchronize()
    assert torch.equal(owned[0], part[0]) and torch.equal(owned[1], part[1])
    assert torch.equal(owned[0], owned[1]) and float(owned[0].abs().max()) > 0
    assert torch.equal(full[0], part[0]) and torch.equal(full[1], part[1])
    assert comm.last_exchange_ms(0) > 0.0 and comm.last_exchange_ms(1) > 0.0
    # what RCCL says about the communicator (a multi-GPU bench line carries this: exchange.rccl), and the link probe
    info = comm.info()
    assert info["ranks"] == 1 and info["rank"] == 0 and info["device"] == torch.cuda.current_device() and info["version"] > 20000, info
    assert comm.link_probe(1 << 20, 1, 2) == 0.0  # (one rank: nobody to send to)
    with pytest.raises(capi.InvalidArgument):
        comm.gather_device(0, owned[0].data_ptr(), None, pad, total, root=0)  # the root must have a buffer
    with pytest.raises(capi.InvalidArgument):
        comm.gather_device(0, owned[0].data_ptr(), full[0].data_ptr(), pad, total, root=1)  # no such rank
    comm.close()
    r.close()
    for bad in ((10, 4, 4), (0, 0, 1)):
        with pytest.raises(capi.InvalidArgument):
            capi.Comm.channel_range(*bad)
    assert [capi.Comm.channel_range(10, rk, 4)[1:] for rk in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
