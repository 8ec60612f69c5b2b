"""The N > 1 path on CPU: world_size 2 over gloo.  Objects are sharded by rank, every rank
renders its shard completely, the partial loudspeaker buses are summed by the one exchange step
(libear_amd/distributed.py) and every rank ends up owning its slice of the channels.  The CPU
oracle stands in for the device renderer here (the collective, the sharding arithmetic and the
linearity argument are what is under test); the GPU suite checks the same decomposition with the
real renderer on one device (tests/test_gpu_render.py::test_c4_size_vs_oracle_and_shard_linearity)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    """a port nobody listens on, taken BELOW the kernel's ephemeral range (32768-60999): a port from that range (bind to 0) can be
    handed to somebody's outgoing connection between this check and the rendezvous' own bind (seen once: EADDRINUSE)"""
    import random
    for _ in range(64):
        port = random.randint(15000, 30000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", port))
        except OSError:
            continue
        finally:
            s.close()
        return port
    raise RuntimeError("no free port found")


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _oracle
    import scenes
    from layouts import LAYOUTS
    from libear_amd.distributed import channel_range, exchange, padded_channels, shard_range

    names = LAYOUTS["4+5+0"]
    m, n, block, nblocks = 24, len(names), 512, 3
    dec = _oracle.design_decorrelators(names)
    curves = scenes.dense_curves(m, n, block, nblocks)
    x = scenes.audio(m, block * nblocks)
    lo, hi = shard_range(m, rank, world)
    o = _oracle.ObjectsRenderer(hi - lo, n, block, dec, 255)
    for i, (t, d, f) in enumerate(curves[lo:hi]):
        o.set_points(i, 0, t, d)
        o.set_points(i, 1, t, f)
    # exchange buffers hold the channel count rounded up to a multiple of the ranks (10 channels, 3 ranks:
    # 12 rows, the last rank owns 2 real channels)
    partial = torch.zeros((padded_channels(n, world), block * nblocks), dtype=torch.float32)
    partial[:n] = torch.from_numpy(o.process(x[lo:hi]))
    owned, work = exchange(partial, async_op=True)
    work.wait()
    clo, chi = channel_range(n, rank, world)
    assert owned.shape[0] == padded_channels(n, world) // world
    assert not owned[chi - clo:].any()  # padding rows stay zero
    np.save(os.path.join(tmp, f"owned_{rank}.npy"), owned[:chi - clo].numpy())
    if rank == 0:
        full = _oracle.ObjectsRenderer(m, n, block, dec, 255)
        for i, (t, d, f) in enumerate(curves):
            full.set_points(i, 0, t, d)
            full.set_points(i, 1, t, f)
        np.save(os.path.join(tmp, "full.npy"), full.process(x))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_and_exchange(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    got = np.concatenate([np.load(tmp_path / f"owned_{r}.npy") for r in range(world)], axis=0)
    assert got.shape == full.shape
    err = np.linalg.norm(got.astype(np.float64) - full) / np.linalg.norm(full)
    assert err <= 1e-6, err


def _time_worker(rank, world, port, tmp, block):
    """--shard time on CPU: rank r renders ALL objects for blocks [b0, b1) behind `lead` blocks from the zero state; the
    pieces, put side by side, are the G = 1 render BIT FOR BIT (no sum is regrouped; the oracle transforms single blocks)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _oracle
    import scenes
    from layouts import LAYOUTS
    from libear_amd.distributed import time_range

    names = LAYOUTS["0+5+0"]
    m, n, nblocks = 12, len(names), 9
    dec = _oracle.design_decorrelators(names)
    curves = scenes.adm_curves(m, n, block * nblocks, period=300, ramp=100, seed=3)
    x = scenes.audio(m, block * nblocks)
    b0, b1, lead = time_range(nblocks, rank, world, partitions=-(-512 // block))  # (512-tap FIRs: ceil(512 / block) partitions)
    lo, hi = (b0 - lead) * block, b1 * block
    o = _oracle.ObjectsRenderer(m, n, block, dec, 255)
    for i, (t, d, f) in enumerate(scenes.window_curves(curves, lo, hi)):
        o.set_points(i, 0, t, d)
        o.set_points(i, 1, t, f)
    np.save(os.path.join(tmp, f"piece_{rank}.npy"), o.process(x[:, lo:hi])[:, lead * block:])
    if rank == 0:
        full = _oracle.ObjectsRenderer(m, n, block, dec, 255)
        for i, (t, d, f) in enumerate(curves):
            full.set_points(i, 0, t, d)
            full.set_points(i, 1, t, f)
        np.save(os.path.join(tmp, "full.npy"), full.process(x))
    dist.barrier()  # (the only collective of this mode: none on the data path)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,block", [(2, 512), (3, 512), (2, 256)])
def test_time_sharding_is_bit_identical_to_one_rank(tmp_path, world, block):
    mp.spawn(_time_worker, args=(world, _free_port(), str(tmp_path), block), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    got = np.concatenate([np.load(tmp_path / f"piece_{r}.npy") for r in range(world)], axis=1)
    assert got.shape == full.shape
    assert np.array_equal(got, full)


def _repeat_worker(rank, world, port, tmp):
    """the reduction order across ranks is deterministic — the same partials exchanged twice give bit-identical
    owned slices (objects / objects-nogather: the reduce-scatter is the only step that adds)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libear_amd.distributed import exchange, padded_channels
    g = torch.Generator().manual_seed(100 + rank)
    part = torch.rand((padded_channels(10, world), 4096), generator=g) * 2 - 1
    outs = []
    for _ in range(2):
        buf = part.clone()
        owned, work = exchange(buf, async_op=True)
        work.wait()
        outs.append(owned.clone())
    assert torch.equal(outs[0], outs[1])
    np.save(os.path.join(tmp, f"ok_{rank}.npy"), np.ones(1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_twice_is_bit_identical(tmp_path, world):
    mp.spawn(_repeat_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok_{r}.npy").exists() for r in range(world))


def test_exchange_model_per_mode():
    sys.path.insert(0, ROOT)
    from libear_amd.distributed import exchange_model, time_range
    n_pad, samples = 24, 524288
    for world, comp in ((2, 0.265), (4, 0.176), (8, 0.125)):
        a = exchange_model("objects", world, n_pad, samples, comp, 50.0, 4)
        b = exchange_model("objects-nogather", world, n_pad, samples, comp, 50.0, 4)
        c = exchange_model("time", world, n_pad, samples, comp, 50.0, 1)
        assert a["exchange_ms_per_step"] == pytest.approx(2 * b["exchange_ms_per_step"], rel=1e-3)
        assert c["exchange_ms_per_step"] == 0 and c["predicted_ms_per_step"] == pytest.approx(comp, rel=1e-3)
        assert a["predicted_ms_per_step"] >= b["predicted_ms_per_step"] >= 0
        assert all("unmeasured" in m["status"] for m in (a, b, c))
    assert exchange_model("objects", 2, 24, 524288, 0.265, 50.0, 4)["exchange_ms_per_step"] == pytest.approx(1.0066, rel=1e-3)
    cover = []
    for r in range(3):
        b0, b1, lead = time_range(10, r, 3)
        assert lead == (1 if r else 0)
        cover.extend(range(b0, b1))
    assert cover == list(range(10))


def test_lead_in_follows_the_reach_of_the_firs_and_the_delay():
    """a time-sharded rank (and a parity window inside a stream) starts as many blocks early as the DSP state reaches back: the
    512-tap decorrelators 511 samples, the compensation delay 255 — one block from 512 samples per block, two at 256, eight at
    64; a longer delay asks for its own"""
    sys.path.insert(0, ROOT)
    from libear_amd.distributed import lead_blocks, time_range
    assert [lead_blocks(b) for b in (1024, 512, 511, 256, 128, 64)] == [1, 1, 1, 2, 4, 8]
    assert lead_blocks(512, n_taps=1, delay=0) == 1 and lead_blocks(512, n_taps=512, delay=2000) == 4
    assert time_range(64, 2, 4, partitions=2, delay_blocks=1)[2] == 2 and time_range(64, 2, 4, partitions=1, delay_blocks=4)[2] == 4
    assert time_range(64, 0, 4, partitions=2, delay_blocks=4) == (0, 16, 0)  # (the stream's first rank has nothing in front of it)


def test_shard_ranges_cover_everything():
    sys.path.insert(0, ROOT)
    from libear_amd.distributed import channel_range, shard_range
    for m in (1, 7, 64, 1024, 1023):
        for world in (1, 2, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(m, r, world)
                cover.extend(range(lo, hi))
            assert cover == list(range(m))
    for n in (24, 10, 6, 2):  # 9+10+3, 4+5+0, 0+5+0, 0+2+0: ragged ownership where the ranks do not divide n
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = channel_range(n, r, world)
                assert 0 <= hi - lo <= -(-n // world)
                cover.extend(range(lo, hi))
            assert cover == list(range(n))
