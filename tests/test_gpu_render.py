"""The composed Objects render block on the GPU (K0 segment prep -> K1 gain_mix -> K2
decorrelate/delay/mix) vs the CPU oracle's composition of the restated libear components
(docs/dsp.rst:40-71).

Bars:
  * gain stage alone (n_buses = 1), strict mode: bit-exact vs the oracle;
  * full chain, default mode: relative RMS <= 1e-6 vs the oracle over all outputs (north_star's
    tolerance; relative as BASELINE.md §2 argues), and both close to a float64 evaluation;
  * streamed (T blocks per call) == block-at-a-time within 1e-6 relative RMS; state carries over;
  * full BASELINE sizes: size-independent properties (shard linearity, silence, impulse -> FIR).
"""
import os

import numpy as np
import pytest

import _oracle
import scenes
from _hip import ctx, set_oracle_curves, set_renderer_curves, with_options
from layouts import LAYOUTS

pytestmark = pytest.mark.gpu


def decorrelators(layout):
    """the FIRs come from the product's native design (libearhip group G), as a caller's would; they equal
    the oracle's (tests/test_decorrelate_native.py, test_gpu_render_full.py)"""
    from libear_amd import capi
    return capi.design_decorrelators(LAYOUTS[layout])


def run_hip(curves, x, n_out, block, dec, delay, calls, strict=False):
    from libear_amd import capi
    two = dec is not None
    ctx().set_strict(strict)
    try:
        r = capi.Renderer(ctx(), x.shape[0], n_out, block, dec, delay, max_blocks=max(calls))
        set_renderer_curves(r, curves, two)
        out = np.zeros((n_out, x.shape[1]), np.float32)
        ofs = 0
        for nb in calls:
            n = nb * block
            out[:, ofs:ofs + n] = r.process(x[:, ofs:ofs + n])
            ofs += n
        r.close()
    finally:
        ctx().set_strict(False)
    return out


def run_oracle(curves, x, n_out, block, dec, delay):
    if dec is None:  # direct bus only: zero decorrelators, no delay => out == direct bus exactly
        o = _oracle.ObjectsRenderer(x.shape[0], n_out, block, np.zeros((n_out, 1), np.float32), 0)
        set_oracle_curves(o, curves, two_bus=False)
    else:
        o = _oracle.ObjectsRenderer(x.shape[0], n_out, block, dec, delay)
        set_oracle_curves(o, curves)
    return o.process(x)


def _with_env(env, fn):
    return with_options(env, fn)


@pytest.mark.parametrize("m,n,block,nblocks", [(1, 6, 512, 3), (64, 10, 512, 4), (5, 3, 64, 9), (33, 24, 256, 5)])
def test_gain_stage_strict_bit_exact(m, n, block, nblocks):
    """C1/C2-shaped: ramped gains only; strict mode reproduces the CPU path bit for bit."""
    curves = scenes.dense_curves(m, n, block, nblocks)
    x = scenes.audio(m, block * nblocks)
    want = run_oracle(curves, x, n, block, None, 0)
    got = run_hip(curves, x, n, block, None, 0, [nblocks], strict=True)
    assert np.array_equal(got, want)
    got2 = run_hip(curves, x, n, block, None, 0, [1] * nblocks, strict=True)
    assert np.array_equal(got2, want)


def test_gain_stage_ragged_curves_strict_bit_exact():
    """points inside tiles, steps, equal neighbours, single points, ramps crossing the call"""
    m, n, block, nblocks = 40, 7, 128, 11
    total = block * nblocks
    curves = scenes.ragged_curves(m, n, total)
    x = scenes.audio(m, total)
    want = run_oracle(curves, x, n, block, None, 0)
    got = run_hip(curves, x, n, block, None, 0, [3, 1, 7], strict=True)
    assert np.array_equal(got, want)
    fast = run_hip(curves, x, n, block, None, 0, [nblocks])
    assert scenes.rel_rms(fast, want) <= 1e-6


@pytest.mark.parametrize("m,layout,block,nblocks,kind",
                         [(1, "0+5+0", 512, 4, "dense"),  # BASELINE config 1 on the HIP path (1 -> N: exact arithmetic)
                          (64, "4+5+0", 512, 4, "dense"), (256, "9+10+3", 512, 3, "dense"),
                          (16, "0+5+0", 1024, 3, "dense"), (48, "9+10+3", 512, 4, "sparse"),
                          (20, "4+5+0", 128, 9, "ragged"), (12, "0+5+0", 2048, 2, "constant")])
def test_full_chain_vs_oracle(m, layout, block, nblocks, kind):
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    if kind == "dense":
        curves = scenes.dense_curves(m, n, block, nblocks)
    elif kind == "sparse":
        lfe = [i for i, nm in enumerate(LAYOUTS[layout]) if nm.startswith("LFE")]
        curves = scenes.sparse_curves(m, n, block, nblocks, lfe)
    elif kind == "ragged":
        curves = scenes.ragged_curves(m, n, total)
    else:
        curves = scenes.constant_curves(m, n)
    x = scenes.audio(m, total)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks])
    assert scenes.rel_rms(got, want) <= 1e-6
    if m == 1:  # config 1: the gain stage alone is libear's LinearInterpVector — bit for bit in strict mode (the default
        # kernel for fewer than 32 objects contracts the ramp into fused multiply-adds: within 1e-6, above)
        direct = run_hip(curves, x, n, block, None, 0, [nblocks], strict=True)
        assert np.array_equal(direct, run_oracle(curves, x, n, block, None, 0))
    step = run_hip(curves, x, n, block, dec, 255, [1] * nblocks)
    assert scenes.rel_rms(step, want) <= 1e-6
    assert scenes.rel_rms(step, got) <= 1e-6
    if m * total * n <= 64 * 2048 * 24:
        truth = scenes.render_f64(curves, x, n, dec, 255)
        assert scenes.rel_rms(got, truth) <= 1e-6
        assert scenes.rel_rms(want, truth) <= 1e-6


@pytest.mark.parametrize("m,nblocks,calls", [(64, 8, [8]), (65, 8, [3, 5]), (200, 4, [4]), (33, 2, [1, 1])])
def test_unaligned_metadata_piece_lists(m, nblocks, calls):
    """ADM-like curves whose points ignore the tile grid: a quarter of all (tile, object) pairs
    has a curve point inside the tile and is rendered from the per-tile piece lists (K0b); odd
    object counts put such an object last; several calls restart the lists mid-curve."""
    layout, block = "9+10+3", 512
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    curves = scenes.adm_curves(m, n, total, period=300, ramp=77, seed=m)
    x = scenes.audio(m, total, seed=m)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, calls)
    assert scenes.rel_rms(got, want) <= 1e-6
    strict = run_hip(curves, x, n, block, None, 0, calls, strict=True)
    assert np.array_equal(strict, run_oracle(curves, x, n, block, None, 0))


def test_long_host_call_threaded_staging():
    """>= 16 MB of input per call: the host-pointer entry point gathers the channels with several
    threads, group by group, overlapping the transfers; same result as the oracle and as short calls."""
    m, layout, block, nblocks = 64, "0+5+0", 512, 128  # 64 x 65536 x 4 B = 16.8 MB
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    curves = scenes.adm_curves(m, n, total, seed=3)
    x = scenes.audio(m, total, seed=4)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks])
    assert scenes.rel_rms(got, want) <= 1e-6
    short = run_hip(curves, x, n, block, dec, 255, [32] * 4)
    assert scenes.rel_rms(short, got) <= 1e-6


def test_piece_list_overflow_takes_generic_path():
    """More pieces in one tile than its list holds (cap = number of objects): the objects that
    do not fit are flagged by K0b and rendered by the generic path; the result is the same."""
    m, n, block, nblocks = 16, 24, 512, 2
    total = block * nblocks
    rng = np.random.default_rng(5)
    curves = []
    for i in range(m):
        t = np.sort(rng.integers(0, total, 60)).astype(np.int64)  # ~7 points per 128-sample tile
        curves.append((t, rng.uniform(0, 1, (60, n)).astype(np.float32),
                       rng.uniform(0, 1, (60, n)).astype(np.float32)))
    x = scenes.audio(m, total, seed=3)
    dec = decorrelators("9+10+3")
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks])
    assert scenes.rel_rms(got, want) <= 1e-6


def test_mixed_hoa_bed_and_objects_scene_block_1024():
    """BASELINE config 5 shape at reduced object count: a 16-channel HOA bed through a constant
    16 x 24 decode matrix (LinearInterpMatrix, never interpolated: docs/dsp.rst:73-89) plus ramped
    objects, block 1024 (FFT 2048).  The bed channels are rendered as inputs with a single gain
    point on the direct bus and zero diffuse gains, so they share the 255-sample compensation delay
    with the objects' direct path."""
    layout, block, nblocks, n_hoa, n_obj = "9+10+3", 1024, 3, 16, 48
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    rng = np.random.default_rng(55)
    decode = rng.uniform(-0.5, 0.5, (n_hoa, n)).astype(np.float32)
    curves = [(np.zeros(1, np.int64), decode[c:c + 1], np.zeros((1, n), np.float32)) for c in range(n_hoa)]
    curves += scenes.dense_curves(n_obj, n, block, nblocks)
    x = scenes.audio(n_hoa + n_obj, block * nblocks)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks])
    assert scenes.rel_rms(got, want) <= 1e-6
    step = run_hip(curves, x, n, block, dec, 255, [1, 2])
    assert scenes.rel_rms(step, want) <= 1e-6
    # the bed alone: out = delayed(decode^T . hoa), bit-exact in strict mode against the oracle
    bed = run_hip(curves[:n_hoa], x[:n_hoa], n, block, None, 0, [nblocks], strict=True)
    assert np.array_equal(bed, run_oracle(curves[:n_hoa], x[:n_hoa], n, block, None, 0))


def test_odd_sizes_and_unaligned_tails():
    """object counts that are odd / not multiples of the wave split, 5 loudspeakers, 7 blocks of 64:
    exercises the single-object path, partial column tiles and short calls"""
    for m, n, block, nblocks in ((1, 5, 64, 7), (3, 5, 64, 7), (17, 9, 128, 5), (130, 24, 64, 3)):
        dec = np.random.default_rng(m).uniform(-0.1, 0.1, (n, min(block, 64))).astype(np.float32)
        curves = scenes.ragged_curves(m, n, block * nblocks, seed=m)
        x = scenes.audio(m, block * nblocks, seed=m)
        want = run_oracle(curves, x, n, block, dec, 31)
        got = run_hip(curves, x, n, block, dec, 31, [nblocks])
        assert scenes.rel_rms(got, want) <= 1e-6, (m, n, block)
        got1 = run_hip(curves, x, n, block, dec, 31, [2, 1, nblocks - 3])
        assert scenes.rel_rms(got1, want) <= 1e-6, (m, n, block)


def test_strict_full_chain_matches_oracle_closely():
    """strict gain stage + device FFT: only the FFT rounding differs from the CPU path"""
    m, layout, block, nblocks = 32, "9+10+3", 512, 3
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    curves = scenes.dense_curves(m, n, block, nblocks)
    x = scenes.audio(m, block * nblocks)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks], strict=True)
    assert scenes.rel_rms(got, want) <= 3e-7


def test_c4_size_vs_oracle_and_shard_linearity():
    """1024 objects -> 9+10+3, block 512 (BASELINE config 4), two blocks against the oracle, and
    the multi-GPU decomposition: rendering two object shards separately and summing the outputs
    equals rendering all objects (the chain after the buses is linear)."""
    m, layout, block, nblocks = 1024, "9+10+3", 512, 2
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    curves = scenes.dense_curves(m, n, block, nblocks)
    x = scenes.audio(m, block * nblocks)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks])
    assert scenes.rel_rms(got, want) <= 1e-6
    half = m // 2
    a = run_hip(curves[:half], x[:half], n, block, dec, 255, [nblocks])
    b = run_hip(curves[half:], x[half:], n, block, dec, 255, [nblocks])
    assert scenes.rel_rms(a + b, got) <= 1e-6
    assert scenes.rel_rms(a + b, want) <= 1e-6


def test_long_stream_properties():
    """64 blocks per call at 256 objects: silence in -> exact silence out; a unit impulse on one
    object with a constant diffuse-only gain reproduces the decorrelator FIR; direct-only gain
    reproduces the 255-sample delay."""
    from libear_amd import capi
    m, layout, block, nblocks = 256, "9+10+3", 512, 64
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
    set_renderer_curves(r, scenes.dense_curves(m, n, block, nblocks))
    assert not np.any(r.process(np.zeros((m, total), np.float32)))
    r.close()

    r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
    one = np.zeros((1, n), np.float32)
    sel = np.zeros((1, n), np.float32)
    sel[0, 5] = 1.0
    r.set_object_points(3, [0], one, sel)       # object 3: diffuse only, loudspeaker 5
    r.set_object_points(7, [0], sel * 0.5, one)  # object 7: direct only, gain 0.5
    x = np.zeros((m, total), np.float32)
    x[3, 1000] = 1.0
    x[7, 20000] = 1.0
    out = r.process(x)
    want = np.zeros((n, total), np.float32)
    want[5, 1000:1512] += dec[5]
    want[5, 20255] += 0.5
    assert np.max(np.abs(out - want)) < 1e-6
    r.close()


@pytest.mark.parametrize("m,layout,two_bus", [(64, "9+10+3", True), (40, "4+5+0", False), (1, "0+5+0", True)])
def test_pinned_and_registered_channel_matrices_skip_the_staging_copies(m, layout, two_bus):
    """earhip_host_alloc / earhip_host_register (include/earhip.h): channel pointers evenly spaced inside memory
    the device reaches take the short-call path without gather, D2H copy or scatter — same bits as the staged
    path, block by block with state carried across calls; pointers of any other shape keep working."""
    from libear_amd import capi
    block, nblocks = 512, 6
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout) if two_bus else None
    total = block * nblocks
    curves = scenes.adm_curves(m, n, total, period=600, ramp=150, seed=m)
    x = scenes.audio(m, total, seed=m + 1)
    calls = [1, 1, 2, 1, 1]  # (the same partition on both paths: two blocks share one transform in K2)
    want = run_hip(curves, x, n, block, dec, 255 if two_bus else 0, calls)
    single = run_hip(curves, x, n, block, dec, 255 if two_bus else 0, [1] * nblocks)

    c = capi.Context(0)
    try:
        r = capi.Renderer(c, m, n, block, dec, 255 if two_bus else 0, max_blocks=2)
        set_renderer_curves(r, curves, two_bus)
        xin, yout = c.pinned_array((m, block)), c.pinned_array((n, block))  # earhip_host_alloc
        xreg, yreg = np.zeros((m, 2 * block), np.float32), np.zeros((n, 2 * block), np.float32)
        c.register(xreg)                                                      # earhip_host_register
        c.register(yreg)
        got = np.zeros((n, total), np.float32)
        b = 0
        for nb in calls:
            if nb == 2:  # two blocks through the registered arrays
                xreg[...] = x[:, b * block:(b + 2) * block]
                r.process_into(xreg, yreg)
                got[:, b * block:(b + 2) * block] = yreg
            else:
                xin[...] = x[:, b * block:(b + 1) * block]
                r.process_into(xin, yout)
                got[:, b * block:(b + 1) * block] = yout
            b += nb
        # rows that are NOT evenly spaced inside a reachable range: the staged path, same results
        r.reset(0)
        odd = c.pinned_array((m + 1, block + 4))
        for b in range(nblocks):
            odd[:m, :block] = x[:, b * block:(b + 1) * block]
            ptr_rows = np.ascontiguousarray(odd[:m, :block])  # (a copy: ordinary memory)
            out = r.process(ptr_rows)
            assert np.array_equal(out, single[:, b * block:(b + 1) * block])
        c.release(yreg)
        c.release(xreg)
        with pytest.raises(capi.InvalidArgument):
            c.release(xreg)
        r.close()
    finally:
        c.close()
    assert np.array_equal(got, want)


def test_render_errors():
    from libear_amd import capi
    dec = decorrelators("0+5+0")
    with pytest.raises(capi.InvalidArgument):
        capi.Renderer(ctx(), 4, 6, 8192, dec, 255)  # too large
    r = capi.Renderer(ctx(), 4, 6, 512, dec, 255, max_blocks=2)
    with pytest.raises(capi.InvalidArgument):
        r.process(np.zeros((4, 512 * 3), np.float32))  # more blocks than max_blocks
    with pytest.raises(capi.InvalidArgument):
        r.set_object_points(9, [0], np.zeros((1, 6)), np.zeros((1, 6)))
    with pytest.raises(capi.InvalidArgument):
        r.set_object_points(0, [5, 1], np.zeros((2, 6)), np.zeros((2, 6)))
    r.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EARHIP_FUZZ_SEEDS", "16")))))
def test_random_scenes_vs_oracle(seed):
    """Randomised shapes: object count (odd, below/above the 32-object chunk of the split-operand kernels),
    layout, block size, call partition, curve families (aligned ramps, ADM-like, ragged, static) and
    call start times; every output within 1e-6 relative RMS of the oracle."""
    rng = np.random.default_rng(1000 + seed)
    layout = ["0+5+0", "4+5+0", "9+10+3"][int(rng.integers(0, 3))]
    n = len(LAYOUTS[layout])
    block = int([128, 256, 512, 1024][int(rng.integers(0, 4))])
    nblocks = int(rng.integers(2, 9))
    m = int(rng.choice([3, 17, 31, 32, 33, 47, 64, 65, 100, 129]))
    total = block * nblocks
    kind = int(rng.integers(0, 5))
    if kind == 0:
        curves = scenes.dense_curves(m, n, block, nblocks, seed=seed)
    elif kind == 1:
        curves = scenes.adm_curves(m, n, total, period=int(rng.integers(100, 900)), ramp=int(rng.integers(1, 99)),
                                   seed=seed)
    elif kind == 2:
        curves = scenes.ragged_curves(m, n, total, seed=seed)
    elif kind == 3:
        curves = scenes.constant_curves(m, n, seed=seed)
    else:  # aligned to 256 but sparse in time: mostly constant with occasional ramps
        curves = []
        for i in range(m):
            grid = np.arange(0, total + 256, 256)
            k = min(int(rng.integers(2, 6)), len(grid))
            t = np.sort(rng.choice(grid, size=k, replace=False)).astype(np.int64)
            curves.append((t, rng.uniform(0, 1, (k, n)).astype(np.float32), rng.uniform(0, 1, (k, n)).astype(np.float32)))
    dec = decorrelators(layout)
    x = scenes.audio(m, total, seed=seed)
    # random partition of the blocks into calls
    calls = []
    left = nblocks
    while left > 0:
        c = int(rng.integers(1, left + 1))
        calls.append(c)
        left -= c
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, calls)
    assert scenes.rel_rms(got, want) <= 1e-6, (layout, block, nblocks, m, kind, calls)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EARHIP_FUZZ_SEEDS", "16")))))
def test_random_aligned_scenes_split_operand_kernels(seed):
    """Randomised scenes the split-operand kernels take (no curve point inside a tile): points on a
    256 or 512 grid with a random mix of ramps, constant stretches and steps (duplicate times), object
    counts around the 32-object chunk, random call partitions (short calls split the objects over
    workgroups), random signal and gain levels, both tiles of the f16x2 kernel."""
    rng = np.random.default_rng(5000 + seed)
    layout = ["0+5+0", "4+5+0", "9+10+3"][int(rng.integers(0, 3))]
    n = len(LAYOUTS[layout])
    block = int([256, 512, 1024][int(rng.integers(0, 3))])
    nblocks = int(rng.integers(2, 9))
    m = int(rng.choice([32, 33, 47, 63, 64, 65, 96, 100, 129, 200]))
    total = block * nblocks
    grid_step = int([256, 512][int(rng.integers(0, 2))])
    grid = np.arange(0, total + grid_step, grid_step)
    glevel = np.float32(10.0 ** rng.uniform(-3, 2))
    curves = []
    for i in range(m):
        k = int(rng.integers(1, min(len(grid), 8) + 1))
        t = np.sort(rng.choice(grid, size=k, replace=False)).astype(np.int64)
        if k >= 3 and rng.random() < 0.3:
            t[1] = t[0]  # a step
        d = (rng.uniform(0, 1, (k, n)) * glevel).astype(np.float32)
        f = (rng.uniform(0, 1, (k, n)) * glevel).astype(np.float32)
        if k >= 2 and rng.random() < 0.4:
            d[-1], f[-1] = d[-2], f[-2]  # a constant stretch
        curves.append((t, d, f))
    dec = decorrelators(layout)
    x = (scenes.audio(m, total, seed=seed) * np.float32(10.0 ** rng.uniform(-5, 2))).astype(np.float32)
    calls = []
    left = nblocks
    while left > 0:
        c = int(rng.integers(1, left + 1))
        calls.append(c)
        left -= c
    tile = ["256", "512", None][int(rng.integers(0, 3))]
    want = run_oracle(curves, x, n, block, dec, 255)
    got = _with_env({"EARHIP_H2_TILE": tile}, lambda: run_hip(curves, x, n, block, dec, 255, calls))
    assert np.isfinite(got).all()
    assert scenes.rel_rms(got, want) <= 1e-6, (layout, block, nblocks, m, grid_step, calls, tile)


@pytest.mark.parametrize("seed", list(range(4)))
def test_random_large_scenes_vs_oracle(seed):
    """Like test_random_scenes_vs_oracle at object counts and stream lengths where the kernels run
    many pipeline groups per wave and short calls split the objects over workgroups."""
    rng = np.random.default_rng(7000 + seed)
    layout, block = "9+10+3", 512
    n = len(LAYOUTS[layout])
    m = int([257, 600, 1023, 1100][seed])
    nblocks = int(rng.integers(12, 33))
    total = block * nblocks
    if seed % 3 == 0:
        curves = scenes.adm_curves(m, n, total, period=int(rng.integers(300, 1500)), ramp=int(rng.integers(10, 290)), seed=seed)
    elif seed % 3 == 1:
        curves = scenes.dense_curves(m, n, block, nblocks, seed=seed)
    else:
        curves = scenes.ragged_curves(m, n, total, seed=seed)
    dec = decorrelators(layout)
    x = scenes.audio(m, total, seed=seed)
    calls, left = [], nblocks
    while left > 0:
        c = int(rng.integers(1, min(left, 16) + 1))
        calls.append(c)
        left -= c
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, calls)
    assert scenes.rel_rms(got, want) <= 1e-6, (m, nblocks, calls)
    whole = run_hip(curves, x, n, block, dec, 255, [nblocks])
    assert scenes.rel_rms(whole, want) <= 1e-6


@pytest.mark.parametrize("block,nblocks,m", [(64, 12, 40), (4096, 2, 24), (2048, 3, 33)])
def test_block_size_limits(block, nblocks, m):
    """smallest and largest block sizes of the fused render (FFT lengths 128 and 8192)"""
    layout = "4+5+0"
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    curves = scenes.adm_curves(m, n, total, period=777, ramp=100, seed=block)
    x = scenes.audio(m, total, seed=block)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks])
    assert scenes.rel_rms(got, want) <= 1e-6
    step = run_hip(curves, x, n, block, dec, 255, [1] * nblocks)
    assert scenes.rel_rms(step, want) <= 1e-6


@pytest.mark.parametrize("block,nblocks,calls,m", [(480, 9, [9], 40), (480, 9, [1, 2, 6], 40), (960, 5, [2, 3], 33), (1920, 3, [3], 24),
                                                   (96, 15, [4, 11], 40), (120, 12, [12], 70), (45, 20, [7, 13], 12),
                                                   (441, 6, [1, 5], 20), (1000, 4, [4], 64), (3000, 2, [1, 1], 24)])
def test_block_sizes_that_are_not_powers_of_two(block, nblocks, calls, m):
    """libear's kissfft factorises any length (submodules/kissfft/kissfft.hh:34-51: radix 4, 2, 3, 5, generic),
    so the renderer's block size need not be a power of two: 10 ms at 48 kHz (480) and its multiples, odd sizes
    (45, 441 = 3^2 7^2 through the generic butterfly).  The mixed-radix K2 kernel and the gain kernels on a block
    grid that no tile size divides, against the oracle; the 512-tap decorrelators span several partitions at the
    small sizes."""
    layout = "4+5+0"
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    curves = scenes.adm_curves(m, n, total, period=700, ramp=min(block // 3 + 5, 300), seed=block)
    x = scenes.audio(m, total, seed=block)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, calls)
    assert scenes.rel_rms(got, want) <= 1e-6
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6
    aligned = scenes.dense_curves(m, n, block, nblocks, seed=block + 1)  # a new gain vector every block
    want = run_oracle(aligned, x, n, block, dec, 255)
    got = run_hip(aligned, x, n, block, dec, 255, calls)
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6


@pytest.mark.parametrize("block,nblocks,calls", [(1024, 5, [2, 3]), (2048, 3, [1, 2]), (4096, 2, [2])])
def test_decorrelator_partition_size_is_the_renderers_choice(block, nblocks, calls):
    """Blocks of 1024, 2048, 4096 samples run their 512-tap decorrelators in 512-sample partitions (the fast
    wave kernel; a linear convolution does not depend on its partitioning), EARHIP_K2_OWN_BLOCK=1 keeps
    libear's own partition size = block size (block_convolver_impl.cpp:16-41): both against the oracle, whose
    BlockConvolver uses the block size, and against each other."""
    layout, m = "9+10+3", 40
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    curves = scenes.adm_curves(m, n, total, period=900, ramp=200, seed=block)
    x = scenes.audio(m, total, seed=block + 3)
    want = run_oracle(curves, x, n, block, dec, 255)
    fast = run_hip(curves, x, n, block, dec, 255, calls)
    own = _with_env({"EARHIP_K2_OWN_BLOCK": "1"}, lambda: run_hip(curves, x, n, block, dec, 255, calls))
    assert scenes.rel_rms_per_channel(fast, want) <= 1e-6
    assert scenes.rel_rms_per_channel(own, want) <= 1e-6
    assert scenes.rel_rms(fast, own) <= 5e-7


@pytest.mark.parametrize("m,nblocks,calls,run", [(24, 37, [37], None), (24, 37, [1, 20, 3, 13], "3"),
                                                 (200, 9, [9], "5"), (24, 40, [40], "1")])
def test_block_512_decorrelator_kernels_agree(m, nblocks, calls, run):
    """block 512 has two K2 kernels: one wave per run of blocks (default) and the workgroup
    kernel (option K2_WG, looked at per launch).  Both against the oracle, over run boundaries (several
    runs per call, odd run lengths through EARHIP_RUN) and over object splits of short calls."""
    layout, block = "4+5+0", 512
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    curves = scenes.adm_curves(m, n, total, period=1500, ramp=300, seed=nblocks)
    x = scenes.audio(m, total, seed=m)
    want = run_oracle(curves, x, n, block, dec, 255)
    outs = []
    for wg in (False, True):
        outs.append(with_options({"EARHIP_K2_WG": "1" if wg else None, "EARHIP_RUN": run},
                                 lambda: run_hip(curves, x, n, block, dec, 255, calls)))
    assert scenes.rel_rms(outs[0], want) <= 1e-6
    assert scenes.rel_rms(outs[1], want) <= 1e-6
    assert scenes.rel_rms(outs[0], outs[1]) <= 1e-6


@pytest.mark.parametrize("tile", ["256", "512"])
@pytest.mark.parametrize("m,layout", [(96, "9+10+3"), (40, "0+5+0"), (130, "4+5+0")])
def test_f16x2_kernel_both_tiles_vs_oracle(tile, m, layout):
    """k_gain_mix_h2 on 256-sample tiles (4 waves) and 512-sample tiles (8 waves; chosen by itself only
    for long calls) against the oracle, block-aligned ramps, object counts that are not multiples of 32."""
    if os.environ.get("EARHIP_MFMA") not in (None, "3", "4"):
        pytest.skip("kernel forced by EARHIP_MFMA")
    block, nblocks = 512, 5
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    curves = scenes.dense_curves(m, n, block, nblocks)
    x = scenes.audio(m, block * nblocks, seed=m)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = _with_env({"EARHIP_H2_TILE": tile}, lambda: run_hip(curves, x, n, block, dec, 255, [nblocks]))
    assert scenes.rel_rms(got, want) <= 1e-6
    parts = _with_env({"EARHIP_H2_TILE": tile}, lambda: run_hip(curves, x, n, block, dec, 255, [2, 1, 2]))
    assert scenes.rel_rms(parts, want) <= 1e-6


@pytest.mark.parametrize("amp,gamp", [(1e-4, 1.0), (300.0, 1.0), (1.0, 2000.0), (1.0, 3e-4), (1e-3, 50.0),
                                      (1e-7, 1.0), (3e4, 1e-3), (1e-9, 1e-6)])
def test_f16x2_kernel_operand_ranges(amp, gamp):
    """The f16x2 kernel scales the gains by their own maximum and the inputs by a power of two from the
    level K0 probes in the call's inputs: 1e-6 holds at any level; inputs beyond the f16 range after the
    prescale (peaks far above the probed level) take the exact in-kernel fallback."""
    if os.environ.get("EARHIP_XSCALE") is not None and not 1e-5 <= amp <= 3.0:
        pytest.skip("fixed input scale (EARHIP_XSCALE): levels far from full scale lose precision by design")
    layout, block, nblocks, m = "4+5+0", 512, 3, 64
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    base = scenes.dense_curves(m, n, block, nblocks)
    curves = [(t, (d * np.float32(gamp)).astype(np.float32), (f * np.float32(gamp)).astype(np.float32))
              for t, d, f in base]
    x = (scenes.audio(m, block * nblocks, seed=7) * np.float32(amp)).astype(np.float32)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks])
    assert np.isfinite(got).all()
    assert scenes.rel_rms(got, want) <= 1e-6


def test_f16x2_kernel_peak_far_above_the_probed_level():
    """a burst 10^4 above the rest of the call, in a sample the level probe does not see: the wave tiles
    that hold it overflow the f16 range and are recomputed exactly"""
    layout, block, nblocks, m = "4+5+0", 512, 3, 64
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    curves = scenes.dense_curves(m, n, block, nblocks)
    x = (scenes.audio(m, block * nblocks, seed=9) * np.float32(1e-3)).astype(np.float32)
    x[:, 700:703] *= np.float32(1e4)   # (the probe reads samples 4 ((m + 5 tile) & 15) .. +3 of a tile)
    want = run_oracle(curves, x, n, block, dec, 255)
    for tile in ("256", "512"):
        got = _with_env({"EARHIP_H2_TILE": tile}, lambda: run_hip(curves, x, n, block, dec, 255, [nblocks]))
        assert np.isfinite(got).all()
        assert scenes.rel_rms(got, want) <= 1e-6


def test_f16x2_kernel_non_finite_input_stays_local():
    """an infinite input sample poisons the outputs of the blocks that see it (through the exact fallback
    of the gain kernel) and at most the block before (K2 transforms two blocks as one complex signal),
    nothing earlier"""
    layout, block, nblocks, m = "0+5+0", 512, 4, 64
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    curves = scenes.dense_curves(m, n, block, nblocks)
    x = scenes.audio(m, block * nblocks, seed=3)
    x[5, 3 * block + 17] = np.inf
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks])
    clean = slice(0, 2 * block)
    assert np.isfinite(got[:, clean]).all()
    assert scenes.rel_rms(got[:, clean], want[:, clean]) <= 1e-6
    assert (~np.isfinite(got[:, 3 * block:])).any()


@pytest.mark.parametrize("layout,tile,wgs", [("4+5+0", "512", "8"), ("4+5+0", "512", "16"), ("0+5+0", "256", "8"), ("9+10+3", "512", "8"),
                                             ("4+5+0", "256", "8")])
def test_grid_kernel_several_tiles_per_workgroup(layout, tile, wgs):
    """The grid kernel works through a workgroup's tiles in one software pipeline (gain_h2.h; the forms with a tile per
    workgroup: gain_h2_t1.h): a call of 41 blocks on 8 or 16 workgroups (EARHIP_H2_WGS), with everything that happens at
    the END of a tile happening in tiles that are not a workgroup's first or last — objects off the grid (exact path
    inside the tile), bursts 10^4 above the probed level (the tile is noted and redone exactly behind the loop) — and a
    ragged last tile."""
    from libear_amd import capi
    if os.environ.get("EARHIP_MFMA") not in (None, "3", "4"):
        pytest.skip("kernel forced by EARHIP_MFMA")
    block, nblocks, m = 512, 41, 128  # (two objects off the grid: the planner's bound for the grid kernel, M / 64)
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks - 0  # (whole blocks; the ragged tile comes from the 256-/512-sample tiles of a 41-block call)
    curves = scenes.dense_curves(m, n, block, nblocks, seed=5)
    odd = scenes.adm_curves(2, n, total, period=700, ramp=150, seed=3)
    curves[17], curves[90] = odd[0], odd[1]
    x = (scenes.audio(m, total, seed=23) * np.float32(1e-3)).astype(np.float32)

    def probed(obj):  # the samples k_level_probe reads of this object's row (gain_kernels.h: 16 runs of 4 float4s)
        nvec, per, nrun = total >> 2, 4, 16
        stretch = nvec // nrun
        out = set()
        for g in range(nrun):
            h = ((obj * 2654435761) & 0xffffffff) ^ (g * 40503)
            room = stretch - per + 1 if stretch > per else 1
            for sub in range(per):
                at = min(g * stretch + ((h * room) >> 32) + sub, nvec - 1)
                out.update(range(4 * at, 4 * at + 4))
        return out

    for b0 in (3, 17, 18, 33, 40):  # bursts in samples the level probe does not read: their tiles overflow the f16 range
        obj, s0 = (b0 * 7) % m, b0 * block + 201
        while probed(obj) & set(range(s0, s0 + 3)):
            s0 += 8
        x[obj, s0:s0 + 3] *= np.float32(1e4)
    want = run_oracle(curves, x, n, block, dec, 255)

    def render():
        r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
        set_renderer_curves(r, curves, True)
        out = r.process(x)
        plan = r.last_plan()
        r.close()
        return out, plan

    got, plan = _with_env({"EARHIP_H2_TILE": tile, "EARHIP_H2_WGS": wgs}, render)
    assert plan["kernel"] == 3 and plan["tile"] == int(tile), plan
    assert np.isfinite(got).all()
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6, (scenes.rel_rms_per_channel(got, want), plan)
    # and with the whole call on its usual number of workgroups: the same values
    got2, _ = _with_env({"EARHIP_H2_TILE": tile}, render)
    assert np.array_equal(got, got2)


@pytest.mark.parametrize("tile", [None, "256", "512"])
def test_mostly_aligned_scene_keeps_the_split_operand_kernel(tile):
    """A few objects whose metadata ignores the block grid (2 of 128 here; up to M / 64) do not move the
    scene to the slot kernel: they take the exact slow path in the tiles where their points fall."""
    from libear_amd import capi
    if os.environ.get("EARHIP_MFMA") not in (None, "3"):
        pytest.skip("kernel forced by EARHIP_MFMA")
    layout, block, nblocks, m = "4+5+0", 512, 6, 128
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    curves = scenes.dense_curves(m, n, block, nblocks)
    odd = scenes.adm_curves(2, n, total, period=700, ramp=150, seed=3)
    curves[17], curves[90] = odd[0], odd[1]
    x = scenes.audio(m, total, seed=21)
    want = run_oracle(curves, x, n, block, dec, 255)

    def render():
        r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
        set_renderer_curves(r, curves, True)
        out = r.process(x)
        kind = r.gain_kernel()
        r.close()
        return out, kind

    got, kind = _with_env({"EARHIP_H2_TILE": tile}, render)
    assert kind == 3
    assert scenes.rel_rms(got, want) <= 1e-6
    # five such objects are more than M / 64: the piece-list kernel takes over (cost proportional to the
    # curve points, whatever their times)
    more = scenes.adm_curves(3, n, total, period=500, ramp=100, seed=4)
    curves[5], curves[40], curves[77] = more[0], more[1], more[2]
    want = run_oracle(curves, x, n, block, dec, 255)
    got, kind = _with_env({"EARHIP_H2_TILE": tile}, render)
    assert kind == 4
    assert scenes.rel_rms(got, want) <= 1e-6


def test_gain_kernel_choice_follows_the_curves():
    """f16x2 kernel (3) for curves without points inside the tiles (block-aligned ramps, static gains), hinge kernel (5)
    for curves that ramp all the time off the tile grid, f16x2 piece-list kernel (4) for the other curves that ignore
    it (ramp, then hold), VALU kernel (0) in strict mode; fewer than 32 objects: f32 slot kernel (1)."""
    from libear_amd import capi
    layout, block, nblocks = "0+5+0", 512, 4
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks

    def kernel_for(m, curves, strict=False, t0=0):
        ctx().set_strict(strict)
        try:
            r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
            set_renderer_curves(r, curves, True)
            r.reset(t0)
            r.process(scenes.audio(m, total))
            k = r.gain_kernel()
            r.close()
        finally:
            ctx().set_strict(False)
        return k

    forced = os.environ.get("EARHIP_MFMA")
    if forced not in (None, "3") or os.environ.get("EARHIP_HINGE") is not None:
        pytest.skip("kernel forced by EARHIP_MFMA / EARHIP_HINGE")
    dense = scenes.dense_curves(64, n, block, nblocks)
    assert kernel_for(64, dense) == 3
    assert kernel_for(64, dense, t0=17) == 5          # same curves, call grid shifted off the points: always ramping
    assert kernel_for(64, dense, strict=True) == 0
    assert kernel_for(16, scenes.dense_curves(16, n, block, nblocks)) == 1   # fewer than 32 objects
    assert kernel_for(64, scenes.adm_curves(64, n, total, seed=1)) == 4
    assert kernel_for(20, scenes.adm_curves(20, n, total, seed=1)) == 1
    assert kernel_for(64, scenes.constant_curves(64, n)) == 3   # static gains: no point inside any tile


@pytest.mark.parametrize("pairs", ["0", "1", None])
@pytest.mark.parametrize("tile", ["256", "512", None])
@pytest.mark.parametrize("kind,m,layout,block,nblocks,calls",
                         [("adm", 64, "9+10+3", 512, 8, [8]), ("adm", 200, "4+5+0", 512, 6, [1, 2, 3]),
                          ("ragged", 100, "9+10+3", 256, 9, [9]), ("dense", 96, "9+10+3", 512, 5, [5]),
                          ("constant", 130, "0+5+0", 1024, 3, [3]), ("short", 64, "9+10+3", 512, 4, [4]),
                          ("adm", 1100, "9+10+3", 512, 12, [5, 7]), ("busy", 70, "4+5+0", 512, 4, [1, 3])])
def test_piece_list_kernel_vs_oracle(pairs, tile, kind, m, layout, block, nblocks, calls):
    """k_gain_mix_p2 forced for every curve family (EARHIP_MFMA=5): metadata that ignores the tile grid,
    irregular curves with steps and dense points, block-aligned ramps, static gains, ramps of a few samples
    deep inside a tile (p0 far outside [0, 1] before the clamp), per channel; at both tile sizes and in both
    layouts of the piece lists (EARHIP_P2_PAIRS: packed, singles + pairs; None: the library's own choice)."""
    from libear_amd import capi
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    if kind == "adm":
        curves = scenes.adm_curves(m, n, total, period=700, ramp=150, seed=m)
    elif kind == "ragged":
        curves = scenes.ragged_curves(m, n, total, seed=m)
    elif kind == "dense":
        curves = scenes.dense_curves(m, n, block, nblocks, seed=m)
    elif kind == "constant":
        curves = scenes.constant_curves(m, n, seed=m)
    elif kind == "busy":  # a few objects with more ramps in one tile than a list takes per object (15; paired lists 7):
        # the exact path; and some with 4-7 (more than the list builder keeps in registers: its second walk)
        curves = scenes.adm_curves(m, n, total, period=700, ramp=150, seed=m)
        rng = np.random.default_rng(m)
        for i, step in ((3, 5), (40, 11), (69, 16), (10, 45), (20, 70), (21, 37)):
            t = np.arange(-40, total + 40, step, dtype=np.int64)
            curves[i] = (t, rng.uniform(0, 1, (len(t), n)).astype(np.float32), rng.uniform(0, 1, (len(t), n)).astype(np.float32))
    else:  # 9-sample ramps at arbitrary times
        curves = scenes.adm_curves(m, n, total, period=333, ramp=9, seed=m)
    x = scenes.audio(m, total, seed=m)
    want = run_oracle(curves, x, n, block, dec, 255)

    def render():
        c = capi.Context(0)  # (the kernel choice is read when a context is created)
        try:
            r = capi.Renderer(c, m, n, block, dec, 255, max_blocks=max(calls))
            set_renderer_curves(r, curves, True)
            out = np.zeros((n, total), np.float32)
            ofs = 0
            for nb in calls:
                out[:, ofs:ofs + nb * block] = r.process(x[:, ofs:ofs + nb * block])
                ofs += nb * block
            plan = r.last_plan()
            r.close()
        finally:
            c.close()
        return out, plan

    got, plan = _with_env({"EARHIP_MFMA": "5", "EARHIP_P2_TILE": tile, "EARHIP_P2_PAIRS": pairs}, render)
    assert plan["kernel"] == 4, plan
    assert np.isfinite(got).all()
    assert scenes.rel_rms(got, want) <= 1e-6, (scenes.rel_rms(got, want), plan)
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6, (scenes.rel_rms_per_channel(got, want), plan)


@pytest.mark.parametrize("gsplit", ["1", None])
@pytest.mark.parametrize("pairs", ["0", "1"])
@pytest.mark.parametrize("tile", ["256", "512"])
@pytest.mark.parametrize("kind,m,block,nblocks,calls",
                         [("adm", 200, 512, 24, [24]), ("adm", 40, 512, 21, [10, 11]), ("ragged", 100, 256, 37, [37]),
                          ("short", 330, 512, 16, [16]), ("constant", 130, 512, 9, [9]), ("mixed", 400, 512, 20, [13, 7])])
def test_piece_list_pipeline_runs_through_a_workgroups_tiles(gsplit, pairs, tile, kind, m, block, nblocks, calls):
    """k_gain_mix_p2 carries its pipeline (piece words, gain rows, B fragments) from one tile's list into the next tile of the
    same workgroup (option P2_WGS: workgroups of the launch; 0: one per tile, nothing carried).  With 1, 3 and 8 workgroups
    for 9-74 tiles every workgroup crosses several list boundaries — odd and even chunk counts, lists too short to be
    carried into (40 objects: two chunks), calls that end inside a tile, lists whose length changes from tile to tile
    ("mixed": a quarter of the objects ramps only in the first third of the call), whole lists per workgroup (GSPLIT 1) and
    the planner's split of a short call's lists into parts: the same bits as a workgroup per tile, and the oracle within 1e-6."""
    from libear_amd import capi
    layout = "9+10+3"
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    if kind == "adm":
        curves = scenes.adm_curves(m, n, total, period=700, ramp=150, seed=m)
    elif kind == "ragged":
        curves = scenes.ragged_curves(m, n, total, seed=m)
    elif kind == "constant":
        curves = scenes.constant_curves(m, n, seed=m)
    elif kind == "mixed":
        curves = scenes.adm_curves(m, n, total, period=700, ramp=150, seed=m)
        const = scenes.constant_curves(m, n, seed=m + 1)
        for i in range(0, m, 4):  # these stop moving after the first third
            t, d, f = curves[i]
            keep = t < total // 3
            curves[i] = (t[keep], d[keep], f[keep]) if keep.sum() >= 2 else const[i]
    else:
        curves = scenes.adm_curves(m, n, total, period=333, ramp=9, seed=m)
    x = scenes.audio(m, total, seed=m)
    want = run_oracle(curves, x, n, block, dec, 255)

    def render():
        c = capi.Context(0)
        try:
            r = capi.Renderer(c, m, n, block, dec, 255, max_blocks=max(calls))
            set_renderer_curves(r, curves, True)
            out = np.zeros((n, total), np.float32)
            ofs = 0
            for nb in calls:
                out[:, ofs:ofs + nb * block] = r.process(x[:, ofs:ofs + nb * block])
                ofs += nb * block
            plan = r.last_plan()
            r.close()
        finally:
            c.close()
        return out, plan

    outs = {}
    for wgs in ("0", "1", "3", "8"):
        got, plan = _with_env({"EARHIP_MFMA": "5", "EARHIP_P2_TILE": tile, "EARHIP_P2_PAIRS": pairs, "EARHIP_P2_WGS": wgs,
                               "EARHIP_TAILCUT": "0", "EARHIP_GSPLIT": gsplit}, render)
        assert plan["kernel"] == 4, plan
        outs[wgs] = got
    assert scenes.rel_rms_per_channel(outs["0"], want) <= 1e-6, (scenes.rel_rms_per_channel(outs["0"], want), plan)
    for wgs in ("1", "3", "8"):
        assert np.array_equal(outs[wgs], outs["0"]), (wgs, float(np.abs(outs[wgs] - outs["0"]).max()))


@pytest.mark.parametrize("block,n_taps,nblocks,calls", [(64, 512, 24, [24]), (128, 512, 10, [1, 2, 7]), (256, 512, 8, [3, 5]),
                                                        (512, 1300, 6, [2, 4]), (1024, 1025, 4, [1, 3]), (256, 700, 9, [9])])
def test_decorrelator_firs_longer_than_a_block(block, n_taps, nblocks, calls):
    """FIRs of several partitions in the fused render (the 512-tap decorrelators at blocks of 64..256, longer
    custom FIRs at any block size): one K2 launch per partition over the diffuse bus delayed by that many
    blocks, history and per-partition overlap-add tails carried across calls — against the oracle's
    partitioned BlockConvolver (block_convolver_impl.cpp:16-41,193-215)."""
    layout, m = "4+5+0", 40
    n = len(LAYOUTS[layout])
    if n_taps == 512:
        dec = decorrelators(layout)
    else:
        dec = (np.random.default_rng(n_taps).uniform(-1, 1, (n, n_taps)) * np.exp(-np.arange(n_taps) / 300.0)).astype(np.float32)
    total = block * nblocks
    curves = scenes.adm_curves(m, n, total, period=500, ramp=120, seed=block)
    x = scenes.audio(m, total, seed=n_taps)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, calls)
    assert scenes.rel_rms(got, want) <= 1e-6
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6
    whole = run_hip(curves, x, n, block, dec, 255, [nblocks])
    assert scenes.rel_rms(whole, want) <= 1e-6


def test_limits_of_the_curve_store():
    """the most points an object may carry (2^18 - 1, one every 3 samples: 1536 blocks in one call) beside an
    ordinary object, against the oracle; one point more, a ramp of 2^31 samples: invalid_argument"""
    from libear_amd import capi
    layout, block = "0+5+0", 512
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    npts = (1 << 18) - 1  # (the segment index field of a descriptor: 18 bits)
    nblocks = (3 * npts + block - 1) // block
    total = block * nblocks
    rng = np.random.default_rng(4)
    t0 = np.arange(npts, dtype=np.int64) * 3
    d0 = rng.uniform(0, 1, (npts, n)).astype(np.float32)
    f0 = rng.uniform(0, 1, (npts, n)).astype(np.float32)
    d0[1000:1010] = d0[999]  # (a few constant stretches)
    t1 = np.array([0, 700, 700, total - 5], np.int64)
    d1 = rng.uniform(0, 1, (4, n)).astype(np.float32)
    f1 = rng.uniform(0, 1, (4, n)).astype(np.float32)
    x = scenes.audio(2, total, seed=6)
    r = capi.Renderer(ctx(), 2, n, block, dec, 255, max_blocks=nblocks)
    r.set_object_points(0, t0, d0, f0)
    r.set_object_points(1, t1, d1, f1)
    got = r.process(x)
    with pytest.raises(capi.InvalidArgument):
        r.set_object_points(0, np.arange(npts + 1, dtype=np.int64), np.zeros((npts + 1, n), np.float32),
                            np.zeros((npts + 1, n), np.float32))
    with pytest.raises(capi.InvalidArgument):
        r.set_object_points(0, [0, 1 << 31], np.zeros((2, n)), np.zeros((2, n)))
    r.close()
    w = _oracle.ObjectsRenderer(2, n, block, dec, 255)
    for i, (t, d, f) in enumerate(((t0, d0, f0), (t1, d1, f1))):
        w.set_points(i, 0, t, d)
        w.set_points(i, 1, t, f)
    want = w.process(x)
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6


@pytest.mark.parametrize("start", [(1 << 40) + 12345, -(1 << 33) - 7])
@pytest.mark.parametrize("kind", ["dense", "adm"])
def test_sample_times_far_from_zero(start, kind):
    """SampleIndex is a long (gain_interpolator.hpp:16): the same scene shifted by 2^40 samples (and to negative
    times) gives the same bits — only differences of times enter the arithmetic (:225-227)"""
    from libear_amd import capi
    m, layout, block, nblocks = 48, "4+5+0", 512, 6
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    curves = scenes.dense_curves(m, n, block, nblocks, seed=3) if kind == "dense" else scenes.adm_curves(m, n, total, seed=3)
    x = scenes.audio(m, total, seed=8)
    outs = []
    for shift in (0, start):
        r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
        for i, (t, d, f) in enumerate(curves):
            r.set_object_points(i, np.asarray(t, np.int64) + shift, d, f)
        r.reset(shift)
        outs.append(r.process(x))
        r.close()
    assert np.array_equal(outs[0], outs[1])


def test_contexts_on_concurrent_threads():
    """libear's objects are single-owner, but different objects may live on different threads: four threads, each
    with its own context (its own stream), renderer and interpolation-policy calls (a table shared by all
    contexts sits behind those), working at the same time give the bits a single thread gives"""
    import threading
    from libear_amd import capi
    layout, block, nblocks = "4+5+0", 512, 8
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks

    def work(seed, context, rounds):
        m = 24 + 8 * seed
        curves = scenes.adm_curves(m, n, total, seed=seed) if seed % 2 else scenes.dense_curves(m, n, block, nblocks, seed=seed)
        x = scenes.audio(m, total, seed=10 + seed)
        r = capi.Renderer(context, m, n, block, dec, 255, max_blocks=nblocks)
        set_renderer_curves(r, curves)
        outs = []
        rng = np.random.default_rng(seed)
        sp, ep = rng.uniform(0, 1, (3, n)).astype(np.float32), rng.uniform(0, 1, (3, n)).astype(np.float32)
        xi = scenes.audio(3, 700, seed=20 + seed)
        for _ in range(rounds):
            r.reset(0)
            outs.append(r.process(x))
            o = np.zeros((n, 700), np.float32)
            context.apply_interp(xi, o, 10, 650, 1000, 990, 1800, sp, ep)  # (LinearInterpMatrix policy)
            outs.append(o)
        r.close()
        return outs

    serial = [work(s, ctx(), 1) for s in range(4)]
    results, errors = [None] * 4, []

    def run(s):
        try:
            c = capi.Context(0)
            results[s] = work(s, c, 6)
            c.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=run, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for s in range(4):
        for k, o in enumerate(results[s]):
            ref = serial[s][k % 2]
            if not np.array_equal(o, ref):  # (say where: which thread, which call, which channels and samples, by how much)
                d = np.abs(o - ref)
                ch, sm = np.nonzero(d.max(axis=1))[0], np.nonzero(d.max(axis=0))[0]
                raise AssertionError(f"thread {s} output {k} ({'renderer' if k % 2 == 0 else 'policy'}): max |diff| {d.max():.3e} of {np.abs(ref).max():.2f}, "
                                     f"channels {ch.tolist()}, samples {sm.min()}..{sm.max()} ({len(sm)} differ), nan {int(np.isnan(o).sum())}")


def test_one_context_shared_by_renderers_of_growing_size():
    """One context, gain stages of different object counts, both list-building and grid kernels: the per-object
    level words the probes leave behind must fit the largest stage and must not leak between the kernels' forms
    (round-3 review: a stage of M objects sized them, a later stage of up to 2 M + 64 wrote past the end)."""
    layout, block, nblocks = "4+5+0", 512, 4
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks

    def check(m, shift):
        curves = [(t + shift, d, f) for t, d, f in scenes.dense_curves(m, n, block, nblocks, seed=m)]
        x = scenes.audio(m, total, seed=m)
        got = run_hip(curves, x, n, block, dec, 255, [nblocks])
        assert scenes.rel_rms_per_channel(got, run_oracle(curves, x, n, block, dec, 255)) <= 1e-6, (m, shift)

    check(96, 37)    # piece lists, small
    check(96, 0)     # grid kernel right after them, same size
    check(200, 37)   # piece lists: > M + 16 of the first stage, <= 2 M + 64
    check(520, 37)   # piece lists, far beyond
    check(520, 0)
    check(64, 0)     # and smaller again
    check(64, 37)


def test_render_block_size_with_a_large_prime_factor():
    """block 1019 (prime): the transform of 2038 points is one radix-2 pass and kissfft's generic butterfly of
    radix 1019 (kissfft.hh:321-352) — every output a sum of 1019 products, thousands of times the work of a
    neighbouring size, and the only path such sizes have: checked through the fused render against the oracle"""
    layout, block, nblocks, m = "0+5+0", 1019, 3, 5
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    curves = scenes.dense_curves(m, n, block, nblocks)
    x = scenes.audio(m, block * nblocks)
    want = run_oracle(curves, x, n, block, dec, 255)
    got = run_hip(curves, x, n, block, dec, 255, [nblocks])
    # (every output of that butterfly is a float32 sum of 1019 products: the CPU path itself is ~1e-6 from a float64
    # render here, so the bar is the float64 render — the GPU no further from it than the CPU path)
    truth = scenes.render_f64(curves, x, n, dec, 255)
    e_gpu, e_cpu = scenes.rel_rms_per_channel(got, truth), scenes.rel_rms_per_channel(want, truth)
    assert e_cpu <= 1e-5, e_cpu
    assert e_gpu <= max(1.25 * e_cpu, 1e-6), (e_gpu, e_cpu)
    assert scenes.rel_rms_per_channel(got, want) <= e_gpu + e_cpu


@pytest.mark.parametrize("kind,m,layout,block,nblocks,calls",
                         [("moving", 64, "9+10+3", 512, 8, [8]), ("moving", 200, "4+5+0", 512, 6, [1, 2, 3]),
                          ("moving", 1100, "9+10+3", 512, 12, [5, 7]), ("moving-300", 96, "9+10+3", 256, 9, [9]),
                          ("moving-130", 70, "0+5+0", 512, 5, [5]), ("adm", 64, "9+10+3", 512, 8, [3, 5]),
                          ("ragged", 100, "9+10+3", 256, 9, [9]), ("dense", 96, "9+10+3", 512, 5, [5]),
                          ("constant", 130, "0+5+0", 1024, 3, [3]), ("short", 64, "9+10+3", 512, 4, [4]),
                          ("mixed", 160, "4+5+0", 512, 7, [7]), ("moving", 40, "9+10+3", 100, 13, [13])])
@pytest.mark.parametrize("hg_tile", [512, 256])
def test_hinge_kernel_vs_oracle(kind, m, layout, block, nblocks, calls, hg_tile):
    """k_gain_mix_hg forced for every curve family (EARHIP_MFMA=6), in both its forms (8 waves on 512-sample tiles with
    up to two kinks on either side of a tile's centre, 4 waves on 256 with one): curves that ramp all the time with their points
    off the tile grid (what it is for: a line per object and tile plus a hinge per curve point), long ramps with holds,
    block-aligned ramps and static gains (no hinges at all), and curves it can only send through its exact path (short
    ramps, steps, dense points), per channel.  `mixed`: every fourth object of a moving scene on ADM-like short ramps."""
    from libear_amd import capi
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    if kind == "moving":
        curves = scenes.adm_curves(m, n, total, period=240, ramp=240, seed=m)
    elif kind == "moving-300":
        curves = scenes.adm_curves(m, n, total, period=300, ramp=300, seed=m)
    elif kind == "moving-130":  # ramps barely long enough (128): two points per tile for most objects
        curves = scenes.adm_curves(m, n, total, period=130, ramp=130, seed=m)
    elif kind == "adm":
        curves = scenes.adm_curves(m, n, total, period=700, ramp=150, seed=m)
    elif kind == "ragged":
        curves = scenes.ragged_curves(m, n, total, seed=m)
    elif kind == "dense":
        curves = scenes.dense_curves(m, n, block, nblocks, seed=m)
    elif kind == "constant":
        curves = scenes.constant_curves(m, n, seed=m)
    elif kind == "mixed":
        curves = scenes.adm_curves(m, n, total, period=240, ramp=240, seed=m)
        odd = scenes.adm_curves(m, n, total, period=333, ramp=9, seed=m + 1)
        for i in range(0, m, 4):
            curves[i] = odd[i]
    else:  # 9-sample ramps at arbitrary times: all of them the exact path
        curves = scenes.adm_curves(m, n, total, period=333, ramp=9, seed=m)
    x = scenes.audio(m, total, seed=m)
    want = run_oracle(curves, x, n, block, dec, 255)

    def render():
        c = capi.Context(0)  # (the kernel choice is read when a context is created)
        try:
            r = capi.Renderer(c, m, n, block, dec, 255, max_blocks=max(calls))
            set_renderer_curves(r, curves, True)
            out = np.zeros((n, total), np.float32)
            ofs = 0
            for nb in calls:
                out[:, ofs:ofs + nb * block] = r.process(x[:, ofs:ofs + nb * block])
                ofs += nb * block
            plan = r.last_plan()
            r.close()
        finally:
            c.close()
        return out, plan

    got, plan = _with_env({"EARHIP_MFMA": "6", "EARHIP_HG_TILE": str(hg_tile)}, render)
    assert plan["kernel"] == 5 and plan["tile"] == hg_tile, plan
    assert np.isfinite(got).all()
    assert scenes.rel_rms(got, want) <= 1e-6, (scenes.rel_rms(got, want), plan)
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6, (scenes.rel_rms_per_channel(got, want), plan)


def test_hinge_kernel_is_chosen_for_always_ramping_curves_off_the_grid():
    """plan_mix: curves that ramp all the time in ramps of half a tile or more, points off the grid -> the hinge kernel;
    the same points on the block grid -> the grid kernel; ADM-like hold-and-ramp metadata -> the piece lists"""
    from libear_amd import capi
    if any(os.environ.get(k) is not None for k in ("EARHIP_MFMA", "EARHIP_HINGE")):
        pytest.skip("kernel forced")
    layout, block, nblocks, m = "9+10+3", 512, 6, 64
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    x = scenes.audio(m, total)
    for curves, kernel in ((scenes.adm_curves(m, n, total, period=240, ramp=240, seed=3), 5),
                           (scenes.dense_curves(m, n, block, nblocks), 3),
                           (scenes.adm_curves(m, n, total, period=960, ramp=240, seed=3), 4)):
        r = capi.Renderer(ctx(), m, n, block, dec, 255, max_blocks=nblocks)
        set_renderer_curves(r, curves, True)
        got = r.process(x)
        plan = r.last_plan()
        r.close()
        assert plan["kernel"] == kernel, plan
        assert scenes.rel_rms_per_channel(got, run_oracle(curves, x, n, block, dec, 255)) <= 1e-6


@pytest.mark.parametrize("force", [None, "6"])
def test_curves_updated_object_by_object_between_block_mode_calls(force):
    """(force = 6: on the hinge kernel, whose kink rows — a row per curve point derived on the device behind every upload —
    have to follow the same updates.)  interp_points is a per-object vector the caller changes freely (gain_interpolator.hpp:42-43): between block-mode
    calls a few objects get a new window of curve points (longer ones too: their region of the curve image moves), most
    keep theirs — the device image follows object by object (CurveSet::commit uploads what changed), the stream equals
    the oracle's with the same updates, and a renderer whose objects are ALL set again gives the same output bit for bit."""
    from libear_amd import capi
    layout, block, nblocks, m = "4+5+0", 256, 14, 96
    n = len(LAYOUTS[layout])
    dec = decorrelators(layout)
    total = block * nblocks
    x = scenes.audio(m, total, seed=5)
    rng = np.random.default_rng(17)

    def window(first_block, length, shift):
        t = (first_block + np.arange(length, dtype=np.int64)) * block + shift
        return t, rng.uniform(0, 1, (length, n)).astype(np.float32), rng.uniform(0, 1, (length, n)).astype(np.float32)

    cur = [window(0, 6, 13 * (i % 3)) for i in range(m)]  # a third of the objects on the block grid, the others off it
    if force is not None and os.environ.get("EARHIP_MFMA") is not None:
        pytest.skip("kernel forced by EARHIP_MFMA")
    c_own = _with_env({"EARHIP_MFMA": force}, lambda: capi.Context(0)) if force else None  # (the choice is read at context creation)
    c_use = c_own if c_own is not None else ctx()
    r = capi.Renderer(c_use, m, n, block, dec, 255, max_blocks=1)
    full = capi.Renderer(c_use, m, n, block, dec, 255, max_blocks=1)
    o = _oracle.ObjectsRenderer(m, n, block, dec, 255)
    for i, (t, d, f) in enumerate(cur):
        r.set_object_points(i, t, d, f)
        o.set_points(i, 0, t, d)
        o.set_points(i, 1, t, f)
    got = np.zeros((n, total), np.float32)
    got_full = np.zeros_like(got)
    want = np.zeros_like(got)
    for b in range(nblocks):
        if b > 0:  # replace 5 objects' windows: same length, longer (the region moves), shorter
            for j in range(5):
                i = (7 * b + 11 * j) % m
                cur[i] = window(b, [6, 40, 3, 9, 6][j], 13 * (i % 3))
                t, d, f = cur[i]
                r.set_object_points(i, t, d, f)
                o.set_points(i, 0, t, d)
                o.set_points(i, 1, t, f)
        for i, (t, d, f) in enumerate(cur):
            full.set_object_points(i, t, d, f)
        sl = slice(b * block, (b + 1) * block)
        got[:, sl] = r.process(x[:, sl])
        got_full[:, sl] = full.process(x[:, sl])
        want[:, sl] = o.process(x[:, sl])
    kernel = r.last_plan()["kernel"]
    r.close()
    full.close()
    if c_own is not None:
        c_own.close()
        assert kernel == 5, kernel
    assert np.array_equal(got, got_full)
    assert scenes.rel_rms_per_channel(got, want) <= 1e-6
