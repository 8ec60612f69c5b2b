"""Synthetic ADM-render scenes shared by the GPU parity tests, the smoke check and bench.py.

All inputs are seeded (numpy PCG64).  A scene = per-object gain curves (times + direct/diffuse
gain vectors) + input audio.  `dense` is the worst case SURVEY §8(d) asks for (every block a
full-length ramp between uniform(0,1) gain vectors on every loudspeaker); `sparse` is the
realistic case (3 non-zero loudspeakers per object, power-normalised, sqrt(1-d)/sqrt(d)
direct/diffuse split, zero LFE columns); `ragged` stresses the segment walk (points inside
blocks, steps from duplicate times, objects with a single point, constant stretches).
"""
import numpy as np


def audio(n_obj, n_samples, seed=1234):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1.0, 1.0, (n_obj, n_samples)).astype(np.float32)


def dense_curves(n_obj, n_out, block, n_blocks, seed=7, t0=0):
    """points at every block boundary: each block is one full-length ramp."""
    rng = np.random.default_rng(seed)
    times = t0 + block * np.arange(n_blocks + 1, dtype=np.int64)
    curves = []
    for _ in range(n_obj):
        d = rng.uniform(0.0, 1.0, (n_blocks + 1, n_out)).astype(np.float32)
        f = rng.uniform(0.0, 1.0, (n_blocks + 1, n_out)).astype(np.float32)
        curves.append((times, d, f))
    return curves


def constant_curves(n_obj, n_out, seed=8):
    rng = np.random.default_rng(seed)
    curves = []
    for _ in range(n_obj):
        d = rng.uniform(0.0, 1.0, (1, n_out)).astype(np.float32)
        f = rng.uniform(0.0, 1.0, (1, n_out)).astype(np.float32)
        curves.append((np.zeros(1, np.int64), d, f))
    return curves


def sparse_curves(n_obj, n_out, block, n_blocks, lfe=(), seed=9):
    """VBAP-like: 3 active loudspeakers per point, unit power, diffuseness d in {0, .5, 1}."""
    rng = np.random.default_rng(seed)
    times = block * np.arange(n_blocks + 1, dtype=np.int64)
    speakers = [c for c in range(n_out) if c not in lfe]
    curves = []
    for _ in range(n_obj):
        d = np.zeros((n_blocks + 1, n_out), np.float32)
        f = np.zeros((n_blocks + 1, n_out), np.float32)
        for k in range(n_blocks + 1):
            idx = rng.choice(speakers, 3, replace=False)
            g = rng.uniform(0.1, 1.0, 3)
            g /= np.sqrt(np.sum(g * g))
            diff = rng.choice([0.0, 0.5, 1.0])
            d[k, idx] = (g * np.sqrt(1.0 - diff)).astype(np.float32)
            f[k, idx] = (g * np.sqrt(diff)).astype(np.float32)
        curves.append((times, d, f))
    return curves


def ragged_curves(n_obj, n_out, total, seed=10):
    """irregular point times incl. duplicates (steps), equal neighbours (constant), single points."""
    rng = np.random.default_rng(seed)
    curves = []
    for m in range(n_obj):
        kind = m % 5
        if kind == 0:  # one point
            t = np.array([rng.integers(0, total)], np.int64)
        elif kind == 1:  # a few random points, may fall anywhere (incl. inside tiles)
            t = np.sort(rng.integers(-50, total + 50, rng.integers(2, 9))).astype(np.int64)
        elif kind == 2:  # steps: duplicated times
            base = np.sort(rng.integers(0, total, 3)).astype(np.int64)
            t = np.sort(np.concatenate([base, base[:2]]))
        elif kind == 3:  # dense points, closer than a tile
            t = np.sort(rng.integers(0, total, 40)).astype(np.int64)
        else:  # long ramp that starts before 0 and ends after the end
            t = np.array([-1000, total + 777], np.int64)
        n = len(t)
        d = rng.uniform(0.0, 1.0, (n, n_out)).astype(np.float32)
        f = rng.uniform(0.0, 1.0, (n, n_out)).astype(np.float32)
        if kind == 3:  # some equal neighbours -> constant segments, per bus
            d[5] = d[4]
            f[9] = f[8]
            d[20] = d[19]
            f[20] = f[19]
        curves.append((t, d, f))
    return curves


def adm_curves(n_obj, n_out, total, period=960, ramp=240, seed=11):
    """ADM-like metadata that ignores the render block grid: every `period` samples (20 ms at
    48 kHz), at a per-object phase, an object gets new gain vectors, reaches them over `ramp`
    samples (interpolationLength) and then holds them until its next metadata block."""
    rng = np.random.default_rng(seed)
    curves = []
    for m in range(n_obj):
        phase = int(rng.integers(0, period))
        starts = np.arange(phase - period, total + period, period, dtype=np.int64)
        t = np.empty(2 * len(starts), np.int64)
        t[0::2] = starts
        t[1::2] = starts + ramp
        vals_d = rng.uniform(0.0, 1.0, (len(starts) + 1, n_out)).astype(np.float32)
        vals_f = rng.uniform(0.0, 1.0, (len(starts) + 1, n_out)).astype(np.float32)
        d = np.empty((2 * len(starts), n_out), np.float32)
        f = np.empty((2 * len(starts), n_out), np.float32)
        d[0::2] = vals_d[:-1]  # block start: still the previous target
        d[1::2] = vals_d[1:]   # end of the ramp: the new target, held until the next block
        f[0::2] = vals_f[:-1]
        f[1::2] = vals_f[1:]
        curves.append((t, d, f))
    return curves


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def render_f64(curves, x, n_out, decorrelators=None, delay=0):
    """float64 evaluation of the composed render (truth for error budgets; small sizes only)."""
    n_obj, total = x.shape
    t = np.arange(total)
    direct = np.zeros((n_out, total))
    diffuse = np.zeros((n_out, total))
    for m, (times, d, f) in enumerate(curves):
        for bus, (g, acc) in enumerate(((d, direct), (f, diffuse))):
            if g is None:
                continue
            k = np.searchsorted(times, t, side="right")  # segment index per sample
            n = len(times)
            km1 = np.clip(k - 1, 0, n - 1)
            kk = np.clip(k, 0, n - 1)
            inside = (k > 0) & (k < n)
            span = np.where(inside, (times[kk] - times[km1]).astype(np.float64), 1.0)
            span[span == 0] = 1.0
            p = np.where(inside, (t - times[km1]) / span, 0.0)
            gs = g[np.where(k == 0, 0, km1)].astype(np.float64)
            ge = g[kk].astype(np.float64)
            gain = gs * (1 - p)[:, None] + ge * p[:, None]
            gain[k == 0] = g[0]
            gain[k == n] = g[n - 1]
            acc += (gain * x[m].astype(np.float64)[:, None]).T
    if decorrelators is None:
        return direct
    out = np.zeros((n_out, total))
    for c in range(n_out):
        dec = np.convolve(diffuse[c], decorrelators[c].astype(np.float64))[:total]
        dl = np.concatenate([np.zeros(delay), direct[c]])[:total]
        out[c] = dec + dl
    return out


def rel_rms_per_channel(a, b):
    """largest per-channel relative RMS error: max_c |a_c - b_c| / |b_c| (a silent reference channel
    must be matched exactly: inf otherwise)"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    num = np.linalg.norm(a - b, axis=1)
    den = np.linalg.norm(b, axis=1)
    worst = 0.0
    for n_, d_ in zip(num, den):
        if d_ == 0.0:
            worst = max(worst, 0.0 if n_ == 0.0 else np.inf)
        else:
            worst = max(worst, n_ / d_)
    return float(worst)


def window_curves(curves, t_lo, t_hi):
    """the curves restricted to what [t_lo, t_hi) can see (the points inside plus one on each side),
    with times relative to t_lo: rendering them from time 0 equals rendering the originals from t_lo"""
    out = []
    for t, d, f in curves:
        t = np.asarray(t, np.int64)
        lo = max(int(np.searchsorted(t, t_lo, side="right")) - 1, 0)
        hi = min(int(np.searchsorted(t, t_hi, side="left")) + 1, len(t))
        hi = max(hi, lo + 1)
        out.append((t[lo:hi] - t_lo, d[lo:hi], None if f is None else f[lo:hi]))
    return out


def mixed_level_sparse(n_obj, n_out, block, n_blocks, lfe=(), span_db=100.0, seed=21):
    """Realistic panning (3 loudspeakers per object, unit power, sqrt(1-d)/sqrt(d) split, zero LFE
    columns) with object LEVELS spread over 0 .. -span_db: the loud objects (above -40 dB) only feed the
    first half of the loudspeakers, so the second half carries quiet objects alone.  Returns
    (curves, levels); the levels are meant for the signals or the gains, as the test chooses."""
    rng = np.random.default_rng(seed)
    times = block * np.arange(n_blocks + 1, dtype=np.int64)
    speakers = [c for c in range(n_out) if c not in lfe]
    first, second = speakers[:len(speakers) // 2], speakers[len(speakers) // 2:]
    levels = (10.0 ** (-rng.uniform(0.0, span_db, n_obj) / 20.0)).astype(np.float32)
    levels[0] = 1.0
    curves = []
    for m in range(n_obj):
        pool = first if levels[m] > 1e-2 else second
        d = np.zeros((n_blocks + 1, n_out), np.float32)
        f = np.zeros((n_blocks + 1, n_out), np.float32)
        for k in range(n_blocks + 1):
            idx = rng.choice(pool, min(3, len(pool)), replace=False)
            g = rng.uniform(0.1, 1.0, len(idx))
            g /= np.sqrt(np.sum(g * g))
            diff = rng.choice([0.0, 0.5, 1.0])
            d[k, idx] = (g * np.sqrt(1.0 - diff)).astype(np.float32)
            f[k, idx] = (g * np.sqrt(diff)).astype(np.float32)
        curves.append((times, d, f))
    return curves, levels


def object_levels(n_obj, span_db=90.0, late_share=0.3, seed=77):
    """per-object signal levels, log-uniform over 0 .. -span_db (object 0 at full scale), and the mask of the objects
    that are digitally silent for the first half of a call (entries): what a real mix looks like at any instant"""
    rng = np.random.default_rng(seed)
    levels = (10.0 ** (-rng.uniform(0.0, span_db, n_obj) / 20.0)).astype(np.float32)
    levels[0] = 1.0
    late = rng.uniform(0.0, 1.0, n_obj) < late_share
    late[0] = False
    return levels, late


def moving_sources(n_obj, total, period=960, seed=31, phase=None, ramp=None):
    """ADM-like trajectories: every `period` samples, at a per-object phase, an object gets a new position
    (a random walk in azimuth / elevation) and diffuseness, reached over a quarter of the period and then
    held.  Returns per object (azimuth, elevation, diffuse, times) arrays of its curve points: the values
    feed a gain producer, the result is a gain curve like adm_curves()."""
    rng = np.random.default_rng(seed)
    az, el, df, times = [], [], [], []
    ramp = max(period // 4, 1) if ramp is None else ramp
    for m in range(n_obj):
        ph = int(rng.integers(0, period)) if phase is None else phase
        starts = np.arange(ph - period, total + period, period, dtype=np.int64)
        k = len(starts) + 1
        a = np.cumsum(rng.normal(0.0, 25.0, k)) + rng.uniform(-180, 180)
        a = (a + 180.0) % 360.0 - 180.0
        e = np.clip(np.cumsum(rng.normal(0.0, 10.0, k)) + rng.uniform(-30, 60), -90, 90)
        d = rng.choice([0.0, 0.0, 0.3, 1.0], k)
        t = np.empty(2 * len(starts), np.int64)
        t[0::2] = starts
        t[1::2] = starts + ramp
        idx = np.empty(2 * len(starts), np.int64)
        idx[0::2] = np.arange(len(starts))      # block start: still the previous target
        idx[1::2] = np.arange(len(starts)) + 1  # end of the ramp: the new target
        az.append(a[idx]); el.append(e[idx]); df.append(d[idx]); times.append(t)
    return az, el, df, times


def bursty_levels(n_obj, n_blocks, solo=8, seed=91):
    """Per-object, per-block signal LEVELS (linear factors, [n_obj][n_blocks]) of non-stationary audio: holds at
    0 .. -30 dB, fades of 3 dB per block down to -90 dB, exact silence, gated bursts (every other block digitally
    silent), single blocks 40 dB ABOVE the level around them.  The first `solo` objects follow a fixed programme that is
    loud at the start of a call and then spends most of it 80 dB down: loud for 16 blocks, -80 dB up to the middle,
    one block at +40 dB, 24 blocks of exact silence, a fade from 0 dB to -90 dB, -90 dB to the end (with solo_curves()
    each of them has a loudspeaker to itself: whatever a level decision gets wrong shows there, in the quiet stretches)."""
    rng = np.random.default_rng(seed)
    lv = np.zeros((n_obj, n_blocks), np.float32)
    db = lambda v: np.float32(10.0 ** (v / 20.0))
    for m in range(n_obj):
        if m < solo:
            mid = n_blocks // 2
            prog = np.full(n_blocks, db(-80.0), np.float32)
            prog[:16] = 1.0
            prog[mid] = db(40.0)
            prog[mid + 1:mid + 25] = 0.0
            fade = np.arange(max(n_blocks - (mid + 25), 0), dtype=np.float64)
            prog[mid + 25:] = np.maximum(10.0 ** (-3.0 * fade / 20.0), 10.0 ** (-90.0 / 20.0)).astype(np.float32)
            lv[m] = prog
            continue
        b = 0
        level = -rng.uniform(0.0, 30.0)
        while b < n_blocks:
            kind = rng.integers(0, 5)
            if kind == 0:  # hold
                n = int(rng.integers(8, 40))
                lv[m, b:b + n] = db(level)
            elif kind == 1:  # fade to -90 dB, then silence
                n = int(np.ceil((90.0 + level) / 3.0))
                steps = level - 3.0 * np.arange(n)
                k = min(n, n_blocks - b)
                lv[m, b:b + k] = (10.0 ** (steps[:k] / 20.0)).astype(np.float32)
                n += int(rng.integers(4, 20))  # (the rest stays exactly zero)
                level = -rng.uniform(0.0, 30.0)
            elif kind == 2:  # gated bursts
                n = 2 * int(rng.integers(5, 12))
                seg = lv[m, b:b + n]
                seg[0::2] = db(level)
            elif kind == 3:  # silence
                n = int(rng.integers(4, 30))
            else:  # one block 40 dB above, then back
                n = int(rng.integers(4, 12))
                lv[m, b:b + n] = db(level)
                lv[m, b] = db(level + 40.0)
            b += n
    return lv


def solo_curves(curves, n_out, solo_speakers, seed=92):
    """give the first len(solo_speakers) objects of a scene a loudspeaker each to themselves: their gains live on that
    column only (both buses), everybody else's gains on those columns are zero; the point times stay the scene's"""
    rng = np.random.default_rng(seed)
    out = []
    for m, (t, d, f) in enumerate(curves):
        d, f = d.copy(), (None if f is None else f.copy())
        if m < len(solo_speakers):
            g = rng.uniform(0.5, 1.0, (len(t), 2)).astype(np.float32)
            for k in range(1, len(t)):  # (metadata that restates a point — same time — restates its gains: no steps)
                if t[k] == t[k - 1]:
                    g[k] = g[k - 1]
            d[:] = 0.0
            d[:, solo_speakers[m]] = g[:, 0]
            if f is not None:
                f[:] = 0.0
                f[:, solo_speakers[m]] = g[:, 1]
        else:
            d[:, solo_speakers] = 0.0
            if f is not None:
                f[:, solo_speakers] = 0.0
        out.append((t, d, f))
    return out
