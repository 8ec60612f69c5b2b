#!/usr/bin/env python3
"""bench.py — the reference's headline workload on MI355X:

    N objects -> 9+10+3 (22.2, 24 ch), block 512, 48 kHz: ramped direct+diffuse gain matrices,
    24 decorrelators (512-tap FIR, FFT 1024), 255-sample compensation delay, mix-down.

One "step" = one stream-mode pass of the hot path over T consecutive blocks (default 1024, i.e.
10.9 s of audio), inputs and gain curves already resident in HBM.  Every block is a full-length ramp
between dense uniform(0,1) gain vectors (worst case).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3

`--config C2|C3|C4|C5` selects a BASELINE.json configuration (default C4 = the one the metric is quoted
on: 1024 objects in total); `--objects/--layout/--block-size/--blocks/--buses/--hoa` override single
fields.  Rank 0 prints ONE JSON line.  `value` = object-samples consumed per second over all GPUs
(Msamples/s = M*B*T / t_step / 1e6); `rtf` = real-time factor (B*T/48000) / t_step.
Multi-GPU: the scene's objects are sharded over the ranks (BASELINE config 4: strong scaling), each
rank renders its shard and the partial loudspeaker buses are summed by one RCCL reduce-scatter over the
channel axis, overlapped with the next step's render; the weak-scaling figure (--objects per rank) is
reported in the same line under `weak_scaling`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3  # vector / f32-MFMA peak, same guide
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 / f16 MFMA peak, same guide
SAMPLE_RATE = 48000.0

PRESETS = {  # BASELINE.json `configs` (C2..C5)
    # (blocks per step: every configuration streams about the same number of object-blocks per step as the
    # headline, 2^20 — a 64-object call of 1024 blocks is 40 us of GPU time, a fifth of it launch overhead)
    "C2": dict(objects=64, hoa=0, layout="4+5+0", block_size=512, buses=1, blocks=8192),
    "C3": dict(objects=256, hoa=0, layout="9+10+3", block_size=512, buses=2, blocks=4096),
    "C4": dict(objects=1024, hoa=0, layout="9+10+3", block_size=512, buses=2, blocks=1024),
    "C5": dict(objects=512, hoa=16, layout="9+10+3", block_size=1024, buses=2, blocks=512),
}


def algorithmic_bytes(m, n, b, k, m_static=0):
    """bytes_gain, bytes_dec, bytes_delay_mix per block.  m ramped objects on k buses plus
    m_static channels through a constant matrix on the direct bus (config 5's HOA bed: one gain row)."""
    gain = 4 * (m * b + 2 * k * m * n + k * n * b)
    if m_static:
        gain += 4 * (m_static * b + m_static * n + n * b)
    p, d = 1, 255
    dec = n * (4 * b + 8 * p * (b + 1) + 2 * 8 * p * (b + 1) + 2 * 4 * b + 4 * b) if k == 2 else 0
    dm = n * (4 * b + 2 * 4 * d + 4 * b + 4 * b) if k == 2 else 0
    return gain, dec, dm


GAIN_KERNELS = {0: "k_gain_mix (VALU, strict)", 1: "k_gain_mix_mfma (f32 MFMA)", 2: "k_gain_mix_f32g (f32 MFMA on the tile grid)",
                3: "k_gain_mix_h2 (f16x2 MFMA)",
                4: "k_gain_mix_p2 (f16x2 MFMA over piece lists)", 5: "k_gain_mix_hg (f16x2 MFMA, hinges)"}
GAIN_DTYPES = {0: "f32 (VALU, libear's exact arithmetic)", 1: "f32 (f32 MFMA, f32 accumulate)", 2: "f32 (f32 MFMA, f32 accumulate)",
               3: "f32 io / f16x2-split MFMA, f32 accumulate", 4: "f32 io / f16x2-split MFMA, f32 accumulate",
               5: "f32 io / f16x2-split MFMA, f32 accumulate"}


def mfma_roofline(kind, macs_per_term, k1_ms):
    """The gain kernel against the matrix pipe it runs on (secondary to the HBM roofline)."""
    if kind in (3, 4):  # 2 operands (B0, B1) x 3 f16 partial products per object, column and sample
        flops, peak, what = 12.0 * macs_per_term, BF16_PEAK_TFLOPS, "f16 MFMA flops: 3 partial products x {gain at tile start, slope}"
    else:           # 2 f32 MACs (start, end gain row) per object, column and sample
        flops, peak, what = 4.0 * macs_per_term, FP32_PEAK_TFLOPS, "f32 flops: 2 MACs per object, column and sample for a ramp"
    ach = flops / (k1_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": GAIN_KERNELS.get(kind, "?"), "achieved": round(ach, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(ach / peak, 4), "note": "executed " + what}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def producer_bench(args):
    """Secondary measurement: gain vectors per second of the Objects gain producer with extent (one GPU)."""
    import numpy as np
    import torch
    from libear_amd import capi
    layout = args.layout or PRESETS[args.config]["layout"]
    n = args.producer
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    pan = capi.Panner(ctx, layout)
    rng = np.random.default_rng(21)
    host = {"az": rng.uniform(-180, 180, n), "el": np.degrees(np.arcsin(rng.uniform(-1, 1, n))),
            "dist": rng.uniform(0.5, 1.5, n), "width": rng.uniform(0, 120, n), "height": rng.uniform(0, 60, n),
            "depth": np.where(rng.uniform(0, 1, n) < 0.25, rng.uniform(0, 0.5, n), 0.0), "gain": rng.uniform(0.1, 1, n),
            "diffuse": rng.choice([0.0, 0.3, 1.0], n)}
    d_in = {k: torch.from_numpy(v).to(dev) for k, v in host.items()}
    direct = torch.empty((n, pan.n_out), dtype=torch.float32, device=dev)
    diffuse = torch.empty_like(direct)

    def step():
        pan.calculate_device(n, d_in["az"].data_ptr(), d_in["el"].data_ptr(), d_in["dist"].data_ptr(), d_in["gain"].data_ptr(),
                             d_in["diffuse"].data_ptr(), direct.data_ptr(), diffuse.data_ptr(), d_in["width"].data_ptr(),
                             d_in["height"].data_ptr(), d_in["depth"].data_ptr())
    for _ in range(max(args.warmup, 3)):
        step()
    torch.cuda.synchronize()
    steps = max(args.steps, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    # parity of what was timed (a sample of it) and the CPU baseline: the oracle's PolarExtent, one thread
    import _oracle
    k = min(n, 2000)
    o = _oracle.PolarExtent(layout)
    t0 = time.perf_counter()
    wd, wf = o.calculate(*(host[q][:k] for q in ("az", "el", "dist", "width", "height", "depth", "gain", "diffuse")))
    cpu_s = time.perf_counter() - t0
    gd, gf = direct[:k].cpu().numpy().astype(np.float64), diffuse[:k].cpu().numpy().astype(np.float64)

    def rel(a, b):
        den = np.maximum(np.minimum(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)), 1e-30)
        return float(np.max(np.linalg.norm(a - b, axis=1) / den))
    md, mf = host["diffuse"][:k] < 1.0, host["diffuse"][:k] > 0.0
    worst = max(rel(gd[md], wd[md].astype(np.float64)), rel(gf[mf], wf[mf].astype(np.float64)))
    ok = worst <= 1e-5
    print(json.dumps({
        "metric": "gain vectors per second of the Objects gain producer (point source + polar extent panners)",
        "value": round(n / ms / 1e3, 2), "unit": "Mpositions/s", "n_gpus": 1, "steps": steps, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "dtype": "f64 set-up / f32 weights and sums (as libear)", "data": "synthetic",
        "config": {"workload": f"{n} (object, metadata block) positions, layout {layout}, widths 0-120, heights 0-60 "
                               "degrees, distances 0.5-1.5, a quarter with depth", "entry": "earhip_panner_calculate_extent_device"},
        "parity": {"against": "oracle PolarExtent (scalar float core)", "sample": k, "max_rel_norm": worst, "tolerance": 1e-5,
                   "pass": ok},
        "cpu_baseline": {"value": round(k / cpu_s / 1e6, 5), "unit": "Mpositions/s", "cores": 1, "kind": "port",
                         "sample": f"{k} of the same positions", "cpu": cpu_model()}}))
    pan.close()
    if not ok:
        print("producer parity FAILED", file=sys.stderr)
        sys.exit(3)


def refbench(args):
    """The reference's only hot-path benchmark DEFINITION (it records no result), timed here on both sides:
    GainInterpolator<LinearInterpMatrix>::process, 32 inputs -> 24 outputs, 1024 samples, a random matrix point every
    100 samples (tests/gain_interpolator_tests.cpp:259-296) — the latency-bound end of the path.  Block mode = that
    very call through the drop-in entry point (host channel pointers: H2D, kernels, D2H, sync); stream mode = the same
    curve density over 2^20 samples per call, device resident; the CPU path (the oracle's restatement, -O3 -DNDEBUG,
    one thread) beside both; and where one host core and the GPU cross over as the call shrinks."""
    import numpy as np
    import torch
    import _oracle
    import scenes
    from libear_amd import capi
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(0, stream.cuda_stream)
    n_out, n = 24, 1024
    rng = np.random.default_rng(5)

    def points(m, total):
        t = np.arange(100, total, 100, dtype=np.int64)
        return t, rng.uniform(0.0, 1.0, (len(t), m, n_out)).astype(np.float32)

    def block_mode(m, calls=200):
        t, v = points(m, n)
        x = rng.uniform(-1.0, 1.0, (m, n)).astype(np.float32)
        g = capi.GainInterp(ctx, m, n_out)
        g.set_points(t, v)
        for _ in range(20):
            got = g.process(0, x)
        lat = []
        for _ in range(calls):
            t0 = time.perf_counter()
            g.process(0, x)
            lat.append(time.perf_counter() - t0)
        g.close()
        lat.sort()
        want = _oracle.gain_interp("matrix", t, v, x, [n])
        reps, c0 = 0, time.perf_counter()
        while time.perf_counter() - c0 < 0.25 or reps < 3:
            _oracle.gain_interp("matrix", t, v, x, [n])
            reps += 1
        cpu = (time.perf_counter() - c0) / reps
        return lat[len(lat) // 2], lat[int(0.95 * len(lat))], cpu, scenes.rel_rms(got, want)

    gpu_ms, gpu_p95, cpu_s, err = block_mode(32)
    sweep = []
    for m in (1, 2, 4, 8, 16, 32, 64, 128):
        g_, _, c_, e_ = block_mode(m, calls=100)
        sweep.append({"inputs": m, "gpu_block_ms": round(g_ * 1e3, 4), "cpu_ms": round(c_ * 1e3, 4), "rel_rms": float(f"{e_:.2e}")})
    cross = next((s["inputs"] for s in sweep if s["gpu_block_ms"] < s["cpu_ms"]), None)
    # stream mode: the same point density over a long device-resident call
    m, total = 32, n * 1024
    t, v = points(m, total)
    g = capi.GainInterp(ctx, m, n_out)
    g.set_points(t, v)
    x = torch.rand((m, total), device=dev, dtype=torch.float32) * 2.0 - 1.0
    out = torch.zeros((n_out, total), device=dev, dtype=torch.float32)
    steps, warm = max(args.steps, 1), max(args.warmup, 2)
    for _ in range(warm):
        g.process_device(0, total, x.data_ptr(), total, out.data_ptr(), total)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.process_device(0, total, x.data_ptr(), total, out.data_ptr(), total)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # parity of the stream call on a window, and the CPU path's rate on the same window
    w = 8 * n
    xs = x[:, :w].cpu().numpy()
    tw = t[t <= w + 100]
    c0 = time.perf_counter()
    want = _oracle.gain_interp("matrix", tw, v[:len(tw)], xs, [n] * 8)
    cpu_stream_s = time.perf_counter() - c0
    serr = scenes.rel_rms(out[:, :w].cpu().numpy(), want)
    g.close()
    bytes_alg = 4 * (m * total + n_out * total) + v.nbytes
    ok = err <= 1e-6 and serr <= 1e-6
    print(json.dumps({
        "metric": "Msamples/s", "value": round(m * total / dt / 1e6, 1), "unit": "Msamples/s", "n_gpus": 1, "steps": steps,
        "warmup": warm, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 io / f16x2-split MFMA, f32 accumulate", "data": "synthetic",
        "config": {"workload": "refbench: GainInterpolator<LinearInterpMatrix>::process, 32 -> 24 channels, a random matrix "
                               "point every 100 samples (the reference's matrix_benchmark, tests/gain_interpolator_tests.cpp:"
                               "259-296); stream mode: 1024 x 1024 samples per call, device resident", "baseline_config": "refbench"},
        "roofline": {"bound": "hbm", "achieved": round(bytes_alg / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(bytes_alg / dt / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                     "note": "algorithmic bytes: inputs + outputs + the matrix points, per call"},
        "block_mode": {"what": "the reference benchmark's own call (1024 samples) through earhip_gain_interp_process: host "
                               "channel pointers, H2D, kernels, D2H, sync", "ms_per_call": round(gpu_ms * 1e3, 4),
                       "p95_ms": round(gpu_p95 * 1e3, 4), "Msamples_per_s": round(32 * n / gpu_ms / 1e6, 1),
                       "rtf": round((n / SAMPLE_RATE) / gpu_ms, 1)},
        "cpu_baseline": {"value": round(32 * n / cpu_s / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
                         "sample": "the same 1024-sample call, oracle restatement of LinearInterpMatrix, -O3 -DNDEBUG, "
                                   "mean over >= 0.25 s", "ms_per_call": round(cpu_s * 1e3, 4),
                         "stream_window_Msamples_per_s": round(32 * w / cpu_stream_s / 1e6, 2), "cpu_model": cpu_model()},
        "crossover": {"sweep_24_outputs_1024_samples": sweep, "gpu_block_mode_faster_from_inputs": cross,
                      "note": "one call of 1024 samples from host pointers against one host core; below the crossover the "
                              "call is launch + PCIe latency on the GPU side"},
        "parity": {"block_rel_rms_vs_cpu": float(f"{err:.3e}"), "stream_window_rel_rms_vs_cpu": float(f"{serr:.3e}"),
                   "tolerance": 1e-6, "pass": bool(ok)}}))
    ctx.close()
    if not ok:
        print("refbench parity FAILED", file=sys.stderr)
        sys.exit(3)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as a FRESH child
    (`python -m torch.distributed.run ... bench.py <the same arguments>`), relay its output (rank 0's JSON line)
    and leave with its exit code.  This process never touches the GPU (no torch import, no HIP call) and does
    not replace itself: the child is an ordinary subprocess."""
    import socket
    import subprocess
    import random
    port = 29500
    for _ in range(64):  # (below the kernel's ephemeral range: a port from there can be taken again before the rendezvous binds it)
        cand = random.randint(15000, 30000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", cand))
        except OSError:
            continue
        finally:
            s.close()
        port = cand
        break
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver only does dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, env=env)
    sys.exit(res.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(PRESETS) + ["refbench"], default="C4",
                    help="BASELINE.json configuration; refbench: the reference's own matrix_benchmark shape (32 -> 24 channels, "
                         "1024 samples, a point every 100 samples), block and stream mode, CPU path beside it")
    ap.add_argument("--objects", type=int, default=None, help="objects of the scene (in total; --scaling weak: per GPU)")
    ap.add_argument("--hoa", type=int, default=None, help="extra input channels through a constant decode matrix (config 5)")
    ap.add_argument("--blocks", type=int, default=None, help="blocks per step (stream length T)")
    ap.add_argument("--block-size", type=int, default=None)
    ap.add_argument("--layout", default=None)
    ap.add_argument("--buses", type=int, choices=(1, 2), default=None,
                    help="1: ramped gains only (config 2); 2: direct + diffuse, decorrelators, delay, mix")
    ap.add_argument("--kernel-timing-every", type=int, default=8,
                    help="record the per-kernel HIP events (roofline, kernels_ms) on every n-th timed step")
    ap.add_argument("--precondition-ms", type=float, default=40.0,
                    help="untimed steps of the same workload before the W warm-up steps, until this much wall time "
                         "has passed: an idle MI355X needs 10-20 ms of load to leave its low-power clocks "
                         "(steps measured right after start-up are ~9 %% slower)")
    ap.add_argument("--cpu-blocks", type=int, default=160, help="blocks of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--strict", action="store_true", help="bit-exact gain arithmetic (slower)")
    ap.add_argument("--scene", choices=("dense", "adm", "static", "moving", "mixed", "panned", "panned-adm", "levels", "levels-adm",
                                        "bursty", "bursty-adm", "bursty-moving"),
                    default="dense",
                    help="dense: block-aligned full-length ramps (headline, SURVEY 8d); adm: metadata blocks of "
                         "960 samples at a random phase per object, 240-sample ramps then constant (not block-aligned); "
                         "panned / panned-adm: moving point sources through the device gain producer (libearhip group "
                         "I: real 3-sparse VBAP gains, sqrt(1-d)/sqrt(d) split, zero LFE columns), a new position every "
                         "block / every 960 samples at a per-object phase; levels / levels-adm: the dense / adm scene with the "
                         "objects' SIGNAL levels log-uniform over 0 .. -90 dB and 30 %% of the objects digitally silent for the "
                         "first half of the call (real mixes: pauses, tails, entries); bursty / bursty-adm / bursty-moving: the "
                         "dense / adm / moving scene with NON-STATIONARY signals (per object and block: holds, fades of 3 dB per "
                         "block to -90 dB, digital silence, gated bursts, single blocks 40 dB above their surroundings), eight "
                         "objects alone on a loudspeaker each and 80 dB down for most of the call")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default, BASELINE config 4): --objects in total, split over the GPUs; "
                         "weak: --objects per GPU")
    ap.add_argument("--shard", choices=("objects", "objects-nogather", "time"), default="objects",
                    help="decomposition over the GPUs (DESIGN 6).  objects (north_star): every rank renders its objects, RCCL "
                         "reduce-scatter of the partial outputs over the channels + gather of the shared bus on rank 0; "
                         "objects-nogather: the same without the gather (a consumer that takes the bus channel-sharded: half "
                         "the exchange); time: every rank renders ALL objects for its T / G blocks of the stream after one lead "
                         "block that recomputes the overlap-add tail and the delay line — no exchange at all, the outputs of "
                         "different time ranges live on different ranks")
    ap.add_argument("--row-pad", type=int, default=256,
                    help="floats of padding between the input rows (row stride = samples + pad; multiple of 4).  Default 256: "
                         "rows a power of two apart (2 MB at the headline's size, 8 MB at config 3's) put the 32 rows a chunk "
                         "reads on the same memory channels — config 3's gain kernel 0.48 (0.456-0.543 from run to run) -> "
                         "0.45 ms, config 5's 0.225 -> 0.220, headline and config 2 unchanged; the caller owns this layout "
                         "(libear takes channel POINTERS: ptr_adapter.hpp:17-24)")
    ap.add_argument("--brief", action="store_true",
                    help="the timed stream-mode steps, their per-kernel times and the parity gate on the timed buffer only "
                         "(no block-mode, exact-f32, strict or CPU-baseline legs): what the `secondary` entries of the default line run")
    ap.add_argument("--no-secondary", action="store_true",
                    help="the default invocation (config C4, dense scene, one GPU) without its `secondary` runs of the other workloads")
    ap.add_argument("--stream-only", action="store_true",
                    help="only the timed stream-mode steps (no block-mode, parity or CPU legs): for PMC profiling passes")
    ap.add_argument("--producer", type=int, default=0, metavar="POSITIONS",
                    help="instead of the render path, time the Objects gain producer (libearhip group I: point source "
                         "+ polar extent panners) on this many (object, metadata block) positions with random extents, "
                         "device pointers; prints its own JSON line with a CPU baseline (the oracle on a sample)")
    args = ap.parse_args()
    if args.producer:
        return producer_bench(args)
    if args.config == "refbench":
        return refbench(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    cfg = dict(PRESETS[args.config])
    for key, val in (("objects", args.objects), ("hoa", args.hoa), ("blocks", args.blocks),
                     ("block_size", args.block_size), ("layout", args.layout), ("buses", args.buses)):
        if val is not None:
            cfg[key] = val

    import numpy as np
    import torch
    import torch.distributed as dist

    parity_failed = False
    secondary_failed = False
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # one process per GPU; EARHIP_BENCH_BACKEND=gloo lets several ranks share one GPU (a functional
    # check of the multi-rank control flow on a single-GPU box, not a measurement)
    backend = os.environ.get("EARHIP_BENCH_BACKEND", "nccl")
    shared_gpus = False
    if world > torch.cuda.device_count() >= 1 and backend == "nccl":
        # fewer GPUs than ranks (RCCL refuses two ranks on one device): the ranks share the GPUs and the
        # collective runs over gloo — every rank sees the same count, so they all decide the same
        backend, shared_gpus = "gloo", True
        if rank == 0:
            print(f"bench.py: {world} ranks on {torch.cuda.device_count()} GPU(s): ranks share devices, exchange over gloo "
                  "(a functional check of the multi-rank path, not a measurement)", file=sys.stderr, flush=True)
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import scenes
    from layouts import LAYOUTS
    from libear_amd import capi
    from libear_amd.distributed import channel_range, exchange, exchange_model, lead_blocks, padded_channels, shard_range, time_range

    names = LAYOUTS[cfg["layout"]]
    N, B, T, K = len(names), cfg["block_size"], cfg["blocks"], cfg["buses"]
    total = B * T
    dec = capi.design_decorrelators(names) if K == 2 else None  # designed natively (libearhip group G)
    delay = capi.compensation_delay() if K == 2 else 0
    n_pad = padded_channels(N, world)

    # ONE stream for everything: the renderer's launches, torch's ops, the events below, and the stream
    # RCCL orders its collectives against (a context given no stream creates its own non-blocking one,
    # which torch's default stream knows nothing about)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = capi.Context(dev_index, stream.cuda_stream)
    ctx.set_strict(args.strict)

    # the exchange step: the library's own RCCL communicator (default on the nccl backend: libearhip group J,
    # what a C++ caller uses — reduce-scatter over the channels + gather of the shared bus on rank 0) or
    # torch.distributed's reduce_scatter_tensor (EARHIP_BENCH_EXCHANGE=torch; the CPU backends).  Should the
    # native communicator fail to come up on any rank, ALL ranks fall back to torch and the line says so.
    native_comm, native_note = None, None
    want_native = os.environ.get("EARHIP_BENCH_EXCHANGE", "native") != "torch"
    if world > 1 and backend == "nccl" and want_native:
        box = [capi.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        err = None
        try:
            native_comm = capi.Comm(ctx, rank, world, box[0])
        except Exception as e:  # noqa: BLE001 - any failure means "not native" for everybody
            err = str(e)
        ok = torch.tensor([0 if err else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if native_comm is not None:
                native_comm.close()
            native_comm = None
            native_note = "earhip_comm_create failed on a rank" + (f" ({err})" if err else "") + ": torch.distributed exchange"
    # what RCCL itself reports (every rank's count / device, gathered on rank 0) and the link rate it gives a 50 MB
    # ncclSend / ncclRecv pair per rank — the exchange model below uses THAT, not an assumed figure
    rccl_info, link_measured = None, None
    if native_comm is not None:
        mine = native_comm.info()
        infos = [None] * world
        dist.all_gather_object(infos, mine)
        rates = []
        for shift in sorted({1, world // 2} - {0}):
            try:  # (a probe must never cost the measurement: a rank that cannot probe reports 0 and every rank sees it)
                rate = native_comm.link_probe(50 << 20, shift, 5)
            except Exception:  # noqa: BLE001
                rate = 0.0
            g = torch.tensor([rate], device=dev, dtype=torch.float64)
            dist.all_reduce(g, op=dist.ReduceOp.MIN)
            rates.append({"shift": shift, "GBps_per_direction_slowest_rank": round(float(g.item()), 1)})
        link_measured = min(r["GBps_per_direction_slowest_rank"] for r in rates) or None  # (0: not measured, the model keeps its assumption)
        rccl_info = {"ranks": mine["ranks"], "version": mine["version"], "devices": [i["device"] for i in infos],
                     "all_ranks_agree": bool(all(i["ranks"] == world for i in infos) and [i["rank"] for i in infos] == list(range(world))),
                     "link_probe": {"bytes": 50 << 20, "pairs": rates,
                                    "note": "every rank sends 50 MB to rank + shift and receives 50 MB from rank - shift at once "
                                            "(ncclSend / ncclRecv), 5 rounds between HIP events: the slowest rank's rate"}}
    gather_root = 0  # the rank that ends up with the whole loudspeaker bus (north_star: "the shared loudspeaker bus")

    class Workload:
        """this rank's shard of a scene of `objects` objects (+ `hoa` bed channels on rank 0), resident
        in HBM, with its renderer and double-buffered output / exchange buffers"""

        def __init__(self, objects, hoa, scaling, seed_base=0, context=None, shard=None):
            self.shard = shard or args.shard
            self.time_sharded = self.shard == "time" and world > 1 and scaling == "strong"
            if self.time_sharded:  # every rank has the whole scene (the same seeds) and renders its blocks of the stream
                assert T % world == 0, "--shard time needs the blocks of a step to divide by the GPUs"
                self.M_obj, self.M_total = objects, objects + hoa
            elif scaling == "strong":  # the scene has `objects` in total; this rank renders its shard
                lo, hi = shard_range(objects, rank, world)
                self.M_obj, self.M_total = hi - lo, objects + hoa
            else:
                self.M_obj, self.M_total = objects, (objects + hoa) * world
            self.M_hoa = hoa if (rank == 0 or scaling == "weak" or self.time_sharded) else 0  # the bed goes to one rank
            self.M = self.M_obj + self.M_hoa
            m, seed = max(self.M_obj, 1), seed_base + (0 if self.time_sharded else rank)
            if args.scene in ("adm", "levels-adm", "bursty-adm"):
                ap_, ar_ = (int(v) for v in os.environ.get("EARHIP_BENCH_ADM", "960,240").split(","))  # (tuning: period, ramp in samples)
                curves = scenes.adm_curves(m, N, total, period=ap_, ramp=ar_, seed=11 + seed)
            elif args.scene in ("moving", "bursty-moving"):  # always ramping: a new target every 240 samples (5 ms) at a per-object phase
                mp = max(64, int(os.environ.get("EARHIP_BENCH_MOVING_PERIOD", "240")))  # (tuning: the update period, samples)
                curves = scenes.adm_curves(m, N, total, period=mp, ramp=mp, seed=12 + seed)
            elif args.scene == "static":  # one gain vector per object and bus, never changing
                curves = scenes.constant_curves(m, N, seed=8 + seed)
            elif args.scene == "mixed":  # the headline scene with 8 of every 1024 objects on ADM-like metadata off the block grid
                base_ = os.environ.get("EARHIP_BENCH_MIXED_BASE")  # (tuning: the other objects' curves)
                curves = (scenes.constant_curves(m, N, seed=8 + seed) if base_ == "static" else scenes.adm_curves(m, N, total, seed=5 + seed)
                          if base_ == "adm" else scenes.adm_curves(m, N, total, period=240, ramp=240, seed=12 + seed) if base_ == "moving"
                          else scenes.dense_curves(m, N, B, T, seed=7 + seed))
                every = max(2, int(os.environ.get("EARHIP_BENCH_MIXED_EVERY", "128")))  # (tuning: one such object in every `every`)
                op_, or_ = (int(v) for v in os.environ.get("EARHIP_BENCH_MIXED_ODD", "960,240").split(","))  # (tuning: their period, ramp)
                odd = scenes.adm_curves(max(m // every, 1), N, total, period=op_, ramp=or_, seed=11 + seed)
                for i, c in enumerate(odd):
                    curves[(every * i + 7) % m] = c
            elif args.scene in ("panned", "panned-adm"):
                # moving point sources: positions -> gain vectors by the device batch panner (earhip group I)
                if args.scene == "panned":  # a new position at every block boundary, reached over the whole block
                    az, el, df, tms = scenes.moving_sources(m, total, period=B, seed=31 + seed, phase=0, ramp=B)
                else:
                    az, el, df, tms = scenes.moving_sources(m, total, period=960, seed=31 + seed)
                pan = capi.Panner(context or ctx, cfg["layout"])
                cat = lambda v: np.concatenate(v)
                d_all, f_all = pan.calculate(cat(az), cat(el), None, None, cat(df))
                pan.close()
                curves, at = [], 0
                for i in range(m):
                    k = len(tms[i])
                    curves.append((tms[i], d_all[at:at + k], f_all[at:at + k]))
                    at += k
            else:
                curves = scenes.dense_curves(m, N, B, T, seed=7 + seed)
            curves = curves[:self.M_obj]
            self.solo = 0
            if args.scene.startswith("bursty") and self.M_obj >= 64:
                self.solo = 8
                spk = [c for c in range(N) if not names[c].startswith("LFE")]
                curves = scenes.solo_curves(curves, N, spk[-self.solo:])
            if self.M_hoa:  # constant decode matrix: one gain point on the direct bus, nothing on the diffuse bus
                decode = np.random.default_rng(55).uniform(-0.5, 0.5, (self.M_hoa, N)).astype(np.float32)
                curves = [(np.zeros(1, np.int64), decode[c:c + 1], np.zeros((1, N), np.float32))
                          for c in range(self.M_hoa)] + curves
            self.curves = curves
            gen = torch.Generator(device=dev)
            gen.manual_seed(1234 + seed)
            # (time-sharded: a rank's blocks [b0, b1) of the stream and `lead` blocks in front of them)
            self.b0, self.b1, self.lead = (time_range(T, rank, world, partitions=(-(-512 // B) if K == 2 else 1), delay_blocks=-(-delay // B))
                                           if self.time_sharded else (0, T, 0))
            self.in_stride = total + args.row_pad
            rows = max(self.M, 1)
            # (the samples do not depend on the padding: the rows are drawn as a [rows][total] block)
            self.x_full = torch.zeros((rows, self.in_stride), device=dev, dtype=torch.float32)
            self.x = self.x_full[:, :total]  # planar rows, in_stride floats apart
            self.x.copy_(torch.rand((rows, total), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0)
            if args.scene.startswith("bursty"):
                blv = scenes.bursty_levels(rows, T, self.solo, seed=91 + seed)
                self.x_full[:, :total].unflatten(1, (T, B)).mul_(torch.as_tensor(blv, device=dev)[:, :, None])
            if args.scene in ("levels", "levels-adm"):
                lv, late = scenes.object_levels(rows, seed=77 + seed)
                self.x_full *= torch.as_tensor(lv, device=dev, dtype=torch.float32)[:, None]
                self.x_full[torch.as_tensor(late, device=dev), :total // 2] = 0.0
            # One rank: the step's output [n_pad][total], double buffered.  Several ranks: a step is rendered in `nch` calls
            # of T / nch blocks, each into one of two chunk buffers [n_pad][clen] and exchanged at once, so that the
            # collectives of a chunk run beside the render of the NEXT chunk of the same step (not only beside the next
            # step); the shared bus lands on the root chunk by chunk, [nch][n_pad][clen] (double buffered over the steps).
            self.nch = 1
            self.gather = self.shard == "objects"
            if world > 1 and not self.time_sharded:
                want = int(os.environ.get("EARHIP_BENCH_CHUNKS", "4"))
                while want > 1 and (T % want or T // want < 8):
                    want //= 2
                self.nch = max(want, 1)
            self.Tc, self.clen = T // self.nch, total // self.nch
            if self.time_sharded:  # one call per step: the lead block + this rank's blocks
                self.Tc = self.b1 - self.b0 + self.lead
                self.clen = self.Tc * B
            self.outs = [torch.zeros((n_pad, self.clen), device=dev, dtype=torch.float32) for _ in range(2)]
            self.owned = [torch.zeros((n_pad // world, self.clen), device=dev, dtype=torch.float32) for _ in range(2)] \
                if world > 1 and not self.time_sharded else None
            self.full = [[torch.zeros((n_pad, self.clen), device=dev, dtype=torch.float32) for _ in range(self.nch)] for _ in range(2)] \
                if world > 1 and rank == gather_root and self.gather and not self.time_sharded else None
            self.head = torch.zeros((n_pad, self.clen), device=dev, dtype=torch.float32) if world > 1 else None
            self.r = self.renderer(context or ctx)
            self.pending = [None, None]
            self.k = 0           # chunks rendered so far: chunk k uses buffer / exchange slot k % 2
            self.last_slot = 0

        def renderer(self, context, max_blocks=None, npoints=None):
            r = capi.Renderer(context, max(self.M, 1), N, B, dec, delay, max_blocks=max_blocks or max(T, self.Tc))
            for m, (t, d, f) in enumerate(self.curves):
                r.set_object_points(m, t[:npoints], d[:npoints], f[:npoints] if K == 2 else None)
            r.commit()
            return r

        def step(self, i, exchange_outputs=True, r=None, keep_head=False):
            r = r or self.r
            if self.time_sharded:
                # this rank's time range: the DSP state at its first block comes from ONE lead block rendered from the zero
                # state (tail: the previous block's; delay line: 255 samples), whose outputs are not used
                slot = self.k % 2
                self.k += 1
                t_lo = (self.b0 - self.lead) * B
                r.reset(t_lo)
                dst = self.head if keep_head else self.outs[slot]
                r.process_device(self.Tc, self.x.data_ptr() + 4 * t_lo, self.in_stride, dst.data_ptr(), self.clen)
                self.last_slot = slot
                return
            r.reset(0)
            for c in range(self.nch):
                slot = self.k % 2
                self.k += 1
                if self.pending[slot] is not None:  # the exchange that last read this buffer must be done
                    self.pending[slot].wait()
                    self.pending[slot] = None
                if native_comm is not None:
                    native_comm.wait(slot)
                dst = self.head if (keep_head and c == 0) else self.outs[slot]  # (keep_head: the step's first blocks for the parity gate)
                r.process_device(self.Tc, self.x.data_ptr() + 4 * c * self.clen, self.in_stride, dst.data_ptr(), self.clen)
                self.last_slot = slot
                if world > 1 and exchange_outputs and self.shard != "time":  # (weak scaling under --shard time: independent streams)
                    per = n_pad // world
                    full_c = self.full[i % 2][c] if self.full is not None else None
                    if native_comm is not None:
                        native_comm.exchange_device(slot, self.outs[slot].data_ptr(), self.owned[slot].data_ptr(), per, self.clen)
                        if self.gather:
                            native_comm.gather_device(slot, self.owned[slot].data_ptr(), full_c.data_ptr() if full_c is not None else None,
                                                      per, self.clen, gather_root)
                    else:
                        _, work = exchange(self.outs[slot], self.owned[slot], async_op=True)
                        if backend == "nccl" and self.gather:  # (gloo: the all-reduce leaves the whole bus on every rank already)
                            work.wait()  # stream-orders the gather behind the reduce-scatter; does not block the host
                            work = dist.gather(self.owned[slot], list(full_c.split(per)) if rank == gather_root else None,
                                               dst=gather_root, async_op=True)
                        self.pending[slot] = work

        def drain(self):
            for b in range(2):
                if self.pending[b] is not None:
                    self.pending[b].wait()
                    self.pending[b] = None
                if native_comm is not None:
                    native_comm.wait(b)

        def timed(self, steps, warmup, precondition_ms=0.0, r=None, timing_every=0):
            """W untimed steps, then exactly `steps` steps between barrier + synchronize on both sides;
            returns (seconds, max over ranks)"""
            pre_steps = 0
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < precondition_ms:
                self.step(pre_steps, exchange_outputs=False, r=r)  # wall-clock bounded: ranks may differ, so no collective here
                pre_steps += 1
                if pre_steps % 8 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            for i in range(warmup):
                self.step(i, r=r)
            self.drain()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            if timing_every:
                (r or self.r).enable_timing(timing_every)
            t0 = time.perf_counter()
            for i in range(steps):
                self.step(i, r=r)
            self.drain()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if world > 1:
                tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            return dt, pre_steps

        def close(self):
            self.r.close()

    # ---- the measured workload ---------------------------------------------------------------------
    wl = Workload(cfg["objects"], cfg["hoa"], args.scaling)
    M, M_total = wl.M, wl.M_total
    # per-kernel HIP events on every TIME_EVERY-th step of the timed region (each timed step records six
    # events = ~20 us of idle GPU; timing every step costs 4 % of the headline value)
    time_every = max(1, min(args.kernel_timing_every, args.steps))
    timed_steps = len(range(0, args.steps, time_every))
    dt, pre_steps = wl.timed(args.steps, args.warmup, args.precondition_ms, timing_every=time_every)
    timing = wl.r.get_timing()
    plan = wl.r.last_plan()
    gain_kernel = plan["kernel"]
    # (a call planned for the hinge kernel is decided on the device: it, or the piece lists standing by)
    scratch_mb = round(wl.r.scratch_bytes() / 1e6, 1)
    wide_form = wl.r.wide_form()  # (the form — plain / wide low pieces — the split-operand kernel picked on the device)
    hinge_robust = gain_kernel == 5 and wl.r.hinge_robust()
    if hinge_robust:
        GAIN_KERNELS[5] = "k_gain_mix_hg (f16x2 MFMA, hinges; kink products in f32: the levels of this call's inputs spread beyond the packed-f16 products' span)"
    hinge_standby = gain_kernel == 5 and wl.r.hinge_standby()
    if hinge_standby:
        GAIN_KERNELS[5] = "k_gain_mix_p2 (f16x2 MFMA over piece lists, standing by for k_gain_mix_hg: the levels of this call's inputs spread beyond its span)"
    wl.r.enable_timing(False)

    # multi-rank self-check (outside the timed region, all ranks take part): the slice this rank
    # owns after the reduce-scatter equals the sum of all ranks' partial outputs
    exchange_err, exchange_repeat_identical, lead_check = None, None, None
    if world > 1 and wl.time_sharded:
        # time sharding's own check: the rank's blocks after ONE lead block equal the same blocks after THREE (the state a
        # lead block leaves is the stream's: nothing older than one block reaches a block's output)
        far = wl.lead + 2  # (the second render starts two blocks further back)
        ncmp = max(1, min(4, wl.Tc - far))  # blocks compared (a rank with few blocks: fewer)
        if wl.b0 >= far and wl.Tc > far:
            wl.step(0, exchange_outputs=False)
            torch.cuda.synchronize()
            one = wl.outs[wl.last_slot][:N, wl.lead * B:(wl.lead + ncmp) * B].clone()
            t_lo = (wl.b0 - far) * B
            wl.r.reset(t_lo)
            wl.r.process_device(far + ncmp, wl.x.data_ptr() + 4 * t_lo, wl.in_stride, wl.outs[wl.last_slot].data_ptr(), wl.clen)
            torch.cuda.synchronize()
            three = wl.outs[wl.last_slot][:N, far * B:(far + ncmp) * B]
            err = ((one - three).abs().max() / three.abs().max().clamp_min(1e-30)).to(torch.float64).reshape(1)
        else:
            err = torch.zeros(1, device=dev, dtype=torch.float64)
        dist.all_reduce(err, op=dist.ReduceOp.MAX)
        lead_check = float(err.item())
    if world > 1 and not wl.time_sharded and (backend == "nccl" or os.environ.get("EARHIP_BENCH_CHECK") == "force"):
        # (on the LAST chunk of the last step: the buffers of its exchange slot still hold it)
        w = min(wl.clen, 4096)
        last = (args.steps - 1) % 2
        per = n_pad // world
        if backend == "nccl":  # the reduce-scatter left the partial output in place
            slot = wl.last_slot
            part = wl.outs[slot][:, :w].clone()
            got = wl.owned[slot][:, :w]
        else:  # (the all-reduce of the CPU backends sums `outs` in place: render the partial again)
            wl.step(last, exchange_outputs=False)
            torch.cuda.synchronize()
            slot = wl.last_slot
            part = wl.outs[slot][:, :w].clone()
            _, work = exchange(wl.outs[slot], None, async_op=True)
            work.wait()
            got = wl.outs[slot][rank * per:(rank + 1) * per, :w]
        ref = part.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        err = ((got - ref[rank * per:(rank + 1) * per]).abs().max() / ref.abs().max().clamp_min(1e-30))
        if backend == "nccl" and rank == gather_root and wl.gather:  # ... and the bus gathered on the root equals all of it
            err = torch.maximum(err, (wl.full[last][wl.nch - 1][:, :w] - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        err = err.to(torch.float64).reshape(1)
        dist.all_reduce(err, op=dist.ReduceOp.MAX)
        exchange_err = float(err.item())
        # run-to-run reproducibility ("reduction order across ranks should be deterministic"): the same partial
        # outputs exchanged a second time give bit-identical owned slices
        first = got.clone()
        again_in = torch.zeros_like(wl.outs[slot])
        again_in[:, :w] = part
        if backend == "nccl":
            again_out = torch.zeros_like(wl.owned[slot])
            if native_comm is not None:
                native_comm.wait(slot)
                native_comm.exchange_device(slot, again_in.data_ptr(), again_out.data_ptr(), per, wl.clen)
                native_comm.wait(slot)
                torch.cuda.synchronize()
            else:
                exchange(again_in, again_out)
            second = again_out[:, :w]
        else:
            _, work = exchange(again_in, None, async_op=True)
            work.wait()
            second = again_in[rank * per:(rank + 1) * per, :w]
        same = torch.tensor([1 if torch.equal(first, second) else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        exchange_repeat_identical = bool(int(same.item()))
    # what the exchange moved and how fast (native communicator: HIP events around the collectives of the last
    # two steps on its own stream — with the renders of the following step running beside them)
    exchange_info = None
    if world > 1 and wl.time_sharded:
        exchange_info = {"mode": "time", "rccl": rccl_info, "reduce_scatter_bytes_per_rank": 0, "gather_bytes_into_root": 0,
                         "blocks_of_this_rank": [wl.b0, wl.b1], "lead_blocks": wl.lead,
                         "note": "no exchange: every rank renders all objects for its blocks of the stream; the outputs of "
                                 "different time ranges live on different ranks"}
    elif world > 1:
        per = n_pad // world
        rs_bytes = (world - 1) * per * total * 4       # reduce-scatter: sent (and received) by every rank
        ga_bytes = (world - 1) * per * total * 4 if wl.gather else 0  # gather: received by the root, one slice from every other rank
        exchange_info = {"mode": args.shard, "reduce_scatter_bytes_per_rank": rs_bytes, "gather_bytes_into_root": ga_bytes,
                         "root": gather_root if wl.gather else None,
                         "rows_per_rank": per, "row_floats": total, "chunks_per_step": wl.nch,
                         "overlap": "a chunk's collectives run beside the render of the step's next chunk (and the last one's beside the next step)"}
        exchange_info["rccl"] = rccl_info
        if native_comm is not None and not wl.time_sharded:
            # (the slots hold the collectives of the last two CHUNKS: a step's are wl.nch of them)
            ms = 0.5 * (native_comm.last_exchange_ms(0) + native_comm.last_exchange_ms(1)) * wl.nch
            tms = torch.tensor([ms], device=dev, dtype=torch.float64)
            dist.all_reduce(tms, op=dist.ReduceOp.MAX)
            ms = float(tms.item())
            exchange_info["collectives_ms"] = round(ms, 4)
            if ms > 0:  # bytes a rank sends in the reduce-scatter + what the root takes in, over the slower of the two
                exchange_info["xgmi_GBps_per_rank_out"] = round(rs_bytes / ms / 1e6, 1)
                exchange_info["xgmi_GBps_root_in"] = round((rs_bytes + ga_bytes) / ms / 1e6, 1)
    t_step = dt / args.steps
    value = M_total * total / t_step / 1e6
    rtf = (total / SAMPLE_RATE) / t_step

    # weak-scaling companion figure (N > 1): --objects per rank, short run, same line
    weak = None
    if world > 1 and args.scaling == "strong" and not args.stream_only:
        ww = Workload(cfg["objects"], cfg["hoa"], "weak", seed_base=100)
        wsteps = max(3, min(args.steps, 10))
        wdt, _ = ww.timed(wsteps, 2)
        weak = {"value": round(ww.M_total * total / (wdt / wsteps) / 1e6, 1), "unit": "Msamples/s",
                "objects_per_gpu": ww.M, "objects_total": ww.M_total, "steps": wsteps,
                "ms_per_step": round(wdt / wsteps * 1e3, 4), "scaling": "weak"}
        ww.close()
        del ww

    # the other decompositions of the same scene, short runs in the same line (N > 1: the first contact with a multi-GPU
    # node then measures all three)
    other_modes = None
    if world > 1 and args.scaling == "strong" and not args.stream_only and T % world == 0:
        other_modes = {}
        for mode in ("objects", "objects-nogather", "time"):
            if mode == args.shard:
                continue
            wo = Workload(cfg["objects"], cfg["hoa"], "strong", shard=mode)
            osteps = max(3, min(args.steps, 10))
            odt, _ = wo.timed(osteps, 2)
            other_modes[mode] = {"value": round(wo.M_total * total / (odt / osteps) / 1e6, 1), "unit": "Msamples/s",
                                 "ms_per_step": round(odt / osteps * 1e3, 4), "steps": osteps,
                                 "objects_per_gpu": wo.M, "blocks_per_gpu": wo.Tc * wo.nch,
                                 "note": {"objects": "reduce-scatter + gather of the shared bus on rank 0 (north_star)",
                                          "objects-nogather": "reduce-scatter only: the bus stays channel-sharded",
                                          "time": "no exchange: T / G blocks per rank behind one lead block"}[mode]}
            wo.close()
            del wo

    result = None
    if rank == 0:
        gain_b, dec_b, dm_b = algorithmic_bytes(wl.M_obj, N, B, K, wl.M_hoa)
        # a long call may be cut into several launches: per-step sums for the kernel table, per-launch
        # figures for the roofline (what rocprofv3 averages)
        # (several ranks: a step is wl.nch calls, every time_every-th CALL carries the events)
        timed_calls = len(range(0, args.steps * wl.nch, time_every))
        k1_launches = max(timing["gain_mix_launches"], 1) / timed_calls * wl.nch
        k1_ms = timing["gain_mix_ms"] / timed_calls * wl.nch
        k2_ms = timing["decor_ms"] / timed_calls * wl.nch
        k0_ms = timing["prep_ms"] / timed_calls * wl.nch
        T_rank = wl.Tc * wl.nch  # blocks this rank renders per step (time-sharded: its share of the stream + the lead block)
        achieved = gain_b * T_rank / (k1_ms * 1e-3) / 1e9
        whole = (gain_b + dec_b + dm_b) * T / t_step / 1e9
        if exchange_info is not None:
            # What the step should take under the decomposition that ran, from this rank's measured kernels and an ASSUMED
            # link rate (EARHIP_XGMI_GBPS, per direction and link; 50 of the nominal 64): libear_amd/distributed.py
            # exchange_model.  Objects modes: every rank renders its shard completely (K0 + K1 on M / G objects, K2 on all
            # loudspeakers), the exchange runs chunk by chunk beside the render: predicted = max(compute, exchange) + what the
            # first chunk's render and the last chunk's exchange leave uncovered.  Time mode: compute only.
            link_env = os.environ.get("EARHIP_XGMI_GBPS")
            link = float(link_env) if link_env else (link_measured if link_measured else 50.0)
            t_comp = k0_ms + k1_ms + k2_ms
            exchange_info["model"] = exchange_model(args.shard, world, n_pad, total, t_comp, link, wl.nch)
            exchange_info["model"]["measured_ms_per_step"] = round(t_step * 1e3, 4)
            exchange_info["model"]["link_rate_source"] = ("EARHIP_XGMI_GBPS" if link_env else "measured at start-up: exchange.rccl.link_probe"
                                                          if link_measured else "assumed (no native communicator to probe with)")
            exchange_info["model"]["note"] = "compute = this rank's kernels (HIP events); the exchange figures are a model"
        # what the fused chain itself moves: K1's bytes, plus (two buses) K2 reading the buses back and writing the outputs
        fused_b = gain_b + (4 * (K * N * B) + 4 * N * B + 2 * 4 * N * 255 / max(T, 1) if K == 2 else 0)
        whole_fused = fused_b * T / t_step / 1e9
        # HBM traffic of the dominant kernel: PMC counters cannot be read inside this process; the
        # figure is attached only when profiles/traffic.json was measured (rocprofv3 --pmc passes,
        # tools/profile_passes.sh) for exactly this kernel instantiation, shape and scene
        traffic, traffic_source, l2_requests = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                for tj in json.load(open(tpath)).get("entries", []):
                    same = (tj.get("objects") == M and tj.get("blocks") == T and tj.get("block_size") == B and
                            tj.get("channels") == N and tj.get("buses", 2) == K and tj.get("gain_kernel") == gain_kernel and
                            tj.get("tile") == plan["tile"] and tj.get("scene", "dense") == args.scene)
                    if same:
                        traffic = int(tj.get("gain_mix_hbm_bytes_per_step") / k1_launches)
                        traffic_source = "profiles/traffic.json: " + tj.get("kernel", "") + ", " + tj.get("source", "")
                        nreq = tj.get("raw_per_launch", {}).get("TCP_TCC_READ_REQ_sum")
                        if nreq:  # what the CUs ask L2 for (128 B a request): the inputs AND the gain rows, which hit in L2
                            l2_requests = int(128 * nreq / k1_launches)
            except Exception:
                traffic = None
        # what a kernel that only reads gets out of this box, measured in this run: the 100 % mark
        peak_measured = None
        try:
            if total % 256 == 0 and wl.in_stride % 256 == 0 and M >= 32:
                ms_lin, ms_rows = ctx.read_bandwidth(wl.x_full.data_ptr(), M, wl.in_stride, total, reps=5)
                peak_measured = {"linear_read_GBps": round(M * wl.in_stride * 4 / ms_lin / 1e6, 1),
                                 "row_pattern_read_GBps": round(M * total * 4 / ms_rows / 1e6, 1),
                                 "bytes": M * total * 4,
                                 "note": "read-only kernels over the input buffer of this run, HIP events, 5 launches each"}
        except Exception as e:  # a probe must never cost the measurement
            peak_measured = {"error": str(e)}
        best_read = max(peak_measured.get("linear_read_GBps", 0), peak_measured.get("row_pattern_read_GBps", 0)) \
            if peak_measured and "error" not in peak_measured else None

        workload = (f"BASELINE config {args.config}: {wl.M_obj} objects"
                    + (f" + {wl.M_hoa} bed channels (constant decode matrix)" if wl.M_hoa else "")
                    + f"{'/GPU' if world > 1 else ''} -> {cfg['layout']} ({N} ch), block {B}, 48 kHz: "
                    + ("ramped direct+diffuse gains, " + f"{N} decorrelators (512 taps), delay 255, mix" if K == 2
                       else "ramped gains only (one bus, no decorrelator)")
                    + f"; stream of {T} blocks per step")
        gains_desc = {"dense": "dense uniform(0,1), full-length ramp every block",
                      "adm": "dense uniform(0,1); metadata every 960 samples at a per-object phase, 240-sample ramp then constant",
                      "moving": "dense uniform(0,1); a new target every 240 samples at a per-object phase, always ramping",
                      "static": "dense uniform(0,1), one gain vector per object, never changing",
                      "mixed": "the dense scene with 8 of every 1024 objects on ADM-like metadata off the block grid",
                      "levels": "the dense scene; signal levels log-uniform 0 .. -90 dB, 30 % of the objects silent for the first half",
                      "levels-adm": "the adm scene; signal levels log-uniform 0 .. -90 dB, 30 % of the objects silent for the first half",
                      "bursty": "the dense scene; non-stationary signals (fades to -90 dB, silence, gated bursts, +40 dB blocks), 8 objects alone on a loudspeaker and 80 dB down most of the call",
                      "bursty-adm": "the adm scene; non-stationary signals as in `bursty`",
                      "bursty-moving": "the moving scene; non-stationary signals as in `bursty`",
                      "panned": "moving point sources, a new position every block: gains from the device panner "
                                "(3-sparse VBAP gains, diffuse split, zero LFE columns)",
                      "panned-adm": "moving point sources, a new position every 960 samples at a per-object phase: "
                                    "gains from the device panner"}[args.scene]
        result = {
            "metric": "Msamples/s", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preconditioning_steps": pre_steps,
            "ms_per_step": round(t_step * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": GAIN_DTYPES.get(gain_kernel, "f32"), "data": "synthetic",
            "rtf": round(rtf, 1),
            "config": {
                "workload": workload, "baseline_config": args.config,
                "objects_per_gpu": M, "objects_total": M_total, "channels": N, "block": B,
                "blocks_per_step": T, "buses": K, "scene": args.scene, "gains": gains_desc,
                "input_row_stride_samples": wl.in_stride,
                "buffers": {"input": hex(wl.x_full.data_ptr()), "outputs": [hex(o.data_ptr()) for o in wl.outs]},
                "shard": args.shard if world > 1 else None,
                "parallelism": ((f"time-sharded over {world} GPUs: every rank renders all objects for T / G blocks of the stream "
                                 "behind one lead block; no exchange") if wl.time_sharded else
                                f"objects sharded over {world} GPU(s), reduce-scatter of the bus over channels"
                                + (f" + gather of the owned slices on rank {gather_root}" if world > 1 and backend == "nccl" and wl.gather else "")
                                + (": libearhip's own RCCL communicator (earhip_comm, api_comm.hip)" if native_comm is not None
                                   else f": torch.distributed ({backend})" if world > 1 else "")
                                + (f" [{native_note}]" if native_note else "")
                                + (" [ranks share GPUs: functional check, not a measurement]" if shared_gpus else "")),
                "strict": bool(args.strict)},
            "roofline": {"bound": "hbm", "kernel": GAIN_KERNELS.get(gain_kernel, "?"),
                         "plan": {"tile_samples": plan["tile"], "tiles": plan["ntiles"], "object_splits": plan["gsplit"],
                                  "scratch_MB": scratch_mb,
                                  "form": None if wide_form is None else ("wide, f32 kink products" if hinge_robust else "wide" if wide_form else "plain"),
                                  "handed_to_standby_lists": bool(hinge_standby)},
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "l1_to_l2_read_requests": None if not l2_requests else {
                             "bytes_per_launch": l2_requests,
                             "GBps": round(l2_requests / (k1_ms / k1_launches) / 1e6, 1),
                             "note": "TCP_TCC_READ_REQ x 128 B of the same PMC passes over this run's launch time: every read the "
                                     "CUs send to L2 (inputs from HBM + gain rows that hit in L2); compare with peak_measured — a "
                                     "read-only kernel's rate through the same path"},
                         "peak_measured": peak_measured,
                         "frac_of_measured_read": round(achieved / best_read, 4) if best_read else None,
                         "launches_per_step": round(k1_launches, 2),
                         "algorithmic_bytes_per_launch": int(gain_b * T_rank / k1_launches),
                         "avg_launch_ms": round(k1_ms / k1_launches, 4)},
            "roofline_mfma": mfma_roofline(gain_kernel, K * M * N * B * T_rank, k1_ms),
            "exchange_check": ({"max_rel_diff_one_lead_block_vs_three": float(f"{lead_check:.3e}"),
                                "bit_identical": bool(lead_check == 0.0), "pass": bool(lead_check <= 5e-7)} if lead_check is not None else
                               None if exchange_err is None else
                               {"max_rel_err_owned_slice_vs_all_reduce": float(f"{exchange_err:.3e}"),
                                "same_partials_exchanged_twice_bit_identical": exchange_repeat_identical}),
            "exchange": exchange_info,
            "weak_scaling": weak,
            "other_shard_modes": other_modes,
            "kernel_timing": {"timed_steps": timed_steps, "every": time_every},
            "kernels_ms": {"seg_prep": round(k0_ms, 4), "gain_mix": round(k1_ms, 4),
                           "decorrelate_delay_mix": round(k2_ms, 4)},
            "whole_path": {"fused_path_GBps": round(whole_fused, 1), "frac_of_hbm_peak": round(whole_fused / HBM_PEAK_GBS, 4),
                           "survey_8d_algorithmic_GBps": round(whole, 1),
                           "note": "fused_path: the bytes this implementation's chain has to move per step (inputs, both gain rows of "
                                   "every block, the buses written by K1 and read once by K2, the delay state, the outputs) over the "
                                   "step time - the fraction is of THOSE bytes.  survey_8d_algorithmic: SURVEY 8(d)'s per-block bytes, "
                                   "which also count the BlockConvolver's spectra, filter and queue traffic that the fused K2 keeps in "
                                   "registers and LDS: a rate, not a fraction of anything HBM does",
                           "gain_fp32_equivalent_tflops": round(4.0 * K * M * N * B * T / (k1_ms * 1e-3) / 1e12, 2)},
        }

        # ---- distribution over steps: each step between its own pair of events on the launch stream
        # (the events cost the GPU a few us of idle time per step: these are not the headline figure)
        if world == 1 and not args.stream_only and not args.brief:
            n_ev = max(10, min(args.steps, 100))
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
            for i, (e0, e1) in enumerate(evs):
                e0.record(stream)
                wl.step(i)
                e1.record(stream)
            torch.cuda.synchronize()
            ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
            result["step_ms"] = {"median": round(ms[len(ms) // 2], 4), "p95": round(ms[int(0.95 * (len(ms) - 1))], 4),
                                 "min": round(ms[0], 4), "max": round(ms[-1], 4), "steps": n_ev,
                                 "note": "per-step HIP events in a separate pass after the timed region"}

        # ---- the same workload on the exact-f32 gain kernels (the default kernel splits f32 operands into
        # two f16 pieces; these do not): f32 MFMA and strict VALU (libear's arithmetic, bit-identical)
        if world == 1 and not args.stream_only and not args.strict and not args.brief:
            keep = os.environ.get("EARHIP_MFMA")
            os.environ["EARHIP_MFMA"] = "1"
            try:
                ctx1 = capi.Context(dev_index, stream.cuda_stream)  # (the knob is read when a context is created)
            finally:
                os.environ.pop("EARHIP_MFMA", None)
                if keep is not None:
                    os.environ["EARHIP_MFMA"] = keep
            r1 = wl.renderer(ctx1)
            n1 = max(3, min(args.steps, 10))
            dt1, _ = wl.timed(n1, 2, args.precondition_ms, r=r1, timing_every=1)  # (the GPU idled while the renderer was set up: low-power clocks)
            tm1 = r1.get_timing()
            k1 = r1.last_plan()["kernel"]
            r1.close()
            ctx1.set_strict(True)
            r0 = wl.renderer(ctx1)
            n0 = max(2, min(args.steps, 3))
            dt0, _ = wl.timed(n0, 1, args.precondition_ms, r=r0)
            r0.close()
            ctx1.close()
            result["value_f32_exact"] = {"value": round(M_total * total / (dt1 / n1) / 1e6, 1), "unit": "Msamples/s",
                                         "kernel": GAIN_KERNELS.get(k1, "?"), "steps": n1,
                                         "ms_per_step": round(dt1 / n1 * 1e3, 4)}
            # ... and beside the split-operand figure inside `roofline`, so that a reader of that object alone sees what the same
            # workload does on exact-f32 arithmetic (its gain kernel against the same algorithmic bytes, and against the fp32
            # matrix pipe that bounds it: 2 MACs per object, column and sample)
            f32_k1_ms = tm1["gain_mix_ms"] / max(n1, 1)
            if f32_k1_ms > 0:
                result["roofline"]["f32_exact"] = {
                    "value": result["value_f32_exact"]["value"], "unit": "Msamples/s", "kernel": GAIN_KERNELS.get(k1, "?"),
                    "ms_per_step": result["value_f32_exact"]["ms_per_step"], "gain_kernel_ms": round(f32_k1_ms, 4),
                    "achieved": round(gain_b * T / (f32_k1_ms * 1e-3) / 1e9, 1), "frac": round(gain_b * T / (f32_k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_of_fp32_matrix_peak": round(4.0 * K * M * N * B * T / (f32_k1_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}
            result["value_strict"] = {"value": round(M_total * total / (dt0 / n0) / 1e6, 1), "unit": "Msamples/s",
                                      "kernel": GAIN_KERNELS[0], "steps": n0, "ms_per_step": round(dt0 / n0 * 1e3, 4)}

        # ---- block mode (the latency figure): ONE block per call through the host-pointer
        # entry point, i.e. including H2D of the inputs, K0/K1/K2 and D2H of the outputs.  Reported
        # beside the stream-mode value, never as `value`.
        if world == 1 and not args.stream_only and not args.brief:
            rb = wl.renderer(ctx, max_blocks=1, npoints=34)
            import ctypes
            xb = np.ascontiguousarray(wl.x[:, :B].cpu().numpy())
            ob = np.zeros((N, B), np.float32)
            ip, op = capi._chan_ptrs(xb), capi._chan_ptrs(ob)  # built once: not part of a call
            lib = capi.load()
            for _ in range(23):
                capi.check(lib.earhip_render_process(rb.h, ctypes.c_size_t(1), ip, op))
            nb_calls = 200  # >= 200 timed blocks after 20 warm-up, median and p95
            lat = []
            for _ in range(nb_calls):
                b0 = time.perf_counter()
                capi.check(lib.earhip_render_process(rb.h, ctypes.c_size_t(1), ip, op))
                lat.append(time.perf_counter() - b0)
            lat.sort()
            bdt = lat[len(lat) // 2]
            # the same call with the channel buffers in pinned memory (earhip_host_alloc: the columns of a matrix
            # the device reaches): no gather into a staging buffer, outputs written in place
            xp, yp = ctx.pinned_array((max(M, 1), B)), ctx.pinned_array((N, B))
            xp[...] = xb
            ipp, opp = capi._chan_ptrs(xp), capi._chan_ptrs(yp)
            for _ in range(23):
                capi.check(lib.earhip_render_process(rb.h, ctypes.c_size_t(1), ipp, opp))
            latp = []
            for _ in range(nb_calls):
                b0 = time.perf_counter()
                capi.check(lib.earhip_render_process(rb.h, ctypes.c_size_t(1), ipp, opp))
                latp.append(time.perf_counter() - b0)
            latp.sort()
            rb.close()
            result["block_mode"] = {"ms_per_block": round(bdt * 1e3, 4), "p95_ms": round(lat[int(0.95 * len(lat))] * 1e3, 4),
                                    "mean_ms": round(sum(lat) / len(lat) * 1e3, 4), "calls": nb_calls,
                                    "rtf": round((B / SAMPLE_RATE) / bdt, 1),
                                    "Msamples_per_s": round(M * B / bdt / 1e6, 1),
                                    "note": "one block per call from host channel pointers: staging copy, H2D, K0/K1/K2, D2H, sync",
                                    "pinned_buffers": {"ms_per_block": round(latp[len(latp) // 2] * 1e3, 4),
                                                       "p95_ms": round(latp[int(0.95 * len(latp))] * 1e3, 4),
                                                       "rtf": round((B / SAMPLE_RATE) / latp[len(latp) // 2], 1),
                                                       "note": "channel buffers from earhip_host_alloc: one strided H2D, "
                                                               "kernels, outputs written in place, sync"}}

        # ---- libear's own calling convention in the record: earhip_render_process from HOST channel pointers, stream-length
        # calls (src/dsp/variable_block_size_impl.cpp:44-81: `const float *const *in, float *const *out`) — PCIe-inclusive,
        # never `value`.  The call runs as a pipeline of time chunks (H2D / kernels / D2H on three streams); the 100 % mark is
        # what a plain hipMemcpyAsync of the same bytes from the same kind of memory takes, measured here.
        if world == 1 and not args.stream_only and not args.brief and M >= 16:
            hs = {"what": "earhip_render_process (host channel pointers: staging or strided DMA, H2D, K0-K2, D2H, sync) in calls of "
                          "64 and 256 blocks; frac = input bytes per second over the measured H2D rate of a plain copy from the same memory",
                  "never_value": True, "calls": []}
            hb_max = min(256, T)
            rh = wl.renderer(ctx, max_blocks=hb_max)
            xh_all = np.ascontiguousarray(wl.x[:, :hb_max * B].cpu().numpy())
            xp_all = ctx.pinned_array((max(M, 1), hb_max * B))
            xp_all[...] = xh_all
            h2d_ms_pin, d2h_ms_pin = ctx.copy_bandwidth(xp_all, reps=3)
            h2d_ms_page, _ = ctx.copy_bandwidth(xh_all, reps=2)
            hs["h2d_GBps_measured"] = {"pinned": round(xp_all.nbytes / h2d_ms_pin / 1e6, 1), "pageable": round(xh_all.nbytes / h2d_ms_page / 1e6, 1),
                                       "d2h_pinned": round(xp_all.nbytes / d2h_ms_pin / 1e6, 1), "bytes": int(xp_all.nbytes)}
            xp_all[...] = xh_all  # (the D2H leg wrote the buffer back: the same bytes, restored for clarity)
            for hb in sorted({min(64, hb_max), hb_max}):
                xh = np.ascontiguousarray(xh_all[:, :hb * B])
                xp = ctx.pinned_array((max(M, 1), hb * B))
                xp[...] = xh
                yp = ctx.pinned_array((N, hb * B))
                yh = np.zeros((N, hb * B), np.float32)  # (the caller's own output rows, touched: a C caller has them)
                import ctypes
                for src, (xa, ya) in (("pageable", (xh, yh)), ("pinned", (xp, yp))):
                    ipp_, opp_ = capi._chan_ptrs(xa), capi._chan_ptrs(ya)  # channel-pointer arrays built once: not part of a call

                    def call():
                        rh.reset(0)
                        capi.check(capi.load().earhip_render_process(rh.h, ctypes.c_size_t(hb), ipp_, opp_))
                        return ya
                    first = np.array(call())
                    # (untimed calls for 0.2 s first: the staging threads have just been woken or made — the first ~100 ms of calls
                    # run at 0.6 of the steady rate while the scheduler spreads them and their cores leave their idle states;
                    # an offline render makes such calls back to back for minutes)
                    w0 = time.perf_counter()
                    while time.perf_counter() - w0 < 0.2:
                        call()
                    ts = []
                    for _ in range(7):
                        c0 = time.perf_counter()
                        call()
                        ts.append(time.perf_counter() - c0)
                    hdt = sorted(ts)[len(ts) // 2]
                    gbps = xa.nbytes / hdt / 1e9
                    ref_rate = hs["h2d_GBps_measured"]["pinned"]
                    hs["calls"].append({"source": src, "blocks_per_call": hb, "ms_per_call": round(hdt * 1e3, 3),
                                        "Gsamples_per_s": round(M * hb * B / hdt / 1e9, 2), "GBps_in": round(gbps, 1),
                                        "h2d_GBps_measured": ref_rate, "frac": round(gbps / ref_rate, 3),
                                        "ms_per_call_min_max": [round(min(ts) * 1e3, 3), round(max(ts) * 1e3, 3)],
                                        "frac_of_pageable_copy": round(gbps / hs["h2d_GBps_measured"]["pageable"], 3) if src == "pageable" else None,
                                        # (the same blocks as the timed stream's first ones: held against the CPU path and the device-resident render)
                                        "max_channel_rel_rms_vs_cpu": None, "max_rel_diff_vs_stream_render": None, "_first": first})
                ctx.release(xp)
                ctx.release(yp)
            ctx.release(xp_all)
            rh.close()
            result["host_stream"] = hs

        # ---- parity gate on the TIMED output: every step starts from reset(0), so the first blocks of
        # the buffer the last timed step wrote are the first blocks of the stream, produced by the very
        # launch plan that was timed; the CPU oracle renders the same blocks.  Per channel.
        if not args.stream_only and (world == 1 or backend == "nccl"):
            import _oracle  # the checker; used only below (parity gate and cpu_baseline)
            nb = 4
            last = (args.steps - 1) % 2
            wl.step(last, exchange_outputs=False, keep_head=world > 1)  # the timed call again (the passes above reused its buffer)
            torch.cuda.synchronize()
            parity_plan = wl.r.last_plan()
            buf = wl.head if world > 1 else wl.outs[wl.last_slot]  # (several ranks: the step's first call)
            nblk = wl.Tc
            base = wl.b0 - wl.lead  # absolute block of the buffer's first block (time-sharded ranks: not 0)

            def make_oracle():
                if K == 2:
                    return _oracle.ObjectsRenderer(max(M, 1), N, B, dec, delay)
                return _oracle.ObjectsRenderer(max(M, 1), N, B, np.zeros((N, 1), np.float32), 0)

            def window(b0, count):
                """blocks [b0, b0 + count) of the timed buffer and the CPU path's render of them (one lead block in front of
                a window inside the stream — as many as the FIRs and the delay reach back: tail, history and delay line of the
                oracle are the stream's by then)"""
                a0 = base + b0
                # (the decorrelated bus reaches 511 samples back, the delayed direct bus 255: ceil(511 / B) lead blocks)
                lead = min(a0, lead_blocks(B, 512 if K == 2 else 1, delay))
                lo, hi = (a0 - lead) * B, (a0 + count) * B
                xs_ = wl.x[:, lo:hi].cpu().numpy()
                win_ = scenes.window_curves(wl.curves, lo, hi)
                o = make_oracle()
                for m, (t, d, f) in enumerate(win_):
                    o.set_points(m, 0, t, d)
                    o.set_points(m, 1, t, f if K == 2 else np.zeros_like(d))
                return buf[:N, b0 * B:(b0 + count) * B].cpu().numpy(), o.process(xs_)[:, lead * B:], xs_, win_

            # three windows of the timed buffer: its first, middle and last blocks (tests/test_gpu_render_full.py::check_windows)
            starts = [wl.lead]  # (a time-sharded rank's lead block is not output)
            if nblk >= 3 * nb + 2:
                starts += [nblk // 2 - nb // 2, nblk - nb]
            got, want, xs, win = window(starts[0], nb)
            if result.get("host_stream") and world == 1:
                # the host-pointer calls' own outputs: their first blocks against the CPU path (the gate: per channel, 1e-6 — a call
                # from host pointers runs as chunks of a few blocks, i.e. on the launch plans of short calls), and all of them
                # against the device-resident render of the same blocks (information: two plans of the same arithmetic)
                for e in result["host_stream"]["calls"]:
                    f_ = e.pop("_first")
                    ref_ = buf[:N, :f_.shape[1]].cpu().numpy()
                    e["max_rel_diff_vs_stream_render"] = float(f"{scenes.rel_rms_per_channel(f_, ref_):.3e}")
                    e["max_channel_rel_rms_vs_cpu"] = float(f"{scenes.rel_rms_per_channel(f_[:, :nb * B], want):.3e}")
                result["host_stream"]["pass"] = bool(all(e["max_channel_rel_rms_vs_cpu"] <= 1e-6 and e["max_rel_diff_vs_stream_render"] <= 2e-6
                                                         for e in result["host_stream"]["calls"]))
            truth = scenes.render_f64([(t, d, f if K == 2 else None) for t, d, f in win], xs, N, dec, delay)
            windows = [{"first_block": base + starts[0], "blocks": nb, "rel_rms_vs_cpu": float(f"{scenes.rel_rms(got, want):.3e}"),
                        "max_channel_rel_rms_vs_cpu": float(f"{scenes.rel_rms_per_channel(got, want):.3e}")}]
            for b0 in starts[1:]:
                g_, w_, _, _ = window(b0, nb)
                windows.append({"first_block": base + b0, "blocks": nb, "rel_rms_vs_cpu": float(f"{scenes.rel_rms(g_, w_):.3e}"),
                                "max_channel_rel_rms_vs_cpu": float(f"{scenes.rel_rms_per_channel(g_, w_):.3e}"),
                                "finite": bool(np.isfinite(g_).all())})
            worst_ch = max(w["max_channel_rel_rms_vs_cpu"] for w in windows)
            result["parity"] = {"what": f"{len(windows)} windows of {nb} blocks (first / middle / last) of the timed call's own output buffer",
                                "kernel": GAIN_KERNELS.get(parity_plan["kernel"], "?"),
                                "plan": {"tile_samples": parity_plan["tile"], "tiles": parity_plan["ntiles"],
                                         "object_splits": parity_plan["gsplit"]},
                                "same_plan_as_timed": parity_plan == plan,
                                "rel_rms_vs_cpu": windows[0]["rel_rms_vs_cpu"],
                                "max_channel_rel_rms_vs_cpu": float(f"{worst_ch:.3e}"),
                                "windows": windows,
                                "gpu_rel_rms_vs_float64": float(f"{scenes.rel_rms(got, truth):.3e}"),
                                "cpu_rel_rms_vs_float64": float(f"{scenes.rel_rms(want, truth):.3e}"),
                                "max_channel_gpu_rel_rms_vs_float64": float(f"{scenes.rel_rms_per_channel(got, truth):.3e}"),
                                "max_abs": float(f"{np.max(np.abs(got - want)):.3e}"), "blocks": nb * len(windows),
                                "tolerance": 1e-6}
            # per channel, against the CPU path: a run whose timed output is off is not a measurement.  And the claim that most
            # of that distance is the CPU path's own sequential float32 sum is checked, not narrated: the GPU output must be
            # no further from a float64 render than the CPU path is (headline kernel; the others within 1.25 x: their chains of
            # MFMAs on one accumulator are longer)
            par = result["parity"]
            par["gpu_closer_to_float64_than_cpu"] = bool(par["gpu_rel_rms_vs_float64"] <= par["cpu_rel_rms_vs_float64"])
            # (asserted where the claim is made: the dense scenes, whose channels are sums over all objects — with three
            # loudspeakers per object the CPU path's sums are short and it is the more accurate of the two, both far below 1e-6)
            claim = args.scene in ("dense", "levels", "bursty", "mixed") and parity_plan["kernel"] == 3 and M >= 256 and not args.strict
            f64_ok = not claim or par["gpu_closer_to_float64_than_cpu"]
            par["float64_claim_checked"] = bool(claim)
            par["pass"] = bool(par["max_channel_rel_rms_vs_cpu"] <= 1e-6 and f64_ok and all(w.get("finite", True) for w in windows))

        # ---- CPU baseline: the scalar restatement on this host, bounded sample -----------------
        if world == 1 and args.cpu_blocks > 0 and not args.stream_only and not args.brief:
            import _oracle
            cb = min(args.cpu_blocks, T)
            xc = wl.x[:, :cb * B].cpu().numpy()
            cwin = scenes.window_curves(wl.curves, 0, cb * B)
            fir = dec if K == 2 else np.zeros((N, 1), np.float32)

            def make(lo, hi, native):
                oc = _oracle.ObjectsRenderer(hi - lo, N, B, fir, delay, native=native)
                for j, m in enumerate(range(lo, hi)):
                    t, d, f = cwin[m]
                    oc.set_points(j, 0, t, d)
                    oc.set_points(j, 1, t, f if K == 2 else np.zeros_like(d))
                return oc
            res = {}
            for native in (False, True):
                oc = make(0, M, native)
                c0 = time.perf_counter()
                oc.process(xc)
                cdt = time.perf_counter() - c0
                res[native] = M * cb * B / cdt / 1e6
                del oc
            # generous variant: the objects sharded over host threads (the path is linear in the
            # objects; libear itself has no threading), march=native build, same sample
            from concurrent.futures import ThreadPoolExecutor
            nthreads = max(1, min(os.cpu_count() or 1, 16, M // 32))  # >= 32 objects per shard: each shard also decorrelates
            bounds = [(M * i // nthreads, M * (i + 1) // nthreads) for i in range(nthreads)]
            shards = [(make(lo, hi, True), np.ascontiguousarray(xc[lo:hi])) for lo, hi in bounds]
            c0 = time.perf_counter()
            with ThreadPoolExecutor(nthreads) as ex:  # ctypes releases the GIL during the call
                parts = list(ex.map(lambda s: s[0].process(s[1]), shards))
            total_out = parts[0]
            for p_ in parts[1:]:
                total_out = total_out + p_
            mt = M * cb * B / (time.perf_counter() - c0) / 1e6
            del shards
            # the gain stage alone in libear's two forms: one GainInterpolator<LinearInterpVector> per object and
            # bus + sum (docs/dsp.rst:54-56; what the composition above does) and the fused
            # GainInterpolator<LinearInterpMatrix> per bus (gain_interpolator.hpp:250-278)
            fb = min(cb, 32)
            xf = np.ascontiguousarray(xc[:, :fb * B])
            fwin = scenes.window_curves(wl.curves, 0, fb * B)
            forms = {}
            if args.scene in ("dense", "panned") and not wl.M_hoa:  # (the fused form needs one common time axis)
                tms = fwin[0][0]
                vals = [np.stack([c[1 + b] for c in fwin], axis=1) for b in range(K)]  # [points][M][N]
                c0 = time.perf_counter()
                for b in range(K):
                    _oracle.gain_interp("matrix", tms, vals[b], xf, [B] * fb)
                forms["fused_matrix"] = round(M * fb * B / (time.perf_counter() - c0) / 1e6, 2)
                c0 = time.perf_counter()
                for b in range(K):
                    acc = np.zeros((N, fb * B), np.float32)
                    for m in range(M):
                        acc += _oracle.gain_interp("vector", tms, vals[b][:, m:m + 1, :], xf[m:m + 1], [B] * fb)
                forms["per_object_vector_plus_sum"] = round(M * fb * B / (time.perf_counter() - c0) / 1e6, 2)
                forms["sample"] = f"first {fb} blocks, {K} bus(es), -O3 -DNDEBUG, 1 thread (the per-object form sums in numpy)"
            result["cpu_baseline"] = {
                "value": round(res[False], 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": f"first {cb} blocks of the same scene ({M} inputs -> {N} ch), scalar C++14 restatement "
                          "of libear's per-object GainInterpolator<LinearInterpVector> + bus sum"
                          + (", BlockConvolver, DelayBuffer" if K == 2 else "") + "; -O3 -DNDEBUG (libear Release flags), 1 thread",
                "rtf": round((B / SAMPLE_RATE) / (M * B / (res[False] * 1e6)), 3),
                "value_march_native": round(res[True], 2),
                "value_sharded_threads": round(mt, 1), "threads": nthreads,
                "gain_stage_forms_Msamples_per_s": forms or None,
                "cpu_model": cpu_model(), "host_cpus": os.cpu_count()}
        # ---- the other workloads in the same record (default invocation only): short runs of the other BASELINE
        # configurations and of the scenes whose curves take the other gain kernels, each a child process of its own
        # (`--brief`: timed steps between barriers exactly as above, per-kernel events, parity gate on its own timed buffer)
        default_call = (world == 1 and args.config == "C4" and args.scene == "dense" and not args.strict and not args.brief
                        and not args.stream_only and not args.no_secondary and args.row_pad == 256
                        and all(getattr(args, k) is None for k in ("objects", "hoa", "blocks", "block_size", "layout", "buses")))
        if default_call:
            result["secondary"] = secondary_runs(max(5, min(args.steps, 80)))
            # a secondary workload that errored or failed ITS parity gate is visible in the line and in the exit code (4)
            result["secondary_pass"] = bool(all("error" not in e and e.get("parity", {}).get("pass") for e in result["secondary"]))
            secondary_failed = not result["secondary_pass"]
        for e in (result.get("host_stream") or {}).get("calls", []):
            e.pop("_first", None)
        print(json.dumps(result), flush=True)
        if secondary_failed:
            bad = [e["workload"] for e in result["secondary"] if "error" in e or not e.get("parity", {}).get("pass")]
            print(f"bench.py: secondary workloads failed (error or parity): {bad}", file=sys.stderr, flush=True)
        if lead_check is not None and not lead_check <= 5e-7:  # (time sharding: a rank's blocks must not depend on how far back its lead-in starts)
            print(f"bench.py: time-sharded blocks differ with the length of the lead-in by {lead_check:.3e}", file=sys.stderr, flush=True)
            parity_failed = True
        if result.get("host_stream") and result["host_stream"].get("pass") is False:
            print("bench.py: the host-pointer stream calls are beyond 1e-6 from the CPU path (or 2e-6 from the device-resident render)", file=sys.stderr, flush=True)
            parity_failed = True
        if result.get("parity") and not result["parity"]["pass"]:
            print("bench.py: PARITY FAILED - the timed output differs from the CPU path by "
                  f"{result['parity']['max_channel_rel_rms_vs_cpu']:.3e} (tolerance 1e-6; from a float64 render: GPU "
                  f"{result['parity']['gpu_rel_rms_vs_float64']:.3e}, CPU path {result['parity']['cpu_rel_rms_vs_float64']:.3e}): "
                  "the value above is invalid",
                  file=sys.stderr, flush=True)
            parity_failed = True

    wl.close()
    if native_comm is not None:
        native_comm.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        sys.exit(3)
    if rank == 0 and secondary_failed:
        sys.exit(4)


SECONDARY = (("C2", ["--config", "C2"]), ("C3", ["--config", "C3"]), ("C5", ["--config", "C5"]),
             ("C4 adm", ["--scene", "adm"]), ("C4 moving", ["--scene", "moving"]), ("C4 bursty", ["--scene", "bursty"]),
             ("C4 bursty-moving", ["--scene", "bursty-moving"]), ("128 objects (a rank's share at 8 GPUs)", ["--objects", "128"]))


def secondary_runs(steps):
    """the `secondary` array of the default line: every entry one `bench.py --brief` child (a fresh process: its own context,
    scene and buffers; this process keeps the GPU but is idle meanwhile), reduced to the figures that matter"""
    out = []
    for name, extra in SECONDARY:
        cmd = [sys.executable, os.path.abspath(__file__), "--brief", "--no-secondary", "--steps", str(steps), "--warmup", "20",
               "--kernel-timing-every", "4"] + extra
        t0 = time.perf_counter()
        entry = {"workload": name, "args": " ".join(extra)}
        try:
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
            line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            if not line:
                raise RuntimeError(f"exit code {res.returncode}: {res.stderr[-300:]}")
            d = json.loads(line[-1])
            r, par = d["roofline"], d.get("parity") or {}
            entry.update({"value": d["value"], "unit": d["unit"], "rtf": d["rtf"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                          "kernels_ms": d.get("kernels_ms"),
                          "roofline": {"kernel": r["kernel"], "plan": r["plan"], "frac": r["frac"], "achieved_GBps": r["achieved"],
                                       "avg_launch_ms": r["avg_launch_ms"]},
                          "parity": {"max_channel_rel_rms_vs_cpu": par.get("max_channel_rel_rms_vs_cpu"),
                                     "gpu_rel_rms_vs_float64": par.get("gpu_rel_rms_vs_float64"),
                                     "cpu_rel_rms_vs_float64": par.get("cpu_rel_rms_vs_float64"),
                                     "same_plan_as_timed": par.get("same_plan_as_timed"), "pass": par.get("pass")},
                          "scene": d["config"]["scene"], "workload_detail": d["config"]["workload"]})
        except Exception as e:  # noqa: BLE001 - a secondary run must never cost the headline line
            entry["error"] = str(e)[:400]
        entry["wall_s"] = round(time.perf_counter() - t0, 1)
        out.append(entry)
    return out


if __name__ == "__main__":
    main()
