#!/usr/bin/env python3
"""bench.py — the reference's headline workload on MI355X:

    N objects -> 9+10+3 (22.2, 24 ch), block 512, 48 kHz: ramped direct+diffuse gain matrices,
    24 decorrelators (512-tap FIR, FFT 1024), 255-sample compensation delay, mix-down.

One "step" = one stream-mode pass of the hot path over T consecutive blocks (default 1024, i.e.
10.9 s of audio) for 1024 objects per GPU, inputs and gain curves already resident in HBM.  Every
block is a full-length ramp between dense uniform(0,1) gain vectors (worst case).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3

Rank 0 prints ONE JSON line.  `value` = object-samples consumed per second over all GPUs
(Msamples/s = M*B*T / t_step / 1e6); `rtf` = real-time factor (B*T/48000) / t_step.
Multi-GPU: objects are sharded (1024 per rank, weak scaling; `--scaling strong` splits 1024 objects
over the ranks instead), each rank renders its shard and the partial loudspeaker buses are summed by
one RCCL reduce-scatter over the channel axis, overlapped with the next step's render.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3  # vector / f32-MFMA peak, same guide
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, same guide
SAMPLE_RATE = 48000.0


def algorithmic_bytes(m, n, b, k, ramp=True):
    """bytes_gain, bytes_dec, bytes_delay_mix per block"""
    e = 2 if ramp else 1
    gain = 4 * (m * b + e * k * m * n + k * n * b)
    p, d = 1, 255
    dec = n * (4 * b + 8 * p * (b + 1) + 2 * 8 * p * (b + 1) + 2 * 4 * b + 4 * b) if k == 2 else 0
    dm = n * (4 * b + 2 * 4 * d + 4 * b + 4 * b) if k == 2 else 0
    return gain, dec, dm


GAIN_KERNELS = {0: "k_gain_mix (VALU, strict)", 1: "k_gain_mix_mfma (f32 MFMA)", 2: "k_gain_mix_bf3 (bf16x3 MFMA)",
                3: "k_gain_mix_h2 (f16x2 MFMA)"}


def mfma_roofline(kind, macs_per_term, k1_ms):
    """The gain kernel against the matrix pipe it runs on (secondary to the HBM roofline)."""
    if kind == 2:   # 2 operands (B0, B1) x 6 bf16 partial products per object, column and sample
        flops, peak, what = 24.0 * macs_per_term, BF16_PEAK_TFLOPS, "bf16 MFMA flops: 6 partial products x {gain at tile start, slope}"
    elif kind == 3:  # 2 operands (B0, B1) x 3 f16 partial products per object, column and sample
        flops, peak, what = 12.0 * macs_per_term, BF16_PEAK_TFLOPS, "f16 MFMA flops: 3 partial products x {gain at tile start, slope}"
    else:           # 2 f32 MACs (start, end gain row) per object, column and sample
        flops, peak, what = 4.0 * macs_per_term, FP32_PEAK_TFLOPS, "f32 flops: 2 MACs per object, column and sample for a ramp"
    ach = flops / (k1_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": GAIN_KERNELS.get(kind, "?"), "achieved": round(ach, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(ach / peak, 4), "note": "executed " + what}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--kernel-timing-every", type=int, default=8,
                    help="record the per-kernel HIP events (roofline, kernels_ms) on every n-th timed step")
    ap.add_argument("--precondition-ms", type=float, default=40.0,
                    help="untimed steps of the same workload before the W warm-up steps, until this much wall time "
                         "has passed: an idle MI355X needs 10-20 ms of load to leave its low-power clocks "
                         "(steps measured right after start-up are ~9 %% slower)")
    ap.add_argument("--objects", type=int, default=1024, help="objects per GPU")
    ap.add_argument("--blocks", type=int, default=1024, help="blocks per step (stream length T)")
    ap.add_argument("--block-size", type=int, default=512)
    ap.add_argument("--layout", default="9+10+3")
    ap.add_argument("--cpu-blocks", type=int, default=160, help="blocks of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--strict", action="store_true", help="bit-exact gain arithmetic (slower)")
    ap.add_argument("--scene", choices=("dense", "adm", "static", "moving", "mixed"), default="dense",
                    help="dense: block-aligned full-length ramps (headline, SURVEY 8d); adm: metadata blocks of "
                         "960 samples at a random phase per object, 240-sample ramps then constant (not block-aligned)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --objects per GPU (default); strong: --objects in total, split over the GPUs")
    ap.add_argument("--row-pad", type=int, default=0,
                    help="floats of padding between the input rows (row stride = samples + pad; multiple of 4)")
    ap.add_argument("--stream-only", action="store_true",
                    help="only the timed stream-mode steps (no block-mode, parity or CPU legs): for PMC profiling passes")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # one process per GPU; EARHIP_BENCH_BACKEND=gloo lets several ranks share one GPU (a functional
    # check of the multi-rank control flow on a single-GPU box, not a measurement)
    backend = os.environ.get("EARHIP_BENCH_BACKEND", "nccl")
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
    from libear_amd.distributed import exchange

    from libear_amd.distributed import shard_range
    names = LAYOUTS[args.layout]
    N, B, T, K = len(names), args.block_size, args.blocks, 2
    if args.scaling == "strong":  # the scene has --objects in total; this rank renders its shard
        lo, hi = shard_range(args.objects, rank, world)
        M, M_total = hi - lo, args.objects
    else:
        M, M_total = args.objects, args.objects * world
    total = B * T

    # decorrelator FIRs: designed natively (libearhip group G, setup path)
    dec = capi.design_decorrelators(names)

    # ---- scene: resident in HBM before the timed region --------------------------------------
    if args.scene == "adm":
        curves = scenes.adm_curves(M, N, total, seed=11 + rank)
    elif args.scene == "moving":  # always ramping: a new target every 240 samples (5 ms) at a per-object phase
        curves = scenes.adm_curves(M, N, total, period=240, ramp=240, seed=12 + rank)
    elif args.scene == "static":  # one gain vector per object and bus, never changing
        curves = scenes.constant_curves(M, N, seed=8 + rank)
    elif args.scene == "mixed":  # the headline scene with 8 of every 1024 objects on ADM-like metadata off the block grid
        curves = scenes.dense_curves(M, N, B, T, seed=7 + rank)
        odd = scenes.adm_curves(max(M // 128, 1), N, total, seed=11 + rank)
        for i, c in enumerate(odd):
            curves[(128 * i + 7) % M] = c
    else:
        curves = scenes.dense_curves(M, N, B, T, seed=7 + rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    in_stride = total + args.row_pad
    x_full = torch.rand((M, in_stride), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
    x = x_full[:, :total]  # planar rows, in_stride floats apart
    outs = [torch.zeros((N, total), device=dev, dtype=torch.float32) for _ in range(2)]
    owned = [torch.zeros((N // world, total), device=dev, dtype=torch.float32) for _ in range(2)] \
        if world > 1 else None

    stream = torch.cuda.current_stream(dev)
    ctx = capi.Context(dev_index, stream.cuda_stream)
    ctx.set_strict(args.strict)
    r = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=T)
    for m, (t, d, f) in enumerate(curves):
        r.set_object_points(m, t, d, f)
    r.commit()

    pending = [None, None]

    def step(i, exchange_outputs=True):
        buf = i % 2
        if pending[buf] is not None:  # the exchange that last read this buffer must be done
            pending[buf].wait()
            pending[buf] = None
        r.reset(0)
        r.process_device(T, x.data_ptr(), in_stride, outs[buf].data_ptr(), total)
        if world > 1 and exchange_outputs:
            _, work = exchange(outs[buf], owned[buf], async_op=True)
            pending[buf] = work

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    # bring the device out of its idle power state (setup, like the allocation above: not a step count)
    pre_steps = 0
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.precondition_ms:
        step(pre_steps, exchange_outputs=False)  # wall-clock bounded: ranks may differ, so no collective here
        pre_steps += 1
        if pre_steps % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # per-kernel HIP events on every TIME_EVERY-th step of the timed region (each timed step records six
    # events = ~20 us of idle GPU; timing every step costs 4 % of the headline value)
    time_every = max(1, min(args.kernel_timing_every, args.steps))
    timed_steps = len(range(0, args.steps, time_every))
    r.enable_timing(time_every)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing = r.get_timing()
    gain_kernel = r.gain_kernel()
    r.enable_timing(False)

    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # multi-rank self-check (outside the timed region, all ranks take part): the slice this rank
    # owns after the reduce-scatter equals the sum of all ranks' partial outputs
    exchange_err = None
    if world > 1 and (backend == "nccl" or os.environ.get("EARHIP_BENCH_CHECK") == "force"):
        from libear_amd.distributed import channel_range
        w = min(total, 4096)
        last = (args.steps - 1) % 2
        ref = outs[last][:, :w].clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        lo, hi = channel_range(N, rank, world)
        got = owned[last][:, :w] if backend == "nccl" else outs[last][lo:hi, :w]
        err = ((got - ref[lo:hi]).abs().max() / ref.abs().max().clamp_min(1e-30)).to(torch.float64).reshape(1)
        dist.all_reduce(err, op=dist.ReduceOp.MAX)
        exchange_err = float(err.item())
    t_step = dt / args.steps
    value = M_total * total / t_step / 1e6
    rtf = (total / SAMPLE_RATE) / t_step

    result = None
    if rank == 0:
        gain_b, dec_b, dm_b = algorithmic_bytes(M, N, B, K)
        # a long call is cut into chunks (K1 of chunk c+1 overlaps K2 of chunk c): per-step sums for
        # the kernel table, per-launch figures for the roofline (what rocprofv3 averages)
        k1_launches = max(timing["gain_mix_launches"], 1) / timed_steps
        k1_ms = timing["gain_mix_ms"] / timed_steps
        k2_ms = timing["decor_ms"] / timed_steps
        k0_ms = timing["prep_ms"] / timed_steps
        achieved = gain_b * T / (k1_ms * 1e-3) / 1e9
        whole = (gain_b + dec_b + dm_b) * T / t_step / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("objects") == M and tj.get("blocks") == T and tj.get("block_size") == B:
                    # measured per step (all K1 launches of one pass over T blocks), reported per launch
                    traffic = int(tj.get("gain_mix_hbm_bytes_per_step") / k1_launches)
            except Exception:
                traffic = None
        result = {
            "metric": "Msamples/s", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preconditioning_steps": pre_steps,
            "ms_per_step": round(t_step * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rtf": round(rtf, 1),
            "config": {
                "workload": f"{M} objects/GPU -> {args.layout} ({N} ch), block {B}, 48 kHz: ramped direct+diffuse "
                            f"gains, {N} decorrelators (512 taps), delay 255, mix; stream of {T} blocks per step",
                "objects_per_gpu": M, "objects_total": M_total, "channels": N, "block": B,
                "blocks_per_step": T, "buses": K,
                "gains": "dense uniform(0,1), full-length ramp every block" if args.scene == "dense" else
                         "dense uniform(0,1); metadata every 960 samples at a per-object phase, 240-sample ramp then constant",
                "parallelism": f"objects sharded over {world} GPU(s), reduce-scatter of the bus over channels",
                "strict": bool(args.strict)},
            "roofline": {"bound": "hbm", "kernel": GAIN_KERNELS.get(gain_kernel, "?"), "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "launches_per_step": round(k1_launches, 2),
                         "algorithmic_bytes_per_launch": int(gain_b * T / k1_launches),
                         "avg_launch_ms": round(k1_ms / k1_launches, 4)},
            "roofline_mfma": mfma_roofline(gain_kernel, K * M * N * B * T, k1_ms),
            "exchange_check": None if exchange_err is None else
                              {"max_rel_err_owned_slice_vs_all_reduce": float(f"{exchange_err:.3e}")},
            "kernel_timing": {"timed_steps": timed_steps, "every": time_every},
            "kernels_ms": {"seg_prep": round(k0_ms, 4), "gain_mix": round(k1_ms, 4),
                           "decorrelate_delay_mix": round(k2_ms, 4)},
            "whole_path": {"algorithmic_GBps": round(whole, 1), "frac_of_hbm_peak": round(whole / HBM_PEAK_GBS, 4),
                           "gain_fp32_equivalent_tflops": round(4.0 * K * M * N * B * T / (k1_ms * 1e-3) / 1e12, 2)},
        }

        # ---- block mode (the latency figure): ONE 512-sample block per call through the host-pointer
        # entry point, i.e. including H2D of the inputs, K0/K1/K2 and D2H of the outputs.  Reported
        # beside the stream-mode value, never as `value`.
        if world == 1 and not args.stream_only:
            rb = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=1)
            for m, (t, d, f) in enumerate(curves):
                rb.set_object_points(m, t[:34], d[:34], f[:34])
            rb.commit()
            import ctypes
            xb = np.ascontiguousarray(x[:, :B].cpu().numpy())
            ob = np.zeros((N, B), np.float32)
            ip, op = capi._chan_ptrs(xb), capi._chan_ptrs(ob)  # built once: not part of a call
            lib = capi.load()
            for _ in range(3):
                capi.check(lib.earhip_render_process(rb.h, ctypes.c_size_t(1), ip, op))
            for _ in range(20):
                capi.check(lib.earhip_render_process(rb.h, ctypes.c_size_t(1), ip, op))
            nb_calls = 200  # >= 200 timed blocks after 20 warm-up, median and p95
            lat = []
            for _ in range(nb_calls):
                b0 = time.perf_counter()
                capi.check(lib.earhip_render_process(rb.h, ctypes.c_size_t(1), ip, op))
                lat.append(time.perf_counter() - b0)
            lat.sort()
            bdt = lat[len(lat) // 2]
            rb.close()
            result["block_mode"] = {"ms_per_block": round(bdt * 1e3, 4), "p95_ms": round(lat[int(0.95 * len(lat))] * 1e3, 4),
                                    "mean_ms": round(sum(lat) / len(lat) * 1e3, 4), "calls": nb_calls,
                                    "rtf": round((B / SAMPLE_RATE) / bdt, 1),
                                    "Msamples_per_s": round(M * B / bdt / 1e6, 1),
                                    "note": "one block per call from host channel pointers: staging copy, H2D, K0/K1/K2, D2H, sync"}

        # ---- parity gate in the same run: first two blocks against the CPU oracle -------------
        if not args.stream_only:
            import _oracle  # the checker; used only below (parity gate and cpu_baseline)
            nb = 2
            xs = x[:, :nb * B].cpu().numpy()
            rr = capi.Renderer(ctx, M, N, B, dec, 255, max_blocks=nb)
            for m, (t, d, f) in enumerate(curves):
                rr.set_object_points(m, t[:nb + 1], d[:nb + 1], f[:nb + 1])
            got = rr.process(xs)
            rr.close()
            o = _oracle.ObjectsRenderer(M, N, B, dec, 255)
            for m, (t, d, f) in enumerate(curves):
                o.set_points(m, 0, t[:nb + 1], d[:nb + 1])
                o.set_points(m, 1, t[:nb + 1], f[:nb + 1])
            want = o.process(xs)
            truth = scenes.render_f64([(t[:nb + 1], d[:nb + 1], f[:nb + 1]) for t, d, f in curves], xs, N, dec, 255)
            result["parity"] = {"rel_rms_vs_cpu": float(f"{scenes.rel_rms(got, want):.3e}"),
                                "gpu_rel_rms_vs_float64": float(f"{scenes.rel_rms(got, truth):.3e}"),
                                "cpu_rel_rms_vs_float64": float(f"{scenes.rel_rms(want, truth):.3e}"),
                                "max_abs": float(f"{np.max(np.abs(got - want)):.3e}"), "blocks": nb,
                                "tolerance": 1e-6}

        # ---- CPU baseline: the scalar restatement on this host, bounded sample -----------------
        if world == 1 and args.cpu_blocks > 0 and not args.stream_only:
            cb = min(args.cpu_blocks, T)
            xc = x[:, :cb * B].cpu().numpy()
            res = {}
            for native in (False, True):
                oc = _oracle.ObjectsRenderer(M, N, B, dec, 255, native=native)
                for m, (t, d, f) in enumerate(curves):
                    oc.set_points(m, 0, t[:cb + 1], d[:cb + 1])
                    oc.set_points(m, 1, t[:cb + 1], f[:cb + 1])
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
            shards = []
            for lo, hi in bounds:
                oc = _oracle.ObjectsRenderer(hi - lo, N, B, dec, 255, native=True)
                for j, m in enumerate(range(lo, hi)):
                    t, d, f = curves[m]
                    oc.set_points(j, 0, t[:cb + 1], d[:cb + 1])
                    oc.set_points(j, 1, t[:cb + 1], f[:cb + 1])
                shards.append((oc, np.ascontiguousarray(xc[lo:hi])))
            c0 = time.perf_counter()
            with ThreadPoolExecutor(nthreads) as ex:  # ctypes releases the GIL during the call
                parts = list(ex.map(lambda s: s[0].process(s[1]), shards))
            total_out = parts[0]
            for p_ in parts[1:]:
                total_out = total_out + p_
            mt = M * cb * B / (time.perf_counter() - c0) / 1e6
            del shards
            result["cpu_baseline"] = {
                "value": round(res[False], 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": f"first {cb} blocks of the same scene ({M} objects -> {N} ch), scalar C++14 restatement "
                          "of libear's per-object GainInterpolator<LinearInterpVector> + bus sum, BlockConvolver, "
                          "DelayBuffer; -O3 -DNDEBUG (libear Release flags), 1 thread",
                "rtf": round((B / SAMPLE_RATE) / (M * B / (res[False] * 1e6)), 3),
                "value_march_native": round(res[True], 2),
                "value_sharded_threads": round(mt, 1), "threads": nthreads,
                "host_cpus": os.cpu_count()}
        print(json.dumps(result), flush=True)

    r.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
