"""Object sharding across the GPUs of one node and the one exchange step of the render path.

The loudspeaker bus is a sum over objects and everything after the buses (decorrelation, delay,
mix) is linear and per-channel, so each rank renders its own objects completely and the partial
outputs are summed: one reduce-scatter over the channel axis (every rank ends up owning N/G
loudspeaker channels of the shared bus), RCCL over xGMI on the GPU (`nccl` backend), `gloo` in the
CPU tests.  No other collective exists on the path.
"""
import torch
import torch.distributed as dist


def shard_range(n_objects, rank, world):
    """contiguous object range [lo, hi) rendered by `rank`"""
    lo = (n_objects * rank) // world
    hi = (n_objects * (rank + 1)) // world
    return lo, hi


def padded_channels(n_out, world):
    """rows of the exchange buffers: n_out rounded up to a multiple of the number of ranks (a
    reduce-scatter hands every rank the same number of rows; the rows past n_out stay zero)"""
    return -(-n_out // world) * world


def channel_range(n_out, rank, world):
    """loudspeaker channels [lo, hi) of the shared bus owned by `rank` after the exchange; ragged when
    n_out is not a multiple of the number of ranks (4+5+0 has 10 channels: at 4 ranks 3, 3, 3, 1; at 8
    ranks 2, 2, 2, 2, 2, 0, 0, 0)"""
    per = padded_channels(n_out, world) // world
    return min(rank * per, n_out), min((rank + 1) * per, n_out)


def exchange(partial, owned=None, group=None, async_op=False):
    """Sum the ranks' partial outputs.

    partial: [padded_channels(n_out, world), samples] float32: this rank's render of its objects in the
             first n_out rows, zeros below (n_out defaults to all rows).
    owned:   [padded / world, samples] output slice of this rank (nccl: reduce-scatter; its first
             hi - lo rows are the channels channel_range() names); ignored on backends without
             reduce_scatter_tensor (gloo), where an all-reduce is done in place and the owned slice is a
             view of `partial`.
    Returns (owned_tensor, work_or_None).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return partial, None
    rank = dist.get_rank(group)
    rows = partial.shape[0]
    assert rows % world == 0, "exchange buffers hold padded_channels(n_out, world) rows"
    per = rows // world
    backend = dist.get_backend(group)
    if backend == "nccl":
        if owned is None:
            owned = torch.empty((per, partial.shape[1]), dtype=partial.dtype, device=partial.device)
        work = dist.reduce_scatter_tensor(owned, partial, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return owned, work
    work = dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return partial[rank * per:(rank + 1) * per], work


# ---- the three decompositions of a render over G ranks -----------------------------------------
SHARD_MODES = ("objects", "objects-nogather", "time")


def lead_blocks(block, n_taps=512, delay=255):
    """blocks of lead-in that bring the DSP state to where the stream has it: the decorrelated bus reaches n_taps - 1 samples
    back, the delayed direct bus `delay` samples — ceil(max(n_taps - 1, delay) / block), at least one"""
    reach = max(int(n_taps) - 1, int(delay), 1)
    return max(1, -(-reach // int(block)))


def time_range(n_blocks, rank, world, partitions=1, delay_blocks=1):
    """`time` sharding: rank r renders blocks [b0, b1) of the stream for ALL objects, after `lead` blocks rendered only to
    bring the DSP state to where the stream has it (the overlap-add tail of a one-partition decorrelator depends on
    the previous block only, the 255-sample delay line on less than one: one lead block reproduces both exactly — the
    decorrelator kernel recomputes a run's first tail the same way).  No exchange at all; the outputs of different time
    ranges live on different ranks.  A decorrelator FIR of P partitions (blocks shorter than its 512 taps) reaches P blocks
    back: `partitions` lead blocks; a delay longer than that (`delay_blocks` = ceil(delay / block)) asks for its own
    (lead_blocks() gives both).  Returns (b0, b1, lead)."""
    b0 = (n_blocks * rank) // world
    b1 = (n_blocks * (rank + 1)) // world
    return b0, b1, min(b0, max(1, partitions, delay_blocks))


def exchange_model(mode, world, n_pad, samples, compute_ms, link_gbps, chunks=1):
    """What a step should take under a decomposition, from this rank's measured compute time and an ASSUMED xGMI link
    rate per direction (no multi-GPU box was available to measure one: every figure here is a model, not a measurement).

    objects:          reduce-scatter (a slice of n_pad / G rows to every peer, each over its own link) + gather of the
                      owned slices on one rank: two slices over one link each;
    objects-nogather: the consumer takes the bus channel-sharded: one slice;
    time:             no exchange (a lead block per rank is part of compute_ms)."""
    slice_ms = (n_pad // world) * samples * 4 / (link_gbps * 1e9) * 1e3
    t_ex = {"objects": 2.0 * slice_ms, "objects-nogather": slice_ms, "time": 0.0}[mode]
    pred = max(compute_ms, t_ex) + (min(compute_ms, t_ex) / max(chunks, 1) if t_ex > 0 else 0.0)
    return {"mode": mode, "link_GBps_per_direction": link_gbps, "compute_ms_per_step": round(compute_ms, 4),
            "exchange_ms_per_step": round(t_ex, 4), "predicted_ms_per_step": round(pred, 4),
            "exchange_equals_compute_at_link_GBps": (round(t_ex / slice_ms * (n_pad // world) * samples * 4 / (compute_ms * 1e-3) / 1e9, 1)
                                                     if t_ex > 0 and compute_ms > 0 else None),
            "status": "model, unmeasured (no N > 1 hardware run exists)"}
