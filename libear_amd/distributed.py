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
