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


def channel_range(n_out, rank, world):
    """loudspeaker channels of the shared bus owned by `rank` after the exchange"""
    assert n_out % world == 0, "n_out must be divisible by the number of ranks"
    per = n_out // world
    return rank * per, (rank + 1) * per


def exchange(partial, owned=None, group=None, async_op=False):
    """Sum the ranks' partial outputs.

    partial: [n_out, samples] float32 (this rank's render of its objects).
    owned:   [n_out / world, samples] output slice of this rank (nccl: reduce-scatter);
             ignored on backends without reduce_scatter_tensor (gloo), where an all-reduce is done
             in place and the owned slice is a view of `partial`.
    Returns (owned_tensor, work_or_None).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return partial, None
    rank = dist.get_rank(group)
    lo, hi = channel_range(partial.shape[0], rank, world)
    backend = dist.get_backend(group)
    if backend == "nccl":
        if owned is None:
            owned = torch.empty((hi - lo, partial.shape[1]), dtype=partial.dtype, device=partial.device)
        work = dist.reduce_scatter_tensor(owned, partial, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return owned, work
    work = dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return partial[lo:hi], work
