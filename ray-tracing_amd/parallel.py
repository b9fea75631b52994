"""Multi-GPU plan: samples-per-pixel sharding (one process per GPU) + one reduction of the radiance buffers.

Every (pixel, sample) is independent (the reference parallelises rows with OpenMP, raytracer.cpp:1456);
here rank r of `world` renders its own global sample indices for all pixels into its own FP64
accumulator, and one sum-reduction to rank 0 (RCCL over xGMI with the "nccl" backend, gloo on CPU)
finishes the frame. The random stream is keyed by (seed, pixel, global sample index), so the set of
samples - and the image up to the order of the final additions - does not depend on `world`.
"""


def shard_offset(step, world, rank, spp_step):
    """first global sample index rank `rank` renders in step `step` (steps and ranks never overlap)"""
    return (step * world + rank) * spp_step


def split_samples(total, rank, world):
    """strong scaling: the half-open range of global sample indices [lo, hi) that rank `rank` of `world` renders of an
    image of `total` samples per pixel; the ranges of all ranks are disjoint, cover 0 .. total-1 and differ in length by
    at most one (ranks beyond `total` get an empty range)"""
    return total * rank // world, total * (rank + 1) // world


def total_spp(steps, world, spp_step):
    return steps * world * spp_step


def reduce_to_root(accum, world):
    """sum the per-rank accumulators (torch tensors) onto rank 0; a no-op for one rank. The caller's renders into
    `accum` must have finished (Renderer.sync()); on return the sum is complete, so it may be resolved on any stream."""
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        if accum.is_cuda:
            torch.cuda.synchronize(accum.device)  # the collective is only ordered against torch's current stream
    return accum
