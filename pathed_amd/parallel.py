"""Multi-GPU sharding of the radiance loop: one process per GPU, samples-per-pixel split.

Every (pixel, sample) is independent given the counter-based random stream, and waves are
additive (reference src/integrator.cpp:42-51 sums them into radianceLookup), so rank r renders
a contiguous range of SAMPLE INDICES for all pixels into its own fp32 sum buffer and one
reduce(SUM) to rank 0 merges them (RCCL over xGMI on GPUs, gloo in the CPU tests).  No
collective is needed on the data path itself.
"""
import torch.distributed as dist


def strong_range(rank, world_size, first, count):
    """Split samples [first, first+count) across ranks (total work fixed): returns (begin, n)."""
    base, extra = divmod(count, world_size)
    begin = first + rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def weak_range(rank, samples_per_rank, first=0):
    """Every rank renders `samples_per_rank` samples (per-GPU work fixed): returns (begin, n)."""
    return first + rank * samples_per_rank, samples_per_rank


def reduce_to_root(tensor, root=0):
    """Sum the per-rank radiance sums into `root` (no-op without a process group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(tensor, dst=root, op=dist.ReduceOp.SUM)
    return tensor
