"""Samples-per-pixel sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

Every pixel sample is an independent unit of work and the keyed random numbers depend only on the GLOBAL sample
index, so rank r renders a contiguous block of sample indices into its own float SUM buffer; the buffers are
summed onto rank 0 with one reduce and divided by the total sample count.  The set of samples, and therefore the
image, does not depend on the number of ranks (only the order of float additions does).
The reference has no counterpart: it is a single process with four threads (src/unidirectionalPathTracer.cpp:59).
"""


def sample_range(psamples, rank, world):
    """Contiguous block [begin, begin+count) of global sample indices of `rank`; blocks differ by at most one."""
    if psamples < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad sharding arguments")
    base, extra = divmod(psamples, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def render_sharded(render_accumulate, acc, psamples, rank=0, world=1, group=None, dst=0):
    """One frame of `psamples` samples per pixel over `world` ranks.

    render_accumulate(acc, begin, count) must ADD the radiance sum of samples [begin, begin+count) into the float32
    tensor `acc` (W*H*3) — for the GPU path that is udp_renderer.render_accumulate on acc.data_ptr().
    Returns the mean-radiance tensor on rank `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    acc.zero_()
    if acc.is_cuda:
        torch.cuda.synchronize(acc.device)
    begin, count = sample_range(psamples, rank, world)
    if count:
        render_accumulate(acc, begin, count)
    if world > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if rank == dst:
        return acc / float(psamples)
    return None
