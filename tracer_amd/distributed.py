"""
Multi-GPU driver logic: one process per GPU, scene replicated, rays sharded by stream id, ONE sum-reduction of
the packed tally buffer at the end (the role of the reference's pathos pool + merge,
tracer/tracer_engine_mp.py:19-35, :44-119).  torch.distributed is the transport: backend "nccl" is RCCL over
xGMI on the GPU box, "gloo" on CPU in the tests.  The payload is tiny (NSTTF: 3*219+2 scalars + 2500 flux bins
= 25 KB of float64), i.e. latency-bound: one all-reduce, no bucketing.
"""
import numpy as N


def shard(n_total, rank, world):
    """[begin, end) of the global ray ids traced by `rank`: contiguous, sizes differ by at most one"""
    base, rem = divmod(int(n_total), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def batch_offset(step, rank, world, rays_per_step):
    """first stream id of the batch traced by `rank` at `step` when every rank traces rays_per_step rays per step"""
    return (int(step) * int(world) + int(rank)) * int(rays_per_step)


def all_reduce_sum(array):
    """Sum a float64 numpy array over all ranks (in place semantics: returns the reduced array)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return array
    t = torch.from_numpy(N.ascontiguousarray(array, dtype=N.float64))
    if dist.get_backend() == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def reduce_scene_tallies(dev):
    """
    All-reduce the device tally buffer of a DeviceScene across ranks without leaving the GPU (nccl), or through
    the host (gloo / single process).  After the call every rank holds the job totals.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    if dist.get_backend() == 'nccl':
        # (torch's HIP runtime must have been initialised before the library's context was created: _cabi.get_context sees to
        # it when torch is imported first, and this says so loudly when it was not)
        from . import _cabi
        _cabi.check_torch_order()
        t = torch.empty(dev.tally_size(), dtype=torch.float64, device='cuda')
        dev.export_tallies(out=t.data_ptr())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        dev.import_tallies(t.data_ptr())
    else:
        dev.import_tallies(all_reduce_sum(dev.export_tallies()))
