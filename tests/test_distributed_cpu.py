"""
The N>1 path on CPU: two processes (gloo), rays sharded by stream id, one all-reduce of the tallies.
The oracle stands in for the device engine (no GPU here); what is under test is the product's sharding and
reduction logic (distributed.shard, reduce_scene_tallies' host branch on the packed tally buffer) and the stream-id design:
the summed tallies of 2 ranks equal the single-process result.  tests/test_gpu_stream.py runs bench.py itself as two ranks
on one GPU.
"""
import os
import sys

import numpy as N
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tracer_amd import scenes, distributed
    from tracer_amd.scene import compile_scene
    from oracle import engine
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    plant, field, rec, src = scenes.nsttf_field(n_heliostats=12)
    cs = compile_scene(plant)
    lo, hi = distributed.shard(n_total, rank, world)
    b = scenes.nsttf_source(hi - lo, src, seed=77, ray_offset=lo)
    # energy per ray is flux*area/n of THIS call; rescale to the global ray count
    with N.errstate(all='ignore'):
        res = engine.trace_from_compiled(cs, b.source_args(), reps=20, min_energy=1e-10)
    scale = float(hi - lo) / n_total

    class Tallies(object):
        """what reduce_scene_tallies needs of a DeviceScene: the packed tally buffer [absorbed S | received S | count S | segments,
        hits | ...] exported to and imported from the host (the gloo branch; the nccl branch hands device pointers over)"""
        def __init__(self):
            self.buf = N.concatenate((res['absorbed'] * scale, res['received'] * scale, res['hits'].astype(float),
                                      [res['segments'], res['hits'].sum()]))

        def tally_size(self):
            return len(self.buf)

        def export_tallies(self, out=None):
            assert out is None          # host branch
            return self.buf.copy()

        def import_tallies(self, src):
            assert isinstance(src, N.ndarray) and src.shape == self.buf.shape
            self.buf = src.copy()

    dev = Tallies()
    distributed.reduce_scene_tallies(dev)
    total = dev.buf
    if rank == 0:
        q.put(total)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_two_rank_shards_sum_to_single_process(world):
    import torch.multiprocessing as mp
    from tracer_amd import scenes
    from tracer_amd.scene import compile_scene
    from oracle import engine
    n_total = 6001
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    total = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    plant, field, rec, src = scenes.nsttf_field(n_heliostats=12)
    cs = compile_scene(plant)
    b = scenes.nsttf_source(n_total, src, seed=77, ray_offset=0)
    with N.errstate(all='ignore'):
        res = engine.trace_from_compiled(cs, b.source_args(), reps=20, min_energy=1e-10)
    S = cs.n_surf
    assert N.array_equal(total[2 * S:3 * S], res['hits'])
    assert total[3 * S] == res['segments'] and total[3 * S + 1] == res['hits'].sum()
    assert N.allclose(total[:S], res['absorbed'], rtol=1e-12, atol=1e-9)
    assert N.allclose(total[S:2 * S], res['received'], rtol=1e-12, atol=1e-9)
