"""
TracerEngineMP: the reference's multi-process driver (tracer/tracer_engine_mp.py:9-130) -- `procs` source bundles traced by
`procs` worker processes on copies of the scene, their ray trees concatenated level by level and the hits stored in the
optics managers collected afterwards.

Here the bundles are traced one after another by the one process that owns the GPU (a bundle of 1e5 rays is a few hundred
microseconds of device time; across GPUs the rays are sharded by rank and the tallies reduced once, tracer_amd/distributed.py),
on the scene itself: the accountants of its surfaces accumulate over the bundles, so nothing has to be collected, and the
trees are merged as the reference merges them (parents of a level shifted by the size the merged level before it had).
"""
import numpy as N

from .tracer_engine import TracerEngine
from .ray_bundle import concatenate_rays
from .trace_tree import RayTree


class TracerEngineMP(TracerEngine):
    def multi_ray_sim(self, sources, procs=1, minener=1e-10, reps=1000, tree=True, **kwargs):
        """sources: list of `procs` bundles.  Further keywords go to TracerEngine.ray_tracer (seed, accel, engine ...)."""
        self.minener = minener
        self.reps = reps
        self.tree_switch = tree
        if len(sources) != procs:
            raise Exception('Number of sources and processors do not agree')
        kwargs.setdefault('accel', False)
        seed = kwargs.pop('seed', None)
        merged = None
        for k, source in enumerate(sources):
            kw = dict(kwargs)
            if seed is not None:
                kw['seed'] = seed + k
            self.ray_tracer(source, self.reps, self.minener, self.tree_switch, **kw)
            if not tree:
                continue
            if merged is None:
                merged = list(self.tree._bunds)
                continue
            sizes_before = [b.get_num_rays() for b in merged]
            for level, bundle in enumerate(self.tree._bunds):
                if level > 0 and bundle.get_num_rays():
                    bundle.set_parents(N.asarray(bundle.get_parents()) + sizes_before[level - 1])
                if level == len(merged):
                    merged.append(bundle)
                else:
                    merged[level] = concatenate_rays([merged[level], bundle])
        if tree:
            self.tree = RayTree()
            for bundle in (merged or []):
                self.tree.append(bundle)
        return self
