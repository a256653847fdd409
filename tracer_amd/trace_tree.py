"""
RayTree: the list of bundles recorded by a trace, each pointing at its parents in the previous
one.  Interface of the reference's tracer/trace_tree.py:6-55 (append, indexing, num_bunds,
ordered_parents, ray_history) -- Renderer-style consumers read tree[level].get_vertices() /
get_energy() / get_parents().
"""
import numpy as N


class RayTree(object):
    def __init__(self):
        self._bunds = []

    def __getitem__(self, level):
        return self._bunds[level]

    def __len__(self):
        return len(self._bunds)

    def num_bunds(self):
        return len(self._bunds)

    def append(self, bund):
        self._bunds.append(bund)

    def ordered_parents(self):
        return [b.get_parents() for b in self._bunds[1:]]

    def ray_history(self, ray_index, level=None):
        """Indices of a ray's ancestors from `level` back to the source bundle, newest first."""
        if level is None:
            level = self.num_bunds()
        hist = N.empty(level, dtype=int)
        hist[0] = ray_index
        for k in range(1, level):
            hist[k] = self._bunds[level - k].get_parents()[hist[k - 1]]
        return hist
