"""
Kd-tree over the global AABBs of the objects' BoundaryBoxes.

build: host side, once per scene.  It follows the reference's builder rule for rule
(tracer/accel_tree.py:42-204) so that the node array is identical to the reference's for the same
scene: breadth-first numbering, leaf when count <= min_leaf or level >= max_depth, split on the
longest axis that has any candidate plane, surface-area cost with t_trav/t_isec/empty_bonus, `fast`
limiting the candidates to 12 planes, and the candidate planes taken from
`bounds[:, tile(in_node, 2)]` exactly as written there (:109).

traversal: on the device, inside the trace kernels (csrc/trc_core.h, trc_nearest_kd).  `flat()`
produces the arrays for trc_scene_set_kdtree.
"""
import logging
import time

import numpy as N

from .object import AssembledObject
from .vector_manipulations import AABB


class Node(object):
    """flag 0/1/2: interior split on that axis (split, child); flag 3: leaf (surfaces_idxs)."""
    pass


class KdTree(object):
    def __init__(self, assembly, max_depth=N.inf, min_leaf=1, loglevel=logging.DEBUG, debug=False, fast=False,
                 t_trav=1., t_isec=1000., empty_bonus=0.2, split_threshold=None):
        self.loglevel = loglevel
        self.nodes = []
        self.t_trav = t_trav
        self.t_isec = t_isec
        self.empty_bonus = empty_bonus
        self.split_threshold = split_threshold
        self.fast = fast
        self.min_leaf = min_leaf
        self.max_depth = max_depth
        self.objects = [assembly] if isinstance(assembly, AssembledObject) else assembly.get_objects()
        self.n_surfs = len(assembly.get_surfaces())
        self.debug = False
        self.build_tree()

    # ------------------------------------------------------------------------------------------
    def build_tree(self):
        logging.log(self.loglevel, 'Building tree')
        t0 = time.time()
        per_obj_bounds = [o.get_boundaries() for o in self.objects]
        n_bounds_obj = N.array([len(b) for b in per_obj_bounds])
        if (n_bounds_obj == 0).all():
            raise Exception('No boundary defined in the assembly, please revert to non-accelerated ray-tracing')
        total = int(N.sum(n_bounds_obj))

        # surface indices of each object, in assembly order
        n_surf_obj = [len(o.get_surfaces()) for o in self.objects]
        first = N.concatenate(([0], N.cumsum(n_surf_obj)))
        obj_surfs = [N.arange(first[i], first[i + 1]) for i in range(len(self.objects))]

        minpoints = N.empty((3, total))
        maxpoints = N.empty((3, total))
        bounds = N.empty((3, 2 * total))
        surfs_of_bound = []
        always = []
        k = 0
        for oi, bnds in enumerate(per_obj_bounds):
            if len(bnds) == 0:
                always.append(obj_surfs[oi])
                continue
            for b in bnds:
                minpoints[:, k] = b._minpoint
                maxpoints[:, k] = b._maxpoint
                bounds[:, 2 * k] = b._minpoint
                bounds[:, 2 * k + 1] = b._maxpoint
                surfs_of_bound.append(obj_surfs[oi])
                k += 1
        self.always_relevant = N.hstack(always).astype(int) if always else N.array([], dtype=int)
        self.minpoint, self.maxpoint = AABB(bounds)

        n_cand = 12 if self.fast == True else None
        # breadth-first construction; boxes[i], levels[i] describe node i
        boxes = [(self.minpoint[:, None], self.maxpoint[:, None])]
        levels = [0]
        self.nodes = [Node()]
        idx = 0
        level = 0
        while idx < len(self.nodes):
            lo, hi = boxes[idx]
            level = levels[idx]
            inside = N.logical_and((maxpoints >= lo).all(axis=0), (minpoints <= hi).all(axis=0))
            count = N.count_nonzero(inside)
            node = self.nodes[idx]
            split = None
            if not (count <= self.min_leaf or level >= self.max_depth):
                split = self.determine_split(lo, hi, minpoints[:, inside], maxpoints[:, inside],
                                             bounds[:, N.tile(inside, 2)], n_bounds=n_cand, t_trav=self.t_trav,
                                             t_isec=self.t_isec, empty_bonus=self.empty_bonus)
                if split[0] == 3:
                    split = None
            if split is None:
                node.flag = 3
                node.surfaces_idxs = [surfs_of_bound[i] for i in N.nonzero(inside)[0]]
            else:
                axis, pos = split
                node.flag = int(axis)
                node.split = pos
                node.child = len(self.nodes)
                hi_below = N.copy(hi)
                hi_below[axis] = pos
                lo_above = N.copy(lo)
                lo_above[axis] = pos
                boxes += [(lo, hi_below), (lo_above, hi)]
                levels += [level + 1, level + 1]
                self.nodes += [Node(), Node()]
            idx += 1
        self.build_time = time.time() - t0
        logging.log(self.loglevel, 'build_time: %ss' % self.build_time)
        logging.log(self.loglevel, 'maximum level%s' % level)
        logging.log(self.loglevel, 'Kd-Tree built')

    def determine_split(self, minpoint_parent, maxpoint_parent, minpoints, maxpoints, bounds, n_bounds=None,
                        t_trav=1., t_isec=1000., empty_bonus=0.2):
        """(axis, position) of the best plane on the longest axis that has candidates, or (3, Ns)."""
        Ns = minpoints.shape[1]
        diag = (maxpoint_parent - minpoint_parent).reshape(3)
        lo = N.reshape(minpoint_parent, 3)
        hi = N.reshape(maxpoint_parent, 3)
        S_inv = 1. / (diag[0] * diag[1] + diag[1] * diag[2] + diag[2] * diag[0])
        basecost = t_trav + Ns * t_isec
        for a in N.argsort(diag)[::-1]:
            cand = bounds[a]
            cand = cand[N.logical_and(cand > lo[a], cand < hi[a])]
            if len(cand) == 0:
                continue
            cand = N.unique(cand)
            d0, d1 = diag[(a + 1) % 3], diag[(a + 2) % 3]
            d0td1 = d0 * d1
            d0pd1 = d0 + d1
            if n_bounds is not None and n_bounds < len(cand):
                cand = cand[N.round(N.linspace(0, len(cand) - 1, n_bounds)).astype(int)]
            N_A = N.count_nonzero(maxpoints[a][None, :] >= cand[:, None], axis=1)
            N_B = N.count_nonzero(minpoints[a][None, :] <= cand[:, None], axis=1)
            p_A = S_inv * (d0td1 + (hi[a] - cand) * d0pd1)
            p_B = S_inv * (d0td1 + (cand - lo[a]) * d0pd1)
            b_e = N.logical_or(N_A == 0, N_B == 0) * empty_bonus
            cost = basecost + t_isec * (1. - b_e) * (p_A * N_A + p_B * N_B)
            return int(a), cand[int(N.argmin(cost))]
        return 3, Ns

    # ------------------------------------------------------------------------------------------
    def traversal(self, bundle, lightweight=False):
        """
        Which surfaces each ray of `bundle` has to be tested against (accel_tree.py:213-312): (any_inter, relevancy), relevancy a
        (n_surfs, n_rays) boolean array, True where the ray crosses a leaf holding the surface -- every leaf on its way through the
        root box -- or the surface has no bounds.  Walked on the device, one ray per lane (trc_kdtree_traversal).
        lightweight=True: the reference's list-of-lists scheduling of the same information (:227-233, :281-286, :301-305), see
        _traversal_lightweight.
        """
        if lightweight:
            return self._traversal_lightweight(bundle)
        import ctypes as C
        from . import _cabi
        ctx = _cabi.get_context()
        f = self.flat()
        d = _cabi.KdTreeDesc()
        d.n_nodes, d.n_leaf_surfs, d.n_always = len(f['flag']), len(f['leaf_surfs']), len(f['always_relevant'])
        i32 = C.POINTER(C.c_int32)
        for name in ('flag', 'child', 'leaf_off', 'leaf_cnt', 'leaf_surfs', 'always_relevant'):
            setattr(d, name, f[name].ctypes.data_as(i32))
        d.split = _cabi.ptr(f['split'])
        for i in range(6):
            d.bounds[i] = f['bounds'][i]
        v, dr = _cabi.f64(bundle.get_vertices()), _cabi.f64(bundle.get_directions())
        n = v.shape[1]
        rays = _cabi.make_rays(n, v[0], v[1], v[2], dr[0], dr[1], dr[2])
        rel = N.zeros((self.n_surfs, n), dtype=N.uint8)
        any_inter = C.c_int32(0)
        _cabi.check(ctx.lib.trc_kdtree_traversal(ctx.handle, C.byref(d), self.n_surfs, C.byref(rays), n,
                                                 rel.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(any_inter)))
        return bool(any_inter.value), rel.view(bool)

    # ------------------------------------------------------------------------------------------
    def _traversal_lightweight(self, bundle):
        """
        traversal(bundle, lightweight=True) (accel_tree.py:213-312): (any_inter, surfaces_relevancy) with
        surfaces_relevancy[s] a list of batches of ray numbers -- batch k holds the rays whose k-th leaf on their way through
        the tree lists surface s, a ray only in the first batch of a surface it appears in (:301-305) -- so that
        intersect_ray_accel_seq (tracer_engine.py:66-122) can test a surface's near rays before its far ones.  The engines here
        never consume it (the device walks front to back by itself); it is offered for scripts that read it.  The walk is the
        reference's, ray by ray on the host (a scheduling list is host data by nature); which (surface, ray) pairs appear at
        all is what the device traversal marks, and the two are compared in the tests.  Surfaces without bounds take every
        ray in their first batch: the reference's own line for them (:229-231) appends bare ray numbers next to the batches
        and its clean-up (:301-305) then fails on them, so there is no other behaviour to follow.
        """
        f = self.flat()
        flag, split, child = f['flag'], f['split'], f['child']
        leaf_off, leaf_cnt, leaf_surfs = f['leaf_off'], f['leaf_cnt'], f['leaf_surfs']
        poss = N.asarray(bundle.get_vertices(), dtype=float)
        dirs = N.asarray(bundle.get_directions(), dtype=float)
        nrays = poss.shape[1]
        with N.errstate(all='ignore'):
            inv = 1. / dirs
            bounds = N.array([N.ravel(self.minpoint), N.ravel(self.maxpoint)], dtype=float)
            neg = N.array(dirs < 0, dtype=int)
            t_mins, t_maxs = N.zeros(nrays), N.full(nrays, N.inf)
            for i in range(3):                                  # intersect_bounds (:314-330)
                lo = (bounds[neg[i], i] - poss[i]) * inv[i]
                hi = (bounds[1 - neg[i], i] - poss[i]) * inv[i]
                swap = lo > hi
                lo[swap], hi[swap] = hi[swap], lo[swap]
                t_mins, t_maxs = N.maximum(t_mins, lo), N.minimum(t_maxs, hi)
            inters = t_maxs > 0
            inters[t_mins > t_maxs] = False
        rel = [[[]] for _ in range(self.n_surfs)]
        for s in N.asarray(self.always_relevant, dtype=int):
            rel[s][0] = list(range(nrays))
        any_inter = False
        order = N.zeros(nrays, dtype=int)
        if inters.any() or len(self.always_relevant):
            for r in N.nonzero(inters)[0]:
                t_min, t_max = t_mins[r], t_maxs[r]
                stack = []
                node = 0
                while True:
                    if t_maxs[r] < t_min:
                        break
                    if flag[node] != 3:
                        ax = flag[node]
                        with N.errstate(all='ignore'):
                            t_plane = (split[node] - poss[ax, r]) * inv[ax, r]
                        c1, c2 = child[node], child[node] + 1
                        below = (poss[ax, r] < split[node]) or (poss[ax, r] == split[node] and dirs[ax, r] <= 0.)
                        if not below:
                            c1, c2 = c2, c1
                        if t_plane > t_max or t_plane <= 0.:
                            node = c1
                        elif t_plane < t_min:
                            node = c2
                        else:
                            stack.append((c2, t_plane, t_max))
                            node = c1
                            t_max = t_plane
                    else:
                        for s in leaf_surfs[leaf_off[node]:leaf_off[node] + leaf_cnt[node]]:
                            while len(rel[s]) <= order[r]:
                                rel[s].append([])
                            if r not in rel[s][order[r]]:
                                rel[s][order[r]].append(int(r))
                        order[r] += 1
                        if not stack:
                            break
                        node, t_min, t_max = stack.pop()
            for s in rel:                                        # a ray is tested once per surface: in its first batch (:301-305)
                seen = set()
                for batch in s:
                    batch[:] = [r for r in batch if r not in seen]
                    seen.update(batch)
            any_inter = True
        return any_inter, rel

    def flat(self):
        """Arrays for trc_kdtree_desc."""
        n = len(self.nodes)
        flag = N.empty(n, dtype=N.int32)
        split = N.zeros(n)
        child = N.zeros(n, dtype=N.int32)
        leaf_off = N.zeros(n, dtype=N.int32)
        leaf_cnt = N.zeros(n, dtype=N.int32)
        leaf_surfs = []
        for i, nd in enumerate(self.nodes):
            flag[i] = nd.flag
            if nd.flag == 3:
                s = N.unique(N.concatenate([N.ravel(x) for x in nd.surfaces_idxs])) if len(nd.surfaces_idxs) else []
                leaf_off[i] = len(leaf_surfs)
                leaf_cnt[i] = len(s)
                leaf_surfs.extend(int(v) for v in s)
            else:
                split[i] = nd.split
                child[i] = nd.child
        return dict(flag=flag, split=split, child=child, leaf_off=leaf_off, leaf_cnt=leaf_cnt,
                    leaf_surfs=N.array(leaf_surfs, dtype=N.int32),
                    always_relevant=N.array(self.always_relevant, dtype=N.int32),
                    bounds=N.concatenate((N.ravel(self.minpoint), N.ravel(self.maxpoint))).astype(float))
