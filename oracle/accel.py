"""
TEST INFRASTRUCTURE (CPU oracle, never shipped or measured): KdTree.traversal of the reference restated for small cases.

tracer/accel_tree.py:213-312 (traversal) on :314-330 (intersect_bounds), the non-lightweight form: a per-ray walk with an explicit
stack that marks every leaf the ray crosses inside the root box; surfaces without bounds are always relevant.  The tree comes as
the flat arrays of trc_kdtree_desc (tracer_amd.accel_tree.KdTree.flat(), or the fixture tests/golden/kdtree_nsttf.npz).
Pinned by the relevancy matrix the reference itself returns for 700 rays on the NSTTF tree (tests/test_oracle_golden.py).
"""
import numpy as N


def intersect_bounds(poss, dirs, inv_dirs, bounds):
    """accel_tree.py:314-330"""
    neg = N.array(dirs < 0, dtype=int)
    t_mins = N.zeros(poss.shape[1])
    t_maxs = N.ones(poss.shape[1]) * N.inf
    for i in range(3):
        a = (bounds[neg[i], i] - poss[i]) * inv_dirs[i]
        b = (bounds[1 - neg[i], i] - poss[i]) * inv_dirs[i]
        swap = a > b
        a[swap], b[swap] = b[swap], a[swap]
        t_mins = N.maximum(t_mins, a)
        t_maxs = N.minimum(t_maxs, b)
    inters = t_maxs > 0
    inters[t_mins > t_maxs] = False
    return inters, t_mins, t_maxs


def traversal(tree, n_surf, poss, dirs):
    """(any_inter, relevancy (n_surf, n) bool) -- accel_tree.py:213-312 with lightweight=False"""
    flag, split, child = tree['flag'], tree['split'], tree['child']
    leaf_off, leaf_cnt, leaf_surfs = tree['leaf_off'], tree['leaf_cnt'], tree['leaf_surfs']
    always = N.asarray(tree['always_relevant'], dtype=int)
    n = poss.shape[1]
    with N.errstate(all='ignore'):
        inv = 1. / dirs                                                     # :224
        bounds = N.array([tree['bounds'][:3], tree['bounds'][3:]])
        inters, t_mins, t_maxs = intersect_bounds(poss, dirs, inv, bounds)
    rel = N.zeros((n_surf, n), dtype=bool)
    rel[always] = True                                                      # :236
    if not (inters.any() or always.any()):                                  # :238 (any() of the index array, as written)
        return False, rel
    for r in range(n):
        if not inters[r]:
            continue
        t_min, t_max = t_mins[r], t_maxs[r]
        todo = []
        node = 0
        while True:
            if t_maxs[r] < t_min:                                           # :243
                break
            f = flag[node]
            if f != 3:
                with N.errstate(all='ignore'):
                    t_plane = (split[node] - poss[f, r]) * inv[f, r]        # :249
                c1, c2 = child[node], child[node] + 1
                below = (poss[f, r] < split[node]) or (poss[f, r] == split[node] and dirs[f, r] <= 0.)
                if not below:
                    c1, c2 = c2, c1
                if t_plane > t_max or t_plane <= 0.:                        # :258-259
                    node = c1
                elif t_plane < t_min:
                    node = c2
                else:
                    todo.append((c2, t_plane, t_max))
                    node = c1
                    t_max = t_plane
            else:
                rel[leaf_surfs[leaf_off[node]:leaf_off[node] + leaf_cnt[node]], r] = True      # :288
                if todo:
                    node, t_min, t_max = todo.pop()
                else:
                    break
    return True, rel
