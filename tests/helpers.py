"""Shared helpers of the test-suite: fixture loading and scene tables."""
import os

import numpy as N

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return N.load(os.path.join(GOLDEN, name), allow_pickle=False)


def case_names(npz):
    return [str(x) for x in npz['names']]


def oracle_scene(npz, pre):
    """list-of-dicts scene for oracle.engine from the 'scene_*' arrays of an engine fixture case"""
    kinds = npz[pre + 'scene_gm_kind']
    extra = npz[pre + 'scene_extra']
    out = []
    for i in range(len(kinds)):
        off, ln = int(npz[pre + 'scene_extra_off'][i]), int(npz[pre + 'scene_extra_len'][i])
        out.append(dict(kind=int(kinds[i]), opt_kind=int(npz[pre + 'scene_optics_kind'][i]), frame=npz[pre + 'scene_frames'][i],
                        gm=list(npz[pre + 'scene_gm'][i]), opt=list(npz[pre + 'scene_opt'][i]),
                        extra=extra[off:off + ln] if off >= 0 else None))
    return out


def table_scene(npz, pre):
    """tracer_amd TableScene (ctypes table for the C-ABI) from the same arrays"""
    from tracer_amd.scene import TableScene
    return TableScene(npz[pre + 'scene_gm_kind'], npz[pre + 'scene_optics_kind'], npz[pre + 'scene_frames'], npz[pre + 'scene_gm'],
                      npz[pre + 'scene_opt'], npz[pre + 'scene_extra'], npz[pre + 'scene_extra_off'], npz[pre + 'scene_extra_len'])


def same_misses(t_a, t_b):
    return N.array_equal(N.isfinite(t_a), N.isfinite(t_b))


def source_dict(npz, pre):
    """oracle.sources source dict from the 'desc_*' arrays of a sources fixture case"""
    return dict(kind=int(npz[pre + 'desc_kind']), center=npz[pre + 'desc_center'], rot_pos=npz[pre + 'desc_rot_pos'],
                rot_dir=npz[pre + 'desc_rot_dir'], p=list(npz[pre + 'desc_p']), energy=float(npz[pre + 'desc_energy']),
                buie=npz[pre + 'desc_buie'])
