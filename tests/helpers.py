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


def plates_scene():
    """the scene of the reference's examples/accel_tree_example.py:20-53 on tracer_amd's classes: two slabs (absorptivity 0.6) and ten
    layers of 10 x 10 Lambertian plates (0.9), every object with its BoundaryBox; returns (assembly, layers, side of the slabs)"""
    from tracer_amd.assembly import Assembly
    from tracer_amd.object import AssembledObject
    from tracer_amd.surface import Surface
    from tracer_amd.flat_surface import RectPlateGM
    from tracer_amd.boundary_shape import BoundaryBox
    from tracer_amd.optics_callables import LambertianReceiver
    n = 10
    side = n + 1.
    objects = []
    for z in (-1., None):
        slab = AssembledObject(Surface(geometry=RectPlateGM(side, side), optics=LambertianReceiver(0.6)),
                               bounds=BoundaryBox([[-side / 2., -side / 2., 0.], [side / 2., side / 2., 0.]]))
        if z is not None:
            slab.set_location(N.array([0., 0., z]))
        objects.append(slab)
    for k in range(n):
        for i in range(n):
            for j in range(n):
                plate = AssembledObject(Surface(geometry=RectPlateGM(.8, .8), optics=LambertianReceiver(0.9)),
                                        bounds=BoundaryBox([[-.4, -.4, 0.], [.4, .4, 0.]]))
                plate.set_location(N.array([i + 0.5 - n / 2., j + 0.5 - n / 2., k + 1.]))
                objects.append(plate)
    return Assembly(objects=objects), n, side


def plates_source(rays, n, side, seed):
    from tracer_amd.sources import oblique_solar_rect_bundle
    return oblique_solar_rect_bundle(num_rays=rays, center=N.vstack([0, 0, n + 1]), source_direction=N.hstack([0, 0, -1]),
                                     rays_direction=N.hstack([0, 0, -1]), x=side, y=side, ang_range=4.65e-3, flux=1000., seed=seed)

