"""
Triangle meshes from STL files as Tracer objects (reference: ray_trace_utils/stl_utils.py:156-235).
The reference reads and writes STL through the numpy-stl package; here the two formats are parsed with NumPy alone.
Every triangle becomes a native flat surface -- a `FlatSimplePolygonGM` or a `TriangularFace`, in the triangle's own frame
(origin at its first vertex, z along its normal) -- with a BoundaryBox for the Kd-tree, as the reference builds them.
"""
import functools
import struct

import numpy as N

from .polygon import FlatSimplePolygonGM
from .triangular_face import TriangularFace
from .surface import Surface
from .object import AssembledObject
from .boundary_shape import BoundaryBox
from .spatial_geometry import roty, rotz
from .vector_manipulations import AABB
from .face_set import FaceSet, LazyBounds


def load_stl(stl_file):
    """(n, 3, 3) array: the three vertices of every triangle of a binary or ASCII STL file."""
    with open(stl_file, 'rb') as f:
        raw = f.read()
    if len(raw) >= 84:
        count = struct.unpack('<I', raw[80:84])[0]
        if len(raw) == 84 + 50 * count:                # binary: 80-byte header, count, 50 bytes per facet
            rec = N.dtype([('normal', '<f4', 3), ('vertices', '<f4', (3, 3)), ('attr', '<u2')])
            return N.frombuffer(raw, dtype=rec, count=count, offset=84)['vertices'].astype(float)
    vertices = [line.split()[1:4] for line in raw.decode('ascii', errors='replace').splitlines() if line.strip().startswith('vertex')]
    if len(vertices) == 0 or len(vertices) % 3:
        raise ValueError('%s is neither a binary nor an ASCII STL file' % stl_file)
    return N.array(vertices, dtype=float).reshape(-1, 3, 3)


def make_stl(verts, faces, filename):
    """Binary STL file of the indexed triangles: verts (n, 3), faces (m, 3) integer indices."""
    verts, faces = N.asarray(verts, dtype=float), N.asarray(faces, dtype=int)
    tri = verts[faces]                                  # (m, 3, 3)
    normals = N.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 1])
    with N.errstate(invalid='ignore', divide='ignore'):
        normals = N.nan_to_num(normals / N.sqrt(N.sum(normals ** 2, axis=1))[:, None])
    rec = N.zeros(len(faces), dtype=N.dtype([('normal', '<f4', 3), ('vertices', '<f4', (3, 3)), ('attr', '<u2')]))
    rec['normal'], rec['vertices'] = normals, tri
    with open(filename, 'wb') as f:
        f.write(b'tracer_amd'.ljust(80, b' '))
        f.write(struct.pack('<I', len(faces)))
        f.write(rec.tobytes())


def stl_to_tracer_geom(triangles, option='polygon'):
    """
    triangles: (n, 3, 3) vertices A, B, C per triangle.  option: 'polygon' or 'triangle' (the geometry manager used).
    Returns (geoms, locs, rots): the frame of a triangle has its origin at A and z along (B - A) x (C - B), reached by a
    rotation about y then z, as in the reference (:193-210).
    """
    if option not in ('polygon', 'triangle'):
        raise ValueError("option is 'polygon' or 'triangle'")
    geoms, locs, rots = [], [], []
    for A, B, C in N.asarray(triangles, dtype=float):
        normal = N.cross(B - A, C - B)
        length = N.sqrt(N.sum(normal ** 2))
        normal = normal / length if length > 0. else N.array([1., 0., 0.])
        azimuth, polar = N.arctan2(normal[1], normal[0]), N.arccos(N.clip(normal[2], -1., 1.))
        to_local = N.dot(roty(-polar), rotz(-azimuth))[:3, :3]
        flat = N.dot(to_local, N.array([A - A, B - A, C - A]).T)          # columns: the vertices in the triangle's plane
        if option == 'polygon':
            geoms.append(FlatSimplePolygonGM(flat[:2]))
        else:
            geoms.append(TriangularFace(N.vstack((flat[:2, 1:], N.zeros((1, 2))))))
        locs.append(A)
        rots.append(N.dot(rotz(azimuth), roty(polar))[:3, :3])
    return geoms, locs, rots


def stl_triangle_frames(triangles):
    """
    The frames stl_to_tracer_geom(option='triangle') gives the faces, for all of them at once: (origins (n, 3), rotations (n, 3, 3),
    local edges (n, 2, 3)).  A face's rotation is rotz(azimuth) roty(polar) of its normal (:193-210); its edges B - A and C - A taken
    into that frame have no z component.
    """
    triangles = N.asarray(triangles, dtype=float)
    A, B, C = triangles[:, 0], triangles[:, 1], triangles[:, 2]
    normal = N.cross(B - A, C - B)
    length = N.sqrt(N.sum(normal ** 2, axis=1))
    ok = length > 0.
    normal = N.where(ok[:, None], normal / N.where(ok, length, 1.)[:, None], N.array([1., 0., 0.]))
    azimuth, polar = N.arctan2(normal[:, 1], normal[:, 0]), N.arccos(N.clip(normal[:, 2], -1., 1.))
    ca, sa, cp, sp = N.cos(azimuth), N.sin(azimuth), N.cos(polar), N.sin(polar)
    z, o = N.zeros_like(ca), N.ones_like(ca)
    Rz = N.array([[ca, -sa, z], [sa, ca, z], [z, z, o]]).transpose(2, 0, 1)
    Ry = N.array([[cp, z, sp], [z, o, z], [-sp, z, cp]]).transpose(2, 0, 1)
    rots = N.einsum('fij,fjk->fik', Rz, Ry)
    # to_local = roty(-polar) rotz(-azimuth), as the reference multiplies it (not the transpose of the product above to the last bit)
    Rzm = N.array([[ca, sa, z], [-sa, ca, z], [z, z, o]]).transpose(2, 0, 1)
    Rym = N.array([[cp, z, -sp], [z, o, z], [sp, z, cp]]).transpose(2, 0, 1)
    to_local = N.einsum('fij,fjk->fik', Rym, Rzm)
    edges = N.stack((N.einsum('fij,fj->fi', to_local, B - A), N.einsum('fij,fj->fi', to_local, C - A)), axis=1)
    edges[:, :, 2] = 0.
    return A.copy(), rots, edges


def make_stl_tracer_object(triangles, optics, optics_args, option='polygon'):
    """AssembledObject of one Surface per triangle, each with optics(**optics_args) and the triangle's bounding box.
    option='triangle': the faces are kept as arrays (face_set.FaceSet), frames, optics instances and boxes made per face on demand."""
    triangles = N.asarray(triangles, dtype=float)
    if option == 'triangle':
        origins, rots, edges = stl_triangle_frames(triangles)
        faces = FaceSet(origins, rots, edges, optics_factory=functools.partial(optics, **optics_args))      # (picklable: TracerEngineMP)
        return AssembledObject(surfs=faces, bounds=LazyBounds(triangles.min(axis=1), triangles.max(axis=1)))
    geoms, locs, rots = stl_to_tracer_geom(triangles, option=option)
    surfs = [Surface(geometry=g, optics=optics(**optics_args), location=l, rotation=r) for g, l, r in zip(geoms, locs, rots)]
    bounds = [BoundaryBox(AABB(t.T)) for t in triangles]
    return AssembledObject(surfs=surfs, bounds=bounds)


def load_stl_into_tracer(stl_file, optics, optics_args, option='polygon'):
    return make_stl_tracer_object(load_stl(stl_file), optics, optics_args, option)
