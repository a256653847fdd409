"""
The faces of a triangle mesh as ONE object holding arrays.

The reference makes a Surface per face (tracer/models/triangulated_surface.py:12-52, ray_trace_utils/stl_utils.py:178-235): a
hundred thousand Python objects for a mesh of 1e5 triangles, each with its frame, a TriangularFace and -- from an STL file -- its
own optics instance and BoundaryBox.  Scripts see the same thing here -- `obj.get_surfaces()` has one entry per face, `[k]` is a
Surface with a TriangularFace in the face's own frame -- but what is kept are the arrays the faces are made from:

    origins (m, 3), rotations (m, 3, 3), local edges (m, 2, 3)      and one optics manager, or a recipe for one per face

A Surface (and its optics, its BoundaryBox) is made when a script asks for face k, and remembered.  The scene compiler
(scene.CompiledScene) takes the rows of all faces from the arrays at once, the object moves its faces by one matrix product.

FaceSet       the faces of one object;  LazyBounds: their bounding boxes, made on demand alike
SurfaceSeq    Assembly.get_surfaces() of an assembly that holds such objects: the concatenation, without materialising anything
"""
import numpy as N

from . import _cabi
from .surface import Surface
from .triangular_face import TriangularFace


class FaceSet(object):
    def __init__(self, origins, rotations, local_edges, optics=None, optics_factory=None):
        """
        origins (m, 3): first vertex of each face in the object's frame; rotations (m, 3, 3): columns = the face's axes (x along its
        first edge, z its normal); local_edges (m, 2, 3): the two edges from the first vertex in the face's frame.
        optics: one optics manager shared by all faces, or optics_factory(): a new one for each face (made when face k is).
        """
        self.origins = N.ascontiguousarray(origins, dtype=float)
        self.rotations = N.ascontiguousarray(rotations, dtype=float)
        self.local_edges = N.ascontiguousarray(local_edges, dtype=float)
        if (optics is None) == (optics_factory is None):
            raise ValueError("give one optics manager for all faces, or a factory of them")
        self.optics = optics
        self.optics_factory = optics_factory
        self._made = {}                 # face number -> Surface, for the faces a script has asked for
        self._parent = N.eye(4)         # frame of the owning object in global coordinates
        self._global = None             # (m, 3, 4): upper rows of the faces' global frames, made when first needed
        self._template = None           # optics instance the per-face ones are copies of in all but identity (optics_factory)
        self._rows = None               # (key, rows): the device-table rows of the faces as last compiled (scene.CompiledScene)
        self.stamp = 0                  # counts what could change those rows behind the key's back (nothing, so far: the arrays are fixed)

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_rows'] = state['_global'] = None        # (made again when needed: not worth 38 MB per copy of a mesh of 1e5 faces)
        return state

    # -- a sequence of Surfaces -------------------------------------------------------------------------------------
    def __len__(self):
        return len(self.origins)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        k = int(k)
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        s = self._made.get(k)
        if s is None:
            opt = self.optics if self.optics is not None else self.optics_factory()
            s = Surface(TriangularFace(self.local_edges[k].T), opt, location=self.origins[k], rotation=self.rotations[k])
            s.transform_frame(self._parent)
            self._made[k] = s
        return s

    def __iter__(self):
        for k in range(len(self)):
            yield self[k]

    def index(self, surface):
        for k, s in self._made.items():
            if s is surface:
                return k
        raise ValueError("the surface is not one of these faces")

    # -- what the object and the scene compiler ask ------------------------------------------------------------------
    def transform_frames(self, parent):
        """the owning object has moved: `parent` is its frame in global coordinates"""
        self._parent = N.array(parent, dtype=float)
        self._global = None
        for s in self._made.values():
            s.transform_frame(self._parent)

    def global_frames12(self):
        """(m, 12): rows 0..2 of every face's global frame, row-major (what trc_surface_desc.frame holds)"""
        if self._global is None:
            # parent x own frame for every face, with the same product (and so the same rounding) as HasFrame.transform_frame
            local = N.zeros((len(self), 4, 4))
            local[:, :3, :3], local[:, :3, 3], local[:, 3, 3] = self.rotations, self.origins, 1.
            self._global = N.ascontiguousarray(N.matmul(self._parent, local)[:, :3, :]).reshape(len(self), 12)
        return self._global

    def optics_template(self):
        """an optics manager with the parameters of every face's (the shared one, or one made by the factory and kept for this)"""
        if self.optics is not None:
            return self.optics
        if self._template is None:
            self._template = self.optics_factory()
        return self._template

    def optics_of(self, k):
        """face k's optics manager (a face nobody has asked for has no state of its own yet: the template stands for it)"""
        if self.optics is not None:
            return self.optics
        s = self._made.get(int(k))
        return s.get_optics_manager() if s is not None else self[int(k)].get_optics_manager()

    def distinct_optics(self):
        if self.optics is not None:
            return [self.optics]
        return [s.get_optics_manager() for s in self._made.values()]

    def compiled_rows(self, key):
        return self._rows[1] if self._rows is not None and self._rows[0] == key else None

    def keep_compiled_rows(self, key, rows):
        self._rows = (key, rows)

    def gm_rows(self):
        """(kind, (m, 6) parameters) of the faces' geometry managers: TriangularFace._native() of each"""
        return _cabi.GM_TRIANGLE, self.local_edges.reshape(len(self), 6)


class LazyBounds(object):
    """the BoundaryBoxes of a mesh's faces (stl_utils.py:230-232), each made when asked for"""
    def __init__(self, lo, hi):
        self.lo, self.hi = N.asarray(lo, dtype=float), N.asarray(hi, dtype=float)
        self._made = {}
        self._parent = N.eye(4)

    def __len__(self):
        return len(self.lo)

    def __getitem__(self, k):
        from .boundary_shape import BoundaryBox
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        k = int(k)
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        b = self._made.get(k)
        if b is None:
            b = BoundaryBox([self.lo[k], self.hi[k]])
            b.transform_frame(self._parent)
            self._made[k] = b
        return b

    def __iter__(self):
        for k in range(len(self)):
            yield self[k]

    def transform_frames(self, parent):
        self._parent = N.array(parent, dtype=float)
        for b in self._made.values():
            b.transform_frame(self._parent)


class SurfaceSeq(object):
    """Assembly.get_surfaces() when some object keeps its faces as a FaceSet: plain lists and FaceSets end to end"""
    def __init__(self, parts):
        self.parts = [p for p in parts if len(p)]
        self.first = N.concatenate(([0], N.cumsum([len(p) for p in self.parts]))).astype(int)

    def __len__(self):
        return int(self.first[-1])

    def _locate(self, k):
        k = int(k)
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        j = int(N.searchsorted(self.first, k, side='right')) - 1
        return j, k - int(self.first[j])

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        j, i = self._locate(k)
        return self.parts[j][i]

    def __iter__(self):
        for p in self.parts:
            for s in p:
                yield s

    def index(self, surface):
        for j, p in enumerate(self.parts):
            try:
                return int(self.first[j]) + p.index(surface)
            except ValueError:
                pass
        raise ValueError("the surface is not in the assembly")

    def __add__(self, other):
        return SurfaceSeq(self.parts + (other.parts if isinstance(other, SurfaceSeq) else [other]))

    def __radd__(self, other):
        return SurfaceSeq([other] + self.parts)

    def segments(self):
        """(first index, part) of every part: lists of Surfaces and FaceSets"""
        return [(int(self.first[j]), p) for j, p in enumerate(self.parts)]

    def optics_of(self, k):
        j, i = self._locate(k)
        p = self.parts[j]
        return p.optics_of(i) if isinstance(p, FaceSet) else p[i].get_optics_manager()

    def distinct_optics(self):
        out, seen = [], set()
        for p in self.parts:
            for o in (p.distinct_optics() if isinstance(p, FaceSet) else [s.get_optics_manager() for s in p]):
                if id(o) not in seen:
                    seen.add(id(o))
                    out.append(o)
        return out
