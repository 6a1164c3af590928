"""Linear simplex element (P1 triangle / tetrahedron).

No counterpart in the reference (``Spectral`` is tensor-product only, src/domain/indices.py:116-122): it
exists for BASELINE.json's unstructured configuration ("tetrahedral mesh, Gmsh import, GMRES+Jacobi").
Same device interface as :class:`Spectral`: three quadrature tables for ``pyn_elem_tables_set`` and the
per-element entry points, all evaluated by the same HIP routines.

Reference simplex: vertices 0, e_1 .. e_dim; basis = barycentric coordinates
``l_0 = 1 - sum(xi), l_i = xi_i``.  Rules: full = degree-2 (3 points / 4 points), reduced = centroid,
nodal = vertex rule (lumped mass).
"""
import numpy as np

from .element import Element
from .spectral import Spectral
from .utilities import GaussPoint2D, GaussPoint3D


class Simplex(Spectral):
    def __init__(self, dim):
        if dim not in (2, 3):
            raise Exception("dim must be 2 or 3")
        Element.__init__(self, dim)
        self.ngl = 2
        self.nnode = dim + 1
        self.nnodedge = self.nnodcell = self.nnodface = 0
        self.elemType = 'Simplex{}D'.format(dim)
        self._dev = None
        if dim == 2:
            self.indWCurl = [[0, 0, 1], [1, 0, 0]]
            self.indCurl = [[0, 1, 0], [0, 0, 1]]
            self.indBdiv = [[0, 1], [1, 2]]
        else:
            self.indWCurl = [[0, 2, 1], [0, 1, 2], [1, 0, 2], [1, 2, 0], [2, 1, 0], [2, 0, 1]]
            self.indCurl = [[0, 2, 1], [0, 1, 2], [1, 0, 2], [1, 2, 0], [2, 1, 0], [2, 0, 1]]
            self.indBdiv = [[0, 1, 5], [1, 2, 3], [5, 3, 4]]
        vol = 0.5 if dim == 2 else 1.0 / 6.0
        if dim == 2:
            full = np.array([[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]])
        else:
            a, b = (5.0 + 3.0 * np.sqrt(5.0)) / 20.0, (5.0 - np.sqrt(5.0)) / 20.0
            full = np.array([[b, b, b], [a, b, b], [b, a, b], [b, b, a]])
        red = np.full((1, dim), 1.0 / (dim + 1))
        nod = np.vstack([np.zeros(dim), np.eye(dim)])
        grad = np.hstack([-np.ones((dim, 1)), np.eye(dim)])          # d l_a / d xi_d  [dim, nn], constant
        GP = GaussPoint2D if dim == 2 else GaussPoint3D

        def tables(pts):
            w = vol / len(pts)
            H = [np.concatenate([[1.0 - p.sum()], p]) for p in pts]
            return H, [grad.copy() for _ in pts], [GP(*p, w) for p in pts]
        self.H, self.Hrs, self.gps = tables(full)
        self.HRed, self.HrsRed, self.gpsRed = tables(red)
        self.HOp, self.HrsOp, self.gpsOp = tables(nod)
        # geometry basis == the element basis (isoparametric, affine map)
        self.HCoo, self.HrsCoo = self.H, self.Hrs
        self.HCooRed, self.HrsCooRed = self.HRed, self.HrsRed
        self.HCooOp, self.HrsCooOp = self.HOp, self.HrsOp

    def _device(self):
        if self._dev is None:
            from pynama_amd import _lib
            ctx = _lib.Context(_lib.default_device())
            conn = np.arange(self.nnode, dtype=np.int32)[None, :]
            ctx.mesh_set(self.dim, conn, np.vstack([np.zeros(self.dim), np.eye(self.dim)]))
            for t in self.deviceTables():
                ctx.tables_set(*t)
            self._dev = ctx
        return self._dev
