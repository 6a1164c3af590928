"""Tensor-product GLL spectral element (Q1 when ngl == 2).

Mirrors ``src/elements/spectral.py`` of the reference: same constructor, attributes
(``H, Hrs, gps | HRed.. | HOp.. | HCoo.. | HCooRed.. | HCooOp.. | HCoo1D``, :45-62 / :70-87),
same local node / Gauss-point ordering (vertices, edges, faces, interior; :220-271, :346-431)
and the same method signatures.  The tables are set-up work and stay on the host (numpy); the
per-element quadrature ``getElemKLEMatrices`` (:89-157) runs on the GPU through
``pyn_elem_local`` / ``pyn_assemble_kle`` -- there is no CPU implementation of it here.
"""
import numpy as np

from .element import Element
from .utilities import (GaussPoint2D, GaussPoint3D, gaussPoints, generateGaussPoints2D,
                        generateGaussPoints3D, lobattoPoints)

# Entity walk of the reference hexahedron / quadrilateral on the index lattice {0..m}^dim
# (axes r, s[, t]).  A "walk" is (start, step_outer, step_inner[, step_innermost]).
_QUAD_VERT = ((1, 1), (0, 1), (0, 0), (1, 0))
_HEX_VERT = ((0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1))
_HEX_EDGE = ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (3, 5), (4, 0), (1, 7), (6, 2))
# faces: (fixed axis, fixed side, outer axis, outer dir, inner axis, inner dir); dir -1 = descending
_HEX_FACE = ((2, 0, 1, -1, 0, +1), (2, 1, 1, +1, 0, -1), (1, 0, 0, -1, 2, +1),
             (1, 1, 0, +1, 2, -1), (0, 1, 2, -1, 1, -1), (0, 0, 2, +1, 1, +1))


def _local_lattice(n, dim):
    """[(i, j[, k])] lattice position of every local point in reference order."""
    if n == 1:
        return [(0,) * dim]
    m = n - 1
    fwd = list(range(1, m))
    seq = {+1: fwd, -1: fwd[::-1]}
    verts = _QUAD_VERT if dim == 2 else _HEX_VERT
    pts = [tuple(m * c for c in v) for v in verts]
    if dim == 2:
        edges = [(a, (a + 1) % 4) for a in range(4)]
    else:
        edges = _HEX_EDGE
    for a, b in edges:
        va, vb = np.array(verts[a]), np.array(verts[b])
        pts += [tuple(int(q) for q in (m * va + s * (vb - va))) for s in fwd]
    if dim == 2:
        pts += [(i, j) for i in fwd for j in fwd[::-1]]
        return pts
    for ax, side, oa, od, ia, idr in _HEX_FACE:
        for o in seq[od]:
            for i in seq[idr]:
                p = [0, 0, 0]
                p[ax], p[oa], p[ia] = side * m, o, i
                pts.append(tuple(p))
    pts += [(i, j, k) for k in fwd[::-1] for j in fwd[::-1] for i in fwd]
    return pts


class Spectral(Element):
    """Spectral element.  Attributes follow the reference (spectral.py:17-37)."""

    def __init__(self, ngl, dim):
        super().__init__(dim)
        self.ngl = ngl
        self.nnode = ngl ** dim
        self.nnodedge = ngl - 2
        self.nnodcell = (ngl - 2) ** dim
        self.elemType = 'Spectral{}D({})'.format(dim, ngl)
        self._dev = None
        if dim == 2:
            self.indWCurl = [[0, 0, 1], [1, 0, 0]]
            self.indCurl = [[0, 1, 0], [0, 0, 1]]
            self.indBdiv = [[0, 1], [1, 2]]
            self.setUpSpectralMats2D(ngl)
        elif dim == 3:
            self.indWCurl = [[0, 2, 1], [0, 1, 2], [1, 0, 2], [1, 2, 0], [2, 1, 0], [2, 0, 1]]
            self.indCurl = [[0, 2, 1], [0, 1, 2], [1, 0, 2], [1, 2, 0], [2, 1, 0], [2, 0, 1]]
            self.indBdiv = [[0, 1, 5], [1, 2, 3], [5, 3, 4]]
            self.nnodface = (ngl - 2) ** 2
            self.setUpSpectralMats3D(ngl)
        else:
            raise Exception("dim must be 2 or 3")

    # ------------------------------------------------------------------ tables (host, set-up)
    def _setUp(self, ngl, compute):
        nodes1D, operWei = lobattoPoints(ngl)
        gps1D, fullWei = gaussPoints(ngl) if ngl <= 3 else lobattoPoints(ngl)
        gps_red1D, redWei = gaussPoints(ngl - 1)
        cnodes1D, _ = lobattoPoints(2)
        (self.H, self.Hrs, self.gps) = compute(nodes1D, gps1D, fullWei)
        (self.HRed, self.HrsRed, self.gpsRed) = compute(nodes1D, gps_red1D, redWei)
        (self.HOp, self.HrsOp, self.gpsOp) = compute(nodes1D, nodes1D, operWei)
        (self.HCoo, self.HrsCoo, self.gpsCoo) = compute(cnodes1D, gps1D, fullWei)
        (self.HCooRed, self.HrsCooRed, self.gpsCooRed) = compute(cnodes1D, gps_red1D, redWei)
        (self.HCooOp, self.HrsCooOp, self.gpsCooOp) = compute(cnodes1D, nodes1D, operWei)
        (self.HCoo1D, _) = self.interpFun1D(cnodes1D, nodes1D)

    def setUpSpectralMats2D(self, ngl):
        self._setUp(ngl, self.computeMats2D)

    def setUpSpectralMats3D(self, ngl):
        self._setUp(ngl, self.computeMats3D)

    def _computeMats(self, nodes1D, gps1D, gps1Dwei, dim):
        h, dh = self.interpFun1D(nodes1D, gps1D)             # [ngp1, nn1]
        nod = np.array(_local_lattice(len(nodes1D), dim))    # [nn, dim]
        gpl = np.array(_local_lattice(len(gps1D), dim))      # [ngp, dim]
        # per-axis factors F[d][g, a] = h[g_d, a_d], D[d] likewise with dh
        F = [h[gpl[:, d]][:, nod[:, d]] for d in range(dim)]
        D = [dh[gpl[:, d]][:, nod[:, d]] for d in range(dim)]
        Hall = np.prod(F, axis=0)
        Hrs_all = np.stack([np.prod([D[e] if e == d else F[e] for e in range(dim)], axis=0)
                            for d in range(dim)], axis=1)    # [ngp, dim, nn]
        x = np.asarray(gps1D)
        w = np.asarray(gps1Dwei)
        if dim == 2:
            gps = [GaussPoint2D(r=x[i], s=x[j], w=w[i] * w[j]) for i, j in gpl]
        else:
            gps = [GaussPoint3D(r=x[i], s=x[j], t=x[k], w=w[i] * w[j] * w[k]) for i, j, k in gpl]
        return (list(Hall), list(Hrs_all), gps)

    def computeMats2D(self, nodes1D, gps1D, gps1Dwei):
        return self._computeMats(nodes1D, gps1D, gps1Dwei, 2)

    def computeMats3D(self, nodes1D, gps1D, gps1Dwei):
        return self._computeMats(nodes1D, gps1D, gps1Dwei, 3)

    @staticmethod
    def getSpectralOrder(nPoints):
        """invPerm: tensor index (r slowest, t fastest) of each local point (spectral.py:346-431)."""
        return [(i * nPoints + j) * nPoints + k for i, j, k in _local_lattice(nPoints, 3)]

    # ------------------------------------------------------------------ device tables
    def deviceTables(self):
        """[(which, w, H, Hrs, HrsCoo)] for pyn_elem_tables_set: full, reduced, nodal."""
        out = []
        for which, (H, Hrs, gps, HrsCoo) in enumerate((
                (self.H, self.Hrs, self.gps, self.HrsCoo),
                (self.HRed, self.HrsRed, self.gpsRed, self.HrsCooRed),
                (self.HOp, self.HrsOp, self.gpsOp, self.HrsCooOp))):
            out.append((which, np.array([g.w for g in gps]), np.array(H), np.array(Hrs), np.array(HrsCoo)))
        return out

    def _device(self):
        """Private single-element GPU context used by the per-element entry points."""
        if self._dev is None:
            from pynama_amd import _lib
            ctx = _lib.Context(_lib.default_device())
            nc = 2 ** self.dim
            lat = np.array(_local_lattice(2, self.dim), dtype=np.float64)
            conn = np.arange(self.nnode, dtype=np.int32)[None, :]
            xyz = np.zeros((self.nnode, self.dim))
            xyz[:nc] = lat
            ctx.mesh_set(self.dim, conn, xyz)
            for t in self.deviceTables():
                ctx.tables_set(*t)
            self._dev = ctx
        return self._dev

    # ------------------------------------------------------------------ hot path (GPU)
    def getElemKLEMatrices(self, coords):
        """Elementary matrices of the KLE method (spectral.py:89-157), computed on the GPU by the
        same device routine the global assembly uses.  Returns (K_e, Rw_e, Rd_e).
        Like the reference, reshapes the caller's `coords` in place (spectral.py:92)."""
        coords.shape = (int(len(coords) / self.dim), self.dim)
        from pynama_amd import _lib
        return self._device().elem_local(_lib.FORM_KLE, coords, alpha_d=1e3, alpha_w=1e2)

    def getElemLaplace(self, coords):
        """Scalar stiffness block L_e (K_e == kron(L_e, I) + penalties)."""
        from pynama_amd import _lib
        return self._device().elem_local(_lib.FORM_LAPLACE, np.asarray(coords, dtype=np.float64))

    def getElemMass(self, coords, nodal=True):
        """Scalar mass matrix: nodal rule = the reference's elWeigMat (spectral.py:215)."""
        from pynama_amd import _lib
        form = _lib.FORM_MASS_NODAL if nodal else _lib.FORM_MASS_FULL
        return self._device().elem_local(form, np.asarray(coords, dtype=np.float64))

    # ------------------------------------------------------------------ operators (GPU)
    def operatorTerms(self):
        """Term tables (row comp, col comp, derivative axis) + coefficients of the three first-order
        operators, as pyn_assemble_operator consumes them.  Restates the B-matrix fill of the reference
        (spectral.py:189-207): strain-rate rows through indBdiv plus the sign overrides, the
        divergence of the (Voigt-ordered) tensor, and the curl through indCurl."""
        dim, ds = self.dim, self.dim_s
        srt = {}
        for x in range(dim):
            for i in range(dim):
                srt[(self.indBdiv[x][i], i)] = (x, 1.0)
        srt[(0, 1)] = (1, -1.0)
        srt[(2, 0)] = (0, -1.0)
        for i in range(ds - 4):
            srt[(4, i)] = (i, -1.0)
            srt[(2 * i, 2)] = (2, -1.0)
        div = {}
        for x in range(dim):
            for i in range(dim):
                div[(i, self.indBdiv[x][i])] = (x, 1.0)
        curl = {}
        for n, (row, comp, der) in enumerate(self.indCurl):
            curl[(row, comp)] = (der, -1.0 if n % 2 else 1.0)

        def pack(d, scale):
            keys = sorted(d)
            return (np.array([[r, c, d[(r, c)][0]] for r, c in keys], dtype=np.int32),
                    np.array([scale * d[k][1] for k in keys], dtype=np.float64))
        return {"SrT": (ds, dim) + pack(srt, 0.5), "DivSrT": (dim, ds) + pack(div, 1.0),
                "Curl": (self.dim_w, dim) + pack(curl, 1.0)}

    def getElemKLEOperators(self, coords):
        """SrT_e, DivSrT_e, Curl_e and the lumped nodal weights (spectral.py:159-218) on the GPU, at
        the nodal GLL rule.  Returns (elSTensorMat, elDivSTMat, elCurlMat, elWeigVec)."""
        coords.shape = (int(len(coords) / self.dim), self.dim)
        from pynama_amd import _lib
        dev = self._device()
        out = []
        ops = self.operatorTerms()
        for name in ("SrT", "DivSrT", "Curl"):
            br, bc, terms, coef = ops[name]
            out.append(dev.elem_operator_local(_lib.Q_NODAL, br, bc, terms, coef, coords))
        wei = dev.elem_local(_lib.FORM_MASS_NODAL, coords).sum(axis=1)
        return (out[0], out[1], out[2], wei)
