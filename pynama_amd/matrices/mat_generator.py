"""Sparse containers of the KLE method on the GPU.

Mirrors ``src/matrices/mat_generator.py``: ``Mat`` (``createEmptyKLEMats`` :32-93, ``assembleAll``
:14-17, ``setIndices2One`` :113-118, ``createNNZWithArray`` :106-111, ``createNonZeroIndex``
:101-104, ``printMatsInfo`` :120-130) and ``Operators`` (:133-190).  PETSc's AIJ preallocation
becomes: all matrices share the device node graph built by ``DMPlexDom.getMatIndices`` and store
(br x bc) blocks per graph edge (layout in include/pynama_hip.h).  Values are filled by ONE fused
device pass (``assembleKLE``) instead of per-cell ``setValues`` calls.
"""
import logging

import numpy as np

from pynama_amd import _lib
from pynama_amd.common.comm import get_world
from pynama_amd.vectors import Vec


class DeviceMat:
    """PETSc.Mat look-alike bound to a device matrix handle."""

    def __init__(self, ctx, br, bc, name=None, rhs=False):
        """rhs: an imposed-column matrix (Krhs, Krhsfs: -K_e[free, bc] + the unit diagonal of the imposed DOFs) kept COMPACT -- only the
        node rows next to an imposed node are stored, the preallocation of mat_generator.py:42-58, 91 (`drhs_nnz`); laid out for the
        Dirichlet mask current on the device (Context.bc_set), again by an assembly under another one"""
        self.ctx, self.br, self.bc = ctx, br, bc
        self.id = ctx.mat_create_rhs(br, bc) if rhs else ctx.mat_create(br, bc)
        self._gen = getattr(ctx, "graph_gen", 0)
        self._name = name
        self._assembled = False

    def destroy(self):
        """PETSc.Mat.destroy(): release the device arrays (values, solver-side image, Jacobi data)."""
        # (a new pyn_csr_symbolic drops every matrix of the old graph and hands the ids out again)
        if self.id is not None and getattr(self.ctx, "h", None) and getattr(self.ctx, "graph_gen", 0) == self._gen:
            self.ctx.mat_destroy(self.id)
        self.id = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:       # interpreter shutdown / context already closed
            pass

    def setName(self, name):
        self._name = name

    def getName(self):
        return self._name

    def setUp(self):
        return self

    def assemble(self):
        self._assembled = True

    assemblyBegin = assemblyEnd = assemble

    def getOwnershipRange(self):
        r0 = getattr(self.ctx, "row_start", 0) * self.br
        return (r0, r0 + self.ctx.n_owned * self.br)

    def getSizes(self):
        return ((self.ctx.n_owned * self.br, None), (self.ctx.n_owned * self.bc, None))

    def createVecRight(self):
        return Vec(self.ctx, self.bc)

    def createVecLeft(self):
        return Vec(self.ctx, self.br)

    def mult(self, x, y):
        self.ctx.spmv(self.id, x.id, y.id)

    def __mul__(self, x):
        if isinstance(x, Vec):
            y = Vec(self.ctx, self.br)
            self.ctx.spmv(self.id, x.id, y.id)
            return y
        return NotImplemented

    def __add__(self, other):
        out = DeviceMat(self.ctx, self.br, self.bc, name=f"({self._name}+{other._name})")
        self.ctx.mat_axpy(out.id, 1.0, self.id)
        self.ctx.mat_axpy(out.id, 1.0, other.id)
        return out

    def setValues(self, rows, cols, values, addv=None):
        """PETSc.Mat.setValues: a dense block of values at scalar DOF indices (node * block + component), added when `addv`
        is true (base_problem.py:531-547, mat_generator.py:113-118, 157-170).  The compatibility path for code written against
        the reference's per-cell loops: one small upload + launch per call; the fused assemblies (Mat.assembleKLE,
        Operators.assembleOperators) are the fast path."""
        r0 = getattr(self.ctx, "row_start", 0)
        rows = np.atleast_1d(np.asarray(rows, dtype=np.int64)) - r0 * self.br
        cols = np.atleast_1d(np.asarray(cols, dtype=np.int64))
        if r0:
            raise NotImplementedError("Mat.setValues with global indices on more than one rank: use the fused assemblies")
        self.ctx.mat_add_values(self.id, rows, cols, values, insert=not addv)
        self._host_values = True
        self.matfree = None              # no longer the operator a fused assembly tagged (KspSolver would multiply with its shell)

    def setValue(self, row, col, value, addv=None):
        self.setValues([row], [col], [value], addv)

    def zeroEntries(self):
        self.ctx.mat_zero(self.id)
        self.matfree = None

    def diagonalScale(self, L=None, R=None):
        if R is not None:
            raise NotImplementedError("column scaling is not used by the reference path")
        if L is not None:
            self.ctx.mat_row_scale(self.id, L.id)
            self.matfree = None

    def getDiagonal(self, result=None):
        v = result or Vec(self.ctx, self.br)
        self.ctx.mat_diagonal(self.id, v.id)
        return v

    def getInfo(self):
        blocks, _ = self.ctx.mat_stored(self.id)
        nnz = blocks * self.br * self.bc
        return {"memory": nnz * 8 + self.ctx.nnzb * 4, "nz_allocated": nnz, "nz_used": nnz, "nz_unneeded": 0}


    def toScipy(self):
        """host copy as scipy CSR (diagnostics / tests only)"""
        import scipy.sparse as sp
        rp, ci = self.ctx.csr_get()
        val = self.ctx.mat_values(self.id, self.br, self.bc)
        lens = np.diff(rp).astype(np.int64)
        n = len(lens)
        node = np.repeat(np.arange(n), lens * self.br * self.bc)
        # storage order within a node: p, k, q
        within = np.concatenate([np.arange(l * self.br * self.bc) for l in lens]) if n < 200000 else None
        if within is None:
            raise MemoryError("toScipy is a diagnostic for small problems")
        ln = np.repeat(lens, lens * self.br * self.bc)
        p = within // (ln * self.bc)
        rem = within - p * ln * self.bc
        k = rem // self.bc
        q = rem - k * self.bc
        rows = node * self.br + p
        cols = ci[np.repeat(rp[:-1].astype(np.int64), lens * self.br * self.bc) + k].astype(np.int64) * self.bc + q
        return sp.coo_matrix((val, (rows, cols)), shape=(n * self.br, self.ctx.n_node * self.bc)).tocsr()


class Mat:
    def __init__(self, dim, comm=None):
        self.dim = dim
        self.comm = comm or get_world()
        self.logger = logging.getLogger(f"[{self.comm.rank}]:MatClass")
        self.dim_w = 1 if self.dim == 2 else 3
        self.dim_s = 3 if self.dim == 2 else 6
        self.mats = list()
        self.ctx = None

    def assembleAll(self):
        for m in self.mats:
            m.assemble()
            self.logger.debug(f"Mat {m.getName()} Assembled")

    def isParallel(self):
        return self.comm.size > 1

    def getGlobalIndices(self, localIndices):
        # borders are closed-form, so every rank already holds the global set (cf. :20-30)
        return set(localIndices)

    def createEmptyKLEMats(self, rStart, rEnd, d_nnz_ind, o_nnz_ind, ind_d, ind_o, indicesDIR):
        """Allocate K, Rw, Rd, Krhs on the shared device graph (mat_generator.py:32-93).
        `ind_d` is the DeviceGraph returned by DMPlexDom.getMatIndices and carries the context."""
        self.ctx = ind_d.ctx
        self.dom = ind_d.dom
        nodesDIR = self.getGlobalIndices(indicesDIR)
        tmp = np.array(sorted(nodesDIR), dtype=np.int64)
        self.globalIndicesDIR = (np.repeat(tmp * self.dim, self.dim)
                                 + np.tile(np.arange(self.dim), len(tmp))).astype(np.int64)
        self.K = DeviceMat(self.ctx, self.dim, self.dim, "K")
        self.Rw = DeviceMat(self.ctx, self.dim, self.dim_w, "Rw")
        self.Rd = DeviceMat(self.ctx, self.dim, 1, "Rd")
        # Krhs: rows next to an imposed node only, as the reference preallocates it (:42-58, 91) -- the Dirichlet mask goes to the device first
        self.ctx.bc_set(self.dim, self.dom.dirichletMaskLocal(self.dim))
        self.Krhs = DeviceMat(self.ctx, self.dim, self.dim, "Krhs", rhs=True)
        self.mats = [self.K, self.Rw, self.Rd, self.Krhs]

    def createEmptyMat(self, rows, cols, d_nonzero, offset_nonzero):
        raise NotImplementedError("matrices are allocated on the device graph: see createEmptyKLEMats")

    def assembleKLE(self, elem, alpha_d=1e3, alpha_w=1e2, with_rd=True, variant=1):
        """The fused device pass that replaces the cell loop of FreeSlip.buildKLEMats
        (base_problem.py:504-547) AND setIndices2One (:549)."""
        for t in elem.deviceTables():
            self.ctx.tables_set(*t)
        self.ctx.bc_set(self.dim, self.dom.dirichletMaskLocal(self.dim))
        # (the library picks the kernel: plan-free on lattices of parallelepipeds, 3x3x3-node patch plans otherwise)
        self.ctx.assemble_kle(alpha_d, alpha_w, self.K.id, self.Krhs.id, self.Rw.id,
                              self.Rd.id if with_rd else -1, variant)
        # K also exists in matrix-free form on structured Q1 hex meshes (KspSolver -pynama_mat_free)
        self.K.matfree = None
        if self.dim == 3 and elem.nnode == 8 and self.ctx.mesh_topology()[0] == "lattice":
            self.ctx.matfree_set(_lib.MATFREE_KLE, alpha_d, alpha_w)      # snapshot of THIS assembly's Dirichlet mask
            self.K.matfree = _lib.MATFREE_KLE

    def createNonZeroIndex(self, d_nnz, o_nnz, dim1, dim2):
        di_nnz = [x * dim1 for x in d_nnz for d in range(dim2)]
        oi_nnz = [x * dim1 for x in o_nnz for d in range(dim2)]
        return di_nnz, oi_nnz

    def createNNZWithArray(self, d_nnz, o_nnz, dim1: int, dim2: int):
        d_nnz = np.array(d_nnz, dtype=np.int32)
        o_nnz = np.array(o_nnz, dtype=np.int32)
        return np.repeat(d_nnz * dim1, dim2), np.repeat(o_nnz * dim1, dim2)

    def setIndices2One(self, indices2one):
        """Unit diagonal on imposed DOFs (:113-118).  Already applied on the device by the fused assembleKLE; inserted here
        when the matrices were filled through the host insertion path."""
        if getattr(self.K, "_host_values", False):     # matrices filled through Mat.setValues (per-cell loop of the reference)
            for indd in np.asarray(indices2one, dtype=np.int64):
                self.Krhs.setValues(indd, indd, 1, addv=True)
                self.K.setValues(indd, indd, 1, addv=True)
        self.Krhs.assemble()
        self.K.assemble()

    def printMatsInfo(self):
        print(" MATS INFO ")
        print("Mat   | Memory Used [B]  | NZ Unneeded")
        print("--------------------------------------")
        for m in self.mats:
            print(self.formatMatInfo(m.getName(), m.getInfo()))

    @staticmethod
    def formatMatInfo(name, info):
        return f"{name:{5}} | {info['memory']:{16}} | {info['nz_unneeded']:{10}}"


class Operators(Mat):
    """Curl / DivSrT / SrT operators with lumped-weight row scaling (mat_generator.py:133-190).
    The per-cell ``setValues`` loop of the reference (base_problem.py:132-140) is one device pass per
    operator (``pyn_assemble_operator``); unlike the reference, which integrates cell 0 once and reuses
    its blocks for every cell (valid on uniform meshes only), every cell is integrated."""
    _weights = _wvec = None

    def createAll(self, rStart, rEnd, d_nnz_ind, o_nnz_ind, graph=None):
        if graph is not None:
            self.ctx, self.dom = graph.ctx, graph.dom
        self.Curl = self.SrT = self.DivSrT = None
        self._weights = self._wvec = None

    def bind(self, graph):
        self.ctx, self.dom = graph.ctx, graph.dom

    def assembleOperators(self, elem):
        from pynama_amd import _lib
        ctx = self.ctx
        for t in elem.deviceTables():
            ctx.tables_set(*t)
        ops = elem.operatorTerms()
        mats = {}
        for name in ("Curl", "DivSrT", "SrT"):
            br, bc, terms, coef = ops[name]
            m = DeviceMat(ctx, br, bc, name)
            ctx.assemble_operator(_lib.Q_NODAL, terms, coef, m.id)
            mats[name] = m
        self.Curl, self.DivSrT, self.SrT = mats["Curl"], mats["DivSrT"], mats["SrT"]
        # lumped nodal weights = diagonal of the nodal-rule mass matrix (spectral.py:215-217)
        ctx.bc_set(1, None)
        mass = DeviceMat(ctx, 1, 1, "nodal-mass")
        ctx.assemble_scalar(_lib.FORM_MASS_NODAL, mass.id, -1, 0)
        self._wvec = mass.getDiagonal()                   # stays on the device; `weights` is its host copy on request
        self._weights = None
        rec = self._wvec.copy()                           # device copy
        rec.reciprocal()
        for m in (self.SrT, self.DivSrT, self.Curl):      # mat_generator.py:172-186: one factor per node for all of its rows
            m.diagonalScale(L=rec)
            m.assemble()
        mass.destroy()
        self.mats = [self.Curl, self.DivSrT, self.SrT]

    @property
    def weights(self):
        if self._weights is None:
            self._weights = self._wvec.getArray()
        return self._weights

    @weights.setter
    def weights(self, w):
        self._weights = w

    def lumpedWeights(self, bs):
        """The lumped nodal weights repeated per DOF (the reference keeps their reciprocal in `weigCurl` etc.,
        mat_generator.py:172-186, and inverts it again to weight its error norms, custom_func.py:145)."""
        v = Vec(self.ctx, bs)
        v.setArray(np.repeat(self.weights, bs))
        return v

    def setValues(self, localOperators, nodes):
        """per-cell insertion of the reference (mat_generator.py:157-170): the blocks of one cell into Curl / SrT / DivSrT and
        the lumped weights.  Compatibility path (Operators.assembleOperators does all cells in three device passes); call
        `createAll`, `setValues` per cell, `assembleAll`."""
        locSrT, locDivSrT, locCurl, locWei = localOperators
        nodes = np.asarray(list(nodes), dtype=np.int64)
        if self.Curl is None:
            self.Curl = DeviceMat(self.ctx, self.dim_w, self.dim, "Curl")
            self.SrT = DeviceMat(self.ctx, self.dim_s, self.dim, "SrT")
            self.DivSrT = DeviceMat(self.ctx, self.dim, self.dim_s, "DivSrT")
            self._wsum = np.zeros(self.ctx.n_owned)
        iv = (nodes[:, None] * self.dim + np.arange(self.dim)).ravel()
        iw = (nodes[:, None] * self.dim_w + np.arange(self.dim_w)).ravel()
        isr = (nodes[:, None] * self.dim_s + np.arange(self.dim_s)).ravel()
        self.Curl.setValues(iw, iv, locCurl, True)
        self.SrT.setValues(isr, iv, locSrT, True)
        self.DivSrT.setValues(iv, isr, locDivSrT, True)
        np.add.at(self._wsum, nodes, np.asarray(locWei, dtype=float).ravel())

    def assembleAll(self):
        """reciprocal lumped weights as row scaling (mat_generator.py:172-190) -- only after per-cell setValues; the fused
        assembleOperators has applied it already"""
        w = getattr(self, "_wsum", None)
        if w is None:
            return None
        self.weights = w
        for m in (self.SrT, self.DivSrT, self.Curl):
            wv = Vec(self.ctx, m.br)
            wv.setArray(np.repeat(w, m.br))
            wv.reciprocal()
            m.diagonalScale(L=wv)
            m.assemble()
        self.mats = [self.Curl, self.DivSrT, self.SrT]
        self._wsum = None
