"""Structured box-mesh domain with the reference's DMPlexDom interface.

Mirrors ``src/domain/dmplex.py`` (``DMPlexDom(PETSc.DMPlex)``): constructor keywords (:8-23),
``setFemIndexing`` (:42-61), ``computeFullCoordinates`` (:66-95), ``getCellCornersCoords``
(:97-104), border / label queries (:113-195), ``getGlobalNodesFromCell`` (:197-200), DOF index
helpers (:210-220), ``getMatIndices`` (:305-333) and the ``apply*ToVec`` helpers (:262-296).

What is different by design (SURVEY.md section 7 "hard parts"):
  * PETSc's DMPlex topology is replaced by a closed-form structured lattice: global node id =
    lexicographic lattice index (x fastest; matches src/tests/test_domain.py:197-201), element-
    local node order = the reference's vertex/edge/face/interior order, cell-0
    corner order as asserted in test_domain.py:26-30,94-104.
  * The per-cell closure lookups of the hot loop become ONE connectivity array uploaded to the
    GPU (`pyn_mesh_set`); `getMatIndices` becomes the device symbolic phase (`pyn_csr_symbolic`).
  * Parallel layout = contiguous node-plane slabs along the slowest axis, one process per GPU
    (same row-block ownership shape as PETSc MPIAIJ, dmplex.py:307 / mat_generator.py:96).
"""
import logging
import os
from math import floor

import numpy as np

from pynama_amd import _lib
from pynama_amd.common.comm import get_world
from pynama_amd.elements.spectral import _local_lattice


class IndicesManager:
    """Node / DOF bookkeeping of a domain (the role of src/domain/indices.py).  The reference's entity -> node maps driven by
    PetscSection offsets (:66-114) are the closed-form lattice numbering of DMPlexDom here; what is left is the DOF
    interleave (:90-92) and the Dirichlet / no-slip node sets (:45-64), which every rank holds globally (borders are closed
    form: no allgather)."""

    def __init__(self, dim, ngl, comm):
        self.logger = logging.getLogger(f"[{comm.rank}] IndicesManager Class")
        self.comm, self.dim, self._ngl = comm, dim, ngl
        self._sets = {"dirichlet": set(), "noslip": set()}

    def getNGL(self):
        return self._ngl

    def getNumCompAndNumDof(self, componentsPerField, numFields):
        """components per field, DOFs per mesh entity (vertex, edge, face[, cell]): an entity of dimension k carries (ngl - 2)^k nodes"""
        inner = self._ngl - 2
        return [componentsPerField] * numFields, [componentsPerField * inner ** k for k in range(self.dim + 1)]

    def setDirichletNodes(self, nodes: set):
        self._sets["dirichlet"].update(nodes)

    def getDirichletNodes(self):
        self.globalIndicesDIR = set(self._sets["dirichlet"])
        return self._sets["dirichlet"]

    def setNoSlipNodes(self, nodes: set):
        self._sets["noslip"].update(nodes)

    def getNoSlipNodes(self):
        self.globalIndicesNS = set(self._sets["noslip"])
        return self._sets["noslip"]

    def mapNodesToIndices(self, nodes, dof):
        """global DOF = node * dof + component, node-major (indices.py:90-92)"""
        return (np.asarray(list(nodes), dtype=np.int64)[:, None] * dof + np.arange(dof)).ravel().tolist()


class SlabPartition:
    """Node-plane slabs along the slowest lattice axis: rank r owns planes [a_r, b_r)."""

    def __init__(self, n_planes, m, nel_slow, size):
        if size > 1 and n_planes // size < m:
            raise ValueError(f"{n_planes} node planes cannot be split over {size} ranks with ngl-1={m}")
        self.size, self.m, self.nel = size, m, nel_slow
        self.bounds = [(r * n_planes) // size for r in range(size + 1)]

    def owned(self, r):
        return self.bounds[r], self.bounds[r + 1]

    def elem_layers(self, r):
        """element layers [k0, k1) that touch an owned plane"""
        a, b = self.owned(r)
        m = self.m
        k0 = max(0, -(-(a - m) // m))            # ceil((a-m)/m)
        k1 = min(self.nel - 1, (b - 1) // m) + 1
        return k0, k1

    def local_planes(self, r):
        k0, k1 = self.elem_layers(r)
        return k0 * self.m, k1 * self.m + 1       # [lo, hi)

    def owner_of_plane(self, p):
        return int(np.searchsorted(self.bounds, p, side="right") - 1)


class DeviceGraph:
    """Lazy stand-in for the per-row index sets returned by the reference's getMatIndices
    (dmplex.py:314-322).  Carries the device context; materialises python sets only on demand."""

    def __init__(self, dom, diag):
        self.dom, self.diag = dom, diag

    @property
    def ctx(self):
        return self.dom.ctx

    def __len__(self):
        return self.dom.nOwned

    def __getitem__(self, row):
        rp, ci = self.dom._hostGraph()
        cols = self.dom._local2global(ci[rp[row]:rp[row + 1]])
        r0, r1 = self.dom.rStart, self.dom.rEnd
        inside = (cols >= r0) & (cols < r1)
        return set(cols[inside if self.diag else ~inside].tolist())

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


def _morton_order(xyz):
    """argsort of the nodes along a Z-order curve (21 bits per axis)"""
    lo, hi = xyz.min(axis=0), xyz.max(axis=0)
    q = ((xyz - lo) / np.where(hi > lo, hi - lo, 1.0) * ((1 << 21) - 1)).astype(np.uint64)
    key = np.zeros(xyz.shape[0], dtype=np.uint64)
    for d in range(xyz.shape[1]):
        v = q[:, d]
        if xyz.shape[1] == 3:                 # spread 21 bits to every third position
            v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
            v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
            v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
            v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
            v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        else:                                 # every second position
            v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
            v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
            v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
            v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
            v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
        key |= v << np.uint64(d)
    return np.argsort(key, kind="stable")


def _lift_high_order(conn, xyz, ngl, dim):
    """Corner-node quadrilateral / hexahedral mesh -> ngl^dim nodes per cell in the reference's local order (vertices, edges,
    faces, interior; src/elements/spectral.py:346-431), shared entities numbered once.  What src/domain/indices.py:66-88 does
    with DMPlex's edge / face entities and their orientations is done with exact integer keys: a node on an edge is (the edge's
    two vertex ids, its distance from the smaller one); a node on a face is (the face's four vertex ids, its distances from the
    smallest vertex along the two face axes, the axis towards the smaller neighbour first); interior nodes belong to their
    cell.  Coordinates: the multilinear map of the corners at the GLL points (HCooOp, src/domain/dmplex.py:66-95).
    Returns (conn [E, ngl^dim], xyz [N, dim], {"ext_nodes": [F, ngl^(dim-1)] nodes of every exterior facet, "ext_corners":
    [F, 2^(dim-1)] their corner ids})."""
    from itertools import product

    from pynama_amd.elements.spectral import Spectral, _local_lattice
    E, m = conn.shape[0], ngl - 1
    lat = np.array(_local_lattice(ngl, dim))                 # tensor position of every local node
    clat = np.array(_local_lattice(2, dim))                  # ... of the corners ({0, 1}^dim)
    corner_at = {tuple(int(v) for v in c): k for k, c in enumerate(clat)}
    nn = lat.shape[0]
    keys = np.zeros((E, nn, 7), dtype=np.int64)
    el = np.arange(E, dtype=np.int64)
    for ln in range(nn):
        p = lat[ln]
        free = [d for d in range(dim) if 0 < p[d] < m]
        if not free:                                         # vertex
            keys[:, ln, 0], keys[:, ln, 1] = 0, conn[:, corner_at[tuple(int(v) // m for v in p)]]
            continue
        if len(free) == dim:                                 # interior of the cell
            keys[:, ln, 0], keys[:, ln, 1], keys[:, ln, 2] = 3, el, ln
            continue
        combos = list(product((0, 1), repeat=len(free)))     # corners of the edge / face the node sits on
        bits = np.array(combos)
        ids = np.empty((E, len(combos)), dtype=np.int64)
        for q, cb in enumerate(combos):
            c = [int(v) // m for v in p]
            for d, b in zip(free, cb):
                c[d] = b
            ids[:, q] = conn[:, corner_at[tuple(c)]]
        o = np.argmin(ids, axis=1)                           # the smallest vertex of the entity: its origin
        obit = bits[o]                                       # [E, len(free)]
        dist = np.abs(np.array([p[d] for d in free])[None, :] - obit * m)
        srt = np.sort(ids, axis=1)
        keys[:, ln, 0] = len(free)
        keys[:, ln, 1:1 + srt.shape[1]] = srt
        if len(free) == 1:
            keys[:, ln, 5] = dist[:, 0]
        else:                                                # face: the axis towards the smaller neighbour of the origin first
            nb = np.empty((E, 2), dtype=np.int64)
            for a in range(2):
                flip = obit.copy()
                flip[:, a] ^= 1
                nb[:, a] = ids[el, flip[:, 0] * 2 + flip[:, 1]]
            swap = nb[:, 1] < nb[:, 0]
            keys[:, ln, 5] = np.where(swap, dist[:, 1], dist[:, 0])
            keys[:, ln, 6] = np.where(swap, dist[:, 0], dist[:, 1])
    uniq, inv = np.unique(keys.reshape(-1, 7), axis=0, return_inverse=True)
    conn_ho = inv.reshape(E, nn)
    nv = int((uniq[:, 0] == 0).sum())
    assert np.array_equal(uniq[:nv, 1], np.arange(nv)), "every vertex of the file must belong to a cell"
    H = np.asarray(Spectral(ngl, dim).HCooOp)                # [nn, 2^dim]: corner basis at the nodal points
    xyz_ho = np.empty((uniq.shape[0], dim))
    xyz_ho[conn_ho.ravel()] = np.einsum("gc,ecd->egd", H, xyz[conn[:, :2 ** dim]]).reshape(-1, dim)
    xyz_ho[:nv] = xyz[:nv]                                   # vertices keep the file's coordinates bit for bit
    # exterior facets (belong to one cell) with all their nodes
    fn, fc = [], []
    for d in range(dim):
        for side in (0, 1):
            loc = np.nonzero(lat[:, d] == side * m)[0]
            cor = [k for k, c in enumerate(clat) if c[d] == side]
            fn.append(conn_ho[:, loc])
            fc.append(conn[:, cor])
    fn, fc = np.concatenate(fn), np.concatenate(fc)
    _, first, counts = np.unique(np.sort(fc, axis=1), axis=0, return_index=True, return_counts=True)
    sel = first[counts == 1]
    return conn_ho, xyz_ho, {"ext_nodes": fn[sel], "ext_corners": fc[sel]}


class DMPlexDom(object):
    def __init__(self, **kwargs):
        self.comm = kwargs.get('comm') or get_world()
        if 'boxMesh' in kwargs or 'nelem' in kwargs:
            meshData = kwargs.get('boxMesh') or {}
            lower = kwargs['lower'] if 'lower' in kwargs else meshData.get('lower')
            upper = kwargs['upper'] if 'upper' in kwargs else meshData.get('upper')
            faces = kwargs['nelem'] if 'nelem' in kwargs else meshData.get('nelem')
            if isinstance(lower[0], str):      # reference: eval() of string bounds (dmplex.py:18-20)
                from math import pi  # noqa: F401  (names usable inside the yaml expressions)
                lower = [eval(v) for v in lower]
                upper = [eval(v) for v in upper]
            self.nelem = [int(n) for n in faces]
            self.lower = [float(v) for v in lower[:len(self.nelem)]]
            self.upper = [float(v) for v in upper[:len(self.nelem)]]
        elif 'fileName' in kwargs or 'mesh' in kwargs:
            # Gmsh quad/hex mesh (dmplex.py:22-23): explicit connectivity, file node numbering.  `mesh=` takes what
            # read_msh returns ({"dim", "xyz", "conn", "facets", "cell"}): an imported mesh that is already in memory
            from pynama_amd.domain.gmsh import read_msh
            self._msh = kwargs['mesh'] if 'mesh' in kwargs else read_msh(kwargs['fileName'])
            dimf = self._msh["dim"]
            self.nelem = [0] * dimf
            self.lower = [float(v) for v in self._msh["xyz"].min(axis=0)]
            self.upper = [float(v) for v in self._msh["xyz"].max(axis=0)]
        else:
            raise ValueError("DMPlexDom needs boxMesh=... or nelem/lower/upper")
        self.dim = len(self.nelem)
        self.dim_w = 1 if self.dim == 2 else 3
        self.dim_s = 3 if self.dim == 2 else 6
        self.logger = logging.getLogger(f"[{self.comm.rank}] Class")
        if self.dim == 2:
            self.namingConvention = ["down", "right", "up", "left"]
            self._border_axis = {"down": (1, 0), "right": (0, 1), "up": (1, 1), "left": (0, 0)}
        elif self.dim == 3:
            self.namingConvention = ["back", "front", "down", "up", "right", "left"]
            self._border_axis = {"back": (2, 0), "front": (2, 1), "down": (1, 0), "up": (1, 1),
                                 "right": (0, 1), "left": (0, 0)}
        else:
            raise ValueError("dim must be 2 or 3")
        self._ctx = None
        self._unstructured = hasattr(self, "_msh")
        self.reorder = kwargs.get('reorder', 'morton')
        self.jitter = float(kwargs.get('jitter', 0.0))
        self.jitterSeed = int(kwargs.get('jitterSeed', 12345))
        self._graph = None
        self._fullCoordVec = None
        self._conn = self._xyz = self._lat_idx_ = None

    @property
    def ctx(self):
        """Device context of this rank: created, bound to the communicator and loaded with the local
        mesh on first use.  Raises if no MI355X is visible (there is no CPU fallback)."""
        if self._ctx is None:
            if not hasattr(self, "indicesManager"):
                raise RuntimeError("setFemIndexing(ngl) first")
            ctx = _lib.Context(_lib.default_device())
            if self.comm.size > 1 and os.environ.get("PYNAMA_SHM_TRANSPORT"):
                # TEST transport: ranks share one GPU, collectives go through the shared-memory file (tests/test_gpu_dist.py)
                ctx.comm_init_shm(self.comm.rank, self.comm.size, os.environ["PYNAMA_SHM_TRANSPORT"],
                                  int(os.environ.get("PYNAMA_SHM_CAP", str(8 << 20))))
                plan = self._halo_plan()
                self._neigh_ranks = list(plan[2])
                ctx.halo_set(*plan)
            elif self.comm.size > 1:
                uid = self.comm.unique_id(_lib.Context.unique_id)          # a fresh id per communicator
                with self.comm.bounded(f"ncclCommInitRank ({self.comm.size} ranks)"):
                    ctx.comm_init(self.comm.rank, self.comm.size, uid)
                plan = self._halo_plan()
                self._neigh_ranks = list(plan[2])
                ctx.halo_set(*plan)
            elif os.environ.get("PYNAMA_FORCE_COMM") == "1":
                # one rank, but with a real RCCL communicator: the collective code paths of a multi-GPU
                # run (single-reduction CG, all-reduced scalars) can be timed and tested on one GPU
                ctx.comm_init(0, 1, _lib.Context.unique_id())
            if self._unstructured or self.jitter > 0.0 or os.environ.get("PYNAMA_HOST_MESH"):
                ctx.mesh_set(self.dim, self.conn, self.xyz)
            else:
                # box mesh: connectivity and coordinates are generated where they are used (pyn_mesh_box); the host copies
                # (`conn`, `xyz`) are built only if a caller asks for them
                k0, k1 = self._layers
                ctx.mesh_box(self.dim, self.ngl, self.nelem[:-1] + [k1 - k0], k0, self.lattice, self._loc,
                             self._local_plane_ids(), self._axes())
            ctx.row_start = self.rStart
            self._ctx = ctx
        return self._ctx

    @property
    def fullCoordVec(self):
        """device copy of the nodal coordinates (dmplex.py:73-92), uploaded on first use"""
        if self._fullCoordVec is None:
            from pynama_amd.vectors import Vec
            self._fullCoordVec = Vec(self.ctx, self.dim, name='NodeCoordinates')
            self._fullCoordVec.setArray(self.xyz[:self.nOwned].ravel())
        return self._fullCoordVec

    # ------------------------------------------------------------------ basic queries
    def getDimension(self):
        return self.dim

    def getBoundingBox(self):
        return tuple((self.lower[d], self.upper[d]) for d in range(self.dim))

    def getNGL(self):
        return self.indicesManager.getNGL()

    def getHeightStratum(self, h):
        assert h == 0
        return self.cellStart, self.cellEnd

    # ------------------------------------------------------------------ indexing / mesh build
    def setFemIndexing(self, ngl):
        """Build the GLL lattice, the connectivity in reference local order, the slab partition,
        and upload the local mesh to the GPU (replaces PetscSection set-up, dmplex.py:42-61)."""
        dim = self.dim
        self.indicesManager = IndicesManager(dim, ngl, self.comm)
        if self._unstructured:
            return self._setUnstructuredIndexing(ngl)
        m = ngl - 1
        self.ngl = ngl
        self.lattice = tuple(m * n + 1 for n in self.nelem)
        self.strides = [int(np.prod(self.lattice[:d])) for d in range(dim)]
        self.nNodesGlobal = int(np.prod(self.lattice))
        self.part = SlabPartition(self.lattice[-1], m, self.nelem[-1], self.comm.size)
        rank = self.comm.rank
        a, b = self.part.owned(rank)
        lo, hi = self.part.local_planes(rank)
        k0, k1 = self.part.elem_layers(rank)
        plane = self.strides[-1]
        self.rStart, self.rEnd = a * plane, b * plane
        self.nOwned = self.rEnd - self.rStart
        # local numbering: owned planes, then ghost planes below, then ghost planes above
        self._ghost_lo = (lo, a)
        self._ghost_hi = (b, hi)
        self.nGhost = ((a - lo) + (hi - b)) * plane
        self.nLocal = self.nOwned + self.nGhost
        # element range of this rank (all cells of the layers that touch owned planes)
        cells_per_layer = int(np.prod(self.nelem[:-1]))
        self.cellStart, self.cellEnd = 0, (k1 - k0) * cells_per_layer     # local numbering (dmplex.py:99)
        self._cell0_global = k0 * cells_per_layer
        self._layers = (k0, k1)
        # connectivity (vectorised): conn[e, a] = base(e) + offset(a)
        loc = np.array(_local_lattice(ngl, dim), dtype=np.int64)
        if dim == 2:                      # x ~ -r, y ~ -s in 2D
            loc = m - loc
        self._loc = loc
        # the host copies of connectivity / coordinates are built on first use (`conn`, `xyz`): the device generates its own
        self._conn = self._xyz = self._lat_idx_ = None
        # ---- device: created lazily (first use of .ctx) so that the host logic runs without a GPU
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = None
        self._graph = None
        if not self.comm.rank:
            self.logger.debug("FEM/SEM Indexing SetUp")

    # host copies of the local mesh: conn[e, a] = base(e) + offset(a) (vectorised), coordinates GLL spaced inside each element
    @property
    def conn(self):
        if self._conn is None:
            dim, m = self.dim, self.ngl - 1
            k0, k1 = self._layers
            shape = tuple(reversed(self.nelem[:-1] + [k1 - k0]))
            eidx = np.indices(shape).reshape(dim, -1)[::-1].astype(np.int64)
            eidx[-1] += k0
            base = sum(eidx[d] * m * self.strides[d] for d in range(dim))
            off = sum(self._loc[:, d] * self.strides[d] for d in range(dim))
            if self.comm.size == 1 and self.nNodesGlobal < 2 ** 31:      # one rank: local == lattice ids, built in int32 at once
                self._conn = base.astype(np.int32)[:, None] + off.astype(np.int32)[None, :]
            else:
                self._conn = self._global2local(base[:, None] + off[None, :]).astype(np.int32)
        return self._conn

    @conn.setter
    def conn(self, v):
        self._conn = v

    @property
    def xyz(self):
        if self._xyz is None:
            self._xyz = self._lattice_coordinates()
        return self._xyz

    @xyz.setter
    def xyz(self, v):
        self._xyz = v

    @property
    def _lat_idx(self):
        """lattice index per axis of every local node"""
        if self._lat_idx_ is None:
            rem = self._local2global(np.arange(self.nLocal))
            idx = []
            for d in reversed(range(self.dim)):
                q = rem // self.strides[d]
                rem = rem - q * self.strides[d]
                idx.append(q)
            self._lat_idx_ = idx[::-1]
        return self._lat_idx_

    def _axes(self):
        """coordinate of every lattice line, per axis (GLL spaced inside each element)"""
        from pynama_amd.elements.utilities import lobattoPoints
        m = self.ngl - 1
        gll, _ = lobattoPoints(self.ngl)
        axes = []
        for d in range(self.dim):
            h = (self.upper[d] - self.lower[d]) / self.nelem[d]
            e = np.arange(self.nelem[d])
            ax = np.empty(self.lattice[d])
            ax[:-1] = (self.lower[d] + h * (e[:, None] + 0.5 * (1.0 + np.asarray(gll)[None, :m]))).ravel()
            ax[-1] = self.upper[d]
            axes.append(ax)
        return axes

    def _setUnstructuredIndexing(self, ngl):
        """explicit (Gmsh) mesh: first-order cells.  Nodes are renumbered along a Morton curve (locality for
        the SpMV gathers and the scatter; PETSc renumbers imported meshes too), rows are split in contiguous
        blocks over the ranks, every rank keeps the cells touching an owned node (owner-computes) and the
        ghost nodes are addressed through an index list."""
        from pynama_amd.domain.gmsh import exterior_facets
        rank, size = self.comm.rank, self.comm.size
        self.ngl = ngl
        self.cellType = self._msh["cell"]
        conn = self._msh["conn"].astype(np.int64)
        xyz = self._msh["xyz"]
        facets = self._msh["facets"]
        ho = None
        if ngl != 2:
            if self.cellType != "tensor":
                raise NotImplementedError("high-order nodes are generated on quadrilateral / hexahedral cells only")
            # the file carries corner nodes: edge / face / interior nodes are generated here, as IndicesManager does from the
            # DMPlex entities (src/domain/indices.py:66-88)
            conn, xyz, ho = _lift_high_order(conn, xyz, ngl, self.dim)
        n = xyz.shape[0]
        if self.reorder == "morton":
            new_of_old = np.empty(n, dtype=np.int64)
            new_of_old[_morton_order(xyz)] = np.arange(n)
            xyz = xyz[np.argsort(new_of_old)]
            conn = new_of_old[conn]
            facets = [(p, new_of_old[f]) for p, f in facets]
        self.nNodesGlobal = n
        if ho is not None:
            # exterior facets with ALL their nodes (corners + the generated edge / face nodes), in the new numbering
            ext = (new_of_old[ho["ext_nodes"]] if self.reorder == "morton" else ho["ext_nodes"])
            if facets:     # physical tags of the file's boundary facets (corner tuples) -> every node of that facet
                tag_of = {tuple(sorted(int(v) for v in f)): p for p, f in facets}
                cor = new_of_old[ho["ext_corners"]] if self.reorder == "morton" else ho["ext_corners"]
                facets = [(tag_of.get(tuple(sorted(int(v) for v in c)), 0), nodes) for c, nodes in zip(cor, ext)]
        else:
            ext = exterior_facets(conn, self.dim)
        ext_mask = np.zeros(n, dtype=bool)
        ext_mask[ext.ravel()] = True
        # named borders: physical tag k of a boundary facet <-> namingConvention[k-1] ("Face Sets", dmplex.py:168-171);
        # without tagged facets, exterior facets lying in a bounding-box plane are assigned by position
        self._border_ids = {name: set() for name in self.namingConvention}
        if facets:
            for phys, nodes in facets:
                if 1 <= phys <= len(self.namingConvention):
                    self._border_ids[self.namingConvention[phys - 1]].update(int(v) for v in nodes)
        else:
            tol = 1e-9 * max(u - l for l, u in zip(self.lower, self.upper))
            for name, (d, hi) in self._border_axis.items():
                ref = self.upper[d] if hi else self.lower[d]
                flat = np.all(np.abs(xyz[ext][:, :, d] - ref) < tol, axis=1)
                self._border_ids[name].update(int(v) for v in ext[flat].ravel())
        # ---- row blocks + owner-computes cell sets
        bounds = np.array([(r * n) // size for r in range(size + 1)], dtype=np.int64)
        if size > 1 and np.any(np.diff(bounds) == 0):
            raise ValueError(f"{n} nodes cannot be split over {size} ranks")
        self._bounds = bounds
        self.rStart, self.rEnd = int(bounds[rank]), int(bounds[rank + 1])
        self.nOwned = self.rEnd - self.rStart
        owner = np.searchsorted(bounds, conn, side="right") - 1                 # [E, nn]
        mine = np.any(owner == rank, axis=1)
        lconn = conn[mine]
        ghosts = np.unique(lconn[(lconn < self.rStart) | (lconn >= self.rEnd)])    # sorted => grouped by owner
        self._ghost_gids = ghosts
        self.nGhost = int(ghosts.size)
        self.nLocal = self.nOwned + self.nGhost
        self.cellStart, self.cellEnd = 0, int(mine.sum())
        self.conn = self._global2local(lconn).astype(np.int32)
        gids = self._local2global(np.arange(self.nLocal))
        self.xyz = xyz[gids]
        self._ext_mask = ext_mask[gids]
        self._ext_ids_global = np.nonzero(ext_mask)[0].tolist()
        # ---- halo plan: node g is a ghost on rank k iff some cell holds g and a node owned by k != owner(g)
        self._unstructured_plan = None
        if size > 1:
            iface = np.any(owner != owner[:, :1], axis=1)
            ic, io = conn[iface], owner[iface]
            nn = ic.shape[1]
            g = np.repeat(ic, nn, axis=1).ravel()                 # node a, paired with ...
            go = np.repeat(io, nn, axis=1).ravel()
            k = np.tile(io, (1, nn)).ravel()                      # ... the owner of node b
            sel = go != k
            pairs = np.unique(np.stack([k[sel], g[sel]], axis=1), axis=0)       # (rank holding the ghost, node)
            pown = np.searchsorted(bounds, pairs[:, 1], side="right") - 1
            send = pairs[pown == rank]                            # my nodes that are ghosts elsewhere
            recv = pairs[pairs[:, 0] == rank]                     # my ghosts, with their owners
            assert np.array_equal(np.sort(recv[:, 1]), ghosts)
            rown = pown[pairs[:, 0] == rank]
            neigh = sorted(set(send[:, 0].tolist()) | set(rown.tolist()))
            send_ptr, recv_ptr, send_idx = [0], [0], []
            for nb in neigh:
                ids = np.sort(send[send[:, 0] == nb][:, 1]) - self.rStart
                send_idx.append(ids)
                send_ptr.append(send_ptr[-1] + ids.size)
                recv_ptr.append(recv_ptr[-1] + int((rown == nb).sum()))
            sidx = np.concatenate(send_idx).astype(np.int32) if send_idx else np.zeros(0, np.int32)
            self._unstructured_plan = (self.nOwned, self.nGhost, np.array(neigh, np.int32),
                                       np.array(send_ptr, np.int64), sidx, np.array(recv_ptr, np.int64))
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = None
        self._graph = None

    def _local_plane_ids(self):
        a, b = self.part.owned(self.comm.rank)
        return list(range(a, b)) + list(range(*self._ghost_lo)) + list(range(*self._ghost_hi))

    def _global2local(self, g):
        """global lattice node id -> local id (owned first, ghosts below, ghosts above)"""
        g = np.asarray(g, dtype=np.int64)
        if self._unstructured:
            out = np.full(g.shape, -1, dtype=np.int64)
            own = (g >= self.rStart) & (g < self.rEnd)
            out[own] = g[own] - self.rStart
            if self.nGhost:
                pos = np.minimum(np.searchsorted(self._ghost_gids, g), self.nGhost - 1)
                gh = (~own) & (self._ghost_gids[pos] == g)
                out[gh] = self.nOwned + pos[gh]
            return out
        if self.comm.size == 1:          # one rank: local ids ARE the lattice ids
            return g
        plane = self.strides[-1]
        a, b = self.part.owned(self.comm.rank)
        lo, hi = self._ghost_lo[0], self._ghost_hi[1]
        p = g // plane
        inpl = g - p * plane
        out = np.full(g.shape, -1, dtype=np.int64)
        own = (p >= a) & (p < b)
        glo = (p >= lo) & (p < a)
        ghi = (p >= b) & (p < hi)
        out[own] = (p[own] - a) * plane + inpl[own]
        out[glo] = self.nOwned + (p[glo] - lo) * plane + inpl[glo]
        out[ghi] = self.nOwned + (a - lo) * plane + (p[ghi] - b) * plane + inpl[ghi]
        return out

    def _local2global(self, l):
        l = np.asarray(l, dtype=np.int64)
        if self._unstructured:
            out = l + self.rStart
            gh = l >= self.nOwned
            out[gh] = self._ghost_gids[l[gh] - self.nOwned]
            return out
        if self.comm.size == 1:
            return l
        plane = self.strides[-1]
        a, b = self.part.owned(self.comm.rank)
        lo = self._ghost_lo[0]
        n_lo = (a - lo) * plane
        out = np.empty(l.shape, dtype=np.int64)
        own = l < self.nOwned
        glo = (~own) & (l < self.nOwned + n_lo)
        ghi = (~own) & (~glo)
        out[own] = l[own] + a * plane
        out[glo] = l[glo] - self.nOwned + lo * plane
        out[ghi] = l[ghi] - self.nOwned - n_lo + b * plane
        return out

    def _halo_plan(self):
        """(n_owned, n_ghost, neigh, send_ptr, send_idx, recv_ptr) for pyn_halo_set."""
        if self._unstructured:
            return self._unstructured_plan
        rank, size = self.comm.rank, self.comm.size
        plane = self.strides[-1]
        a, b = self.part.owned(rank)
        neigh, send_ptr, send_idx, recv_ptr = [], [0], [], [0]
        for nb, (g0, g1) in ((rank - 1, self._ghost_lo), (rank + 1, self._ghost_hi)):
            if nb < 0 or nb >= size:
                assert g1 - g0 <= 0
                continue
            # what we receive from nb: our ghost planes [g0, g1) (all owned by nb: slabs >= m planes)
            assert self.part.owner_of_plane(g0) == nb and self.part.owner_of_plane(g1 - 1) == nb
            # what nb needs from us: its ghost planes on our side
            nlo, nhi = self.part.local_planes(nb)
            na, nb_b = self.part.owned(nb)
            s0, s1 = (nb_b, nhi) if nb < rank else (nlo, na)
            assert s0 >= a and s1 <= b
            neigh.append(nb)
            ids = np.arange((s0 - a) * plane, (s1 - a) * plane, dtype=np.int64)
            send_idx.append(ids)
            send_ptr.append(send_ptr[-1] + ids.size)
            recv_ptr.append(recv_ptr[-1] + (g1 - g0) * plane)
        sidx = np.concatenate(send_idx).astype(np.int32) if send_idx else np.zeros(0, np.int32)
        return (self.nOwned, self.nGhost, np.array(neigh, np.int32), np.array(send_ptr, np.int64), sidx,
                np.array(recv_ptr, np.int64))

    def _lattice_coordinates(self):
        dim = self.dim
        axes = self._axes()
        idx = self._lat_idx
        xyz = np.empty((self.nLocal, dim))
        for d in range(dim):
            xyz[:, d] = axes[d][idx[d]]
        if self.jitter > 0.0:
            if self.ngl != 2:
                raise ValueError("jitter needs ngl == 2 (geometry is multilinear from the corners)")
            # same stream for every rank: draw for the whole lattice, pick the local nodes
            rng = np.random.default_rng(self.jitterSeed)
            hmin = min((self.upper[d] - self.lower[d]) / self.nelem[d] for d in range(dim))
            move = self.jitter * hmin * rng.uniform(-1, 1, size=(self.nNodesGlobal, dim))
            on = np.zeros(self.nLocal, dtype=bool)
            for d in range(dim):
                on |= (idx[d] == 0) | (idx[d] == self.lattice[d] - 1)
            mv = move[self._local2global(np.arange(self.nLocal))]
            mv[on] = 0.0
            xyz = xyz + mv
        return xyz

    def _coords_of_local(self, ln):
        """coordinates of the local nodes `ln` without the full array (box meshes: from the lattice lines)"""
        if self._unstructured or self.jitter > 0.0 or self._xyz is not None:
            return self.xyz[ln]
        axes = self._axes()
        rem = self._local2global(np.asarray(ln, dtype=np.int64))
        out = np.empty((rem.size, self.dim))
        for d in reversed(range(self.dim)):
            q = rem // self.strides[d]
            rem = rem - q * self.strides[d]
            out[:, d] = axes[d][q]
        return out

    # ------------------------------------------------------------------ coordinates
    def computeFullCoordinates(self, spElem):
        """Nodal coordinates of every (owned) node.  The reference interpolates the corner
        coordinates with HCooOp per cell (dmplex.py:66-95); on a box mesh that is exactly the GLL
        lattice built in setFemIndexing."""
        self.nodes = range(self.rStart, self.rEnd)

    def getCellCornersCoords(self, cell):
        if cell + self.cellStart >= self.cellEnd:
            raise Exception('elem parameter must be in local numbering!')
        nc = 2 ** self.dim
        return self.xyz[self.conn[cell, :nc]].reshape(nc * self.dim).copy()

    # ------------------------------------------------------------------ nodes / DOFs
    def getGlobalNodesFromCell(self, cell, shared):
        nodes = self._local2global(self.conn[cell])
        if not shared:
            nodes = nodes[(nodes >= self.rStart) & (nodes < self.rEnd)]
        return [int(n) for n in nodes]

    def getGlobalNodesFromEntities(self, entities, shared):
        nodes = set()
        for cell in entities:
            nodes |= set(self.getGlobalNodesFromCell(cell, shared))
        return nodes

    def getVelocityIndex(self, nodes):
        return self.indicesManager.mapNodesToIndices(nodes, self.dim)

    def getVorticityIndex(self, nodes):
        return self.indicesManager.mapNodesToIndices(nodes, self.dim_w)

    def getSrtIndex(self, nodes):
        return self.indicesManager.mapNodesToIndices(nodes, self.dim_s)

    def getAllNodes(self):
        return range(self.rStart, self.rEnd)

    def getNodesCoordinates(self, nodes=None, indices=None):
        dim = self.dim
        if nodes is None:
            assert indices is not None
            nodes = [int(i // dim) for i in list(indices)[::dim]][:floor(len(indices) / dim)]
        loc = self._global2local(np.asarray(nodes, dtype=np.int64))
        if np.any(loc < 0):
            raise IndexError("node not stored on this rank")
        return self._coords_of_local(loc).reshape((len(nodes), dim))

    # ------------------------------------------------------------------ borders / labels
    def _on_border_mask(self, name):
        if self._unstructured:
            ids = np.fromiter(self._border_ids[name], dtype=np.int64, count=len(self._border_ids[name]))
            return np.isin(self._local2global(np.arange(self.nLocal)), ids)
        d, hi = self._border_axis[name]
        return self._lat_idx[d] == (self.lattice[d] - 1 if hi else 0)

    def _global_border_nodes(self, name):
        """all (global) nodes of a border, computed in closed form on every rank"""
        if self._unstructured:
            return np.array(sorted(self._border_ids[name]), dtype=np.int64)
        d, hi = self._border_axis[name]
        others = [k for k in range(self.dim) if k != d]
        shape = tuple(self.lattice[k] for k in reversed(others))
        idx = np.indices(shape).reshape(len(others), -1)[::-1].astype(np.int64)     # lowest axis fastest
        ids = (self.lattice[d] - 1 if hi else 0) * self.strides[d]
        ids = ids + sum(idx[j] * self.strides[k] for j, k in enumerate(others))
        return np.sort(ids)

    def setLabelToBorders(self):
        """dmplex.py:113-131 builds a bit-label per entity; here borders are closed-form masks."""
        self._labels_set = True

    def getBordersNames(self):
        return self.namingConvention

    def getBorderNodes(self, name):
        return self._global_border_nodes(name).tolist()

    def getBordersNodes(self) -> set:
        nodes = set()
        for faceName in self.namingConvention:
            nodes |= set(self.getBorderNodes(faceName))
        return nodes

    def getNodesFromLabel(self, label, shared=False) -> set:
        if label != "External Boundary":
            self.logger.warning(f"Label >> {label} << found")
            return set()
        if self._unstructured:
            return set(self._ext_ids_global)
        return self.getBordersNodes()

    def boundaryMaskLocal(self):
        """uint8 [nLocal]: 1 on 'External Boundary' nodes (owned + ghost)."""
        if self._unstructured:
            return self._ext_mask.astype(np.uint8)
        # local nodes = local planes (owned first, then ghosts) of the lattice's leading axes: the borders are slices
        planes = np.asarray(self._local_plane_ids())
        on = np.zeros((planes.size,) + tuple(reversed(self.lattice[:-1])), dtype=np.uint8)
        on[(planes == 0) | (planes == self.lattice[-1] - 1)] = 1
        on[:, ..., 0] = 1
        on[:, ..., -1] = 1
        if self.dim == 3:
            on[:, 0, :] = 1
            on[:, -1, :] = 1
        return on.reshape(-1)

    def getGlobalIndicesDirichlet(self):
        return self.indicesManager.getDirichletNodes()

    def getGlobalIndicesNoSlip(self):
        return self.indicesManager.getNoSlipNodes()

    def setBoundaryCondition(self, freeSlipFaces=[], noSlipFaces=[]):
        if len(freeSlipFaces) or len(noSlipFaces):
            for fsFace in freeSlipFaces:
                self.indicesManager.setDirichletNodes(set(self.getBorderNodes(fsFace)))
            for nsFace in noSlipFaces:
                self.indicesManager.setNoSlipNodes(set(self.getBorderNodes(nsFace)))
        else:
            self.indicesManager.setDirichletNodes(self.getBordersNodes())

    def dirichletMaskLocal(self, ndof):
        """uint8 [nLocal*ndof] from the Dirichlet node set (all DOFs of the node imposed,
        base_problem.py:516-520)."""
        nodes = np.fromiter(self.indicesManager.getDirichletNodes(), dtype=np.int64)
        mask = np.zeros((self.nLocal, ndof), dtype=np.uint8)
        if nodes.size:
            loc = self._global2local(nodes)
            mask[loc[loc >= 0]] = 1
        return mask

    def patchPlan(self, tile=(7, 7, 7)):
        """Partition of the OWNED rows into lattice tiles for the tiled device assembly
        (pyn_patch_plan_set): returns (patch_ptr [P+1], patch_rows [nOwned]) in local row ids."""
        dim = self.dim
        if self._unstructured:      # consecutive-row patches in the file's numbering
            chunk = int(np.prod(tile))
            ptr = np.minimum(np.arange(0, self.nOwned + chunk, chunk), self.nOwned).astype(np.int32)
            return ptr, np.arange(self.nOwned, dtype=np.int32)
        a, b = self.part.owned(self.comm.rank)
        shape = list(self.lattice[:-1]) + [b - a]            # owned lattice, x fastest
        idx = np.indices(shape[::-1]).reshape(dim, -1)[::-1]  # idx[d][row], rows in local order
        ntile = [-(-shape[d] // tile[d]) for d in range(dim)]
        tid = np.zeros(idx[0].shape, dtype=np.int64)
        for d in reversed(range(dim)):
            tid = tid * ntile[d] + idx[d] // tile[d]
        order = np.argsort(tid, kind="stable").astype(np.int32)
        counts = np.bincount(tid, minlength=int(np.prod(ntile)))
        ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        return ptr, order

    # ------------------------------------------------------------------ matrix indices
    def _hostGraph(self):
        if self._graph is None:
            self.ctx.csr_symbolic()
            self._graph = self.ctx.csr_get()
        return self._graph

    def getMatIndices(self):
        """Symbolic phase on the GPU (replaces the per-row Python loop of dmplex.py:305-333).
        Returns the reference's tuple; ind_d / ind_o are lazy DeviceGraph views."""
        if self.comm.size == 1:        # every column is on-process: the row offsets are all that is needed
            self.ctx.csr_symbolic()
            rp, _ = self.ctx.csr_get(cols=False)
            return (self.rStart, self.rEnd, np.diff(rp).astype(np.int32), np.zeros(self.nOwned, np.int32),
                    DeviceGraph(self, True), DeviceGraph(self, False))
        rp, ci = self._hostGraph()
        cols = self._local2global(ci)
        inside = (cols >= self.rStart) & (cols < self.rEnd)
        row_of = np.repeat(np.arange(self.nOwned), np.diff(rp))
        alt_d = np.bincount(row_of[inside], minlength=self.nOwned).astype(np.int32)
        alt_o = np.bincount(row_of[~inside], minlength=self.nOwned).astype(np.int32)
        return self.rStart, self.rEnd, alt_d, alt_o, DeviceGraph(self, True), DeviceGraph(self, False)

    # ------------------------------------------------------------------ vec helpers
    def createGlobalVec(self, bs=None):
        from pynama_amd.vectors import Vec
        return Vec(self.ctx, bs or self.dim)

    def _owned_local(self, nodes):
        if isinstance(nodes, range):
            nodes = np.arange(nodes.start, nodes.stop, nodes.step, dtype=np.int64)
        else:
            nodes = np.asarray(list(nodes), dtype=np.int64)
        keep = (nodes >= self.rStart) & (nodes < self.rEnd)
        return nodes[keep], nodes[keep] - self.rStart

    def applyFunctionVecToVec(self, nodes, f_vec, vec, dof):
        gn, ln = self._owned_local(nodes)
        coords = self._coords_of_local(ln)
        values = np.array(list(map(f_vec, coords)), dtype=np.float64).reshape(len(ln), dof)
        inds = (ln[:, None] * dof + np.arange(dof)[None, :]).ravel()
        vec.setValuesLocal(inds, values.ravel(), addv=False)
        return vec

    def applyFunctionScalarToVec(self, nodes, f_scalar, vec):
        gn, ln = self._owned_local(nodes)
        values = np.array(list(map(f_scalar, self._coords_of_local(ln))), dtype=np.float64)
        vec.setValuesLocal(ln, values, addv=False)
        return vec

    def applyValuesToVec(self, nodes, values, vec):
        dof = len(values)
        assert dof <= self.dim
        if dof == 1:
            vec.set(values[0])
        else:
            gn, ln = self._owned_local(nodes)
            inds = (ln[:, None] * dof + np.arange(dof)[None, :]).ravel()
            vec.setValuesLocal(inds, np.tile(np.asarray(values, dtype=np.float64), len(ln)), addv=False)
        return vec

    def view(self):
        kind = "gmsh" if self._unstructured else f"box {self.nelem}"
        return f"DMPlexDom({kind}, ngl={getattr(self, 'ngl', None)}, rank {self.comm.rank}/{self.comm.size})"
