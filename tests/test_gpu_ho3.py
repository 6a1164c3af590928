"""GPU parity of the second-order (ngl = 3) structured path -- the element order of every case of the reference
(src/cases/*.yaml: `ngl: 3`; rules of src/elements/spectral.py:41-43): closed-form graph, row-run assembly kernels
(pynama_amd/csrc/pyn_assemble_ho3.hip) for K, Krhs, Rw and the scalar Laplacian, against the CPU oracle
(oracle/fem_oracle.py), the generic atomics kernel (variant 0) and the reference's own element fixtures (tests/golden/g3_elem.npz).
Everything goes through the C ABI (pynama_amd._lib.Context == include/pynama_hip.h)."""
import os

import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests.util import mat_to_scipy, rel_err, sp_rel_err

pytestmark = pytest.mark.gpu

FP_TOL = 2e-13      # relative; summation order differs from numpy (closed-form blocks, LDS adds)


@pytest.fixture(scope="module")
def lib():
    from pynama_amd import _lib
    assert _lib.device_count() > 0, "GPU tests need an MI355X"
    return _lib


def make_ctx(lib, mesh, mask=None, ndof=None, ngl=3):
    from pynama_amd.elements.spectral import Spectral
    ctx = lib.Context(0)
    ctx.mesh_set(mesh.dim, mesh.conn, mesh.xyz)
    for t in Spectral(ngl, mesh.dim).deviceTables():
        ctx.tables_set(*t)
    if mask is not None:
        ctx.bc_set(ndof, mask)
    ctx.csr_symbolic()
    return ctx


def oracle_kle(mesh, mask, ngl=3):
    """assemble_kle_freeslip with a per-DOF mask [n_node, dim] (the oracle's own routine takes node sets)"""
    import scipy.sparse as sp
    dim = mesh.dim
    tb = fo.Tables(ngl, dim)
    dw = tb.dim_w
    Ke, Rwe, _ = fo.elem_kle_matrices(tb, mesh.corners())
    n = mesh.n_node
    vdof, wdof = fo.dof_indices(mesh.conn, dim), fo.dof_indices(mesh.conn, dw)
    is_bc = np.asarray(mask, bool).reshape(-1)
    rfree, cbc = ~is_bc[vdof], is_bc[vdof]
    R = np.broadcast_to(vdof[:, :, None], Ke.shape)
    C = np.broadcast_to(vdof[:, None, :], Ke.shape)
    mff = rfree[:, :, None] & rfree[:, None, :]
    mfb = rfree[:, :, None] & cbc[:, None, :]
    K = fo._scatter((n * dim, n * dim), R[mff], C[mff], Ke[mff])
    Krhs = fo._scatter((n * dim, n * dim), R[mfb], C[mfb], -Ke[mfb])
    bc_idx = np.nonzero(is_bc)[0]
    ident = sp.coo_matrix((np.ones(len(bc_idx)), (bc_idx, bc_idx)), shape=K.shape).tocsr()
    Rr = np.broadcast_to(vdof[:, :, None], Rwe.shape)
    Rc = np.broadcast_to(wdof[:, None, :], Rwe.shape)
    mrow = np.broadcast_to(rfree[:, :, None], Rwe.shape)
    return {"K": (K + ident).tocsr(), "Krhs": (Krhs + ident).tocsr(), "Rw": fo._scatter((n * dim, n * dw), Rr[mrow], Rc[mrow], Rwe[mrow])}


def boundary_mask(mesh):
    m = np.zeros((mesh.n_node, mesh.dim), np.uint8)
    m[mesh.boundary] = 1
    return m


@pytest.mark.parametrize("nelem", [[3, 4], [19, 5], [1, 1], [2, 3, 2], [5, 1, 2], [1, 1, 1]])
def test_topology_and_closed_form_graph(lib, nelem):
    dim = len(nelem)
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, 3)
    ctx = make_ctx(lib, mesh)
    lat = [2 * n + 1 for n in nelem]
    assert ctx.mesh_topology() == ("lattice-ngl3", lat[0], lat[1], lat[2] if dim == 3 else 1)
    rp, ci = ctx.csr_get()
    rp_o, ci_o = fo.node_graph(mesh)
    assert np.array_equal(rp, rp_o) and np.array_equal(ci, ci_o)          # bit exact, integer work
    os.environ["PYNAMA_NO_HO3_SYMBOLIC"] = "1"                            # ... and equal to the sort-based graph
    try:
        ctx.csr_symbolic()
    finally:
        del os.environ["PYNAMA_NO_HO3_SYMBOLIC"]
    rp_s, ci_s = ctx.csr_get()
    assert np.array_equal(rp, rp_s) and np.array_equal(ci, ci_s)
    ctx.close()


@pytest.mark.parametrize("nelem,upper", [([3, 4], [1.0, 0.8]), ([19, 5], [2.0, 0.5]), ([40, 33], [1.0, 1.0]), ([1, 1], [1.0, 1.0]),
                                         ([2, 3, 2], [1.0, 0.8, 1.2]), ([5, 3, 2], [1.0, 1.0, 1.0]), ([1, 1, 1], [1.0, 1.0, 1.0]),
                                         ([6, 5, 4], [1.0, 0.8, 1.2])])
def test_kle_vs_oracle_and_generic(lib, nelem, upper):
    """K, Krhs, Rw of FreeSlip.buildKLEMats (External Boundary imposed): row-run kernels == oracle == generic atomics kernel"""
    dim = len(nelem)
    dw = 1 if dim == 2 else 3
    mesh = fo.box_mesh(nelem, [0.0] * dim, upper, 3)
    ctx = make_ctx(lib, mesh, boundary_mask(mesh), dim)
    K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1, variant=1)
    ref = fo.assemble_kle_freeslip(mesh, fo.Tables(3, dim))
    got = {"K": mat_to_scipy(ctx, K, dim, dim), "Krhs": mat_to_scipy(ctx, Krhs, dim, dim), "Rw": mat_to_scipy(ctx, Rw, dim, dw)}
    for k in ("K", "Krhs", "Rw"):
        assert sp_rel_err(got[k], ref[k]) < FP_TOL, k
    K0, Kr0, Rw0 = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
    ctx.assemble_kle(1e3, 1e2, K0, Kr0, Rw0, -1, variant=0)
    for a, b, br, bc in ((K, K0, dim, dim), (Krhs, Kr0, dim, dim), (Rw, Rw0, dim, dw)):
        assert rel_err(ctx.mat_values(a, br, bc), ctx.mat_values(b, br, bc)) < FP_TOL       # entry by entry, storage order
    ctx.close()


@pytest.mark.parametrize("nelem,run", [([19, 5], 16), ([70, 3], 64), ([5, 3, 2], 2), ([9, 2, 2], 8)])
def test_run_lengths(lib, nelem, run):
    """every run length of the row-run kernel gives the oracle's matrices (partial last runs included)"""
    dim = len(nelem)
    dw = 1 if dim == 2 else 3
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, 3)
    ctx = make_ctx(lib, mesh, boundary_mask(mesh), dim)
    K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
    os.environ["PYNAMA_HO3_RUN"] = str(run)
    try:
        ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
    finally:
        del os.environ["PYNAMA_HO3_RUN"]
    ref = fo.assemble_kle_freeslip(mesh, fo.Tables(3, dim))
    assert sp_rel_err(mat_to_scipy(ctx, K, dim, dim), ref["K"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Krhs, dim, dim), ref["Krhs"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Rw, dim, dw), ref["Rw"]) < FP_TOL
    ctx.close()


@pytest.mark.parametrize("nelem,ngl,grid", [([19, 6], 3, 1), ([19, 6], 3, 3), ([5, 4, 3], 3, 2), ([5, 4, 3], 3, 7), ([11, 9], 2, 2)])
def test_workgroups_walk_many_runs(lib, nelem, ngl, grid):
    """the row-run kernel is persistent: a workgroup handles runs w, w + grid, ... with the next run's row offsets, Dirichlet bits and
    geometry requested one run ahead.  PYNAMA_HO3_GRID forces 1 / 2 / 3 / 7 workgroups per launch, i.e. tens of runs each (odd and even
    counts, so both LDS sets end a launch), with imposed DOFs scattered inside the mesh: the matrices are the oracle's"""
    dim = len(nelem)
    dw = 1 if dim == 2 else 3
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.7, 1.3][:dim], ngl)
    mask = boundary_mask(mesh)
    rng = np.random.default_rng(17)
    mask[rng.choice(mesh.n_node, max(3, mesh.n_node // 40), replace=False), rng.integers(0, dim)] = 1
    ctx = make_ctx(lib, mesh, boundary_mask(mesh), dim, ngl=ngl)
    ctx.bc_set(dim, mask)
    K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create_rhs(dim, dim), ctx.mat_create(dim, dw)
    os.environ["PYNAMA_HO3_GRID"] = str(grid)
    os.environ["PYNAMA_HO3_REQUIRE"] = "1"
    try:
        ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
        ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
    finally:
        del os.environ["PYNAMA_HO3_GRID"], os.environ["PYNAMA_HO3_REQUIRE"]
    ref = oracle_kle(mesh, mask, ngl=ngl)
    assert sp_rel_err(mat_to_scipy(ctx, K, dim, dim), ref["K"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Krhs, dim, dim), ref["Krhs"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Rw, dim, dw), ref["Rw"]) < FP_TOL
    ctx.close()


@pytest.mark.parametrize("nelem", [[9, 7], [4, 3, 3]])
@pytest.mark.parametrize("kind", ["none", "per_dof_random", "interior_nodes"])
def test_dirichlet_routing(lib, nelem, kind):
    """no mask, a random PER-DOF mask (imposed DOFs anywhere, also inside), imposed interior nodes: K / Krhs / unit diagonal"""
    dim = len(nelem)
    dw = 1 if dim == 2 else 3
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.9, 1.1][:dim], 3)
    rng = np.random.default_rng(7)
    if kind == "none":
        mask = None
    elif kind == "per_dof_random":
        mask = (rng.uniform(size=(mesh.n_node, dim)) < 0.15).astype(np.uint8)
    else:
        mask = np.zeros((mesh.n_node, dim), np.uint8)
        mask[rng.choice(mesh.n_node, mesh.n_node // 10, replace=False)] = 1
    ctx = make_ctx(lib, mesh, mask, dim)
    K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
    ref = oracle_kle(mesh, mask if mask is not None else np.zeros((mesh.n_node, dim), np.uint8))
    assert sp_rel_err(mat_to_scipy(ctx, K, dim, dim), ref["K"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Krhs, dim, dim), ref["Krhs"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Rw, dim, dw), ref["Rw"]) < FP_TOL
    ctx.close()


@pytest.mark.parametrize("dim", [2, 3])
def test_sheared_mesh_and_nonaffine_fallback(lib, dim):
    """an affine image of the box keeps every cell a parallelogram / parallelepiped (row-run kernels, full J^-1); moving ONE
    vertex makes its cells non-affine and the whole assembly falls back to the generic quadrature kernel -- both equal the oracle"""
    nelem = [5, 4] if dim == 2 else [3, 2, 3]
    dw = 1 if dim == 2 else 3
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, 3)
    A = np.eye(dim) + 0.25 * np.random.default_rng(3).standard_normal((dim, dim))
    assert np.linalg.det(A) > 0
    mesh.xyz = mesh.xyz @ A.T + 0.3
    for bend in (False, True):
        if bend:
            corner_nodes = np.unique(mesh.conn[:, :2 ** dim])
            inner = np.setdiff1d(corner_nodes, mesh.boundary)
            mesh.xyz[inner[len(inner) // 2]] += 0.02
        ctx = make_ctx(lib, mesh, boundary_mask(mesh), dim)
        K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
        ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
        ref = fo.assemble_kle_freeslip(mesh, fo.Tables(3, dim))
        assert sp_rel_err(mat_to_scipy(ctx, K, dim, dim), ref["K"]) < FP_TOL, bend
        assert sp_rel_err(mat_to_scipy(ctx, Krhs, dim, dim), ref["Krhs"]) < FP_TOL, bend
        assert sp_rel_err(mat_to_scipy(ctx, Rw, dim, dw), ref["Rw"]) < FP_TOL, bend
        ctx.close()


@pytest.mark.parametrize("nelem", [[7, 5], [3, 4, 3]])
@pytest.mark.parametrize("switch", ["PYNAMA_HO3_NO_DIAG", "PYNAMA_HO3_NO_PSTD", "PYNAMA_HO3_DENSE_TABLES"])
def test_general_forms_on_box_meshes(lib, nelem, switch):
    """what an axis-aligned box mesh normally skips -- the dense J^-1 contraction, plane ids read from memory, table records per node
    pair instead of the 1-D factors -- gives the same matrices as the default (diagonal forms, closed-form plane ids, 1-D factors)"""
    dim = len(nelem)
    dw = 1 if dim == 2 else 3
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.6, 1.4][:dim], 3)
    ctx = make_ctx(lib, mesh, boundary_mask(mesh), dim)
    K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create_rhs(dim, dim), ctx.mat_create(dim, dw)
    K1, Kr1, Rw1 = ctx.mat_create(dim, dim), ctx.mat_create_rhs(dim, dim), ctx.mat_create(dim, dw)
    os.environ["PYNAMA_HO3_REQUIRE"] = "1"
    try:
        ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
        os.environ[switch] = "1"
        ctx.assemble_kle(1e3, 1e2, K1, Kr1, Rw1, -1)
    finally:
        os.environ.pop(switch, None)
        del os.environ["PYNAMA_HO3_REQUIRE"]
    ref = fo.assemble_kle_freeslip(mesh, fo.Tables(3, dim))
    for a, b, name, bc in ((K, K1, "K", dim), (Krhs, Kr1, "Krhs", dim), (Rw, Rw1, "Rw", dw)):
        assert sp_rel_err(mat_to_scipy(ctx, a, dim, bc), ref[name]) < FP_TOL, name
        assert sp_rel_err(mat_to_scipy(ctx, b, dim, bc), ref[name]) < FP_TOL, name
    ctx.close()


def test_run_flags_follow_the_dirichlet_set_and_the_run_length(lib):
    """the per-run flags (which runs see an imposed DOF) are rebuilt when the Dirichlet set or the run length changes"""
    dim, dw = 3, 3
    mesh = fo.box_mesh([5, 3, 3], [0.0] * dim, [1.0] * dim, 3)
    ctx = make_ctx(lib, mesh, boundary_mask(mesh), dim)
    K, Krhs = ctx.mat_create(dim, dim), ctx.mat_create_rhs(dim, dim)
    ref = fo.assemble_kle_freeslip(mesh, fo.Tables(3, dim))
    ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1)
    assert sp_rel_err(mat_to_scipy(ctx, K, dim, dim), ref["K"]) < FP_TOL
    mask = np.zeros((mesh.n_node, dim), np.uint8)                      # another set: a few interior nodes, one component each
    rng = np.random.default_rng(23)
    mask[rng.choice(mesh.n_node, 9, replace=False), rng.integers(0, dim, 9)] = 1
    ctx.bc_set(dim, mask)
    ref2 = oracle_kle(mesh, mask)
    for run in (None, "8", "2", None):
        if run:
            os.environ["PYNAMA_HO3_RUN"] = run
        try:
            ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1)
        finally:
            os.environ.pop("PYNAMA_HO3_RUN", None)
        assert sp_rel_err(mat_to_scipy(ctx, K, dim, dim), ref2["K"]) < FP_TOL, run
        assert sp_rel_err(mat_to_scipy(ctx, Krhs, dim, dim), ref2["Krhs"]) < FP_TOL, run
    ctx.close()


@pytest.mark.parametrize("dim", [2, 3])
def test_single_cell_mesh_reproduces_reference_element(lib, golden, dim):
    """the row-run kernels on a ONE-cell mesh against the reference's own K_e / Rw_e (tests/golden/g3_elem.npz, written by
    src/elements/spectral.py:89-157): the affine fixtures"""
    g = golden["g3_elem"]
    dw = 1 if dim == 2 else 3
    mesh = fo.box_mesh([1] * dim, [0.0] * dim, [1.0] * dim, 3)
    for case in ("unit", "reftest", "brick128", "stretched"):
        key = f"d{dim}_n3_{case}"
        coords = g[key + "_coords"].reshape(2 ** dim, dim)
        mesh.xyz[mesh.conn[0, :2 ** dim]] = coords           # geometry is multilinear from the corners (spectral.py:44, 120)
        ctx = make_ctx(lib, mesh)
        assert ctx.mesh_topology()[0] == "lattice-ngl3"
        K, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
        ctx.assemble_kle(1e3, 1e2, K, -1, Rw, -1)
        # one cell: the global matrices ARE the element matrices, permuted to node order
        perm = np.argsort(mesh.conn[0])
        pk = (perm[:, None] * dim + np.arange(dim)).ravel()
        pw = (perm[:, None] * dw + np.arange(dw)).ravel()
        assert rel_err(mat_to_scipy(ctx, K, dim, dim).toarray(), g[key + "_K"][np.ix_(pk, pk)]) < FP_TOL, case
        assert rel_err(mat_to_scipy(ctx, Rw, dim, dw).toarray(), g[key + "_Rw"][np.ix_(pk, pw)]) < FP_TOL, case
        ctx.close()


@pytest.mark.parametrize("nelem", [[11, 6], [4, 3, 2]])
def test_scalar_laplacian(lib, nelem):
    dim = len(nelem)
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.7, 1.3][:dim], 3)
    mask = np.zeros(mesh.n_node, np.uint8)
    mask[mesh.boundary] = 1
    ctx = make_ctx(lib, mesh, mask, 1)
    A, Ar = ctx.mat_create(1, 1), ctx.mat_create(1, 1)
    ctx.assemble_scalar(lib.FORM_LAPLACE, A, Ar)
    ref = fo.assemble_scalar(mesh, fo.Tables(3, dim), "laplace", dirichlet=mesh.boundary)
    assert sp_rel_err(mat_to_scipy(ctx, A, 1, 1), ref["A"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Ar, 1, 1), ref["Arhs"]) < FP_TOL
    ctx.close()


def test_imposed_column_matrix_across_dirichlet_sets(lib):
    """Krhs is rewritten where a NEW Dirichlet set needs it (and cleared where the old one left values)"""
    nelem = [6, 5]
    mesh = fo.box_mesh(nelem, [0.0, 0.0], [1.0, 1.0], 3)
    ctx = make_ctx(lib, mesh, boundary_mask(mesh), 2)
    K, Krhs = ctx.mat_create(2, 2), ctx.mat_create(2, 2)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1)                    # second call: Krhs known clean for this set
    ref = fo.assemble_kle_freeslip(mesh, fo.Tables(3, 2))
    assert sp_rel_err(mat_to_scipy(ctx, Krhs, 2, 2), ref["Krhs"]) < FP_TOL
    mask = np.zeros((mesh.n_node, 2), np.uint8)
    mask[np.random.default_rng(5).choice(mesh.n_node, 12, replace=False)] = 1
    ctx.bc_set(2, mask)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1)
    ref2 = oracle_kle(mesh, mask)
    assert sp_rel_err(mat_to_scipy(ctx, K, 2, 2), ref2["K"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Krhs, 2, 2), ref2["Krhs"]) < FP_TOL
    ctx.close()


@pytest.mark.parametrize("nelem,size", [([5, 8], 2), ([4, 9], 3), ([3, 2, 6], 2), ([2, 2, 7], 3)])
def test_rank_slabs(lib, nelem, size):
    """a rank's slab (owned planes first, ghost planes with the LAST ids): closed-form graph, K, Krhs, Rw and the product equal the
    owned rows of the serial oracle"""
    from pynama_amd.common.comm import Comm
    from pynama_amd.domain.dmplex import DMPlexDom
    from pynama_amd.elements.spectral import Spectral
    dim = len(nelem)
    dw = 1 if dim == 2 else 3
    lo, up = [0.0] * dim, [1.0, 0.8, 1.2][:dim]
    glob = fo.box_mesh(nelem, lo, up, 3)
    ref = fo.assemble_kle_freeslip(glob, fo.Tables(3, dim))
    rp_g, ci_g = fo.node_graph(glob)
    xg = np.random.default_rng(11).standard_normal(glob.n_node * dim)
    yg = ref["K"] @ xg
    for r in range(size):
        dom = DMPlexDom(boxMesh={'nelem': nelem, 'lower': lo, 'upper': up}, comm=Comm(r, size))
        dom.setFemIndexing(3)
        ctx = lib.Context(0)
        ctx.comm_init(r, size, None)                      # detached
        ctx.halo_set(*dom._halo_plan())
        ctx.mesh_set(dim, dom.conn, dom.xyz)
        for t in Spectral(3, dim).deviceTables():
            ctx.tables_set(*t)
        ctx.bc_set(dim, np.repeat(dom.boundaryMaskLocal()[:, None], dim, axis=1))
        ctx.csr_symbolic()
        assert ctx.mesh_topology()[0] == "lattice-ngl3"
        rp, ci = ctx.csr_get()
        l2g = dom._local2global(np.arange(dom.nLocal))
        assert np.array_equal(np.diff(rp), np.diff(rp_g)[dom.rStart:dom.rEnd])
        for i in range(0, dom.nOwned, max(1, dom.nOwned // 50)):
            gi = dom.rStart + i
            assert np.array_equal(np.sort(l2g[ci[rp[i]:rp[i + 1]]]), ci_g[rp_g[gi]:rp_g[gi + 1]])
            assert np.all(np.diff(ci[rp[i]:rp[i + 1]]) > 0)
        K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
        ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
        rows = (np.arange(dom.rStart, dom.rEnd)[:, None] * dim + np.arange(dim)).ravel()
        cv = (l2g[:, None] * dim + np.arange(dim)).ravel()
        cw = (l2g[:, None] * dw + np.arange(dw)).ravel()
        assert sp_rel_err(mat_to_scipy(ctx, K, dim, dim), ref["K"][rows][:, cv]) < FP_TOL
        assert sp_rel_err(mat_to_scipy(ctx, Krhs, dim, dim), ref["Krhs"][rows][:, cv]) < FP_TOL
        assert sp_rel_err(mat_to_scipy(ctx, Rw, dim, dw), ref["Rw"][rows][:, cw]) < FP_TOL
        vx, vy = ctx.vec_create(dim), ctx.vec_create(dim)
        ctx.vec_set_local(vx, xg[cv])
        ctx.spmv(K, vx, vy)
        assert rel_err(ctx.vec_get(vy, dim), yg[rows]) < 1e-13
        ctx.close()


@pytest.mark.parametrize("dim", [2, 3])
def test_uniform_flow_solve(lib, dim):
    """the reference's analytic assertion (src/tests/test_solver.py:20-27, 52-62: uniform flow is reproduced exactly) on matrices of
    the row-run kernels, at the C ABI: 10 x 10 (2-D) / 3 x 3 x 3 (3-D) cells, ngl 3"""
    nelem = [10, 10] if dim == 2 else [3, 3, 3]
    dw = 1 if dim == 2 else 3
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, 3)
    ctx = make_ctx(lib, mesh, boundary_mask(mesh), dim)
    K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
    cte = np.array([1.0, 4.0, -2.0][:dim])
    vel = np.zeros((mesh.n_node, dim))
    vel[mesh.boundary] = cte
    vv, vr, vx = ctx.vec_create(dim), ctx.vec_create(dim), ctx.vec_create(dim)
    ctx.vec_set(vv, vel.ravel())
    ctx.spmv(Krhs, vv, vr)                                  # rhs = Rw * 0 + Krhs * vel (base_problem.py:481)
    info = ctx.solve_direct(K, vr, vx)                      # the reference's default: -ksp_type preonly -pc_type lu (ksp_solver.py:13-16)
    err = np.linalg.norm(ctx.vec_get(vx, dim) - np.tile(cte, mesh.n_node))
    assert err < (1e-12 if dim == 2 else 2e-13), (err, info.true_resid)
    info = ctx.solve(K, vr, vx, rtol=1e-14, atol=1e-300, dtol=1e8, norm_type=lib.NORM_UNPRECONDITIONED, maxit=200000)
    err = np.linalg.norm(ctx.vec_get(vx, dim) - np.tile(cte, mesh.n_node))
    assert err < 5e-12, (err, info.iters, info.reason)      # Jacobi-PCG to round-off on the same system
    ctx.close()


@pytest.mark.parametrize("nelem", [[9, 7], [4, 3, 3]])
@pytest.mark.parametrize("lanes", [0, 8, 16, 32, 64])
def test_block_csr_product(lib, nelem, lanes):
    """y = A x straight from the block-CSR values (bcsr_spmv_kernel) for every block shape of the path: K (dim x dim), Rw (dim x dim_w), the
    scalar Laplacian with its long rows, the operator shapes SrT / DivSrT / Curl -- against scipy on the matrix read back, for every
    lanes-per-node setting, and equal to the SELL-image product (PYNAMA_BLOCK_SELL=1)"""
    from pynama_amd.elements.spectral import Spectral
    dim = len(nelem)
    dw, ds = (1, 3) if dim == 2 else (3, 6)
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.9, 1.1][:dim], 3)
    ctx = make_ctx(lib, mesh, boundary_mask(mesh), dim)
    K, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
    ctx.assemble_kle(1e3, 1e2, K, -1, Rw, -1)
    mats = [(K, dim, dim), (Rw, dim, dw)]
    ops = Spectral(3, dim).operatorTerms()
    for name in ("SrT", "DivSrT", "Curl"):
        br, bc, terms, coef = ops[name]
        M = ctx.mat_create(br, bc)
        ctx.assemble_operator(lib.Q_NODAL, terms, coef, M)
        mats.append((M, br, bc))
    ctx.bc_set(1, boundary_mask(mesh)[:, 0].copy())
    A = ctx.mat_create(1, 1)
    ctx.assemble_scalar(lib.FORM_LAPLACE, A, -1)
    mats.append((A, 1, 1))
    rng = np.random.default_rng(5)
    if lanes:
        os.environ["PYNAMA_BCSR_LANES"] = str(lanes)
    try:
        for mid, br, bc in mats:
            S = mat_to_scipy(ctx, mid, br, bc)
            xv = rng.standard_normal(mesh.n_node * bc)
            vx, vy = ctx.vec_create(bc), ctx.vec_create(br)
            ctx.vec_set(vx, xv)
            ctx.spmv(mid, vx, vy)
            y = ctx.vec_get(vy, br)
            assert rel_err(y, S @ xv) < 1e-13, (br, bc)
            if not lanes:
                os.environ["PYNAMA_BLOCK_SELL"] = "1"
                try:
                    ctx.spmv(mid, vx, vy)
                finally:
                    del os.environ["PYNAMA_BLOCK_SELL"]
                assert rel_err(ctx.vec_get(vy, br), y) < 1e-13, (br, bc)
                ctx.spmv(mid, vx, vy)                       # and back: no stale image state
                assert rel_err(ctx.vec_get(vy, br), y) < 1e-15, (br, bc)
    finally:
        os.environ.pop("PYNAMA_BCSR_LANES", None)
    ctx.close()


# ---- the same row-run kernels on first-order lattices and for the first-order operators -----------------------------------------------
@pytest.mark.parametrize("nelem", [[9, 7], [70, 3], [1, 1], [33, 40]])
def test_q1_quadrilaterals_take_the_row_run_kernels(lib, nelem):
    """2-D Q1 box meshes (the reference's `ngl: 2` runs, BASELINE configs[0]): closed-form graph, K, Krhs (compact and full), Rw and the
    scalar Laplacian by the row-run kernels == oracle == generic kernel"""
    from pynama_amd.elements.spectral import Spectral
    mesh = fo.box_mesh(nelem, [0.0, 0.0], [1.0, 0.7], 2)
    mask = boundary_mask(mesh)
    ctx = lib.Context(0)
    ctx.mesh_set(2, mesh.conn, mesh.xyz)
    for t in Spectral(2, 2).deviceTables():
        ctx.tables_set(*t)
    ctx.bc_set(2, mask)
    ctx.csr_symbolic()
    assert ctx.mesh_topology() == ("lattice-q1-2d", nelem[0] + 1, nelem[1] + 1, 1)
    rp, ci = ctx.csr_get()
    rp_o, ci_o = fo.node_graph(mesh)
    assert np.array_equal(rp, rp_o) and np.array_equal(ci, ci_o)
    K, Kr, Krc, Rw = ctx.mat_create(2, 2), ctx.mat_create(2, 2), ctx.mat_create_rhs(2, 2), ctx.mat_create(2, 1)
    os.environ["PYNAMA_HO3_REQUIRE"] = "1"
    try:
        ctx.assemble_kle(1e3, 1e2, K, Kr, Rw, -1)
        ctx.assemble_kle(1e3, 1e2, K, Krc, -1, -1)
    finally:
        del os.environ["PYNAMA_HO3_REQUIRE"]
    ref = fo.assemble_kle_freeslip(mesh, fo.Tables(2, 2))
    assert sp_rel_err(mat_to_scipy(ctx, K, 2, 2), ref["K"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Kr, 2, 2), ref["Krhs"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Krc, 2, 2), ref["Krhs"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Rw, 2, 1), ref["Rw"]) < FP_TOL
    K0 = ctx.mat_create(2, 2)
    ctx.assemble_kle(1e3, 1e2, K0, -1, -1, -1, variant=0)
    assert rel_err(ctx.mat_values(K, 2, 2), ctx.mat_values(K0, 2, 2)) < FP_TOL
    ctx.bc_set(1, mask[:, 0].copy())
    A, Ar = ctx.mat_create(1, 1), ctx.mat_create(1, 1)
    ctx.assemble_scalar(lib.FORM_LAPLACE, A, Ar)
    refs = fo.assemble_scalar(mesh, fo.Tables(2, 2), "laplace", dirichlet=mesh.boundary)
    assert sp_rel_err(mat_to_scipy(ctx, A, 1, 1), refs["A"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Ar, 1, 1), refs["Arhs"]) < FP_TOL
    ctx.close()


@pytest.mark.parametrize("nelem,ngl", [([9, 7], 2), ([6, 5, 4], 2), ([9, 7], 3), ([3, 2, 3], 3), ([1, 1], 3), ([1, 1, 1], 2), ([40, 3], 3)])
def test_operators_on_lattices(lib, nelem, ngl):
    """SrT / DivSrT / Curl of Spectral.getElemKLEOperators + Operators.setValues (spectral.py:159-218, mat_generator.py:157-190) by the
    row-run kernels on a sheared box (full J^-1): == oracle (after the lumped-weight row scaling) == generic kernel entry by entry"""
    from pynama_amd.elements.spectral import Spectral
    dim = len(nelem)
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.8, 1.2][:dim], ngl)
    Aff = np.eye(dim) + 0.2 * np.random.default_rng(8).standard_normal((dim, dim))
    assert np.linalg.det(Aff) > 0
    mesh.xyz = mesh.xyz @ Aff.T
    ref = fo.assemble_operators(mesh, fo.Tables(ngl, dim))
    ctx = lib.Context(0)
    ctx.mesh_set(dim, mesh.conn, mesh.xyz)
    sp = Spectral(ngl, dim)
    for t in sp.deviceTables():
        ctx.tables_set(*t)
    ctx.csr_symbolic()
    ops = sp.operatorTerms()
    mass = ctx.mat_create(1, 1)
    ctx.assemble_scalar(lib.FORM_MASS_NODAL, mass, -1, 0)
    vw = ctx.vec_create(1)
    ctx.mat_diagonal(mass, vw)
    w = ctx.vec_get(vw, 1)
    for name in ("SrT", "DivSrT", "Curl"):
        br, bc, terms, coef = ops[name]
        m, m0 = ctx.mat_create(br, bc), ctx.mat_create(br, bc)
        ctx.assemble_operator(lib.Q_NODAL, terms, coef, m)
        t_fast = ctx.timers()["assemble_ms"]
        os.environ["PYNAMA_NO_HO3_OPERATOR"] = "1"
        try:
            ctx.assemble_operator(lib.Q_NODAL, terms, coef, m0)
        finally:
            del os.environ["PYNAMA_NO_HO3_OPERATOR"]
        assert rel_err(ctx.mat_values(m, br, bc), ctx.mat_values(m0, br, bc)) < FP_TOL, (name, t_fast)
        vs = ctx.vec_create(br)
        ctx.vec_set(vs, np.repeat(1.0 / w, br))
        ctx.mat_row_scale(m, vs)
        assert sp_rel_err(mat_to_scipy(ctx, m, br, bc), ref[name]) < FP_TOL, name
    ctx.close()


@pytest.mark.parametrize("nelem,ngl", [([7, 6, 5], 2), ([2, 2, 2], 2), ([9, 7], 2), ([9, 7], 3), ([1, 1], 3), ([1, 3, 2], 2)])
def test_closed_form_pattern_dictionary(lib, nelem, ngl):
    """the column-pattern dictionary of a single-rank lattice written in closed form (27 / 9 / 16 patterns) gives the products of the
    hash-based construction (PYNAMA_NO_LATTICE_PATTERNS=1) and of scipy, for the scalar and the block matrices that use it"""
    from pynama_amd.elements.spectral import Spectral
    dim = len(nelem)
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, ngl)
    x1 = np.random.default_rng(3).standard_normal(mesh.n_node)
    xd = np.random.default_rng(4).standard_normal(mesh.n_node * dim)
    res = {}
    for mode in ("closed", "hash"):
        if mode == "hash":
            os.environ["PYNAMA_NO_LATTICE_PATTERNS"] = "1"
        try:
            ctx = lib.Context(0)
            ctx.mesh_set(dim, mesh.conn, mesh.xyz)
            for t in Spectral(ngl, dim).deviceTables():
                ctx.tables_set(*t)
            ctx.csr_symbolic()
            A, K = ctx.mat_create(1, 1), ctx.mat_create(dim, dim)
            ctx.assemble_scalar(lib.FORM_LAPLACE, A, -1)
            ctx.assemble_kle(1e3, 1e2, K, -1, -1, -1)
            v1, w1, vd, wd = ctx.vec_create(1), ctx.vec_create(1), ctx.vec_create(dim), ctx.vec_create(dim)
            ctx.vec_set(v1, x1)
            ctx.vec_set(vd, xd)
            ctx.spmv(A, v1, w1)
            ctx.spmv(K, vd, wd)
            info = ctx.solve(K, vd, wd, fixed_iters=3)          # the solver's product kernels (image / LDS-staged runs)
            res[mode] = (ctx.vec_get(w1, 1), ctx.vec_get(wd, dim), info.rnorm)
            if mode == "closed":
                SA, SK = mat_to_scipy(ctx, A, 1, 1), mat_to_scipy(ctx, K, dim, dim)
                assert rel_err(res[mode][0], SA @ x1) < 1e-13
            ctx.close()
        finally:
            os.environ.pop("PYNAMA_NO_LATTICE_PATTERNS", None)
    assert rel_err(res["closed"][0], res["hash"][0]) < 1e-14
    assert rel_err(res["closed"][1], res["hash"][1]) < 1e-13
    assert abs(res["closed"][2] - res["hash"][2]) <= 1e-12 * abs(res["hash"][2])
