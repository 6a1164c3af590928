"""Compact imposed-column matrices (pyn_mat_create_rhs): Krhs / Krhsfs / Arhs store only the node rows with an imposed node in their
neighbourhood -- the preallocation the reference makes for Krhs (src/matrices/mat_generator.py:42-58, 91: `drhs_nnz`).  Every kernel
family that can be handed one (natively: generic atomics, ngl = 3 row-run, KLE lattice; the others through the imposed-node element
pass), the product Krhs v of solveKLE (src/cases/base_problem.py:481), host insertion, a change of the Dirichlet set, rank slabs."""
import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests.util import mat_to_scipy, rel_err, sp_rel_err

pytestmark = pytest.mark.gpu
FP_TOL = 2e-13


@pytest.fixture(scope="module")
def lib():
    from pynama_amd import _lib
    assert _lib.device_count() > 0, "GPU tests need an MI355X"
    return _lib


def make_ctx(lib, mesh, ngl, mask, ndof):
    from pynama_amd.elements.spectral import Spectral
    ctx = lib.Context(0)
    ctx.mesh_set(mesh.dim, mesh.conn, mesh.xyz)
    for t in Spectral(ngl, mesh.dim).deviceTables():
        ctx.tables_set(*t)
    ctx.bc_set(ndof, mask)
    ctx.csr_symbolic()
    return ctx


def stored_rows_expected(mesh, node_imposed):
    """node rows with an imposed node among their columns (themselves included)"""
    rp, ci = fo.node_graph(mesh)
    hit = np.add.reduceat(node_imposed[ci].astype(np.int64), rp[:-1]) > 0
    return int(hit.sum()), int(np.diff(rp)[hit].sum())


@pytest.mark.parametrize("nelem,ngl,jitter,variant,plan", [
    ([8, 7, 6], 2, 0.0, 1, False),      # KLE lattice kernels, closed-form blocks (native)
    ([8, 7, 6], 2, 0.2, 1, False),      # ... general geometry (native)
    ([8, 7, 6], 2, 0.2, 1, True),       # patch-plan KLE kernels: K alone, Krhs from the elements that hold an imposed node
    ([8, 7, 6], 2, 0.2, 0, False),      # generic atomics kernel (native)
    ([9, 7], 2, 0.2, 1, False),         # 2-D Q1: generic kernel
    ([9, 7], 3, 0.0, 1, False),         # ngl 3 row-run kernels (native)
    ([4, 3, 3], 3, 0.0, 1, False),
    ([2, 2], 6, 0.0, 1, False),         # high order through the generic / matrix-core kernels
    ([2, 2, 2], 4, 0.0, 1, False)])
def test_kle_compact_krhs(lib, nelem, ngl, jitter, variant, plan):
    dim = len(nelem)
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.8, 1.2][:dim], ngl, jitter=jitter)
    mask = np.zeros((mesh.n_node, dim), np.uint8)
    mask[mesh.boundary] = 1
    ctx = make_ctx(lib, mesh, ngl, mask, dim)
    if plan:
        from tests.test_gpu_kernels import tile_plan
        ctx.patch_plan_set(*tile_plan(mesh, (4, 3, 3)), kind=1)
    K, Krhs, Kfull = ctx.mat_create(dim, dim), ctx.mat_create_rhs(dim, dim), ctx.mat_create(dim, dim)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1, variant=variant)
    ref = fo.assemble_kle_freeslip(mesh, fo.Tables(ngl, dim))
    assert sp_rel_err(mat_to_scipy(ctx, K, dim, dim), ref["K"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Krhs, dim, dim), ref["Krhs"]) < FP_TOL      # pyn_mat_get_values returns the graph's layout
    node_imposed = np.zeros(mesh.n_node, bool)
    node_imposed[mesh.boundary] = True
    nrows, nblocks = stored_rows_expected(mesh, node_imposed)
    assert ctx.mat_stored(Krhs) == (nblocks, nrows) and ctx.mat_stored(K) == (ctx.nnzb, mesh.n_node)
    # the product of solveKLE: rhs = Krhs v (rows that are not stored are zero rows)
    v = np.random.default_rng(1).standard_normal(mesh.n_node * dim)
    vx, vy = ctx.vec_create(dim), ctx.vec_create(dim)
    ctx.vec_set(vx, v)
    ctx.vec_fill(vy, 7.0)
    ctx.spmv(Krhs, vx, vy)
    assert rel_err(ctx.vec_get(vy, dim), ref["Krhs"] @ v) < 1e-13
    # a second assembly into the same compact matrix, and the same values as a matrix with the graph's full pattern
    ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1, variant=variant)
    ctx.assemble_kle(1e3, 1e2, K, Kfull, -1, -1, variant=variant)
    assert rel_err(ctx.mat_values(Krhs, dim, dim), ctx.mat_values(Kfull, dim, dim)) < FP_TOL
    with pytest.raises(lib.PynamaHipError):
        ctx.solve(Krhs, vx, vy)
    with pytest.raises(lib.PynamaHipError):
        ctx.mat_axpy(K, 1.0, Krhs)
    ctx.close()


def test_new_dirichlet_set_lays_the_matrix_out_again(lib):
    nelem = [7, 6, 5]
    mesh = fo.box_mesh(nelem, [0, 0, 0], [1, 1, 1], 2, jitter=0.2)
    mask = np.zeros((mesh.n_node, 3), np.uint8)
    mask[mesh.boundary] = 1
    ctx = make_ctx(lib, mesh, 2, mask, 3)
    K, Krhs = ctx.mat_create(3, 3), ctx.mat_create_rhs(3, 3)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1)
    first = ctx.mat_stored(Krhs)
    rng = np.random.default_rng(4)
    mask2 = np.zeros((mesh.n_node, 3), np.uint8)
    mask2[rng.choice(mesh.n_node, 9, replace=False)] = 1            # a handful of imposed nodes anywhere: a much smaller matrix
    mask2[rng.choice(mesh.n_node, 5, replace=False), 1] = 1          # ... and single imposed DOFs
    ctx.bc_set(3, mask2)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1)
    import scipy.sparse as sp
    tb = fo.Tables(2, 3)
    Ke, _, _ = fo.elem_kle_matrices(tb, mesh.corners())
    vdof = fo.dof_indices(mesh.conn, 3)
    is_bc = mask2.astype(bool).reshape(-1)
    rfree, cbc = ~is_bc[vdof], is_bc[vdof]
    R = np.broadcast_to(vdof[:, :, None], Ke.shape)
    C = np.broadcast_to(vdof[:, None, :], Ke.shape)
    mfb = rfree[:, :, None] & cbc[:, None, :]
    ref = fo._scatter((mesh.n_node * 3,) * 2, R[mfb], C[mfb], -Ke[mfb])
    idx = np.nonzero(is_bc)[0]
    ref = (ref + sp.coo_matrix((np.ones(len(idx)), (idx, idx)), shape=ref.shape)).tocsr()
    assert sp_rel_err(mat_to_scipy(ctx, Krhs, 3, 3), ref) < FP_TOL
    second = ctx.mat_stored(Krhs)
    assert second[1] < first[1] and second == stored_rows_expected(mesh, mask2.any(axis=1))[::-1]
    ctx.close()


@pytest.mark.parametrize("nelem,jitter,variant", [([8, 7, 6], 0.0, 1), ([8, 7, 6], 0.2, 1), ([8, 7, 6], 0.2, 0), ([9, 8], 0.2, 1)])
def test_scalar_compact_arhs(lib, nelem, jitter, variant):
    """the scalar lattice / z-marching kernels assemble A alone, Arhs comes from the elements that hold an imposed node"""
    dim = len(nelem)
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, 2, jitter=jitter)
    mask = np.zeros(mesh.n_node, np.uint8)
    mask[mesh.boundary] = 1
    ctx = make_ctx(lib, mesh, 2, mask, 1)
    A, Ar = ctx.mat_create(1, 1), ctx.mat_create_rhs(1, 1)
    ctx.assemble_scalar(lib.FORM_LAPLACE, A, Ar, variant=variant)
    ref = fo.assemble_scalar(mesh, fo.Tables(2, dim), "laplace", dirichlet=mesh.boundary)
    assert sp_rel_err(mat_to_scipy(ctx, A, 1, 1), ref["A"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Ar, 1, 1), ref["Arhs"]) < FP_TOL
    v = np.random.default_rng(2).standard_normal(mesh.n_node)
    vx, vy = ctx.vec_create(1), ctx.vec_create(1)
    ctx.vec_set(vx, v)
    ctx.spmv(Ar, vx, vy)
    assert rel_err(ctx.vec_get(vy, 1), ref["Arhs"] @ v) < 1e-13
    ctx.close()


def test_noslip_split_with_compact_krhs_and_krhsfs(lib):
    nelem = [7, 6, 6]
    mesh = fo.box_mesh(nelem, [0, 0, 0], [1, 1, 1], 2)
    cls = fo.noslip_classes(mesh, ["down", "up"], ["left", "right", "back", "front"])
    ctx = make_ctx(lib, mesh, 2, cls.astype(np.uint8), 3)
    ids = [ctx.mat_create(3, 3), ctx.mat_create_rhs(3, 3), ctx.mat_create(3, 3), ctx.mat_create(3, 1),
           ctx.mat_create(3, 3), ctx.mat_create_rhs(3, 3), ctx.mat_create(3, 3), ctx.mat_create(3, 1)]
    ctx.assemble_kle_noslip(1e3, 1e2, ids)
    ref = fo.assemble_kle_noslip(mesh, fo.Tables(2, 3), cls)
    for k, (name, bc) in enumerate((("K", 3), ("Krhs", 3), ("Rw", 3), ("Rd", 1), ("Kfs", 3), ("Krhsfs", 3), ("Rwfs", 3), ("Rdfs", 1))):
        assert sp_rel_err(mat_to_scipy(ctx, ids[k], 3, bc), ref[name]) < FP_TOL, name
    assert ctx.mat_stored(ids[1])[1] < mesh.n_node and ctx.mat_stored(ids[5])[1] < mesh.n_node
    ctx.close()


def test_host_insertion_into_a_compact_matrix(lib):
    """Mat.setValues on Krhs (the reference's per-cell loop, base_problem.py:531-533): entries next to imposed nodes land, a nonzero
    entry in a row that is not stored is refused"""
    nelem = [6, 5]
    mesh = fo.box_mesh(nelem, [0, 0], [1, 1], 2)
    mask = np.zeros((mesh.n_node, 2), np.uint8)
    mask[mesh.boundary] = 1
    ctx = make_ctx(lib, mesh, 2, mask, 2)
    Krhs = ctx.mat_create_rhs(2, 2)
    tb = fo.Tables(2, 2)
    Ke, _, _ = fo.elem_kle_matrices(tb, mesh.corners())
    vdof = fo.dof_indices(mesh.conn, 2)
    is_bc = mask.astype(bool).reshape(-1)
    for e in range(mesh.n_elem):
        free = [i for i, d in enumerate(vdof[e]) if not is_bc[d]]
        bc = [i for i, d in enumerate(vdof[e]) if is_bc[d]]
        if free and bc:
            ctx.mat_add_values(Krhs, vdof[e][free], vdof[e][bc], -Ke[e][np.ix_(free, bc)])
    for d in np.nonzero(is_bc)[0]:
        ctx.mat_add_values(Krhs, [d], [d], [1.0])
    ref = fo.assemble_kle_freeslip(mesh, tb)["Krhs"]
    assert sp_rel_err(mat_to_scipy(ctx, Krhs, 2, 2), ref) < FP_TOL
    inner = np.setdiff1d(np.arange(mesh.n_node), mesh.boundary)
    far = [n for n in inner if not np.isin(mesh.conn[np.any(mesh.conn == n, axis=1)], mesh.boundary).any()]
    assert far, "mesh too small for a row away from the boundary"
    with pytest.raises(lib.PynamaHipError):
        ctx.mat_add_values(Krhs, [far[0] * 2], [far[0] * 2], [3.0])
    ctx.close()


@pytest.mark.parametrize("ngl,nelem", [(2, [6, 5, 9]), (3, [3, 2, 6])])
def test_rank_slabs_with_compact_krhs(lib, ngl, nelem):
    from pynama_amd.common.comm import Comm
    from pynama_amd.domain.dmplex import DMPlexDom
    from pynama_amd.elements.spectral import Spectral
    size = 2
    glob = fo.box_mesh(nelem, [0, 0, 0], [1, 1, 1], ngl)
    ref = fo.assemble_kle_freeslip(glob, fo.Tables(ngl, 3))
    xg = np.random.default_rng(11).standard_normal(glob.n_node * 3)
    yg = ref["Krhs"] @ xg
    for r in range(size):
        dom = DMPlexDom(boxMesh={'nelem': nelem, 'lower': [0, 0, 0], 'upper': [1, 1, 1]}, comm=Comm(r, size))
        dom.setFemIndexing(ngl)
        ctx = lib.Context(0)
        ctx.comm_init(r, size, None)
        ctx.halo_set(*dom._halo_plan())
        ctx.mesh_set(3, dom.conn, dom.xyz)
        for t in Spectral(ngl, 3).deviceTables():
            ctx.tables_set(*t)
        ctx.bc_set(3, np.repeat(dom.boundaryMaskLocal()[:, None], 3, axis=1))
        ctx.csr_symbolic()
        K, Krhs = ctx.mat_create(3, 3), ctx.mat_create_rhs(3, 3)
        ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1)
        l2g = dom._local2global(np.arange(dom.nLocal))
        rows = (np.arange(dom.rStart, dom.rEnd)[:, None] * 3 + np.arange(3)).ravel()
        cv = (l2g[:, None] * 3 + np.arange(3)).ravel()
        assert sp_rel_err(mat_to_scipy(ctx, Krhs, 3, 3), ref["Krhs"][rows][:, cv]) < FP_TOL
        vx, vy = ctx.vec_create(3), ctx.vec_create(3)
        ctx.vec_set_local(vx, xg[cv])
        ctx.spmv(Krhs, vx, vy)
        assert rel_err(ctx.vec_get(vy, 3), yg[rows]) < 1e-13
        ctx.close()
