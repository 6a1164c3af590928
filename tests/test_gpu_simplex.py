"""Linear triangles / tetrahedra on the device (BASELINE.json configs[4]: "unstructured tetrahedral mesh, Gmsh
import, GMRES+Jacobi -- irregular indexing stress").  No reference counterpart exists (the reference is
tensor-product only); the oracle's P1 tables are pinned by closed forms in tests/test_simplex_host.py."""
import os

import numpy as np
import pytest
import yaml

import pynama_amd
from oracle import fem_oracle as fo
from tests.util import mat_to_scipy, rel_err, sp_rel_err

pytestmark = pytest.mark.gpu
FP_TOL = 2e-13      # relative, FP64 with atomics (summation order differs from numpy)
CASES = os.path.join(os.path.dirname(pynama_amd.__file__), "cases")


@pytest.fixture(scope="module")
def lib():
    from pynama_amd import _lib
    if _lib.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path is the product, there is no CPU fallback")
    return _lib


def _ctx(lib, mesh, bc_ndof=None):
    from pynama_amd.elements.simplex import Simplex
    ctx = lib.Context(0)
    ctx.mesh_set(mesh.dim, mesh.conn, mesh.xyz)
    for t in Simplex(mesh.dim).deviceTables():
        ctx.tables_set(*t)
    if bc_ndof:
        mask = np.zeros((mesh.n_node, bc_ndof), np.uint8)
        mask[mesh.boundary] = 1
        ctx.bc_set(bc_ndof, mask)
    ctx.csr_symbolic()
    return ctx


@pytest.mark.parametrize("dim", [2, 3])
def test_simplex_element_entry_points(lib, dim):
    from pynama_amd.elements.simplex import Simplex
    el, tb = Simplex(dim), fo.SimplexTables(dim)
    m = fo.simplex_box_mesh([2, 3, 2][:dim], [0.0] * dim, [1.0, 0.7, 1.9][:dim], jitter=0.25)
    for e in (0, 5, m.n_elem - 1):
        c = m.corners()[e]
        assert rel_err(el.getElemLaplace(c.copy()), fo.elem_laplace(tb, c)[0]) < FP_TOL
        assert rel_err(el.getElemMass(c.copy(), nodal=False), fo.elem_mass(tb, c, rule="full")[0]) < FP_TOL
        assert rel_err(el.getElemMass(c.copy()), fo.elem_mass(tb, c)[0]) < FP_TOL
        K, Rw, Rd = el.getElemKLEMatrices(c.copy())
        Ko, Rwo, Rdo = fo.elem_kle_matrices(tb, c)
        assert rel_err(K, Ko[0]) < FP_TOL and rel_err(Rw, Rwo[0]) < FP_TOL and rel_err(Rd, Rdo[0]) < FP_TOL


@pytest.mark.parametrize("dim,nelem", [(2, [9, 7]), (3, [7, 6, 5])])
def test_simplex_assembly_and_krylov_vs_oracle(lib, dim, nelem):
    """randomly numbered simplicial mesh: graph, scalar + KLE assembly with elimination, SpMV, and the
    GMRES(30)+Jacobi / CG+Jacobi iterates all equal the oracle's"""
    mesh = fo.simplex_box_mesh(nelem, [0.0] * dim, [1.0] * dim, jitter=0.2, permute_seed=7)
    tb = fo.SimplexTables(dim)
    ctx = _ctx(lib, mesh, bc_ndof=1)
    rp, ci = ctx.csr_get()
    rp_o, ci_o = fo.node_graph(mesh)
    assert np.array_equal(rp, rp_o) and np.array_equal(ci, ci_o)
    A, Ar = ctx.mat_create(1, 1), ctx.mat_create(1, 1)
    ctx.assemble_scalar(lib.FORM_LAPLACE, A, Ar)
    ref = fo.assemble_scalar(mesh, tb, "laplace", dirichlet=mesh.boundary)
    S = mat_to_scipy(ctx, A, 1, 1)
    assert sp_rel_err(S, ref["A"]) < FP_TOL and sp_rel_err(mat_to_scipy(ctx, Ar, 1, 1), ref["Arhs"]) < FP_TOL
    # three independent device paths: LDS patch kernel (3-D default), one-lane-per-cell atomics kernel, and the
    # table-driven generic kernel
    for env in (("PYNAMA_NO_P1_TILED",), ("PYNAMA_NO_P1_TILED", "PYNAMA_NO_P1")):
        for e in env:
            os.environ[e] = "1"
        try:
            A2, Ar2 = ctx.mat_create(1, 1), ctx.mat_create(1, 1)
            ctx.assemble_scalar(lib.FORM_LAPLACE, A2, Ar2)
        finally:
            for e in env:
                del os.environ[e]
        assert sp_rel_err(mat_to_scipy(ctx, A2, 1, 1), ref["A"]) < FP_TOL
        assert sp_rel_err(mat_to_scipy(ctx, Ar2, 1, 1), ref["Arhs"]) < FP_TOL
    rng = np.random.default_rng(1)
    b = rng.standard_normal(mesh.n_node)
    b[mesh.boundary] = 0.0
    vb, vx = ctx.vec_create(1), ctx.vec_create(1)
    ctx.vec_set(vb, b)
    ctx.spmv(A, vb, vx)
    assert rel_err(ctx.vec_get(vx, 1), ref["A"] @ b) < 1e-13
    info = ctx.solve(A, vb, vx, method=lib.KSP_GMRES, pc=lib.PC_JACOBI, rtol=1e-10, restart=30)
    x_o, it_o, _ = fo.gmres(S, b, rtol=1e-10, restart=30)
    assert info.reason == 2 and abs(info.iters - it_o) <= 1
    assert rel_err(ctx.vec_get(vx, 1), x_o) < 1e-7
    assert info.true_resid <= 1e-8 * np.linalg.norm(b)
    info = ctx.solve(A, vb, vx, method=lib.KSP_CG, pc=lib.PC_JACOBI, rtol=1e-10)
    x_c, it_c, _ = fo.pcg(S, b, rtol=1e-10)
    assert info.reason == 2 and abs(info.iters - it_c) <= 1 and rel_err(ctx.vec_get(vx, 1), x_c) < 1e-8
    # vector-valued (KLE) blocks with the free-slip elimination
    mask = np.zeros((mesh.n_node, dim), np.uint8)
    mask[mesh.boundary] = 1
    ctx.bc_set(dim, mask)
    dw = 1 if dim == 2 else 3
    K, Kr, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
    ctx.assemble_kle(1e3, 1e2, K, Kr, Rw, -1)
    kle = fo.assemble_kle_freeslip(mesh, tb)
    assert sp_rel_err(mat_to_scipy(ctx, K, dim, dim), kle["K"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Kr, dim, dim), kle["Krhs"]) < FP_TOL
    assert sp_rel_err(mat_to_scipy(ctx, Rw, dim, dw), kle["Rw"]) < FP_TOL
    ctx.close()


@pytest.mark.parametrize("dim,nelem", [(2, [8, 6]), (3, [5, 4, 6])])
def test_imported_simplex_mesh_uniform_flow_gmres(tmp_path, dim, nelem):
    """DMPlexDom(fileName=...) on triangles / tetrahedra + `-ksp_type gmres -pc_type jacobi`: P1 reproduces the
    uniform field (the reference's analytic assertion, src/tests/test_solver.py:20-27, on a simplicial mesh)"""
    pynama_amd.install_reference_layout()
    from cases.uniform import UniformFlow
    from common.options import Options
    from pynama_amd.domain.gmsh import write_msh
    src = fo.simplex_box_mesh(nelem, [0.0] * dim, [1.0, 0.8, 1.2][:dim], jitter=0.2, permute_seed=3)
    path = str(tmp_path / "s.msh")
    write_msh(path, src.xyz, src.conn)
    with open(os.path.join(CASES, 'uniform.yaml')) as f:
        cfg = yaml.load(f, Loader=yaml.Loader)
    cfg["domain"] = {"ngl": 2, "gmsh-file": path}
    Options(["-ksp_type", "gmres", "-pc_type", "jacobi", "-ksp_rtol", "1e-12"])
    try:
        fem = UniformFlow(cfg, case="uniform")
        fem.setUp()
        fem.setUpSolver()
        assert fem.elemType.elemType.startswith("Simplex") and fem.dom.cellType == "simplex"
        exactVel, exactVort = fem.generateExactVecs()
        fem.solveKLE(time=0.0, vort=exactVort)
        assert fem.solver.getConvergedReason() > 0
        assert (exactVel - fem.vel).norm(norm_type=2) < 1e-9
    finally:
        Options([])


@pytest.mark.parametrize("cell,size", [("tet", 3), ("hex", 4)])
def test_imported_mesh_rank_blocks(lib, tmp_path, cell, size):
    """row-block partition of an imported mesh (Morton numbering, ghost index lists): each rank's owned rows
    of the assembled matrix and of the SpMV equal the one-rank result (detached communicator)"""
    from pynama_amd.common.comm import Comm
    from pynama_amd.domain.dmplex import DMPlexDom
    from pynama_amd.domain.gmsh import write_msh
    from pynama_amd.elements.simplex import Simplex
    from pynama_amd.elements.spectral import Spectral
    nelem = [6, 5, 7]
    src = fo.simplex_box_mesh(nelem, [0, 0, 0], [1, 1, 1], jitter=0.2, permute_seed=5) if cell == "tet" \
        else fo.box_mesh(nelem, [0, 0, 0], [1, 1, 1], 2, jitter=0.2)
    path = str(tmp_path / "m.msh")
    write_msh(path, src.xyz, src.conn)
    el = Simplex(3) if cell == "tet" else Spectral(2, 3)
    tb = fo.SimplexTables(3) if cell == "tet" else fo.Tables(2, 3)
    one = DMPlexDom(fileName=path, comm=Comm())
    one.setFemIndexing(2)
    glob = fo.BoxMesh(3, 2, tuple(nelem), (), one.conn, one.xyz, np.nonzero(one.boundaryMaskLocal())[0], {}, nc=tb.nc)
    ref = fo.assemble_scalar(glob, tb, "laplace", dirichlet=glob.boundary)
    xg = np.random.default_rng(11).standard_normal(glob.n_node)
    yg = ref["A"] @ xg
    for r in range(size):
        dom = DMPlexDom(fileName=path, comm=Comm(r, size))
        dom.setFemIndexing(2)
        ctx = lib.Context(0)
        ctx.comm_init(r, size, None)                      # detached
        ctx.halo_set(*dom._halo_plan())
        ctx.mesh_set(3, dom.conn, dom.xyz)
        for t in el.deviceTables():
            ctx.tables_set(*t)
        ctx.bc_set(1, dom.boundaryMaskLocal())
        ctx.csr_symbolic()
        A = ctx.mat_create(1, 1)
        ctx.assemble_scalar(lib.FORM_LAPLACE, A)          # hexes: tiled kernel on the default plan; tets: generic
        cols = dom._local2global(np.arange(dom.nLocal))
        assert sp_rel_err(mat_to_scipy(ctx, A, 1, 1), ref["A"][dom.rStart:dom.rEnd][:, cols]) < FP_TOL
        vx, vy = ctx.vec_create(1), ctx.vec_create(1)
        ctx.vec_set_local(vx, xg[cols])
        ctx.spmv(A, vx, vy)
        assert rel_err(ctx.vec_get(vy, 1), yg[dom.rStart:dom.rEnd]) < 1e-13
        ctx.close()
