"""Parity at BASELINE.json's full configuration sizes, through size-independent properties (the
oracle cannot assemble 128^3 in seconds): two independent device kernels agree entry by entry, the
operators annihilate what they must, are symmetric, and the solves reach the 1e-10 residual bar and
the analytic solutions."""
import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests.util import mat_to_scipy, rel_err, sp_rel_err

pytestmark = pytest.mark.gpu


def _domain(nelem, jitter=0.0):
    from pynama_amd.domain.dmplex import DMPlexDom
    from pynama_amd.elements.spectral import Spectral
    dim = len(nelem)
    dom = DMPlexDom(boxMesh={'nelem': nelem, 'lower': [0.0] * dim, 'upper': [1.0] * dim}, jitter=jitter)
    dom.setFemIndexing(2)
    ctx = dom.ctx
    for t in Spectral(2, dim).deviceTables():
        ctx.tables_set(*t)
    return dom, ctx


def test_c1_poisson_2d_32x32_vs_oracle():
    """configs[0]: 2D Poisson on 32x32 Q1 quads (the reference's CPU-runnable plumbing case)"""
    from pynama_amd import _lib
    dom, ctx = _domain([32, 32])
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(1, bm)
    ctx.csr_symbolic()
    A, Ar = ctx.mat_create(1, 1), ctx.mat_create(1, 1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, Ar)
    mesh = fo.box_mesh([32, 32], [0, 0], [1, 1], 2)
    ref = fo.assemble_scalar(mesh, fo.Tables(2, 2), "laplace", dirichlet=mesh.boundary)
    assert sp_rel_err(mat_to_scipy(ctx, A, 1, 1), ref["A"]) < 2e-13
    assert sp_rel_err(mat_to_scipy(ctx, Ar, 1, 1), ref["Arhs"]) < 2e-13
    b = np.random.default_rng(0).standard_normal(mesh.n_node)
    b[mesh.boundary] = 0
    vb, vx = ctx.vec_create(1), ctx.vec_create(1)
    ctx.vec_set(vb, b)
    info = ctx.solve(A, vb, vx, rtol=1e-10, norm_type=_lib.NORM_UNPRECONDITIONED)
    x_o, it_o, _ = fo.pcg(ref["A"], b, rtol=1e-10, norm_type=fo.NORM_UNPRECONDITIONED)
    assert abs(info.iters - it_o) <= 1 and info.true_resid <= 1e-10 and rel_err(ctx.vec_get(vx, 1), x_o) < 1e-8
    ctx.close()


def test_c2_poisson_128cubed_properties():
    """configs[1]: 3D Poisson, 128^3 Q1 hexahedra (2,146,689 DOFs, nnz 385^3)"""
    from pynama_amd import _lib
    n = 128
    dom, ctx = _domain([n, n, n], jitter=0.2)
    n_rows, nnz = ctx.csr_symbolic()
    assert n_rows == (n + 1) ** 3 and nnz == (3 * n + 1) ** 3
    ctx.patch_plan_set(*dom.patchPlan((7, 7, 7)))
    npatch, maxrows, maxlen, npe = ctx.patch_plan_info(0)
    assert npatch == 19 ** 3 and maxrows == 343 and maxlen == 27                   # 129 = 18 x 7 + 3 rows per axis; 27-point rows
    assert 1.0 < npe / n ** 3 < 1.6                                                 # elements on tile faces are integrated once per tile
    A, B = ctx.mat_create(1, 1), ctx.mat_create(1, 1)
    # (1) no mask: the two independent kernels (LDS-tiled vs HBM-atomic) agree entry by entry
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1, variant=1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, B, -1, variant=0)
    va, vb_ = ctx.mat_values(A, 1, 1), ctx.mat_values(B, 1, 1)
    assert np.abs(va - vb_).max() < 2e-13 * np.abs(vb_).max()
    del va, vb_
    # (2) the Laplacian annihilates constants and is symmetric
    one, y, u, v, Au, Av = (ctx.vec_create(1) for _ in range(6))
    ctx.vec_fill(one, 1.0)
    ctx.spmv(A, one, y)
    assert ctx.vec_norm(y, 3) < 1e-12
    rng = np.random.default_rng(1)
    ctx.vec_set(u, rng.standard_normal(n_rows))
    ctx.vec_set(v, rng.standard_normal(n_rows))
    ctx.spmv(A, u, Au)
    ctx.spmv(A, v, Av)
    assert abs(ctx.vec_dot(v, Au) - ctx.vec_dot(u, Av)) < 1e-10 * abs(ctx.vec_dot(v, Au))
    # (3) with the Dirichlet mask: identity rows, symmetric, CG to the 1e-10 residual bar
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(1, bm)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1, variant=1)
    e = np.zeros(n_rows)
    e[bm != 0] = rng.standard_normal(int((bm != 0).sum()))
    ctx.vec_set(u, e)
    ctx.spmv(A, u, Au)
    assert rel_err(ctx.vec_get(Au, 1), e) < 1e-15                        # A[bc, :] = identity
    f = (1.0 + dom.xyz[:, 0] + np.exp(dom.xyz[:, 1] * dom.xyz[:, 2])) / n ** 3
    f[bm != 0] = 0.0
    ctx.vec_set(u, f)
    info = ctx.solve(A, u, v, rtol=1e-10, norm_type=_lib.NORM_UNPRECONDITIONED, maxit=5000)
    assert info.reason == 2 and info.true_resid <= 1e-10
    ctx.close()


def test_c3_kle_128cubed_properties():
    """configs[2]: 3D KLE stiffness (3 DOF/node) on 128^3 hexahedra, vector-valued assembly"""
    from pynama_amd import _lib
    n = 128
    dom, ctx = _domain([n, n, n])
    n_rows, nnz = ctx.csr_symbolic()
    ctx.patch_plan_set(*dom.patchPlan((3, 3, 3)), kind=1)
    K, K2 = ctx.mat_create(3, 3), ctx.mat_create(3, 3)
    # (1) no mask: tiled == atomic kernel, rigid translations are in the null space, K symmetric
    ctx.assemble_kle(1e3, 1e2, K, -1, -1, -1, variant=1)
    ctx.assemble_kle(1e3, 1e2, K2, -1, -1, -1, variant=0)
    va, vb_ = ctx.mat_values(K, 3, 3), ctx.mat_values(K2, 3, 3)
    assert np.abs(va - vb_).max() < 2e-13 * np.abs(vb_).max()
    del va, vb_
    t, y, u, v, Ku, Kv = (ctx.vec_create(3) for _ in range(6))
    for comp in range(3):
        tr = np.zeros((n_rows, 3))
        tr[:, comp] = 1.0
        ctx.vec_set(t, tr.ravel())
        ctx.spmv(K, t, y)
        assert ctx.vec_norm(y, 3) < 1e-9 * 1500.5                       # lambda_max of K_e = 1500.5
    rng = np.random.default_rng(2)
    ctx.vec_set(u, rng.standard_normal(n_rows * 3))
    ctx.vec_set(v, rng.standard_normal(n_rows * 3))
    ctx.spmv(K, u, Ku)
    ctx.spmv(K, v, Kv)
    assert abs(ctx.vec_dot(v, Ku) - ctx.vec_dot(u, Kv)) < 1e-10 * abs(ctx.vec_dot(v, Ku))
    # (2) uniform flow: exact solution v == [1, 0, 0] (src/cases/uniform.py:23,35-37,53-62)
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(3, np.repeat(bm[:, None], 3, axis=1))
    Krhs = K2
    ctx.assemble_kle(1e3, 1e2, K, Krhs, -1, -1, variant=1)
    vel = np.zeros((n_rows, 3))
    vel[bm != 0] = [1.0, 0.0, 0.0]
    ctx.vec_set(u, vel.ravel())
    ctx.spmv(Krhs, u, y)
    info = ctx.solve(K, y, v, rtol=1e-10, norm_type=_lib.NORM_UNPRECONDITIONED, maxit=20000)
    x = ctx.vec_get(v, 3).reshape(-1, 3)
    assert info.reason == 2 and info.true_resid <= 1e-10
    assert np.abs(x - [1.0, 0.0, 0.0]).max() < 1e-7
    ctx.close()


def test_c5_unstructured_tets_gmsh_gmres(tmp_path):
    """configs[4]: ~5M tetrahedra (94^3 hexes cut in 6, randomly renumbered nodes), written to and imported
    from a Gmsh file, Poisson with GMRES(30)+Jacobi -- the irregular-indexing configuration"""
    import os
    from pynama_amd import _lib
    from pynama_amd.domain.dmplex import DMPlexDom
    from pynama_amd.domain.gmsh import write_msh
    from pynama_amd.elements.simplex import Simplex
    n = 94
    src = fo.simplex_box_mesh([n, n, n], [0.0] * 3, [1.0] * 3, jitter=0.2, permute_seed=2024)
    assert src.n_elem == 4983504
    path = str(tmp_path / "c5.msh")
    write_msh(path, src.xyz, src.conn)
    dom = DMPlexDom(fileName=path)
    dom.setFemIndexing(2)
    os.remove(path)
    assert dom.cellType == "simplex" and dom.nOwned == (n + 1) ** 3 and dom.conn.shape == (src.n_elem, 4)
    ctx = dom.ctx
    for t in Simplex(3).deviceTables():
        ctx.tables_set(*t)
    n_rows, nnz = ctx.csr_symbolic()
    edges = np.sort(src.conn[:, [[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]]].reshape(-1, 2), axis=1)
    n_edges = len(np.unique(edges[:, 0].astype(np.int64) * src.n_node + edges[:, 1]))
    assert nnz == n_rows + 2 * n_edges                                  # node graph = mesh edges + diagonal
    del edges
    # (1) no mask: the LDS patch kernel and the one-lane-per-cell atomics kernel agree entry by entry
    A, B = ctx.mat_create(1, 1), ctx.mat_create(1, 1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
    os.environ["PYNAMA_NO_P1_TILED"] = "1"          # the one-lane-per-cell atomics kernel instead of the LDS patch kernel
    try:
        ctx.assemble_scalar(_lib.FORM_LAPLACE, B)
    finally:
        del os.environ["PYNAMA_NO_P1_TILED"]
    va, vb_ = ctx.mat_values(A, 1, 1), ctx.mat_values(B, 1, 1)
    assert np.abs(va - vb_).max() < 2e-13 * np.abs(vb_).max()
    del va, vb_
    # (2) constants are annihilated, linear fields are discretely harmonic (P1 exactness), A is symmetric
    bm = dom.boundaryMaskLocal()
    u, v, Au, Av = (ctx.vec_create(1) for _ in range(4))
    ctx.vec_fill(u, 1.0)
    ctx.spmv(A, u, Au)
    assert ctx.vec_norm(Au, 3) < 1e-12
    ctx.vec_set(u, dom.xyz @ np.array([1.0, 2.0, 3.0]))
    ctx.spmv(A, u, Au)
    assert np.abs(ctx.vec_get(Au, 1)[bm == 0]).max() < 1e-12
    rng = np.random.default_rng(3)
    ctx.vec_set(u, rng.standard_normal(n_rows))
    ctx.vec_set(v, rng.standard_normal(n_rows))
    ctx.spmv(A, u, Au)
    ctx.spmv(A, v, Av)
    assert abs(ctx.vec_dot(v, Au) - ctx.vec_dot(u, Av)) < 1e-10 * abs(ctx.vec_dot(v, Au))
    # (3) Dirichlet problem: GMRES(30)+Jacobi to the 1e-10 residual bar; CG lands on the same solution
    ctx.bc_set(1, bm)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
    X = dom.xyz
    f = (1.0 + X[:, 0] + 2.0 * X[:, 1] ** 2 + np.exp(X[:, 0] * X[:, 1] * X[:, 2])) / n ** 3
    f[bm != 0] = 0.0
    ctx.vec_set(u, f)
    # (convergence is declared on the residual recomputed at every restart, in the norm asked for: rtol 1e-10 IS 1e-10)
    info = ctx.solve(A, u, v, method=_lib.KSP_GMRES, pc=_lib.PC_JACOBI, rtol=1e-10, restart=30, maxit=100000,
                     norm_type=_lib.NORM_UNPRECONDITIONED)
    assert info.reason == 2 and info.true_resid <= 1e-10 * (1 + 1e-6)
    xg = ctx.vec_get(v, 1)
    info = ctx.solve(A, u, v, method=_lib.KSP_CG, pc=_lib.PC_JACOBI, rtol=1e-12, norm_type=_lib.NORM_UNPRECONDITIONED)
    assert info.reason == 2
    assert rel_err(xg, ctx.vec_get(v, 1)) < 1e-6
    ctx.close()


def test_c4_poisson_256cubed_single_gpu_properties():
    """configs[3] is 256^3 over 8 GPUs; the same mesh also fits ONE MI355X (16,974,593 DOFs, nnz 769^3 = 454,756,609:
    5.4 GB of CSR values): closed-form graph, plan-free assembly, SELL image and CG at that size"""
    from pynama_amd import _lib
    n = 256
    dom, ctx = _domain([n, n, n])
    n_rows, nnz = ctx.csr_symbolic()
    assert n_rows == (n + 1) ** 3 and nnz == (3 * n + 1) ** 3
    assert ctx.mesh_topology() == ("lattice", n + 1, n + 1, n + 1)
    A = ctx.mat_create(1, 1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
    one, y, u, v, Au, Av = (ctx.vec_create(1) for _ in range(6))
    ctx.vec_fill(one, 1.0)
    ctx.spmv(A, one, y)
    assert ctx.vec_norm(y, 3) < 1e-12                                    # constants annihilated
    rng = np.random.default_rng(4)
    ctx.vec_set(u, rng.standard_normal(n_rows))
    ctx.vec_set(v, rng.standard_normal(n_rows))
    ctx.spmv(A, u, Au)
    ctx.spmv(A, v, Av)
    assert abs(ctx.vec_dot(v, Au) - ctx.vec_dot(u, Av)) < 1e-10 * abs(ctx.vec_dot(v, Au))
    # uniform mesh: the interior stencil is the classical 27-point one, diagonal 8 h / 3
    h = 1.0 / n
    e = np.zeros(n_rows)
    centre = (n // 2) * (n + 1) ** 2 + (n // 2) * (n + 1) + n // 2
    e[centre] = 1.0
    ctx.vec_set(u, e)
    ctx.spmv(A, u, Au)
    col = ctx.vec_get(Au, 1)
    assert abs(col[centre] - 8.0 * h / 3.0) < 1e-14
    assert np.count_nonzero(col) == 27 - 6                               # face neighbours vanish for the trilinear Laplacian
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(1, bm)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
    f = (1.0 + dom.xyz[:, 0] + np.exp(dom.xyz[:, 1] * dom.xyz[:, 2])) * h ** 3
    f[bm != 0] = 0.0
    ctx.vec_set(u, f)
    info = ctx.solve(A, u, v, rtol=1e-10, norm_type=_lib.NORM_UNPRECONDITIONED, maxit=5000)
    assert info.reason == 2 and info.true_resid <= 1e-10
    ctx.close()


@pytest.mark.parametrize("jitter", [0.0, 0.2])
def test_matrix_free_at_full_size(jitter):
    """10 M-DOF Poisson and 128^3 KLE: the matrix-free products equal the assembled ones row by row (two independent
    kernels: SELL/CSR values vs element products recomputed from the coordinates), constants are annihilated, and the
    matrix-free CG reaches the 1e-10 bar of BASELINE.json measured with the ASSEMBLED matrix"""
    from pynama_amd import _lib
    n = 215 if jitter == 0.0 else 128
    dom, ctx = _domain([n, n, n], jitter=jitter)
    bm = dom.boundaryMaskLocal()
    n_rows, _ = ctx.csr_symbolic()
    rng = np.random.default_rng(3)
    vx, vy, vz = ctx.vec_create(1), ctx.vec_create(1), ctx.vec_create(1)
    # no mask: the Laplacian annihilates constants
    ctx.bc_set(1, None)
    ctx.matfree_set(_lib.MATFREE_LAPLACE)
    ctx.vec_set(vx, np.ones(n_rows))
    ctx.matfree_apply(vx, vy)
    assert np.abs(ctx.vec_get(vy, 1)).max() < 1e-12 / n
    # Dirichlet boundary: product and solve
    ctx.bc_set(1, bm)
    A = ctx.mat_create(1, 1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
    ctx.matfree_set(_lib.MATFREE_LAPLACE)
    x = rng.standard_normal(n_rows)
    ctx.vec_set(vx, x)
    ctx.spmv(A, vx, vy)
    ctx.matfree_apply(vx, vz)
    assert rel_err(ctx.vec_get(vz, 1), ctx.vec_get(vy, 1)) < 2e-13
    b = rng.standard_normal(n_rows) / n ** 3
    b[bm != 0] = 0.0
    ctx.vec_set(vx, b)
    i0 = ctx.solve(A, vx, vy, rtol=1e-10, maxit=20000, norm_type=_lib.NORM_UNPRECONDITIONED)
    i1 = ctx.solve(A, vx, vz, rtol=1e-10, maxit=20000, norm_type=_lib.NORM_UNPRECONDITIONED, matfree=_lib.MATFREE_LAPLACE)
    assert i0.reason == 2 and i1.reason == 2 and abs(i0.iters - i1.iters) <= 2 and i1.true_resid <= 1e-10
    assert rel_err(ctx.vec_get(vz, 1), ctx.vec_get(vy, 1)) < 1e-7
    ctx.close()
    # KLE, 128^3 (configs[2])
    dom, ctx = _domain([128, 128, 128], jitter=jitter)
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(3, np.repeat(bm[:, None], 3, axis=1))
    n_rows, _ = ctx.csr_symbolic()
    K = ctx.mat_create(3, 3)
    ctx.assemble_kle(1e3, 1e2, K)
    ctx.matfree_set(_lib.MATFREE_KLE, 1e3, 1e2)
    vx, vy, vz = ctx.vec_create(3), ctx.vec_create(3), ctx.vec_create(3)
    ctx.vec_set(vx, rng.standard_normal(3 * n_rows))
    ctx.spmv(K, vx, vy)
    ctx.matfree_apply(vx, vz, op=_lib.MATFREE_KLE)
    assert rel_err(ctx.vec_get(vz, 3), ctx.vec_get(vy, 3)) < 2e-13
    ctx.close()


# ---- entry-by-entry parity, at full size, of the kernels bench.py actually times (no explicit plan: the plan-free
# ---- lattice / z-marching / KLE lattice kernels; the generic atomics kernel, variant 0, is the independent second opinion) ----------
def _same_entries(ctx, a, b, br, bc, tol=2e-13):
    va, vb = ctx.mat_values(a, br, bc), ctx.mat_values(b, br, bc)
    err = np.abs(va - vb).max() / np.abs(vb).max()
    assert err < tol, err


@pytest.mark.parametrize("jitter", [0.0, 0.2])
def test_c2_bench_kernels_128cubed_entry_by_entry(jitter):
    """128^3 scalar Laplacian: the plan-free lattice kernel (uniform mesh: parallelepiped closed form) and the z-marching kernel
    (jitter 0.2: 2x2x2 rule per cell) against the generic kernel, A and Arhs, with the boundary mask"""
    from pynama_amd import _lib
    n = 128
    dom, ctx = _domain([n, n, n], jitter=jitter)
    ctx.bc_set(1, dom.boundaryMaskLocal())
    ctx.csr_symbolic()
    assert ctx.mesh_topology()[0] == "lattice"
    A, Ar, B, Br = (ctx.mat_create(1, 1) for _ in range(4))
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, Ar, variant=1)            # what bench.py's C2 / headline / general-geometry legs launch
    ctx.assemble_scalar(_lib.FORM_LAPLACE, B, Br, variant=0)
    _same_entries(ctx, A, B, 1, 1)
    _same_entries(ctx, Ar, Br, 1, 1)
    ctx.close()


@pytest.mark.parametrize("jitter", [0.0, 0.2])
def test_c3_bench_kernels_128cubed_entry_by_entry(jitter):
    """128^3 KLE: K, Krhs AND Rw of the plan-free KLE lattice kernels (closed-form blocks on the uniform mesh; element pre-pass +
    closed form of the 2x2x2 rule with jitter 0.2) against the generic kernel"""
    n = 128
    dom, ctx = _domain([n, n, n], jitter=jitter)
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(3, np.repeat(bm[:, None], 3, axis=1))
    ctx.csr_symbolic()
    K, Kr, Rw = (ctx.mat_create(3, 3) for _ in range(3))
    ctx.assemble_kle(1e3, 1e2, K, Kr, Rw, -1, variant=1)                # no plan set: what bench.py's C3 legs launch
    ref = ctx.mat_create(3, 3)
    for target, ids in ((K, (ref, -1, -1)), (Kr, (-1, ref, -1)), (Rw, (-1, -1, ref))):   # one 4.1 GB reference matrix at a time
        k0, kr0, rw0 = ids
        if kr0 >= 0:                      # Krhs alone is not a request of the ABI: K goes along into a scratch matrix
            scratch = ctx.mat_create(3, 3)
            ctx.assemble_kle(1e3, 1e2, scratch, kr0, -1, -1, variant=0)
            ctx.mat_destroy(scratch)
        else:
            ctx.assemble_kle(1e3, 1e2, k0, kr0, rw0, -1, variant=0)
        _same_entries(ctx, target, ref, 3, 3)
    ctx.close()


@pytest.mark.parametrize("dim,nel", [(2, 1024), (3, 64)])
def test_ho3_bench_kernels_entry_by_entry(dim, nel):
    """bench.py's HO3_2D / HO3_3D meshes (1024^2, 64^3 second-order cells): K, Krhs, Rw of the row-run kernels against the generic
    kernel entry by entry; K symmetric, rigid translations in its null space; the three product kernels agree"""
    import os
    from pynama_amd.domain.dmplex import DMPlexDom
    from pynama_amd.elements.spectral import Spectral
    dw = 1 if dim == 2 else 3
    dom = DMPlexDom(boxMesh={'nelem': [nel] * dim, 'lower': [0.0] * dim, 'upper': [1.0] * dim})
    dom.setFemIndexing(3)
    ctx = dom.ctx
    for t in Spectral(3, dim).deviceTables():
        ctx.tables_set(*t)
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(dim, np.repeat(bm[:, None], dim, axis=1))
    n_rows, nnzb = ctx.csr_symbolic()
    assert ctx.mesh_topology()[0] == "lattice-ngl3" and nnzb == (8 * nel + 1) ** dim       # per axis: n - 1 inner vertices x 5, 2 end vertices x 3, n mid nodes x 3
    K, Kr, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
    os.environ["PYNAMA_HO3_REQUIRE"] = "1"
    try:
        ctx.assemble_kle(1e3, 1e2, K, Kr, Rw, -1)
    finally:
        del os.environ["PYNAMA_HO3_REQUIRE"]
    K0, Kr0 = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim)
    ctx.assemble_kle(1e3, 1e2, K0, Kr0, -1, -1, variant=0)
    _same_entries(ctx, K, K0, dim, dim)
    _same_entries(ctx, Kr, Kr0, dim, dim)
    ctx.mat_destroy(K0)
    ctx.mat_destroy(Kr0)
    Rw0 = ctx.mat_create(dim, dw)
    ctx.assemble_kle(1e3, 1e2, -1, -1, Rw0, -1, variant=0)
    _same_entries(ctx, Rw, Rw0, dim, dw)
    ctx.mat_destroy(Rw0)
    # products: default kernel == block-CSR group kernel == SELL image, on the masked K
    rng = np.random.default_rng(3)
    u, v, Ku, Kv = (ctx.vec_create(dim) for _ in range(4))
    ctx.vec_set(u, rng.standard_normal(n_rows * dim))
    ctx.vec_set(v, rng.standard_normal(n_rows * dim))
    ctx.spmv(K, u, Ku)
    y0 = ctx.vec_get(Ku, dim)
    for env in ({"PYNAMA_NO_CSRLB": "1", "PYNAMA_BCSR_MIN_AVG": "0"}, {"PYNAMA_BLOCK_SELL": "1"}):
        os.environ.update(env)
        try:
            ctx.spmv(K, u, Ku)
        finally:
            for k in env:
                del os.environ[k]
        assert rel_err(ctx.vec_get(Ku, dim), y0) < 1e-13, env
    ctx.spmv(K, u, Ku)
    ctx.spmv(K, v, Kv)
    assert abs(ctx.vec_dot(v, Ku) - ctx.vec_dot(u, Kv)) < 1e-10 * abs(ctx.vec_dot(v, Ku))       # symmetric after the elimination
    # no mask: translations are annihilated
    ctx.bc_set(dim, None)
    ctx.assemble_kle(1e3, 1e2, K, -1, -1, -1)
    for comp in range(dim):
        tr = np.zeros((n_rows, dim))
        tr[:, comp] = 1.0
        ctx.vec_set(u, tr.ravel())
        ctx.spmv(K, u, Ku)
        assert ctx.vec_norm(Ku, 3) < 1e-8 * 1500.5
    ctx.close()
