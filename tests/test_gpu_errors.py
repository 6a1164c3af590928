"""Error behaviour of the C ABI (include/pynama_hip.h: every call returns 0 or a negative code with a message;
the Python layer raises PynamaHipError).  The product never falls back to another implementation: a wrong
call fails loudly and leaves the context usable."""
import numpy as np
import pytest

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pynama_amd import _lib
    assert _lib.device_count() > 0, "GPU tests need an MI355X"
    return _lib


def _tables(ctx, ngl, dim):
    from pynama_amd.elements.spectral import Spectral
    for t in Spectral(ngl, dim).deviceTables():
        ctx.tables_set(*t)


def test_call_order_is_enforced(lib):
    mesh = fo.box_mesh([3, 3, 3], [0, 0, 0], [1, 1, 1], 2)
    ctx = lib.Context(0)
    with pytest.raises(lib.PynamaHipError, match="pyn_mesh_set first"):
        ctx.csr_symbolic()
    ctx.mesh_set(3, mesh.conn, mesh.xyz)
    with pytest.raises(lib.PynamaHipError, match="pyn_csr_symbolic first"):
        ctx.mat_create(1, 1)
    ctx.csr_symbolic()
    A = ctx.mat_create(1, 1)
    with pytest.raises(lib.PynamaHipError, match="tables"):
        ctx.assemble_scalar(lib.FORM_LAPLACE, A)          # element tables not uploaded yet
    _tables(ctx, 2, 3)
    ctx.assemble_scalar(lib.FORM_LAPLACE, A)              # ... and the context still works
    assert np.isfinite(ctx.mat_values(A, 1, 1)).all()
    ctx.close()


def test_bad_arguments(lib):
    mesh = fo.box_mesh([3, 2, 2], [0, 0, 0], [1, 1, 1], 2)
    ctx = lib.Context(0)
    bad = mesh.conn.copy()
    bad[1, 3] = mesh.n_node                                # out-of-range node id
    with pytest.raises(lib.PynamaHipError, match="out of range"):
        ctx.mesh_set(3, bad, mesh.xyz)
    with pytest.raises(lib.PynamaHipError, match="neither"):
        ctx.mesh_set(3, mesh.conn[:, :5].copy(), mesh.xyz)   # 5 nodes per cell: no such element
    ctx.mesh_set(3, mesh.conn, mesh.xyz)
    _tables(ctx, 2, 3)
    ctx.csr_symbolic()
    A, K = ctx.mat_create(1, 1), ctx.mat_create(3, 3)
    with pytest.raises(lib.PynamaHipError, match="block shape"):
        ctx.assemble_scalar(lib.FORM_LAPLACE, K)           # 3x3 matrix for a scalar form
    with pytest.raises(lib.PynamaHipError):
        ctx.assemble_scalar(lib.FORM_LAPLACE, 99)          # no such matrix
    mask3 = np.ones((mesh.n_node, 3), np.uint8)
    ctx.bc_set(3, mask3)
    with pytest.raises(lib.PynamaHipError, match="ndof"):
        ctx.assemble_scalar(lib.FORM_LAPLACE, A)           # vector mask, scalar form
    ctx.bc_set(1, None)
    ctx.assemble_scalar(lib.FORM_LAPLACE, A)
    v1, v3 = ctx.vec_create(1), ctx.vec_create(3)
    with pytest.raises(lib.PynamaHipError, match="block size"):
        ctx.solve(A, v3, v1)
    with pytest.raises(lib.PynamaHipError, match="differ"):
        ctx.solve(A, v1, v1)
    with pytest.raises(lib.PynamaHipError):
        ctx.spmv(K, v1, v3)                                # column block size 3, x has 1
    with pytest.raises(lib.PynamaHipError, match="cover"):
        ctx.patch_plan_set(np.array([0, 5], np.int32), np.arange(mesh.n_node, dtype=np.int32))
    ctx.close()


def test_solver_reports_breakdown_instead_of_garbage(lib):
    """zero matrix rows / indefinite systems end with a negative converged reason, not with NaNs passed on"""
    mesh = fo.box_mesh([3, 3, 3], [0, 0, 0], [1, 1, 1], 2)
    ctx = lib.Context(0)
    ctx.mesh_set(3, mesh.conn, mesh.xyz)
    _tables(ctx, 2, 3)
    ctx.csr_symbolic()
    A = ctx.mat_create(1, 1)
    ctx.assemble_scalar(lib.FORM_LAPLACE, A)               # singular (pure Neumann) matrix
    b = np.ones(mesh.n_node)
    vb, vx = ctx.vec_create(1), ctx.vec_create(1)
    ctx.vec_set(vb, b)                                      # inconsistent right-hand side
    info = ctx.solve(A, vb, vx, rtol=1e-12, maxit=200)
    assert info.reason < 0 or info.true_resid > 1e-6        # did not pretend to converge
    ctx.close()


def test_missing_library_is_loud(monkeypatch, tmp_path):
    """the Python layer refuses to run without the HIP library instead of falling back to anything"""
    from pynama_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libpynama_hip.so"))
    with pytest.raises(_lib.PynamaHipError, match="no CPU fallback"):
        _lib.load_library()
