"""The C restatement of the oracle (oracle/c/fem_oracle.c, the bench's CPU baseline) against the
numpy oracle, which is itself pinned to the reference's golden vectors."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import fem_oracle as fo
from tests.util import block_csr_to_scipy, rel_err, sp_rel_err


@pytest.mark.parametrize("dim,ngl", [(2, 2), (3, 2), (2, 3), (3, 3)])
def test_c_elem_kle_vs_golden(golden, dim, ngl):
    g = golden["g3_elem"]
    tb = fo.Tables(ngl, dim)
    for case in ("unit", "jitter", "stretched"):
        key = f"d{dim}_n{ngl}_{case}"
        K, Rw, Rd = co.elem_kle(tb, g[key + "_coords"])
        assert rel_err(K, g[key + "_K"]) < 2e-13
        assert rel_err(Rw, g[key + "_Rw"]) < 2e-13
        assert rel_err(Rd, g[key + "_Rd"]) < 2e-13


@pytest.mark.parametrize("nelem", [[7, 5], [5, 4, 6]])
def test_c_global_vs_numpy_oracle(nelem):
    dim = len(nelem)
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, 2, jitter=0.2)
    tb = fo.Tables(2, dim)
    rp, ci = co.csr_pattern(mesh.conn, mesh.n_node)
    rp_o, ci_o = fo.node_graph(mesh)
    assert np.array_equal(rp, rp_o) and np.array_equal(ci, ci_o)
    # scalar
    mask = np.zeros(mesh.n_node, np.uint8)
    mask[mesh.boundary] = 1
    A, Ar = co.assemble_laplace(mesh, tb, rp, ci, mask)
    ref = fo.assemble_scalar(mesh, tb, "laplace", dirichlet=mesh.boundary)
    assert sp_rel_err(block_csr_to_scipy(rp, ci, A, 1, 1), ref["A"]) < 2e-13
    assert sp_rel_err(block_csr_to_scipy(rp, ci, Ar, 1, 1), ref["Arhs"]) < 2e-13
    # KLE
    vmask = np.zeros((mesh.n_node, dim), np.uint8)
    vmask[mesh.boundary] = 1
    K, Kr, Rw = co.assemble_kle(mesh, tb, rp, ci, vmask)
    refk = fo.assemble_kle_freeslip(mesh, tb)
    assert sp_rel_err(block_csr_to_scipy(rp, ci, K, dim, dim), refk["K"]) < 2e-13
    assert sp_rel_err(block_csr_to_scipy(rp, ci, Kr, dim, dim), refk["Krhs"]) < 2e-13
    assert sp_rel_err(block_csr_to_scipy(rp, ci, Rw, dim, tb.dim_w), refk["Rw"]) < 2e-13
    # PCG
    rng = np.random.default_rng(3)
    b = rng.standard_normal(mesh.n_node)
    b[mesh.boundary] = 0
    x, it, rn = co.pcg(rp, ci, A, b, rtol=1e-10, norm_type=1)
    x_o, it_o, hist = fo.pcg(ref["A"], b, rtol=1e-10, norm_type=1)
    assert abs(it - it_o) <= 1 and rel_err(x, x_o) < 1e-8
    y = co.spmv(rp, ci, K, np.ones(mesh.n_node * dim), dim, dim)
    assert rel_err(y, refk["K"] @ np.ones(mesh.n_node * dim)) < 1e-12
