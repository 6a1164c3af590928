"""Linear simplices (BASELINE.json configs[4]; no reference counterpart): the oracle's SimplexTables are
pinned by the textbook closed forms, and the product's tables must equal the oracle's.  Host only."""
import numpy as np
import pytest

from oracle import fem_oracle as fo


@pytest.mark.parametrize("dim", [2, 3])
def test_oracle_simplex_closed_forms(dim):
    m = fo.simplex_box_mesh([3, 2, 4][:dim], [0.0] * dim, [1.0, 2.0, 0.5][:dim], jitter=0.25, permute_seed=1)
    tb = fo.SimplexTables(dim)
    X = m.corners().reshape(-1, dim + 1, dim)
    J = X[:, 1:] - X[:, :1]
    det = np.linalg.det(J)
    fact = 2 if dim == 2 else 6
    assert det.min() > 0 and abs(det.sum() / fact - np.prod([1.0, 2.0, 0.5][:dim])) < 1e-13
    V = det / fact
    Ji = np.linalg.inv(J)                                           # [e, x, d] = d xi_d / d x
    gl = np.concatenate([-Ji.sum(axis=2)[:, :, None], Ji], axis=2)   # grad l_a  [e, x, a]
    L = V[:, None, None] * np.einsum("exa,exb->eab", gl, gl)
    M = V[:, None, None] * (1 + np.eye(dim + 1))[None] / ((dim + 1) * (dim + 2))
    assert np.abs(fo.elem_laplace(tb, m.corners()) - L).max() < 1e-13 * np.abs(L).max()
    assert np.abs(fo.elem_mass(tb, m.corners(), rule="full") - M).max() < 1e-15
    lump = np.eye(dim + 1)[None] * V[:, None, None] / (dim + 1)
    assert np.abs(fo.elem_mass(tb, m.corners()) - lump).max() < 1e-15
    # KLE blocks: K_e = kron(L_e, I) + penalties with constant gradients; check the curl/div parts
    # through their action on linear fields (P1 reproduces them exactly)
    K, Rw, Rd = fo.elem_kle_matrices(tb, m.corners())
    rigid = np.tile(np.arange(1.0, dim + 1), dim + 1)                # constant velocity: zero energy
    assert np.abs(K @ rigid).max() < 1e-9 * np.abs(K).max()
    A = fo.assemble_scalar(m, tb, "laplace")["A"]
    assert abs(A.sum(axis=1)).max() < 1e-12 and abs(A - A.T).max() < 1e-12
    lin = m.xyz @ np.arange(1.0, dim + 1)                            # harmonic: interior rows vanish
    interior = np.setdiff1d(np.arange(m.n_node), m.boundary)
    assert np.abs((A @ lin)[interior]).max() < 1e-12


@pytest.mark.parametrize("dim", [2, 3])
def test_product_tables_equal_oracle(dim):
    from pynama_amd.elements.simplex import Simplex
    el = Simplex(dim)
    tb = fo.SimplexTables(dim)
    assert el.nnode == dim + 1
    for (which, w, H, Hrs, HrsCoo), q in zip(el.deviceTables(), (tb.full, tb.red, tb.op)):
        np.testing.assert_allclose(w, q.w, atol=1e-16)
        np.testing.assert_allclose(H, q.H, atol=1e-15)
        np.testing.assert_allclose(Hrs, q.Hrs, atol=0)
        np.testing.assert_allclose(HrsCoo, q.Hrs, atol=0)
    # degree-2 exactness of the full rule, degree-1 of the others
    pts = np.array([[g[d] for d in range(dim)] for g in el.gps])
    w = np.array([g.w for g in el.gps])
    vol = 0.5 if dim == 2 else 1 / 6
    assert abs(w.sum() - vol) < 1e-16
    assert abs((w * pts[:, 0] ** 2).sum() - (1 / 12 if dim == 2 else 1 / 60)) < 1e-16
    assert abs((w * pts[:, 0] * pts[:, 1]).sum() - (1 / 24 if dim == 2 else 1 / 120)) < 1e-16
