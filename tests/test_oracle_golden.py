"""Pin the CPU oracle against the reference's own outputs (tests/golden/*.npz, produced by
tests/golden/make_golden.py from /root/reference/src/elements) and the known answers of
/root/reference/src/tests/test_element.py."""
import math

import numpy as np
import pytest

from oracle import fem_oracle as fo


# ---- known answers: src/tests/test_element.py:176-229 --------------------------------
def test_gauss_known_answers():
    x, w = fo.gauss_legendre(2)
    np.testing.assert_allclose(x, [-1 / math.sqrt(3), 1 / math.sqrt(3)], rtol=0, atol=1e-15)
    np.testing.assert_allclose(w, [1, 1], rtol=0, atol=1e-15)
    x, w = fo.gauss_legendre(3)
    np.testing.assert_allclose(x, [-math.sqrt(3 / 5), 0, math.sqrt(3 / 5)], atol=1e-12)
    np.testing.assert_allclose(w, [5 / 9, 8 / 9, 5 / 9], atol=1e-12)


def test_lobatto_known_answers():
    exp = {2: ([-1, 1], [1, 1]), 3: ([-1, 0, 1], [1 / 3, 4 / 3, 1 / 3]),
           4: ([-1, -1 / math.sqrt(5), 1 / math.sqrt(5), 1], [1 / 6, 5 / 6, 5 / 6, 1 / 6])}
    for n, (xe, we) in exp.items():
        x, w = fo.gauss_lobatto(n)
        np.testing.assert_allclose(x, xe, atol=1e-12)
        np.testing.assert_allclose(w, we, atol=1e-12)


# ---- G1 ------------------------------------------------------------------------------
@pytest.mark.parametrize("n", range(2, 13))
def test_rules_vs_reference(golden, n):
    g = golden["g1_rules"]
    x, w = fo.gauss_legendre(n)
    np.testing.assert_allclose(x, g[f"gauss_x_{n}"], rtol=0, atol=5e-15)
    np.testing.assert_allclose(w, g[f"gauss_w_{n}"], rtol=0, atol=5e-15)
    x, w = fo.gauss_lobatto(n)
    np.testing.assert_allclose(x, g[f"lobatto_x_{n}"], rtol=0, atol=5e-15)
    np.testing.assert_allclose(w, g[f"lobatto_w_{n}"], rtol=0, atol=5e-15)


# ---- G2 ------------------------------------------------------------------------------
VARIANT = {"": "full", "Red": "red", "Op": "op", "Coo": "coo", "CooRed": "coo_red", "CooOp": "coo_op"}


@pytest.mark.parametrize("dim,ngl", [(2, 2), (2, 3), (2, 4), (2, 5), (3, 2), (3, 3), (3, 4)])
def test_tables_vs_reference(golden, dim, ngl):
    g = golden["g2_tables"]
    tb = fo.Tables(ngl, dim)
    for suffix, attr in VARIANT.items():
        q = getattr(tb, attr)
        ref_gps = g[f"d{dim}_n{ngl}_gps{suffix}"]
        np.testing.assert_allclose(q.pts, ref_gps[:, :dim], rtol=0, atol=1e-14, err_msg=f"gps{suffix}")
        np.testing.assert_allclose(q.w, ref_gps[:, dim], rtol=0, atol=1e-14, err_msg=f"w{suffix}")
        np.testing.assert_allclose(q.H, g[f"d{dim}_n{ngl}_H{suffix}"], rtol=0, atol=2e-14, err_msg=f"H{suffix}")
        np.testing.assert_allclose(q.Hrs, g[f"d{dim}_n{ngl}_Hrs{suffix}"], rtol=0, atol=2e-13, err_msg=f"Hrs{suffix}")
    np.testing.assert_allclose(tb.HCoo1D, g[f"d{dim}_n{ngl}_HCoo1D"], rtol=0, atol=1e-14)


@pytest.mark.parametrize("dim,ngl", [(2, 6), (2, 7), (2, 11), (3, 5), (3, 6)])
def test_orderings_high_order(golden, dim, ngl):
    g = golden["g2_tables"]
    tb = fo.Tables(ngl, dim)
    for suffix, attr in (("Op", "op"), ("", "full"), ("Red", "red")):
        ref = g[f"d{dim}_n{ngl}_gps{suffix}"]
        q = getattr(tb, attr)
        np.testing.assert_allclose(q.pts, ref[:, :dim], rtol=0, atol=1e-13)
        np.testing.assert_allclose(q.w, ref[:, dim], rtol=0, atol=1e-13)


# ---- G3 ------------------------------------------------------------------------------
CASES = ["unit", "reftest", "brick128", "stretched", "jitter"]


@pytest.mark.parametrize("dim,ngl", [(2, 2), (2, 3), (2, 5), (3, 2), (3, 3)])
@pytest.mark.parametrize("case", CASES)
def test_element_matrices_vs_reference(golden, dim, ngl, case):
    g = golden["g3_elem"]
    key = f"d{dim}_n{ngl}_{case}"
    tb = fo.Tables(ngl, dim)
    K, Rw, Rd = fo.elem_kle_matrices(tb, g[key + "_coords"])
    for name, got in (("K", K[0]), ("Rw", Rw[0]), ("Rd", Rd[0])):
        ref = g[f"{key}_{name}"]
        scale = np.abs(ref).max()
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-13 * scale, err_msg=name)
    SrT, Div, Curl, wei = fo.elem_kle_operators(tb, g[key + "_coords"])
    for name, got in (("SrT", SrT[0]), ("DivSrT", Div[0]), ("Curl", Curl[0]), ("wei", wei[0])):
        ref = g[f"{key}_{name}"]
        scale = max(np.abs(ref).max(), 1e-300)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-13 * scale, err_msg=name)


@pytest.mark.parametrize("dim", [2, 3])
def test_kron_identity(golden, dim):
    """K_e == kron(L_e, I_dim) + penalty; penalty vanishes for alpha=0."""
    g = golden["g3_elem"]
    tb = fo.Tables(2, dim)
    c = g[f"d{dim}_n2_jitter_coords"]
    K0, _, _ = fo.elem_kle_matrices(tb, c, alpha_d=0.0, alpha_w=0.0)
    L = fo.elem_laplace(tb, c)
    np.testing.assert_allclose(K0[0], np.kron(L[0], np.eye(dim)), rtol=0, atol=1e-13)
    # sanity values from SURVEY.md A.1
    if dim == 3:
        K, _, _ = fo.elem_kle_matrices(tb, g["d3_n2_unit_coords"])
        ev = np.linalg.eigvalsh(K[0])
        assert abs(ev.max() - 1500.5) < 1e-9 and np.sum(np.abs(ev) < 1e-9) == 3
    else:
        K, _, _ = fo.elem_kle_matrices(tb, g["d2_n2_unit_coords"])
        assert abs(K[0][0, 0] - 275.666666666666) < 1e-9 and abs(K[0][0, 1] - 225) < 1e-9
