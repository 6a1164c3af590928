"""The PRODUCT's host tables against the reference's own outputs (tests/golden/g1_rules.npz, g2_tables.npz, written by
tests/golden/make_golden.py from the imported /root/reference/src/elements): `gaussPoints` / `lobattoPoints`
(src/elements/utilities.py:43-92), `Element.interpFun1D` (element.py:17-49) and all 19 tables of `Spectral(ngl, dim)`
(spectral.py:39-87, 220-431) -- the analogue of src/tests/test_element.py:9-133, 176-229 for the drop-in classes.  No GPU:
the tables are host set-up (the device receives them through `deviceTables()`)."""
import math

import numpy as np
import pytest

from pynama_amd.elements.spectral import Spectral
from pynama_amd.elements.utilities import gaussPoints, lobattoPoints


# ---- known answers: src/tests/test_element.py:176-229 ------------------------------------------------------------------
def test_gauss_known_answers():
    x, w = gaussPoints(2)
    np.testing.assert_allclose(x, [-1 / math.sqrt(3), 1 / math.sqrt(3)], rtol=0, atol=1e-15)
    np.testing.assert_allclose(w, [1, 1], rtol=0, atol=1e-15)
    x, w = gaussPoints(3)
    np.testing.assert_allclose(x, [-math.sqrt(3 / 5), 0, math.sqrt(3 / 5)], atol=1e-14)
    np.testing.assert_allclose(w, [5 / 9, 8 / 9, 5 / 9], atol=1e-14)


def test_lobatto_known_answers():
    exp = {2: ([-1, 1], [1, 1]), 3: ([-1, 0, 1], [1 / 3, 4 / 3, 1 / 3]),
           4: ([-1, -1 / math.sqrt(5), 1 / math.sqrt(5), 1], [1 / 6, 5 / 6, 5 / 6, 1 / 6])}
    for n, (xe, we) in exp.items():
        x, w = lobattoPoints(n)
        np.testing.assert_allclose(x, xe, atol=1e-14)
        np.testing.assert_allclose(w, we, atol=1e-14)


# ---- G1: the rules, N = 2..12 --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", range(2, 13))
def test_rules_vs_reference(golden, n):
    g = golden["g1_rules"]
    x, w = gaussPoints(n)
    np.testing.assert_allclose(x, g[f"gauss_x_{n}"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(w, g[f"gauss_w_{n}"], rtol=0, atol=1e-14)
    x, w = lobattoPoints(n)
    np.testing.assert_allclose(x, g[f"lobatto_x_{n}"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(w, g[f"lobatto_w_{n}"], rtol=0, atol=1e-14)


# ---- G2: H, Hrs, gps for the full / reduced / nodal rules of the element and of the corner geometry, HCoo1D ---------------
SUFFIXES = ["", "Red", "Op", "Coo", "CooRed", "CooOp"]


def _gps_array(gps, dim):
    return np.array([[p.r, p.s, p.w] if dim == 2 else [p.r, p.s, p.t, p.w] for p in gps])


@pytest.mark.parametrize("dim,ngl", [(2, 2), (2, 3), (2, 4), (2, 5), (3, 2), (3, 3), (3, 4)])
def test_spectral_tables_vs_reference(golden, dim, ngl):
    g = golden["g2_tables"]
    sp = Spectral(ngl, dim)
    for suffix in SUFFIXES:
        ref_gps = g[f"d{dim}_n{ngl}_gps{suffix}"]
        np.testing.assert_allclose(_gps_array(getattr(sp, "gps" + suffix), dim), ref_gps, rtol=0, atol=1e-14, err_msg=f"gps{suffix}")
        np.testing.assert_allclose(np.asarray(getattr(sp, "H" + suffix)), g[f"d{dim}_n{ngl}_H{suffix}"], rtol=0, atol=1e-14,
                                   err_msg=f"H{suffix}")
        ref_hrs = g[f"d{dim}_n{ngl}_Hrs{suffix}"]
        np.testing.assert_allclose(np.asarray(getattr(sp, "Hrs" + suffix)), ref_hrs, rtol=0,
                                   atol=1e-14 * max(1.0, np.abs(ref_hrs).max()), err_msg=f"Hrs{suffix}")
    np.testing.assert_allclose(np.asarray(sp.HCoo1D), g[f"d{dim}_n{ngl}_HCoo1D"], rtol=0, atol=1e-14)


@pytest.mark.parametrize("dim,ngl", [(2, 6), (2, 7), (2, 11), (3, 5), (3, 6)])
def test_spectral_orderings_high_order(golden, dim, ngl):
    """vertex -> edge -> face -> interior ordering of the points (spectral.py:346-431) up to the Taylor-Green order"""
    g = golden["g2_tables"]
    sp = Spectral(ngl, dim)
    for suffix in ("Op", "", "Red"):
        np.testing.assert_allclose(_gps_array(getattr(sp, "gps" + suffix), dim), g[f"d{dim}_n{ngl}_gps{suffix}"], rtol=0, atol=1e-13)


def test_partition_of_unity_and_kronecker():
    """src/tests/test_element.py:60-133: sum_a H = 1, sum_a Hrs = 0 at every point; nodal rule: H is the identity"""
    for dim, ngl in ((2, 3), (2, 5), (3, 2), (3, 4)):
        sp = Spectral(ngl, dim)
        for suffix in SUFFIXES:
            H, Hrs = np.asarray(getattr(sp, "H" + suffix)), np.asarray(getattr(sp, "Hrs" + suffix))
            np.testing.assert_allclose(H.sum(axis=1), 1.0, atol=1e-13)
            np.testing.assert_allclose(Hrs.sum(axis=2), 0.0, atol=1e-12)
        np.testing.assert_allclose(np.asarray(sp.HOp), np.eye(ngl ** dim), atol=1e-13)
