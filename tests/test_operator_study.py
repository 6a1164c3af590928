"""Known answers of the KLE operator chain: the reference's operator study (`run_case.py -test operators`,
src/cases/custom_func.py:110-170) applies Curl, Curl Div(v (x) v) and Curl Div(2 mu S(v)) / rho to analytic fields and
reports the lumped-mass L2 errors.  Here the same study is restated on the sinusoidal 2-D field of custom_func.py:276-310
(v = (sin 2 pi y, sin 4 pi x); its curl, (v . grad) w and nu lap(w) written out in this file, not copied) and on the 3-D
Taylor-Green vortex (:196-272 with unit box), with the spectral convergence the reference plots as the assertion.
CPU half: the oracle's operators.  GPU half: the product's CustomFuncCase.OperatorsTests equals the oracle's numbers."""
import os

import numpy as np
import pytest
import yaml

from oracle import fem_oracle as fo
from tests.util import device_available

PI = np.pi
MU, RHO = 0.01, 0.5          # cases/taylor-green.yaml: material-properties


def sinus_fields(X, nu):
    x, y = X[:, 0], X[:, 1]
    vel = np.stack([np.sin(2 * PI * y), np.sin(4 * PI * x)], 1)
    vort = 4 * PI * np.cos(4 * PI * x) - 2 * PI * np.cos(2 * PI * y)                         # dx v_y - dy v_x
    conv = ((2 * PI) ** 2 - (4 * PI) ** 2) * np.sin(2 * PI * y) * np.sin(4 * PI * x)           # (v . grad) w
    diff = nu * ((2 * PI) ** 3 * np.cos(2 * PI * y) - (4 * PI) ** 3 * np.cos(4 * PI * x))      # nu lap(w)
    return vel, vort, conv, diff


def oracle_errors(nel, ngl):
    mesh = fo.box_mesh([nel, nel], [0, 0], [1, 1], ngl)
    ops = fo.assemble_operators(mesh, fo.Tables(ngl, 2))
    vel, vort, conv, diff = sinus_fields(mesh.xyz, MU / RHO)
    v = vel.ravel()
    vv = np.stack([vel[:, 0] ** 2, vel[:, 0] * vel[:, 1], vel[:, 1] ** 2], 1).ravel()            # base_problem.py:234-252
    l2 = lambda e: float(np.sqrt((e * e) @ ops["weights"]))
    return (l2(ops["Curl"] @ (ops["DivSrT"] @ vv) - conv),
            l2(ops["Curl"] @ ((ops["DivSrT"] @ (2 * MU * (ops["SrT"] @ v))) / RHO) - diff),
            l2(ops["Curl"] @ v - vort))


def test_oracle_operator_chain_converges_spectrally():
    """p-refinement on 4 x 4 elements: every two orders gain about two digits (what the reference's loglog plots show)"""
    e7, e9, e11 = oracle_errors(4, 7), oracle_errors(4, 9), oracle_errors(4, 11)
    for k, bound in enumerate((1e-5, 1e-4, 1e-7)):        # convective, diffusive, curl at ngl 11 (field maxima 118 / 45 / 19)
        assert e11[k] < bound
        assert e9[k] < e7[k] / 30 and e11[k] < e9[k] / 30


@pytest.mark.gpu
@pytest.mark.skipif(not device_available(), reason="needs an MI355X")
def test_product_operator_study_equals_oracle():
    import pynama_amd
    pynama_amd.install_reference_layout()
    from cases.custom_func import CustomFuncCase
    with open(os.path.join(os.path.dirname(pynama_amd.__file__), "cases", "taylor-green.yaml")) as f:
        cfg = yaml.load(f, Loader=yaml.Loader)
    cfg["name"] = "senoidal"
    fem = CustomFuncCase(cfg, case="senoidal", nelem=[4, 4], ngl=9)
    fem.setUp()
    fem.setUpSolver()
    got = fem.OperatorsTests()
    want = oracle_errors(4, 9)
    # the convective chain acts on the KLE SOLUTION (custom_func.py:137-138), which carries its own discretisation error
    assert got[0] < 5 * want[0] + 1e-3
    assert abs(got[1] - want[1]) < 1e-8 * (1 + want[1]) and abs(got[2] - want[2]) < 1e-8 * (1 + want[2])
    # the fields themselves: diffusive chain and Curl of the exact velocity against the oracle's operators, entry by entry
    exactVel, exactVort, exactConv, exactDiff = fem.generateExactOperVecs(0.0)
    mesh = fo.box_mesh([4, 4], [0, 0], [1, 1], 9)
    ops = fo.assemble_operators(mesh, fo.Tables(9, 2))
    v = exactVel.getArray()
    np.testing.assert_allclose((fem.operator.Curl * exactVel).getArray(), ops["Curl"] @ v, rtol=0, atol=1e-9)
    np.testing.assert_allclose(fem.getDiffusive(exactVel, exactDiff).getArray(),
                               ops["Curl"] @ ((ops["DivSrT"] @ (2 * MU * (ops["SrT"] @ v))) / RHO), rtol=0, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.skipif(not device_available(), reason="needs an MI355X")
def test_product_taylor_green_3d_operators():
    """3-D Taylor-Green (custom_func.py:196-272): Curl, convective and diffusive chains at ngl 7 on 2^3 elements"""
    import pynama_amd
    pynama_amd.install_reference_layout()
    from cases.custom_func import CustomFuncCase
    with open(os.path.join(os.path.dirname(pynama_amd.__file__), "cases", "taylor-green.yaml")) as f:
        cfg = yaml.load(f, Loader=yaml.Loader)
    fem = CustomFuncCase(cfg, case="taylor-green", nelem=[2, 2, 2], lower=[0, 0, 0], upper=[1, 1, 1], ngl=7)
    fem.setUp()
    fem.setUpSolver()
    exactVel, exactVort, exactConv, exactDiff = fem.generateExactOperVecs(0.0)
    wei = fem.operator.lumpedWeights(3)
    l2 = lambda a, b: float(np.sqrt(((a - b) * (a - b)).dot(wei)))
    nrm = lambda a: float(np.sqrt((a * a).dot(wei)))
    fem.vel = exactVel.copy() if hasattr(exactVel, "copy") else exactVel
    assert l2(fem.operator.Curl * exactVel, exactVort) < 2e-2 * nrm(exactVort)
    assert l2(fem.getDiffusive(exactVel, exactDiff), exactDiff) < 2e-1 * nrm(exactDiff)
    assert l2(fem.getConvective(exactVel, exactConv), exactConv) < 2e-1 * nrm(exactConv)
