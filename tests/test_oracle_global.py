"""Global-level oracle checks: the analytic assertions the reference makes end-to-end
(/root/reference/src/tests/test_solver.py:20-27, 29-37, 52-62), evaluated on the oracle's
assembly + solve; plus structural checks of the mesh conventions
(/root/reference/src/tests/test_domain.py:26-30,94-104,52-78,138-171,187-201)."""
from math import cos, exp, pi, sin

import numpy as np
import pytest

from oracle import fem_oracle as fo


def _uniform(nelem, ngl, method, **kw):
    dim = len(nelem)
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, ngl)
    tb = fo.Tables(ngl, dim)
    mats = fo.assemble_kle_freeslip(mesh, tb)
    cte = np.array([1.0, 0.0, 0.0][:dim])
    vel = np.zeros(mesh.n_node * dim)
    vel[fo.dof_indices(mesh.boundary[:, None], dim).ravel()] = np.tile(cte, len(mesh.boundary))
    vort = np.zeros(mesh.n_node * tb.dim_w)
    x, rhs = fo.solve_kle(mats, vort, vel, method=method, **kw)
    exact = np.tile(cte, mesh.n_node)
    return np.linalg.norm(x - exact), mats, x, rhs


def test_solveKLE_uniform_2d_lu():                # test_solver.py:20-27
    err, *_ = _uniform([10, 10], 3, "lu")
    assert err < 1e-12


def test_solveKLE_uniform_3d_lu():                # test_solver.py:52-62
    err, *_ = _uniform([3, 3, 3], 3, "lu")
    assert err < 2e-13


def test_solveKLE_uniform_2d_cg_residual():       # BASELINE: residual <= 1e-10
    err, mats, x, rhs = _uniform([10, 10], 3, "cg", rtol=1e-12, norm_type=fo.NORM_UNPRECONDITIONED)
    assert np.linalg.norm(rhs - mats["K"] @ x) / np.linalg.norm(rhs) < 1e-10
    assert err < 1e-8


def test_solveKLE_uniform_2d_gmres():
    err, mats, x, rhs = _uniform([6, 6], 2, "gmres", rtol=1e-12, restart=30)
    assert np.linalg.norm(rhs - mats["K"] @ x) / np.linalg.norm(rhs) < 1e-9


def _tg_vel(c):                                    # cases/custom_func.py:173-183 at t=0
    x_, y_ = 2 * pi * c[0], 2 * pi * c[1]
    return [cos(x_) * sin(y_), -sin(x_) * cos(y_)]


def _tg_vort(c):                                   # cases/custom_func.py:185-193 at t=0
    x_, y_ = 2 * pi * c[0], 2 * pi * c[1]
    return [-2 * pi * 2.0 * cos(x_) * cos(y_)]


def test_solveKLE_taylorgreen_ngl11():            # test_solver.py:29-37
    mesh = fo.box_mesh([2, 2], [0, 0], [1, 1], 11)
    tb = fo.Tables(11, 2)
    mats = fo.assemble_kle_freeslip(mesh, tb)
    exact_v = np.array([_tg_vel(c) for c in mesh.xyz]).ravel()
    exact_w = np.array([_tg_vort(c) for c in mesh.xyz]).ravel()
    vel = np.zeros_like(exact_v)
    bd = fo.dof_indices(mesh.boundary[:, None], 2).ravel()
    vel[bd] = exact_v[bd]
    x, _ = fo.solve_kle(mats, exact_w, vel, method="lu")
    assert np.linalg.norm(x - exact_v) < 2e-8


def test_K_symmetric_spd_on_free():
    mesh = fo.box_mesh([3, 3, 3], [0, 0, 0], [1, 1, 1], 2, jitter=0.2)
    mats = fo.assemble_kle_freeslip(mesh, fo.Tables(2, 3))
    K = mats["K"]
    assert abs(K - K.T).max() < 1e-10
    assert np.linalg.eigvalsh(K.toarray()).min() > 0


# ---- mesh conventions -------------------------------------------------------------------
def test_cell0_corners_2d():                      # test_domain.py:26-30
    m = fo.box_mesh([3, 4], [0, 0], [0.6, 0.8], 3)
    np.testing.assert_allclose(m.corners()[0], [0, 0, 0.2, 0, 0.2, 0.2, 0, 0.2], atol=1e-13)
    assert m.n_elem == 12
    assert len(m.boundary) == 28                   # :32-45
    for b, ids in m.borders.items():               # :47-53
        assert len(ids) == (7 if b in ("up", "down") else 9)


def test_cell0_corners_3d():                      # test_domain.py:94-104
    m = fo.box_mesh([3, 4, 5], [0, 0, 0], [0.6, 0.8, 1.0], 3)
    exp = [0, 0, 0, 0, 0.2, 0, 0.2, 0.2, 0, 0.2, 0, 0, 0, 0, 0.2, 0.2, 0, 0.2, 0.2, 0.2, 0.2, 0, 0.2, 0.2]
    np.testing.assert_allclose(m.corners()[0], exp, atol=1e-13)
    assert len(m.boundary) == 28 * 11 + 35 * 2     # :114-126
    for b, ids in m.borders.items():               # :128-136
        n = 7 * 11 if b in ("up", "down") else 9 * 11 if b in ("left", "right") else 7 * 9
        assert len(ids) == n


@pytest.mark.parametrize("ngl", range(2, 10, 2))
def test_border_counts_3d(ngl):                   # test_domain.py:147-171
    m = fo.box_mesh([2, 3, 4], [0, 0, 0], [0.6, 0.8, 1.0], ngl)
    assert len(m.boundary) == 54 + (36 + 68) * (ngl - 2) + 52 * (ngl - 2) ** 2
    for b, ids in m.borders.items():
        k = ngl - 2
        n = (15 + 22 * k + 8 * k * k if b in ("up", "down") else
             20 + 31 * k + 12 * k * k if b in ("right", "left") else 12 + 17 * k + 6 * k * k)
        assert len(ids) == n


@pytest.mark.parametrize("ngl", range(2, 14))
def test_total_nodes_2d(ngl):                     # test_domain.py:187-195
    m = fo.box_mesh([2, 3], [0, 0], [0.6, 0.8], ngl)
    assert m.n_node == 12 + 17 * (ngl - 2) + 6 * (ngl - 2) ** 2


def test_node_coordinates_lexicographic():        # test_domain.py:197-201
    m = fo.box_mesh([2, 2], [0, 0], [1, 1], 2)
    exp = [[0, 0], [.5, 0], [1, 0], [0, .5], [.5, .5], [1, .5], [0, 1], [.5, 1], [1, 1]]
    np.testing.assert_allclose(m.xyz, exp, atol=1e-15)


def test_node_graph_matches_assembled_pattern():
    m = fo.box_mesh([3, 3, 3], [0, 0, 0], [1, 1, 1], 2)
    rp, ci = fo.node_graph(m)
    # interior node of a Q1 hex mesh has 27 neighbours
    lens = np.diff(rp)
    assert lens.max() == 27 and lens.min() == 8
    assert rp[-1] == (3 * 3 + 1) ** 3              # nnz = (3n+1)^3 (385^3 at n=128)


@pytest.mark.parametrize("dim", [2, 3])
def test_operators_exact_on_linear_fields(dim):
    """Curl / SrT operators (base_problem.py:132-140, mat_generator.py:157-190) reproduce the
    derivatives of a linear velocity field exactly at every node (lumped projection of a constant)."""
    nelem = [4, 3, 3][:dim]
    mesh = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.8, 1.2][:dim], 2, jitter=0.15)
    tb = fo.Tables(2, dim)
    ops = fo.assemble_operators(mesh, tb)
    Agrad = np.array([[0.3, -1.0, 0.2], [1.5, 0.1, -0.4], [0.7, 0.6, -0.4]])[:dim, :dim]
    v = (mesh.xyz @ Agrad.T).ravel()
    curl = (ops["Curl"] @ v).reshape(mesh.n_node, -1)
    if dim == 2:
        np.testing.assert_allclose(curl[:, 0], Agrad[1, 0] - Agrad[0, 1], atol=1e-12)
    else:
        w = [Agrad[2, 1] - Agrad[1, 2], Agrad[0, 2] - Agrad[2, 0], Agrad[1, 0] - Agrad[0, 1]]
        np.testing.assert_allclose(curl, np.tile(w, (mesh.n_node, 1)), atol=1e-12)
    # constant tensor field has zero divergence only in the interior (boundary rows see one-sided sums)
    srt = (ops["SrT"] @ v).reshape(mesh.n_node, -1)
    assert np.abs(srt - srt[0]).max() < 1e-12
    # the reference's shortcut (cell-0 blocks for every cell) agrees on a uniform mesh
    um = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, 2)
    a, b = fo.assemble_operators(um, tb), fo.assemble_operators(um, tb, cell0_only=True)
    for k in ("SrT", "DivSrT", "Curl"):
        assert abs(a[k] - b[k]).max() < 1e-12
