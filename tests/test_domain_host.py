"""Host logic of the product's structured DMPlexDom (no GPU needed): the conventions the reference
asserts in src/tests/test_domain.py, checked on pynama_amd.domain.dmplex AND against the oracle's
independent mesh generator; plus the slab partition / halo plan used for multi-GPU runs."""
import numpy as np
import pytest

from oracle import fem_oracle as fo
from pynama_amd.common.comm import Comm
from pynama_amd.domain.dmplex import DMPlexDom


def make(nelem, lower, upper, ngl, rank=0, size=1, **kw):
    dom = DMPlexDom(boxMesh={'nelem': nelem, 'lower': lower, 'upper': upper}, comm=Comm(rank, size), **kw)
    dom.setFemIndexing(ngl)
    return dom


def test_box_2d():                                 # test_domain.py:14-53
    dom = make([3, 4], [0, 0], [0.6, 0.8], 3)
    assert dom.getDimension() == 2
    assert (dom.cellStart, dom.cellEnd) == (0, 12)
    np.testing.assert_allclose(dom.getCellCornersCoords(0), [0, 0, 0.2, 0, 0.2, 0.2, 0, 0.2], atol=1e-13)
    dom.setLabelToBorders()
    b1, b2 = dom.getBordersNodes(), dom.getNodesFromLabel("External Boundary")
    assert isinstance(b1, set) and len(b1) == 28 and b1 == b2
    for b in dom.getBordersNames():
        assert len(dom.getBorderNodes(b)) == (7 if b in ('up', 'down') else 9)
    with pytest.raises(Exception):
        dom.getCellCornersCoords(12)


def test_box_3d():                                 # test_domain.py:84-136
    dom = make([3, 4, 5], [0, 0, 0], [0.6, 0.8, 1], 3)
    assert (dom.cellStart, dom.cellEnd) == (0, 60)
    exp = [0, 0, 0, 0, 0.2, 0, 0.2, 0.2, 0, 0.2, 0, 0, 0, 0, 0.2, 0.2, 0, 0.2, 0.2, 0.2, 0.2, 0, 0.2, 0.2]
    np.testing.assert_allclose(dom.getCellCornersCoords(0), exp, atol=1e-13)
    names = dom.getBordersNames()
    assert len(names) == 6 and set(names) == {'up', 'down', 'left', 'right', 'front', 'back'}
    assert len(dom.getBordersNodes()) == 28 * 11 + 35 * 2
    for b in names:
        n = 7 * 11 if b in ('up', 'down') else 9 * 11 if b in ('left', 'right') else 7 * 9
        assert len(dom.getBorderNodes(b)) == n


@pytest.mark.parametrize("ngl", range(2, 10, 2))
def test_ngl_indexing(ngl):                        # test_domain.py:55-78, 138-171
    d2 = make([2, 3], [0, 0], [0.6, 0.8], ngl)
    assert len(d2.getBordersNodes()) == 10 + 10 * (ngl - 2)
    for b in d2.getBordersNames():
        assert len(d2.getBorderNodes(b)) == (3 + 2 * (ngl - 2) if b in ('up', 'down') else 4 + 3 * (ngl - 2))
    d3 = make([2, 3, 4], [0, 0, 0], [0.6, 0.8, 1], ngl)
    k = ngl - 2
    assert len(d3.getBordersNodes()) == 54 + (36 + 68) * k + 52 * k * k
    for b in d3.getBordersNames():
        n = (15 + 22 * k + 8 * k * k if b in ('up', 'down') else
             20 + 31 * k + 12 * k * k if b in ('right', 'left') else 12 + 17 * k + 6 * k * k)
        assert len(d3.getBorderNodes(b)) == n


@pytest.mark.parametrize("ngl", range(2, 14))
def test_all_nodes_count(ngl):                     # test_domain.py:187-195
    dom = make([2, 3], [0, 0], [0.6, 0.8], ngl)
    assert len(dom.getAllNodes()) == 12 + 17 * (ngl - 2) + 6 * (ngl - 2) ** 2


def test_node_coordinates():                       # test_domain.py:197-201
    dom = make([2, 2], [0, 0], [1, 1], 2)
    exp = [[0, 0], [.5, 0], [1, 0], [0, .5], [.5, .5], [1, .5], [0, 1], [.5, 1], [1, 1]]
    np.testing.assert_allclose(dom.getNodesCoordinates(dom.getAllNodes()), exp, atol=1e-15)
    assert dom.getVelocityIndex([3, 7]) == [6, 7, 14, 15]        # indices.py:90-92


@pytest.mark.parametrize("nelem,ngl,jit", [([4, 3], 2, 0.0), ([3, 2, 4], 2, 0.2), ([2, 3], 4, 0.0), ([2, 2, 2], 3, 0.0)])
def test_matches_oracle_mesh(nelem, ngl, jit):
    dim = len(nelem)
    up = [1.0, 0.7, 1.3][:dim]
    dom = make(nelem, [0.0] * dim, up, ngl, jitter=jit)
    ref = fo.box_mesh(nelem, [0.0] * dim, up, ngl, jitter=jit)
    assert np.array_equal(dom.conn, ref.conn)
    np.testing.assert_allclose(dom.xyz, ref.xyz, atol=1e-15)
    assert np.array_equal(np.nonzero(dom.boundaryMaskLocal())[0], ref.boundary)
    for name, ids in ref.borders.items():
        assert sorted(dom.getBorderNodes(name)) == list(ids)


@pytest.mark.parametrize("nelem,ngl,size", [([4, 9], 2, 2), ([3, 3, 8], 2, 3), ([3, 7], 3, 2), ([2, 2, 6], 3, 4)])
def test_slab_partition_and_halo_plan(nelem, ngl, size):
    dim = len(nelem)
    doms = [make(nelem, [0.0] * dim, [1.0] * dim, ngl, rank=r, size=size) for r in range(size)]
    glob = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, ngl)
    # ownership: contiguous row blocks covering all nodes
    assert doms[0].rStart == 0 and doms[-1].rEnd == glob.n_node
    for r in range(size - 1):
        assert doms[r].rEnd == doms[r + 1].rStart
    plans = [d._halo_plan() for d in doms]
    for r, d in enumerate(doms):
        n_owned, n_ghost, neigh, sp_, sidx, rp_ = plans[r]
        assert n_owned == d.nOwned and n_ghost == d.nGhost
        # local -> global is a bijection onto owned + ghost; coordinates agree with the global mesh
        g = d._local2global(np.arange(d.nLocal))
        assert len(np.unique(g)) == d.nLocal
        assert np.array_equal(d._global2local(g), np.arange(d.nLocal))
        np.testing.assert_allclose(d.xyz, glob.xyz[g], atol=1e-15)
        # every element touching an owned node is local, with the right connectivity
        touching = np.nonzero(((glob.conn >= d.rStart) & (glob.conn < d.rEnd)).any(axis=1))[0]
        local_as_global = g[d.conn]
        assert len(touching) == d.conn.shape[0]
        assert np.array_equal(np.sort(local_as_global, axis=0), np.sort(glob.conn[touching], axis=0))
        # halo symmetry: what r sends to nb is exactly what nb expects to receive from r, in order
        for k, nb in enumerate(neigh):
            sent_global = d._local2global(sidx[sp_[k]:sp_[k + 1]])
            n2 = plans[nb]
            k2 = list(n2[2]).index(r)
            ghosts = doms[nb]._local2global(np.arange(n2[0] + n2[5][k2], n2[0] + n2[5][k2 + 1]))
            assert np.array_equal(sent_global, ghosts)


def test_ksp_solver_options_host():
    """PETSc option names reach the solver facade; -pynama_mat_free is a flag (no GPU needed: nothing is solved)"""
    from pynama_amd.common.options import Options
    from pynama_amd.solver.ksp_solver import KspSolver
    try:
        Options(["-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1e-9", "-pynama_mat_free"])
        k = KspSolver()
        k.createSolver(None, None)
        assert (k.ksp_type, k.pc_type, k.rtol, k.mat_free) == ("cg", "jacobi", 1e-9, True)
        Options(["-ksp_type", "gmres", "-pc_type", "none", "-pynama_mat_free", "0"])
        k = KspSolver()
        k.createSolver(None, None)
        assert (k.ksp_type, k.pc_type, k.mat_free) == ("gmres", "none", False)
        Options(["-ksp_type", "cg", "-pc_type", "jacobi"])
        k = KspSolver()
        k.createSolver(None, None)
        assert k.mat_free is None                              # automatic: the shell when the matrix carries one
    finally:
        Options([])


@pytest.mark.parametrize("nelem,ngl,size", [([5, 4], 2, 1), ([4, 6], 3, 2), ([3, 4, 5], 2, 1), ([3, 2, 6], 3, 3), ([2, 2, 4], 4, 2)])
def test_boundary_mask_slices_equal_the_per_node_lattice_indices(nelem, ngl, size):
    """boundaryMaskLocal marks the borders as slices of the (planes, y, x) lattice of the local nodes (owned planes first, then the
    ghosts); the per-node lattice indices give the same mask, and host conn / xyz exist only once somebody asks for them"""
    from pynama_amd.common.comm import Comm
    from pynama_amd.domain.dmplex import DMPlexDom
    dim = len(nelem)
    for r in range(size):
        dom = DMPlexDom(boxMesh={'nelem': nelem, 'lower': [0.0] * dim, 'upper': [1.0, 2.0, 0.5][:dim]}, comm=Comm(r, size))
        dom.setFemIndexing(ngl)
        assert dom._conn is None and dom._xyz is None
        bm = dom.boundaryMaskLocal()
        assert dom._conn is None and dom._xyz is None and bm.shape == (dom.nLocal,)
        on = np.zeros(dom.nLocal, bool)
        for d in range(dim):
            on |= (dom._lat_idx[d] == 0) | (dom._lat_idx[d] == dom.lattice[d] - 1)
        assert np.array_equal(bm.astype(bool), on)
        # coordinates of a few nodes without the full array == rows of the full array
        ln = np.array([0, dom.nLocal - 1, dom.nLocal // 2])
        X = dom._coords_of_local(ln)
        assert dom._xyz is None
        assert np.array_equal(X, dom.xyz[ln])
        assert dom.conn.shape == ((dom._layers[1] - dom._layers[0]) * int(np.prod(nelem[:-1])), ngl ** dim)
