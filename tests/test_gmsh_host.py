"""Gmsh import: reader, corner-order mapping, border detection.  Host only."""
import numpy as np
import pytest

from oracle import fem_oracle as fo
from pynama_amd.common.comm import Comm
from pynama_amd.domain.dmplex import DMPlexDom
from pynama_amd.domain.gmsh import exterior_facets, read_msh, write_msh

# two unit hexahedra side by side, written by hand in Gmsh's own corner order, with the six box faces
# tagged 1..6 in the order of the reference's namingConvention (back, front, down, up, right, left)
TWO_HEX = """$MeshFormat
2.2 0 8
$EndMeshFormat
$Nodes
12
1 0 0 0
2 1 0 0
3 2 0 0
4 0 1 0
5 1 1 0
6 2 1 0
7 0 0 1
8 1 0 1
9 2 0 1
10 0 1 1
11 1 1 1
12 2 1 1
$EndNodes
$Elements
12
1 3 2 1 1 1 2 5 4
2 3 2 1 1 2 3 6 5
3 3 2 2 2 7 8 11 10
4 3 2 2 2 8 9 12 11
5 3 2 3 3 1 2 8 7
6 3 2 3 3 2 3 9 8
7 3 2 4 4 4 5 11 10
8 3 2 4 4 5 6 12 11
9 3 2 5 5 3 6 12 9
10 3 2 6 6 1 4 10 7
11 5 2 0 0 1 2 5 4 7 8 11 10
12 5 2 0 0 2 3 6 5 8 9 12 11
$EndElements
"""


def test_read_hand_written_hex_file(tmp_path):
    p = tmp_path / "two.msh"
    p.write_text(TWO_HEX)
    m = read_msh(str(p))
    assert m["dim"] == 3 and m["conn"].shape == (2, 8) and len(m["facets"]) == 10
    # reference corner order: positive Jacobian with the reference's tables
    tb = fo.Tables(2, 3)
    A = fo.elem_mass(tb, m["xyz"][m["conn"]], rule="full")
    assert np.allclose(A.sum(axis=(1, 2)), 1.0)                  # volume of each unit cell
    ref = fo.box_mesh([1, 1, 1], [0, 0, 0], [1, 1, 1], 2)
    assert np.allclose(m["xyz"][m["conn"][0]], ref.xyz[ref.conn[0]])
    d = DMPlexDom(fileName=str(p), comm=Comm(), reorder=None)
    d.setFemIndexing(2)
    assert d.getDimension() == 3 and d.nOwned == 12
    assert d.boundaryMaskLocal().all()                           # no interior node in a 2x1x1 box
    assert sorted(d.getBorderNodes("right")) == [2, 5, 8, 11]    # x = 2
    assert sorted(d.getBorderNodes("left")) == [0, 3, 6, 9]
    assert sorted(d.getBorderNodes("back")) == [0, 1, 2, 3, 4, 5]
    assert sorted(d.getBorderNodes("up")) == [3, 4, 5, 9, 10, 11]


@pytest.mark.parametrize("nelem", [[4, 3], [3, 2, 4]])
def test_roundtrip_permuted_box(tmp_path, nelem):
    dim = len(nelem)
    box = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.8, 1.3][:dim], 2, jitter=0.2)
    rng = np.random.default_rng(4)
    perm = rng.permutation(box.n_node)
    xyz = box.xyz[np.argsort(perm)]
    conn = perm[box.conn][rng.permutation(box.n_elem)]
    p = str(tmp_path / "box.msh")
    write_msh(p, xyz, conn)                                       # no tagged facets: borders by position
    d = DMPlexDom(fileName=p, comm=Comm(), reorder=None)          # keep the file's numbering
    d.setFemIndexing(2)
    assert np.array_equal(d.conn, conn) and np.allclose(d.xyz, xyz)
    assert set(np.nonzero(d.boundaryMaskLocal())[0]) == set(perm[box.boundary])
    for name, ids in box.borders.items():
        assert sorted(d.getBorderNodes(name)) == sorted(perm[ids])
    assert len(exterior_facets(conn, dim)) == 2 * sum(int(np.prod(nelem)) // n for n in nelem)
    ptr, rows = d.patchPlan((7, 7, 7)[:dim])
    assert ptr[0] == 0 and ptr[-1] == d.nOwned and np.array_equal(rows, np.arange(d.nOwned))


def test_rejects_unsupported(tmp_path):
    p = tmp_path / "tri.msh"
    p.write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n3\n1 0 0 0\n2 1 0 0\n3 0 1 0\n$EndNodes\n"
                 "$Elements\n1\n1 9 2 0 0 1 2 3 1 2 3\n$EndElements\n")     # 6-node triangle
    with pytest.raises(ValueError, match="only first-order"):
        read_msh(str(p))
    q = tmp_path / "v4.msh"
    q.write_text("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n")
    with pytest.raises(ValueError, match="MSH 2.x"):
        read_msh(str(q))
    box = fo.box_mesh([2, 2], [0, 0], [1, 1], 2)
    r = str(tmp_path / "b.msh")
    write_msh(r, box.xyz, box.conn)
    d = DMPlexDom(fileName=r, comm=Comm())
    d.setFemIndexing(3)                      # quadrilaterals / hexahedra take any ngl (tests/test_highorder_import_host.py)
    assert d.conn.shape == (4, 9) and d.xyz.shape[0] == 25
    from pynama_amd.domain.gmsh import write_msh as _w
    tri = fo.simplex_box_mesh([2, 2], [0, 0], [1, 1])
    t = str(tmp_path / "t.msh")
    _w(t, tri.xyz, tri.conn)
    with pytest.raises(NotImplementedError, match="quadrilateral / hexahedral"):
        DMPlexDom(fileName=t, comm=Comm()).setFemIndexing(3)


@pytest.mark.parametrize("cell,nelem,size", [("tet", [4, 3, 5], 3), ("hex", [4, 4, 6], 2), ("tri", [7, 6], 4), ("tet", [5, 5, 5], 5)])
def test_row_block_partition_of_imported_mesh(tmp_path, cell, nelem, size):
    """imported meshes: Morton renumbering is a pure permutation; contiguous row blocks; every rank holds the
    cells touching an owned node; ghosts addressed through a sorted index list; halo plans are symmetric
    between ANY pair of ranks (not only rank +-1 as for the box slabs)."""
    dim = len(nelem)
    up = [1.0, 0.8, 1.2][:dim]
    if cell == "hex":
        src = fo.box_mesh(nelem, [0.0] * dim, up, 2, jitter=0.2)
    else:
        src = fo.simplex_box_mesh(nelem, [0.0] * dim, up, jitter=0.2)
    rng = np.random.default_rng(8)
    perm = rng.permutation(src.n_node)
    p = str(tmp_path / "m.msh")
    write_msh(p, src.xyz[np.argsort(perm)], perm[src.conn])
    doms = [DMPlexDom(fileName=p, comm=Comm(r, size)) for r in range(size)]
    for d in doms:
        d.setFemIndexing(2)
    one = DMPlexDom(fileName=p, comm=Comm())
    one.setFemIndexing(2)                                   # the global mesh in the build's (Morton) numbering
    assert one.nOwned == src.n_node and one.nGhost == 0
    # the renumbering only permutes: same node set, same cells
    key = lambda x: np.round(x * 1e9).astype(np.int64)
    assert set(map(tuple, key(one.xyz))) == set(map(tuple, key(src.xyz)))
    cells_a = np.sort(np.sort(key(one.xyz[one.conn]).reshape(one.conn.shape[0], -1), axis=1), axis=0)
    cells_b = np.sort(np.sort(key(src.xyz[src.conn]).reshape(src.n_elem, -1), axis=1), axis=0)
    assert np.array_equal(cells_a, cells_b)
    assert doms[0].rStart == 0 and doms[-1].rEnd == src.n_node
    plans = [d._halo_plan() for d in doms]
    far = False
    for r, d in enumerate(doms):
        if r:
            assert doms[r - 1].rEnd == d.rStart
        n_owned, n_ghost, neigh, sp_, sidx, rp_ = plans[r]
        assert n_owned == d.nOwned and n_ghost == d.nGhost and rp_[-1] == n_ghost
        g = d._local2global(np.arange(d.nLocal))
        assert len(np.unique(g)) == d.nLocal
        assert np.array_equal(d._global2local(g), np.arange(d.nLocal))
        np.testing.assert_array_equal(d.xyz, one.xyz[g])
        touching = ((one.conn >= d.rStart) & (one.conn < d.rEnd)).any(axis=1)
        assert np.array_equal(g[d.conn], one.conn[touching])
        assert np.array_equal(d.boundaryMaskLocal(), one.boundaryMaskLocal()[g])
        for name in one.getBordersNames():
            assert np.array_equal(d._on_border_mask(name), one._on_border_mask(name)[g])
        far |= any(abs(int(nb) - r) > 1 for nb in neigh)
        for k, nb in enumerate(neigh):
            sent_global = d._local2global(sidx[sp_[k]:sp_[k + 1]])
            n2 = plans[nb]
            k2 = list(n2[2]).index(r)
            ghosts = doms[nb]._local2global(np.arange(n2[0] + n2[5][k2], n2[0] + n2[5][k2 + 1]))
            assert np.array_equal(sent_global, ghosts)
    if size >= 4:
        assert far                                          # Morton blocks do touch non-adjacent ranks


def test_simplex_files(tmp_path):
    """tetrahedra keep Gmsh's order; a negatively oriented cell is fixed on import"""
    m = fo.simplex_box_mesh([2, 2, 2], [0, 0, 0], [1, 1, 1])
    conn = m.conn.copy()
    conn[3, [2, 3]] = conn[3, [3, 2]]                       # flip one tetrahedron
    p = str(tmp_path / "t.msh")
    write_msh(p, m.xyz, conn)
    r = read_msh(p)
    assert r["cell"] == "simplex" and r["dim"] == 3
    X = r["xyz"][r["conn"]]
    assert (np.linalg.det(X[:, 1:] - X[:, :1]) > 0).all()
    assert np.array_equal(r["conn"], m.conn)
    assert len(exterior_facets(r["conn"], 3)) == 6 * 2 * 4   # 6 faces x 4 squares x 2 triangles
