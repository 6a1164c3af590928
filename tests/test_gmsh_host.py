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
    d = DMPlexDom(fileName=str(p), comm=Comm())
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
    d = DMPlexDom(fileName=p, comm=Comm())
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
                 "$Elements\n1\n1 2 2 0 0 1 2 3\n$EndElements\n")
    with pytest.raises(ValueError, match="only lines, quadrangles and hexahedra"):
        read_msh(str(p))
    q = tmp_path / "v4.msh"
    q.write_text("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n")
    with pytest.raises(ValueError, match="MSH 2.x"):
        read_msh(str(q))
    box = fo.box_mesh([2, 2], [0, 0], [1, 1], 2)
    r = str(tmp_path / "b.msh")
    write_msh(r, box.xyz, box.conn)
    d = DMPlexDom(fileName=r, comm=Comm())
    with pytest.raises(NotImplementedError, match="ngl must be 2"):
        d.setFemIndexing(3)
