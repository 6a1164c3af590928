"""High-order nodes on imported (corner-node) quadrilateral / hexahedral meshes (IndicesManager.
mapEntitiesToNodes with its orientation reversals, src/domain/indices.py:66-88): `_lift_high_order` numbers edge / face /
interior nodes from exact integer keys.  The check that matters is consistency across cells that see a shared edge or face
in DIFFERENT orientations: every cell's multilinear image of the GLL points must equal the coordinates its connectivity
points at.  Cells are therefore rotated at random (all 4 / 24 orientation-preserving relabelings of the reference cell) and
the vertices renumbered at random.  Host only."""
from itertools import permutations, product

import numpy as np
import pytest

from oracle import fem_oracle as fo
from pynama_amd.domain.dmplex import _lift_high_order
from pynama_amd.elements.spectral import Spectral, _local_lattice


def _rotations(dim):
    """corner permutations of the reference cell induced by its orientation-preserving symmetries"""
    clat = np.array(_local_lattice(2, dim))
    at = {tuple(c): k for k, c in enumerate(clat)}
    out = []
    for perm in permutations(range(dim)):
        for flips in product((0, 1), repeat=dim):
            sign = (-1) ** (sum(flips) + sum(1 for a in range(dim) for b in range(a + 1, dim) if perm[a] > perm[b]))
            if sign < 0:
                continue                                           # keep det J > 0
            img = [at[tuple((c[perm[d]] ^ flips[d]) for d in range(dim))] for c in clat]
            out.append(np.array(img))
    return out


@pytest.mark.parametrize("dim,nelem,ngl", [(2, [5, 4], 3), (2, [3, 3], 6), (3, [3, 2, 4], 3), (3, [2, 3, 2], 4), (3, [2, 2, 2], 5)])
def test_lifted_mesh_is_consistent_under_random_cell_orientations(dim, nelem, ngl):
    rng = np.random.default_rng(7 + dim + ngl)
    m1 = fo.box_mesh(nelem, [0.0] * dim, [1.0, 0.7, 1.3][:dim], 2, jitter=0.15)
    rots = _rotations(dim)
    assert len(rots) == (4 if dim == 2 else 24)
    conn = np.stack([c[rots[rng.integers(len(rots))]] for c in m1.conn.astype(np.int64)])
    perm = rng.permutation(m1.n_node)                              # new id of old vertex
    conn, xyz = perm[conn], m1.xyz[np.argsort(perm)]
    ch, xh, ho = _lift_high_order(conn, xyz, ngl, dim)
    want_nodes = int(np.prod([n * (ngl - 1) + 1 for n in nelem]))
    assert xh.shape[0] == want_nodes and ch.shape == (conn.shape[0], ngl ** dim)
    assert np.array_equal(ch[:, :2 ** dim], conn)                   # corners first, ids kept
    # every cell's own image of the GLL points == the coordinates its connectivity points at
    H = np.asarray(Spectral(ngl, dim).HCooOp)
    img = np.einsum("gc,ecd->egd", H, xyz[conn])
    assert np.abs(img - xh[ch]).max() < 1e-13
    # no node twice: all coordinates distinct
    key = np.round(xh / 1e-9).astype(np.int64)
    assert np.unique(key, axis=0).shape[0] == want_nodes
    # exterior facets: exactly the nodes on the box faces (the jitter keeps boundary nodes on the faces)
    ext = np.unique(ho["ext_nodes"])
    up = np.array([1.0, 0.7, 1.3][:dim])
    on_face = np.any((np.abs(xh) < 1e-12) | (np.abs(xh - up) < 1e-12), axis=1)
    assert np.array_equal(ext, np.nonzero(on_face)[0])
    assert ho["ext_nodes"].shape[1] == ngl ** (dim - 1) and ho["ext_corners"].shape[1] == 2 ** (dim - 1)


def test_lifted_mesh_through_dmplexdom_borders():
    """DMPlexDom(mesh=...) with ngl 3: named borders by position, node counts, corner coordinates untouched"""
    from pynama_amd.domain.dmplex import DMPlexDom
    m1 = fo.box_mesh([3, 2, 2], [0, 0, 0], [1, 1, 1], 2)
    dom = DMPlexDom(mesh={"dim": 3, "xyz": m1.xyz.copy(), "conn": m1.conn.astype(np.int32), "facets": [], "cell": "tensor"})
    dom.setFemIndexing(3)
    assert dom.xyz.shape[0] == 7 * 5 * 5 and dom.conn.shape == (12, 27)
    names = dom.getBordersNames()
    left = dom.getBorderNodes("left")
    assert len(names) == 6 and len(left) == 5 * 5
    allb = dom.getNodesFromLabel("External Boundary")
    assert len(allb) == 7 * 5 * 5 - 5 * 3 * 3
