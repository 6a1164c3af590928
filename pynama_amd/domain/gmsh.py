"""Gmsh MSH 2.2 (ASCII) reader for quadrilateral / hexahedral meshes.

The reference reads Gmsh files through PETSc (``DMPlex.createFromFile``, src/domain/dmplex.py:22-23) and
supports only tensor-product cells (src/domain/indices.py:116-122).  This reader returns what the device
path needs: node coordinates, the cell connectivity in the REFERENCE's local corner order
and the boundary facets with their physical tags (PETSc's "Face Sets")."""
import numpy as np

# Gmsh element type -> (nodes, topological dimension)
_TYPES = {1: (2, 1), 3: (4, 2), 5: (8, 3), 15: (1, 0)}
# Gmsh corner order -> reference / DMPlex closure order
#   quad: (x0,y0),(x1,y0),(x1,y1),(x0,y1) in both
#   hex : Gmsh bottom face counter-clockwise seen from +z, DMPlex closure walks it the other way round
_TO_REF = {3: [0, 1, 2, 3], 5: [0, 3, 2, 1, 4, 5, 6, 7]}
_FACETS = {2: [(0, 1), (1, 2), (2, 3), (3, 0)],
           3: [(0, 1, 2, 3), (4, 5, 6, 7), (0, 3, 5, 4), (1, 2, 6, 7), (2, 3, 5, 6), (0, 1, 7, 4)]}  # reference order


def read_msh(path):
    with open(path) as f:
        tok = f.read().split()
    pos = {t: i for i, t in enumerate(tok) if t.startswith("$") and not t.startswith("$End")}
    if "$MeshFormat" not in pos or not tok[pos["$MeshFormat"] + 1].startswith("2"):
        raise ValueError("only Gmsh MSH 2.x ASCII files are supported")
    if int(tok[pos["$MeshFormat"] + 2]) != 0:
        raise ValueError("binary MSH files are not supported")
    i = pos["$Nodes"] + 1
    nn = int(tok[i])
    raw = np.array(tok[i + 1:i + 1 + 4 * nn], dtype=np.float64).reshape(nn, 4)
    tags = raw[:, 0].astype(np.int64)
    xyz = raw[:, 1:4].copy()
    i = pos["$Elements"] + 1
    ne = int(tok[i])
    i += 1
    cells, facets = {}, {}
    for _ in range(ne):
        etype, ntags = int(tok[i + 1]), int(tok[i + 2])
        nnod, tdim = _TYPES.get(etype, (None, None))
        if nnod is None:
            raise ValueError(f"element type {etype}: only lines, quadrangles and hexahedra are supported")
        phys = int(tok[i + 3]) if ntags > 0 else 0
        nodes = [int(v) for v in tok[i + 3 + ntags:i + 3 + ntags + nnod]]
        (cells if tdim >= 2 else facets).setdefault(etype, []).append((phys, nodes))
        i += 3 + ntags + nnod
    dim = 3 if 5 in cells else 2
    ctype = 5 if dim == 3 else 3
    if ctype not in cells:
        raise ValueError("no quadrangle / hexahedron cells in the file")
    if dim == 3 and 3 in cells:                     # quads of a 3-D mesh are boundary facets
        facets[3] = cells.pop(3)
    order = np.argsort(tags)
    lookup = np.full(int(tags.max()) + 1, -1, dtype=np.int64)
    lookup[tags[order]] = np.arange(nn)[order]      # file order kept: node k of the file -> id k
    lookup[tags] = np.arange(nn)
    conn = lookup[np.array([n for _, n in cells[ctype]], dtype=np.int64)][:, _TO_REF[ctype]]
    ftype = 3 if dim == 3 else 1
    bfac = [(p, lookup[np.array(n)]) for p, n in facets.get(ftype, [])]
    return {"dim": dim, "xyz": xyz[:, :dim].copy(), "conn": conn.astype(np.int32), "facets": bfac}


def exterior_facets(conn, dim):
    """facets (sorted node tuples) that belong to exactly one cell -> array [n, 2^(dim-1)]"""
    loc = np.array(_FACETS[dim])
    f = np.sort(conn[:, loc].reshape(-1, loc.shape[1]), axis=1)
    uniq, counts = np.unique(f, axis=0, return_counts=True)
    return uniq[counts == 1]


def write_msh(path, xyz, conn_ref, facets=()):
    """write cells given in REFERENCE corner order (tests / round trips)"""
    dim = xyz.shape[1]
    ctype = 5 if dim == 3 else 3
    inv = np.argsort(_TO_REF[ctype])
    ftype = 3 if dim == 3 else 1
    with open(path, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % len(xyz))
        for k, p in enumerate(xyz):
            q = list(p) + [0.0] * (3 - dim)
            f.write("%d %.17g %.17g %.17g\n" % (k + 1, q[0], q[1], q[2]))
        f.write("$EndNodes\n$Elements\n%d\n" % (len(conn_ref) + len(facets)))
        eid = 1
        for phys, nodes in facets:
            f.write("%d %d 2 %d %d %s\n" % (eid, ftype, phys, phys, " ".join(str(int(n) + 1) for n in nodes)))
            eid += 1
        for c in conn_ref:
            f.write("%d %d 2 0 0 %s\n" % (eid, ctype, " ".join(str(int(n) + 1) for n in np.asarray(c)[inv])))
            eid += 1
        f.write("$EndElements\n")
