"""Gmsh MSH 2.2 (ASCII) reader for quadrilateral / hexahedral (and linear simplicial) meshes.

The reference reads Gmsh files through PETSc (``DMPlex.createFromFile``, src/domain/dmplex.py:22-23) and
supports only tensor-product cells (src/domain/indices.py:116-122).  This reader returns what the device
path needs: node coordinates, the cell connectivity in the REFERENCE's local corner order
and the boundary facets with their physical tags (PETSc's "Face Sets")."""
import numpy as np

# Gmsh element type -> (nodes, topological dimension)
_TYPES = {1: (2, 1), 2: (3, 2), 3: (4, 2), 4: (4, 3), 5: (8, 3), 15: (1, 0)}
# Gmsh corner order -> reference / DMPlex closure order
#   quad: (x0,y0),(x1,y0),(x1,y1),(x0,y1) in both
#   hex : Gmsh bottom face counter-clockwise seen from +z, DMPlex closure walks it the other way round
#   triangle / tetrahedron (no reference counterpart; BASELINE.json configs[4]): Gmsh order kept
_TO_REF = {2: [0, 1, 2], 3: [0, 1, 2, 3], 4: [0, 1, 2, 3], 5: [0, 3, 2, 1, 4, 5, 6, 7]}
# facets in reference order, keyed by (dim, nodes per cell)
_FACETS = {(2, 4): [(0, 1), (1, 2), (2, 3), (3, 0)],
           (3, 8): [(0, 1, 2, 3), (4, 5, 6, 7), (0, 3, 5, 4), (1, 2, 6, 7), (2, 3, 5, 6), (0, 1, 7, 4)],
           (2, 3): [(0, 1), (1, 2), (2, 0)],
           (3, 4): [(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)]}


def _section(text, name):
    a = text.find("$" + name)
    b = text.find("$End" + name)
    if a < 0 or b < 0:
        raise ValueError(f"no ${name} section")
    return text[a + len(name) + 1:b]


def read_msh(path):
    with open(path) as f:
        text = f.read()
    fmt = _section(text, "MeshFormat").split()
    if not fmt or not fmt[0].startswith("2"):
        raise ValueError("only Gmsh MSH 2.x ASCII files are supported")
    if int(fmt[1]) != 0:
        raise ValueError("binary MSH files are not supported")
    body = _section(text, "Nodes").strip()
    head, _, rest = body.partition("\n")
    nn = int(head)
    raw = np.array(rest.split(), dtype=np.float64).reshape(nn, 4)
    tags = raw[:, 0].astype(np.int64)
    xyz = raw[:, 1:4].copy()
    # elements: rows are ragged (type-dependent), so group consecutive lines of equal token count and parse
    # every group in one vectorised call
    lines = _section(text, "Elements").strip().split("\n")
    del text
    ne = int(lines[0])
    lines = lines[1:ne + 1]
    cnt = np.fromiter((len(l.split()) for l in lines), dtype=np.int64, count=len(lines))
    cuts = np.concatenate([[0], np.nonzero(np.diff(cnt))[0] + 1, [len(lines)]])
    cells, facets = {}, {}
    for a, b in zip(cuts[:-1], cuts[1:]):
        rows = np.array(" ".join(lines[a:b]).split(), dtype=np.int64).reshape(b - a, cnt[a])
        for etype in np.unique(rows[:, 1]):
            nnod, tdim = _TYPES.get(int(etype), (None, None))
            if nnod is None:
                raise ValueError(f"element type {etype}: only first-order lines, triangles, quadrangles, tetrahedra "
                                 "and hexahedra are supported")
            r = rows[rows[:, 1] == etype]
            ntags = int(r[0, 2])
            if not np.all(r[:, 2] == ntags) or 3 + ntags + nnod != r.shape[1]:
                raise ValueError("malformed $Elements section")
            phys = r[:, 3] if ntags > 0 else np.zeros(len(r), dtype=np.int64)
            (cells if tdim >= 2 else facets).setdefault(int(etype), []).append((phys, r[:, 3 + ntags:]))
    cells = {t: (np.concatenate([p for p, _ in v]), np.vstack([n for _, n in v])) for t, v in cells.items()}
    facets = {t: (np.concatenate([p for p, _ in v]), np.vstack([n for _, n in v])) for t, v in facets.items()}
    dim = 3 if (5 in cells or 4 in cells) else 2
    if dim == 3:                                    # surface elements of a 3-D mesh are boundary facets
        for t in (2, 3):
            if t in cells:
                facets[t] = cells.pop(t)
    if len(cells) != 1:
        raise ValueError("no cells, or a mix of cell types, in the file: one of triangles / quadrangles / "
                         "tetrahedra / hexahedra is needed")
    ctype = next(iter(cells))
    lookup = np.full(int(tags.max()) + 1, -1, dtype=np.int64)
    lookup[tags] = np.arange(nn)                    # file order kept: node k of the file -> id k
    conn = lookup[cells[ctype][1]][:, _TO_REF[ctype]]
    xyz = xyz[:, :dim].copy()
    if ctype in (2, 4):                             # simplices: enforce det J > 0 whatever wrote the file
        X = xyz[conn]
        neg = np.linalg.det(X[:, 1:] - X[:, :1]) < 0
        conn[neg, -2], conn[neg, -1] = conn[neg, -1].copy(), conn[neg, -2].copy()
    ftype = {5: 3, 4: 2}.get(ctype, 1)
    bfac = []
    if ftype in facets:
        ph, nodes = facets[ftype]
        nodes = lookup[nodes]
        bfac = [(int(p), nodes[i]) for i, p in enumerate(ph)]
    return {"dim": dim, "xyz": xyz, "conn": conn.astype(np.int32), "facets": bfac,
            "cell": "simplex" if ctype in (2, 4) else "tensor"}


def exterior_facets(conn, dim):
    """facets (sorted node tuples) that belong to exactly one cell -> array [n, 2^(dim-1)]"""
    loc = np.array(_FACETS[(dim, conn.shape[1])])
    f = np.sort(conn[:, loc].reshape(-1, loc.shape[1]), axis=1)
    uniq, counts = np.unique(f, axis=0, return_counts=True)
    return uniq[counts == 1]


def write_msh(path, xyz, conn_ref, facets=()):
    """write cells given in REFERENCE corner order (tests / round trips)"""
    xyz = np.asarray(xyz, dtype=np.float64)
    conn_ref = np.asarray(conn_ref)
    dim = xyz.shape[1]
    ctype = {(2, 3): 2, (2, 4): 3, (3, 4): 4, (3, 8): 5}[(dim, conn_ref.shape[1])]
    inv = np.argsort(_TO_REF[ctype])
    ftype = {5: 3, 4: 2}.get(ctype, 1)
    with open(path, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % len(xyz))
        full = np.zeros((len(xyz), 3))
        full[:, :dim] = xyz
        f.write("\n".join("%d %s %s %s" % (k + 1, repr(float(p[0])), repr(float(p[1])), repr(float(p[2])))
                          for k, p in enumerate(full)))
        f.write("\n$EndNodes\n$Elements\n%d\n" % (len(conn_ref) + len(facets)))
        eid = 1
        for phys, nodes in facets:
            f.write("%d %d 2 %d %d %s\n" % (eid, ftype, phys, phys, " ".join(str(int(n) + 1) for n in nodes)))
            eid += 1
        ne = len(conn_ref)
        rows = np.empty((ne, 5 + conn_ref.shape[1]), dtype=np.int64)
        rows[:, 0] = np.arange(eid, eid + ne)
        rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4] = ctype, 2, 0, 0
        rows[:, 5:] = conn_ref[:, inv] + 1
        try:                                        # C-speed text output for multi-million-cell files
            import pandas as pd
            pd.DataFrame(rows).to_csv(f, sep=" ", header=False, index=False, lineterminator="\n")
        except ImportError:
            f.write("\n".join(" ".join(r) for r in rows.astype(str).tolist()) + "\n")
        f.write("$EndElements\n")
