"""CPU oracle for the Pynama finite/spectral-element hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pynama_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
there only as the checker.  It is a plain numpy/scipy restatement of the reference
algorithm, each function citing the reference lines it follows (paths relative to
``/root/reference/``).

Parity status: PINNED for the element level -- ``tests/test_oracle_golden.py`` checks every
function below against ``tests/golden/g{1,2,3}_*.npz`` (outputs of the reference's own
``elements/{utilities,element,spectral}.py`` run in the build container by
``tests/golden/make_golden.py``) and against the reference tests' known answers
(``src/tests/test_element.py:176-229``).  The global level (mesh, CSR, Dirichlet
elimination, Krylov solve) lives in PETSc in the reference (petsc4py is not installable
here: SURVEY.md section 8c), so there the oracle follows the call sites
(``src/cases/base_problem.py:479-481,499-552``, ``src/matrices/mat_generator.py:113-118``)
and is pinned by the analytic assertions of ``src/tests/test_solver.py:20-62`` -- no
captured PETSc vectors exist ("parity unpinned" w.r.t. PETSc's own iterates).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------------------
# 1-D rules                                                  src/elements/utilities.py
# --------------------------------------------------------------------------------------


def gauss_legendre(n: int):
    """Gauss-Legendre rule by Golub-Welsch, symmetrised.  utilities.py:43-61."""
    k = np.arange(1, n, dtype=np.float64)
    off = 0.5 / np.sqrt(1.0 - 1.0 / (4.0 * k * k))
    jac = np.diag(off, 1) + np.diag(off, -1)
    lam, vec = np.linalg.eigh(jac)
    order = np.argsort(lam)
    x = lam[order]
    w = 2.0 * vec[0, order] ** 2
    return 0.5 * (x - x[::-1]), 0.5 * (w + w[::-1])


def gauss_lobatto(n: int):
    """Gauss-Lobatto-Legendre rule, Newton on (1-x^2)P'_{n-1}.  utilities.py:63-92."""
    x = np.cos(np.linspace(0.0, np.pi, n))
    leg = np.zeros((n, n))
    prev = np.full(n, 2.0)
    while np.max(np.abs(x - prev)) > 1e-15:
        prev = x
        leg[:, 0] = 1.0
        leg[:, 1] = x
        for j in range(2, n):
            leg[:, j] = ((2 * j - 1) * x * leg[:, j - 1] - (j - 1) * leg[:, j - 2]) / j
        x = prev - (x * leg[:, n - 1] - leg[:, n - 2]) / (n * leg[:, n - 1])
    w = 2.0 / ((n - 1) * n * leg[:, n - 1] ** 2)
    return 0.5 * (x[::-1] - x), 0.5 * (w[::-1] + w)


def lagrange_1d(nodes, pts):
    """Lagrange cardinal functions and first derivatives.  elements/element.py:17-49.

    Returns (h, dh), each [len(pts), len(nodes)].
    """
    nodes = np.asarray(nodes, dtype=np.float64)
    pts = np.asarray(pts, dtype=np.float64)
    m = len(nodes)
    h = np.zeros((len(pts), m))
    dh = np.zeros((len(pts), m))
    for a in range(m):
        others = [b for b in range(m) if b != a]
        den = np.prod([nodes[a] - nodes[b] for b in others])
        for ip, x in enumerate(pts):
            h[ip, a] = np.prod([x - nodes[b] for b in others]) / den
            acc = 0.0
            for skip in others:
                acc += np.prod([x - nodes[b] for b in others if b != skip])
            dh[ip, a] = acc / den
    return h, dh


# --------------------------------------------------------------------------------------
# local orderings                         spectral.py:220-271 (2D), spectral.py:346-431 (3D)
# --------------------------------------------------------------------------------------


def _line(p0, p1, n):
    """interior lattice points strictly between p0 and p1 (n = points per side)."""
    p0, p1 = np.array(p0), np.array(p1)
    step = (p1 - p0) // (n - 1)
    return [tuple(p0 + s * step) for s in range(1, n - 1)]


def local_lattice(n: int, dim: int):
    """Lattice index (i[,j[,k]]) along (r,s[,t]) of every local point, in the reference's
    vertex -> edge -> face -> interior order.  Used for the nodes (n = ngl) and, with the
    same permutation, for tensor quadrature points (spectral.py:296-298,340-342)."""
    if n == 1:
        return [(0,) * dim]
    m = n - 1
    inner = range(1, m)
    if dim == 2:
        v = [(m, m), (0, m), (0, 0), (m, 0)]
        out = list(v)
        for a in range(4):
            out += _line(v[a], v[(a + 1) % 4], n)
        out += [(i, j) for i in inner for j in reversed(inner)]
        return out
    v = [(0, 0, 0), (0, m, 0), (m, m, 0), (m, 0, 0), (0, 0, m), (m, 0, m), (m, m, m), (0, m, m)]
    edges = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4),
             (3, 5), (4, 0), (1, 7), (6, 2)]
    out = list(v)
    for a, b in edges:
        out += _line(v[a], v[b], n)
    rin = list(inner)
    rev = rin[::-1]
    out += [(i, j, 0) for j in rev for i in rin]      # face t=-1
    out += [(i, j, m) for j in rin for i in rev]      # face t=+1
    out += [(i, 0, k) for i in rev for k in rin]      # face s=-1
    out += [(i, m, k) for i in rin for k in rev]      # face s=+1
    out += [(m, j, k) for k in rev for j in rev]      # face r=+1
    out += [(0, j, k) for k in rin for j in rin]      # face r=-1
    out += [(i, j, k) for k in rev for j in rev for i in rin]
    return out


# --------------------------------------------------------------------------------------
# tensor tables                                     spectral.py:39-87, 220-344
# --------------------------------------------------------------------------------------


@dataclass
class Quad:
    """One quadrature's tables: H [ngp, nn], Hrs [ngp, dim, nn], pts [ngp, dim], w [ngp]."""
    H: np.ndarray
    Hrs: np.ndarray
    pts: np.ndarray
    w: np.ndarray


def tensor_tables(nodes1d, pts1d, w1d, dim) -> Quad:
    """Shape functions of the tensor Lagrange basis on `nodes1d` evaluated at the tensor
    grid of `pts1d`, both in reference order.  spectral.py:220-300 / 302-344."""
    h, dh = lagrange_1d(nodes1d, pts1d)
    nod = local_lattice(len(nodes1d), dim)
    gp = local_lattice(len(pts1d), dim)
    H = np.zeros((len(gp), len(nod)))
    Hrs = np.zeros((len(gp), dim, len(nod)))
    pts = np.zeros((len(gp), dim))
    w = np.zeros(len(gp))
    for g, gi in enumerate(gp):
        pts[g] = [pts1d[i] for i in gi]
        w[g] = np.prod([w1d[i] for i in gi])
        for a, ai in enumerate(nod):
            vals = [h[gi[d], ai[d]] for d in range(dim)]
            H[g, a] = np.prod(vals)
            for d in range(dim):
                t = list(vals)
                t[d] = dh[gi[d], ai[d]]
                Hrs[g, d, a] = np.prod(t)
    return Quad(H, Hrs, pts, w)


class Tables:
    """All tables of `Spectral(ngl, dim)`.  spectral.py:17-87."""

    def __init__(self, ngl: int, dim: int):
        self.ngl, self.dim = ngl, dim
        self.dim_w = 1 if dim == 2 else 3
        self.dim_s = 3 if dim == 2 else 6
        self.nn = ngl ** dim
        self.nc = 2 ** dim                       # geometry (corner) nodes
        nod, w_nod = gauss_lobatto(ngl)
        full, w_full = gauss_legendre(ngl) if ngl <= 3 else gauss_lobatto(ngl)   # :41-42
        red, w_red = gauss_legendre(ngl - 1)                                      # :43
        cor, _ = gauss_lobatto(2)                                                 # :44
        self.full = tensor_tables(nod, full, w_full, dim)
        self.red = tensor_tables(nod, red, w_red, dim)
        self.op = tensor_tables(nod, nod, w_nod, dim)
        self.coo = tensor_tables(cor, full, w_full, dim)
        self.coo_red = tensor_tables(cor, red, w_red, dim)
        self.coo_op = tensor_tables(cor, nod, w_nod, dim)
        self.HCoo1D, _ = lagrange_1d(cor, nod)
        self.nodes1d = nod


class SimplexTables:
    """P1 triangle / tetrahedron (NO reference counterpart: the reference is tensor-product only,
    src/domain/indices.py:116-122; BASELINE.json configs[4] asks for a tetrahedral mesh).  Textbook linear
    simplex: barycentric basis on the reference simplex {0, e_1..e_dim}; full rule = the symmetric
    degree-2 rule (3 / 4 points), reduced = centroid, nodal = vertex rule.  Pinned by the closed forms
    L_ab = V grad(l_a).grad(l_b) and M_ab = V (1 + delta_ab) / ((dim+1)(dim+2)) in tests/test_simplex_host.py."""

    def __init__(self, dim: int):
        self.ngl, self.dim = 2, dim
        self.dim_w = 1 if dim == 2 else 3
        self.dim_s = 3 if dim == 2 else 6
        self.nn = self.nc = dim + 1
        vol = 0.5 if dim == 2 else 1.0 / 6.0
        if dim == 2:
            full = np.array([[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]])
        else:
            a, b = 0.5854101966249685, 0.1381966011250105
            full = np.array([[b, b, b], [a, b, b], [b, a, b], [b, b, a]])
        grad = np.hstack([-np.ones((dim, 1)), np.eye(dim)])

        def quad(pts):
            H = np.hstack([1.0 - pts.sum(axis=1, keepdims=True), pts])
            return Quad(H, np.repeat(grad[None], len(pts), axis=0), pts, np.full(len(pts), vol / len(pts)))
        self.full = self.coo = quad(full)
        self.red = self.coo_red = quad(np.full((1, dim), 1.0 / (dim + 1)))
        self.op = self.coo_op = quad(np.vstack([np.zeros(dim), np.eye(dim)]))


# --------------------------------------------------------------------------------------
# element forms (batched over elements)                        spectral.py:89-218
# --------------------------------------------------------------------------------------


def _geom(q_geo: Quad, q: Quad, X):
    """J = HrsCoo.X, G = J^-1 Hrs, c = w detJ at every point.  spectral.py:120-122.
    X [E, 2^dim, dim] -> G [E, ngp, dim, nn], c [E, ngp]."""
    J = np.einsum("gdc,ecx->egdx", q_geo.Hrs, X)
    G = np.einsum("egdx,gxa->egda", np.linalg.inv(J), q.Hrs)
    return G, q.w[None, :] * np.linalg.det(J)


def _curl_rows(dim):
    """(row, component, derivative, sign) of the discrete curl.  spectral.py:26-33,128-129,147-148."""
    if dim == 2:
        ind_w = [[0, 0, 1], [1, 0, 0]]
        ind_c = [[0, 1, 0], [0, 0, 1]]
    else:
        ind_w = ind_c = [[0, 2, 1], [0, 1, 2], [1, 0, 2], [1, 2, 0], [2, 1, 0], [2, 0, 1]]
    sg = lambda i: 1.0 if i % 2 == 0 else -1.0
    return ([(r, c, d, sg(i)) for i, (r, c, d) in enumerate(ind_w)],
            [(r, c, d, sg(i)) for i, (r, c, d) in enumerate(ind_c)])


def elem_kle_matrices(tb: Tables, coords, alpha_d=1e3, alpha_w=1e2):
    """K_e, Rw_e, Rd_e of the KLE method for a batch of elements.  spectral.py:89-157.

    coords: [E, 2^dim * dim] (or one flat element) corner coordinates.
    returns K [E, dim nn, dim nn], Rw [E, dim nn, dim_w nn], Rd [E, dim nn, nn].
    """
    dim, dw, nn = tb.dim, tb.dim_w, tb.nn
    X = np.asarray(coords, dtype=np.float64).reshape(-1, tb.nc, dim)
    E = X.shape[0]
    K = np.zeros((E, dim * nn, dim * nn))
    Rw = np.zeros((E, dim * nn, dw * nn))
    Rd = np.zeros((E, dim * nn, nn))
    wcurl, vcurl = _curl_rows(dim)

    G, c = _geom(tb.coo, tb.full, X)
    for g in range(len(tb.full.w)):
        Gg, cg, Hg = G[:, g], c[:, g], tb.full.H[g]
        B_gr = np.zeros((E, dim * dim, dim * nn))
        Hvel = np.zeros((dim, dim * nn))
        for nd in range(dim):
            B_gr[:, dim * nd:dim * nd + dim, nd::dim] = Gg                    # :125
            Hvel[nd, nd::dim] = Hg                                             # :126
        Bw = np.zeros((E, dim, dw * nn))
        for r, comp, d, s in wcurl:
            Bw[:, r, comp::dw] = s * Gg[:, d]                                  # :129
        K += cg[:, None, None] * np.einsum("eki,ekj->eij", B_gr, B_gr)         # :131
        Rw += cg[:, None, None] * np.einsum("ki,ekj->eij", Hvel, Bw)           # :132
        Rd -= cg[:, None, None] * np.einsum("ki,ekj->eij", Hvel, Gg)           # :133

    G, c = _geom(tb.coo_red, tb.red, X)
    for g in range(len(tb.red.w)):
        Gg, cg, Hg = G[:, g], c[:, g], tb.red.H[g]
        B_div = np.zeros((E, 1, dim * nn))
        for nd in range(dim):
            B_div[:, 0, nd::dim] = Gg[:, nd]                                   # :145
        B_curl = np.zeros((E, dw, dim * nn))
        for r, comp, d, s in vcurl:
            B_curl[:, r, comp::dim] = s * Gg[:, d]                             # :148
        Hw = np.zeros((dw, dw * nn))
        for nd in range(dw):
            Hw[nd, nd::dw] = Hg                                                # :150
        K += cg[:, None, None] * (alpha_d * np.einsum("eki,ekj->eij", B_div, B_div)
                                  + alpha_w * np.einsum("eki,ekj->eij", B_curl, B_curl))  # :152-153
        Rw += cg[:, None, None] * alpha_w * np.einsum("eki,kj->eij", B_curl, Hw)  # :155
        flatF = np.transpose(Gg, (0, 2, 1)).reshape(E, dim * nn)               # Hxy.flatten('F')
        Rd += cg[:, None, None] * alpha_d * flatF[:, :, None] * Hg[None, None, :]  # :156
    return K, Rw, Rd


def elem_kle_operators(tb: Tables, coords):
    """SrT_e, DivSrT_e, Curl_e and lumped weights at the nodal (GLL) rule.  spectral.py:159-218."""
    dim, dw, ds, nn = tb.dim, tb.dim_w, tb.dim_s, tb.nn
    X = np.asarray(coords, dtype=np.float64).reshape(-1, tb.nc, dim)
    E = X.shape[0]
    ind_bdiv = [[0, 1], [1, 2]] if dim == 2 else [[0, 1, 5], [1, 2, 3], [5, 3, 4]]  # :28,33
    _, vcurl = _curl_rows(dim)
    SrT = np.zeros((E, ds * nn, dim * nn))
    Div = np.zeros((E, dim * nn, ds * nn))
    Curl = np.zeros((E, dw * nn, dim * nn))
    Wm = np.zeros((E, nn, nn))
    G, c = _geom(tb.coo_op, tb.op, X)
    # NB the reference allocates B_srt / B_div once, OUTSIDE the point loop, and scales
    # B_srt *= 0.5 in place every point (:207); entries rewritten each point are unaffected,
    # and every nonzero entry is rewritten each point, so per-point rebuild is equivalent.
    for g in range(len(tb.op.w)):
        Gg, cg, Hg = G[:, g], c[:, g], tb.op.H[g]
        Hs = np.zeros((ds, ds * nn))
        for k in range(ds):
            Hs[k, k::ds] = Hg                                                  # :190
        B_curl = np.zeros((E, dw, dim * nn))
        for r, comp, d, s in vcurl:
            B_curl[:, r, comp::dim] = s * Gg[:, d]                             # :193
        Hdiv = np.zeros((dim, dim * nn))
        B_div = np.zeros((E, dim, ds * nn))
        B_srt = np.zeros((E, ds, dim * nn))
        for x in range(dim):
            Hdiv[x, x::dim] = Hg                                               # :196
            for i in range(dim):
                B_div[:, i, ind_bdiv[x][i]::ds] = Gg[:, x]                     # :198
                B_srt[:, ind_bdiv[x][i], i::dim] = Gg[:, x]                    # :199
        B_srt[:, 0, 1::dim] = -Gg[:, 1]                                        # :201
        B_srt[:, 2, 0::dim] = -Gg[:, 0]                                        # :202
        for i in range(ds - 4):
            B_srt[:, 4, i::dim] = -Gg[:, i]                                    # :204
            B_srt[:, 2 * i, 2::dim] = -Gg[:, 2]                                # :205
        B_srt *= 0.5                                                           # :207
        Hc = np.zeros((dw, dw * nn))
        for i in range(dw):
            Hc[i, i::dw] = Hg                                                  # :210
        SrT += cg[:, None, None] * np.einsum("ki,ekj->eij", Hs, B_srt)         # :212
        Div += cg[:, None, None] * np.einsum("ki,ekj->eij", Hdiv, B_div)       # :213
        Curl += cg[:, None, None] * np.einsum("ki,ekj->eij", Hc, B_curl)       # :214
        Wm += cg[:, None, None] * np.outer(Hg, Hg)[None]                       # :215
    return SrT, Div, Curl, Wm.sum(2)                                           # :217


def elem_laplace(tb: Tables, coords):
    """Scalar stiffness L_e = sum_full c G^T G -- the `kron(., I_dim)` block of K_e
    (spectral.py:125,131 restricted to one velocity component)."""
    X = np.asarray(coords, dtype=np.float64).reshape(-1, tb.nc, tb.dim)
    G, c = _geom(tb.coo, tb.full, X)
    return np.einsum("eg,egda,egdb->eab", c, G, G)


def elem_mass(tb: Tables, coords, rule="nodal"):
    """Scalar mass matrix.  rule='nodal' is the reference's elWeigMat (spectral.py:215,
    GLL collocation => diagonal); rule='full' uses the full quadrature."""
    X = np.asarray(coords, dtype=np.float64).reshape(-1, tb.nc, tb.dim)
    qg, q = (tb.coo_op, tb.op) if rule == "nodal" else (tb.coo, tb.full)
    _, c = _geom(qg, q, X)
    return np.einsum("eg,ga,gb->eab", c, q.H, q.H)


# --------------------------------------------------------------------------------------
# structured box mesh with the reference's element-local conventions
#   src/domain/dmplex.py:8-40 (box mesh, border names), :97-104 (corner order),
#   src/tests/test_domain.py:26-30,94-104 (cell-0 corners), :187-201 (lexicographic ids)
# --------------------------------------------------------------------------------------


@dataclass
class BoxMesh:
    dim: int
    ngl: int
    nelem: tuple
    lattice: tuple          # nodes per direction
    conn: np.ndarray        # [n_elem, nn] int32, reference local order
    xyz: np.ndarray         # [n_node, dim]
    boundary: np.ndarray    # sorted node ids on "External Boundary"
    borders: dict           # name -> node ids
    nc: int = 0             # geometry nodes per element (0: 2^dim)

    @property
    def n_node(self):
        return self.xyz.shape[0]

    @property
    def n_elem(self):
        return self.conn.shape[0]

    def corners(self):
        """[n_elem, 2^dim * dim] corner coordinates in closure order.  dmplex.py:97-104."""
        nc = self.nc or 2 ** self.dim
        return self.xyz[self.conn[:, :nc]].reshape(self.n_elem, nc * self.dim)


def box_mesh(nelem, lower, upper, ngl=2, jitter=0.0, seed=12345) -> BoxMesh:
    dim = len(nelem)
    m = ngl - 1
    lat = tuple(m * n + 1 for n in nelem)
    gll, _ = gauss_lobatto(ngl)
    # 1-D node coordinates per direction (GLL-spaced inside each element)
    axes = []
    for d in range(dim):
        h = (upper[d] - lower[d]) / nelem[d]
        ax = np.zeros(lat[d])
        for e in range(nelem[d]):
            ax[e * m:e * m + ngl] = lower[d] + h * (e + 0.5 * (1.0 + gll))
        ax[-1] = upper[d]
        axes.append(ax)
    grids = np.meshgrid(*axes, indexing="ij")                   # index order (i, j[, k])
    strides = [int(np.prod(lat[:d])) for d in range(dim)]       # x fastest
    n_node = int(np.prod(lat))
    ids = sum(np.indices(lat)[d] * strides[d] for d in range(dim))
    xyz = np.zeros((n_node, dim))
    for d in range(dim):
        xyz[ids.ravel(), d] = grids[d].ravel()
    loc = np.array(local_lattice(ngl, dim))                     # [nn, dim]
    if dim == 2:                                                # x ~ -r, y ~ -s  (A.2)
        loc = m - loc
    eidx = np.indices(tuple(reversed(nelem))).reshape(dim, -1)[::-1]       # cell (ex,ey[,ez]), x fastest
    base = sum(eidx[d].astype(np.int64) * m * strides[d] for d in range(dim))          # [n_elem]
    off = sum(loc[:, d].astype(np.int64) * strides[d] for d in range(dim))             # [nn]
    conn = (base[:, None] + off[None, :]).astype(np.int32)
    idx = np.indices(lat)
    on = np.zeros(lat, dtype=bool)
    for d in range(dim):
        on |= (idx[d] == 0) | (idx[d] == lat[d] - 1)
    boundary = np.sort(ids[on])
    if dim == 2:                                                # dmplex.py:37-38
        names = {"down": (1, 0), "right": (0, 1), "up": (1, 1), "left": (0, 0)}
    else:                                                       # dmplex.py:39-40
        names = {"back": (2, 0), "front": (2, 1), "down": (1, 0), "up": (1, 1),
                 "right": (0, 1), "left": (0, 0)}
    borders = {k: np.sort(ids[idx[d] == (lat[d] - 1 if hi else 0)]) for k, (d, hi) in names.items()}
    if jitter > 0.0:
        rng = np.random.default_rng(seed)
        hmin = min((upper[d] - lower[d]) / nelem[d] / m for d in range(dim))
        move = jitter * hmin * rng.uniform(-1, 1, size=xyz.shape)
        move[boundary] = 0.0
        if ngl > 2:
            raise ValueError("jitter only for ngl=2 (geometry is multilinear from corners)")
        xyz = xyz + move
    return BoxMesh(dim, ngl, tuple(nelem), lat, conn, xyz, boundary, borders)


# --------------------------------------------------------------------------------------
# global assembly with Dirichlet elimination            base_problem.py:499-552
# --------------------------------------------------------------------------------------


def simplex_box_mesh(nelem, lower, upper, jitter=0.0, seed=12345, permute_seed=None) -> BoxMesh:
    """Build-generated simplicial mesh of a box: the Q1 lattice of `box_mesh` with every
    quad cut into 2 triangles / every hex into 6 tetrahedra (Kuhn's conforming subdivision, positive
    orientation).  `permute_seed` applies a random node permutation to destroy index locality."""
    from itertools import permutations
    dim = len(nelem)
    box = box_mesh(nelem, lower, upper, 2, jitter=jitter, seed=seed)
    lat = box.lattice
    strides = np.cumprod((1,) + lat[:-1])
    cells = np.stack(np.meshgrid(*[np.arange(n) for n in nelem], indexing="ij"), axis=-1).reshape(-1, dim)
    # keep the cell order of box_mesh (x fastest)
    order = np.lexsort([cells[:, d] for d in range(dim)])
    base = cells[order] @ strides
    conn = []
    for perm in permutations(range(dim)):
        offs = [0]
        for d in perm:
            offs.append(offs[-1] + strides[d])
        inv = sum(1 for a in range(dim) for b in range(a + 1, dim) if perm[a] > perm[b])
        if inv % 2:                                   # odd permutation: swap two vertices -> det J > 0
            offs[-1], offs[-2] = offs[-2], offs[-1]
        conn.append(base[:, None] + np.array(offs)[None, :])
    conn = np.stack(conn, axis=1).reshape(-1, dim + 1)
    xyz, boundary, borders = box.xyz, box.boundary, box.borders
    if permute_seed is not None:
        perm = np.random.default_rng(permute_seed).permutation(box.n_node)     # old id -> new id
        xyz = xyz[np.argsort(perm)]
        conn = perm[conn]
        boundary = np.sort(perm[boundary])
        borders = {k: perm[v] for k, v in borders.items()}
    return BoxMesh(dim, 2, tuple(nelem), lat, conn.astype(np.int32), xyz, boundary, borders, nc=dim + 1)


def dof_indices(nodes, ndof):
    """global DOF = node*ndof + d.  src/domain/indices.py:90-92."""
    nodes = np.asarray(nodes)
    return (nodes[..., None] * ndof + np.arange(ndof)).reshape(*nodes.shape[:-1], -1)


def _scatter(shape, rows, cols, vals):
    return sp.coo_matrix((vals.ravel(), (rows.ravel(), cols.ravel())), shape=shape).tocsr()


def assemble_kle_freeslip(mesh: BoxMesh, tb: Tables, alpha_d=1e3, alpha_w=1e2, with_rd=False):
    """K, Krhs, Rw (and Rd) exactly as FreeSlip.buildKLEMats + setIndices2One build them
    (single rank).  base_problem.py:499-552, mat_generator.py:113-118.

      K[free,free] += K_e[free,free]      :540-541      K[bc,bc]    = 1   :549
      Krhs[free,bc] += -K_e[free,bc]      :531-533      Krhs[bc,bc] = 1   :549
      Rw[free,:]  += Rw_e[free,:]         :546-547
    """
    dim, dw = tb.dim, tb.dim_w
    Ke, Rwe, Rde = elem_kle_matrices(tb, mesh.corners(), alpha_d, alpha_w)
    n = mesh.n_node
    vdof = dof_indices(mesh.conn, dim)                          # [E, dim nn]
    wdof = dof_indices(mesh.conn, dw)
    is_bc = np.zeros(n * dim, dtype=bool)
    is_bc[dof_indices(mesh.boundary[:, None], dim).ravel()] = True
    rfree = ~is_bc[vdof]                                        # [E, dim nn]
    cbc = is_bc[vdof]
    R = np.broadcast_to(vdof[:, :, None], Ke.shape)
    C = np.broadcast_to(vdof[:, None, :], Ke.shape)
    mff = rfree[:, :, None] & rfree[:, None, :]
    mfb = rfree[:, :, None] & cbc[:, None, :]
    K = _scatter((n * dim, n * dim), R[mff], C[mff], Ke[mff])
    Krhs = _scatter((n * dim, n * dim), R[mfb], C[mfb], -Ke[mfb])
    bc_idx = np.nonzero(is_bc)[0]
    ident = sp.coo_matrix((np.ones(len(bc_idx)), (bc_idx, bc_idx)), shape=K.shape).tocsr()
    K = (K + ident).tocsr()
    Krhs = (Krhs + ident).tocsr()
    Rr = np.broadcast_to(vdof[:, :, None], Rwe.shape)
    Rc = np.broadcast_to(wdof[:, None, :], Rwe.shape)
    mrow = np.broadcast_to(rfree[:, :, None], Rwe.shape)
    Rw = _scatter((n * dim, n * dw), Rr[mrow], Rc[mrow], Rwe[mrow])
    out = {"K": K, "Krhs": Krhs, "Rw": Rw, "is_bc": is_bc}
    if with_rd:                                                 # NoSlipFreeSlip :435-437
        Dr = np.broadcast_to(vdof[:, :, None], Rde.shape)
        Dc = np.broadcast_to(mesh.conn[:, None, :], Rde.shape)
        md = np.broadcast_to(rfree[:, :, None], Rde.shape)
        out["Rd"] = _scatter((n * dim, n), Dr[md], Dc[md], Rde[md])
    return out


def noslip_classes(mesh: BoxMesh, ns_walls, dir_walls=()):
    """DOF classes of NoSlipFreeSlip.buildKLEMats (base_problem.py:343-379) on a box mesh:
    0 free; 1 tangential DOF of a node on a no-slip wall (dofFreeFSSetNS); 2 imposed in both solves
    (dofSetFSNS: the wall-normal DOF of a no-slip node, every DOF of a Dirichlet-wall node)."""
    axis = ({"down": 1, "up": 1, "left": 0, "right": 0} if mesh.dim == 2 else
            {"back": 2, "front": 2, "down": 1, "up": 1, "left": 0, "right": 0})
    cls = np.zeros((mesh.n_node, mesh.dim), dtype=np.uint8)
    for w in ns_walls:
        nodes = mesh.borders[w]
        tang = [d for d in range(mesh.dim) if d != axis[w]]
        for d in tang:
            cls[nodes, d] = np.maximum(cls[nodes, d], 1)
        cls[nodes, axis[w]] = 2
    for w in dir_walls:
        cls[mesh.borders[w], :] = 2
    return cls


def assemble_kle_noslip(mesh: BoxMesh, tb: Tables, cls, alpha_d=1e3, alpha_w=1e2):
    """The eight matrices of NoSlipFreeSlip.buildKLEMats (base_problem.py:329-454), single rank,
    every cell integrated (the reference reuses cell 0's blocks: uniform meshes only, :333-334)."""
    dim, dw = tb.dim, tb.dim_w
    Ke, Rwe, Rde = elem_kle_matrices(tb, mesh.corners(), alpha_d, alpha_w)
    n = mesh.n_node
    vdof = dof_indices(mesh.conn, dim)
    wdof = dof_indices(mesh.conn, dw)
    c = cls.ravel()[vdof]                                          # [E, dim nn] class of each local DOF
    ci, cj = c[:, :, None], c[:, None, :]
    R = np.broadcast_to(vdof[:, :, None], Ke.shape)
    C = np.broadcast_to(vdof[:, None, :], Ke.shape)

    def mat(mask, vals, rows=R, cols=C, shape=(n * dim, n * dim)):
        mask = np.broadcast_to(mask, vals.shape)
        return _scatter(shape, rows[mask], cols[mask], vals[mask])
    idx1 = np.nonzero(cls.ravel() >= 1)[0]
    idx_fs = np.nonzero(cls.ravel() == 1)[0]
    idx_set = np.nonzero(cls.ravel() == 2)[0]

    def diag(idx, val):
        return sp.coo_matrix((np.full(len(idx), val), (idx, idx)), shape=(n * dim, n * dim)).tocsr()
    out = {}
    out["K"] = (mat((ci == 0) & (cj == 0), Ke) + diag(idx1, 1.0)).tocsr()                     # :426-430, 439
    out["Krhs"] = (mat((ci == 0) & (cj >= 1), -Ke) + diag(idx1, 1.0)).tocsr()                 # :388-395, 439
    out["Kfs"] = (mat(((ci == 1) & (cj <= 1)) | ((ci == 0) & (cj == 1)), Ke) + diag(idx_fs, -1.0)).tocsr()  # :396-407, 441-442
    out["Krhsfs"] = (mat((ci <= 1) & (cj == 2), -Ke) + diag(idx_set, 1.0)).tocsr()            # :417-424, 449-450
    Rr = np.broadcast_to(vdof[:, :, None], Rwe.shape)
    Rc = np.broadcast_to(wdof[:, None, :], Rwe.shape)
    out["Rw"] = mat(ci == 0, Rwe, Rr, Rc, (n * dim, n * dw))                                  # :432-433
    out["Rwfs"] = mat(ci == 1, Rwe, Rr, Rc, (n * dim, n * dw))                                # :412-413
    Dr = np.broadcast_to(vdof[:, :, None], Rde.shape)
    Dc = np.broadcast_to(mesh.conn[:, None, :], Rde.shape)
    out["Rd"] = mat(ci == 0, Rde, Dr, Dc, (n * dim, n))                                       # :435-437
    out["Rdfs"] = mat(ci == 1, Rde, Dr, Dc, (n * dim, n))                                     # :415-416
    return out


def assemble_scalar(mesh: BoxMesh, tb: Tables, form="laplace", dirichlet=None):
    """Scalar operator (BASELINE 'Poisson' = the L_e block) with the SAME elimination rule
    as the KLE path: A[free,free], Arhs[free,bc] = -A_e, unit diagonal on bc rows."""
    Ae = elem_laplace(tb, mesh.corners()) if form == "laplace" else elem_mass(tb, mesh.corners())
    n = mesh.n_node
    is_bc = np.zeros(n, dtype=bool)
    if dirichlet is not None:
        is_bc[np.asarray(dirichlet)] = True
    rn = mesh.conn
    R = np.broadcast_to(rn[:, :, None], Ae.shape)
    C = np.broadcast_to(rn[:, None, :], Ae.shape)
    rfree, cbc = ~is_bc[rn], is_bc[rn]
    mff = rfree[:, :, None] & rfree[:, None, :]
    mfb = rfree[:, :, None] & cbc[:, None, :]
    bc_idx = np.nonzero(is_bc)[0]
    ident = sp.coo_matrix((np.ones(len(bc_idx)), (bc_idx, bc_idx)), shape=(n, n)).tocsr()
    A = (_scatter((n, n), R[mff], C[mff], Ae[mff]) + ident).tocsr()
    Arhs = (_scatter((n, n), R[mfb], C[mfb], -Ae[mfb]) + ident).tocsr()
    return {"A": A, "Arhs": Arhs, "is_bc": is_bc}


def assemble_operators(mesh: BoxMesh, tb: Tables, cell0_only=False):
    """Global SrT, DivSrT, Curl with reciprocal lumped-weight row scaling.
    base_problem.py:132-140 (loop over cells, Operators.setValues mat_generator.py:157-170) and
    Operators.assembleAll (mat_generator.py:172-190: weights assembled, reciprocal, diagonalScale(L=)).
    cell0_only=True reproduces the reference literally (cell 0's blocks reused for every cell,
    base_problem.py:133-134: valid on uniform meshes); the default integrates every cell."""
    dim, dw, ds = tb.dim, tb.dim_w, tb.dim_s
    X = mesh.corners()
    if cell0_only:
        X = np.repeat(X[:1], mesh.n_elem, axis=0)
    SrT, Div, Curl, wei = elem_kle_operators(tb, X)
    n = mesh.n_node
    iv, iw, isr = dof_indices(mesh.conn, dim), dof_indices(mesh.conn, dw), dof_indices(mesh.conn, ds)

    def glob(Me, rows, cols, nr, nc):
        R = np.broadcast_to(rows[:, :, None], Me.shape)
        C = np.broadcast_to(cols[:, None, :], Me.shape)
        return _scatter((nr, nc), R, C, Me)
    w = np.zeros(n)
    np.add.at(w, mesh.conn.ravel(), wei.ravel())
    out = {}
    for name, Me, rows, cols, br, bc in (("SrT", SrT, isr, iv, ds, dim), ("DivSrT", Div, iv, isr, dim, ds),
                                        ("Curl", Curl, iw, iv, dw, dim)):
        M = glob(Me, rows, cols, n * br, n * bc)
        out[name] = sp.diags(1.0 / np.repeat(w, br)) @ M
    out["weights"] = w
    return out


def node_graph(mesh: BoxMesh):
    """Node adjacency CSR pattern (what DM.createMat yields, dmplex.py:300-333)."""
    E, nn = mesh.conn.shape
    R = np.broadcast_to(mesh.conn[:, :, None], (E, nn, nn)).ravel()
    C = np.broadcast_to(mesh.conn[:, None, :], (E, nn, nn)).ravel()
    g = sp.coo_matrix((np.ones(R.size, dtype=np.int8), (R, C)), shape=(mesh.n_node,) * 2).tocsr()
    g.sort_indices()
    return g.indptr.astype(np.int32), g.indices.astype(np.int32)


# --------------------------------------------------------------------------------------
# Krylov solvers with PETSc KSP semantics (KSPCG / KSPGMRES + PCJACOBI, left PC)
#   reference call sites: src/solver/ksp_solver.py:9-19, base_problem.py:479-481
# --------------------------------------------------------------------------------------

NORM_PRECONDITIONED, NORM_UNPRECONDITIONED, NORM_NATURAL = 0, 1, 2


def pcg(A, b, rtol=1e-5, atol=1e-50, dtol=1e5, maxit=10000, jacobi=True,
        norm_type=NORM_PRECONDITIONED, x0=None):
    """Preconditioned conjugate gradients; convergence test of KSPConvergedDefault:
    rnorm <= max(rtol * rnorm_0, atol), on the norm selected by `norm_type`."""
    dinv = 1.0 / A.diagonal() if jacobi else np.ones(A.shape[0])
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x
    z = dinv * r
    rz = r @ z

    def nrm():
        if norm_type == NORM_PRECONDITIONED:
            return np.sqrt(z @ z)
        if norm_type == NORM_UNPRECONDITIONED:
            return np.sqrt(r @ r)
        return np.sqrt(abs(rz))

    r0 = nrm()
    ttol = max(rtol * r0, atol)
    hist = [r0]
    if r0 <= ttol:
        return x, 0, hist
    p = z.copy()
    for it in range(1, maxit + 1):
        Ap = A @ p
        alpha = rz / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        z = dinv * r
        rz_new = r @ z
        rn = nrm() if norm_type != NORM_NATURAL else np.sqrt(abs(rz_new))
        hist.append(rn)
        if rn <= ttol or rn >= dtol * r0:
            return x, it, hist
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, maxit, hist


def gmres(A, b, rtol=1e-5, atol=1e-50, maxit=10000, restart=30, jacobi=True, x0=None):
    """Left-preconditioned restarted GMRES(m), modified Gram-Schmidt, Givens rotations;
    convergence on the preconditioned residual norm (KSPGMRES default)."""
    n = b.size
    dinv = 1.0 / A.diagonal() if jacobi else np.ones(n)
    x = np.zeros_like(b) if x0 is None else x0.copy()
    its = 0
    hist = []
    ttol = None
    while True:
        r = dinv * (b - A @ x)
        beta = np.linalg.norm(r)
        if ttol is None:
            ttol = max(rtol * beta, atol)
            hist.append(beta)
        if beta <= ttol or its >= maxit:
            return x, its, hist
        V = np.zeros((restart + 1, n))
        Hm = np.zeros((restart + 1, restart))
        cs, sn = np.zeros(restart), np.zeros(restart)
        g = np.zeros(restart + 1)
        g[0] = beta
        V[0] = r / beta
        k_used = 0
        for k in range(restart):
            w = dinv * (A @ V[k])
            for j in range(k + 1):
                Hm[j, k] = w @ V[j]
                w -= Hm[j, k] * V[j]
            Hm[k + 1, k] = np.linalg.norm(w)
            if Hm[k + 1, k] > 0:
                V[k + 1] = w / Hm[k + 1, k]
            for j in range(k):
                t = cs[j] * Hm[j, k] + sn[j] * Hm[j + 1, k]
                Hm[j + 1, k] = -sn[j] * Hm[j, k] + cs[j] * Hm[j + 1, k]
                Hm[j, k] = t
            den = np.hypot(Hm[k, k], Hm[k + 1, k])
            cs[k], sn[k] = Hm[k, k] / den, Hm[k + 1, k] / den
            Hm[k, k] = den
            Hm[k + 1, k] = 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            its += 1
            k_used = k + 1
            hist.append(abs(g[k + 1]))
            if abs(g[k + 1]) <= ttol or its >= maxit:
                break
        y = np.linalg.solve(np.triu(Hm[:k_used, :k_used]), g[:k_used])
        x = x + y @ V[:k_used]
        if hist[-1] <= ttol or its >= maxit:
            return x, its, hist


def solve_kle(mats, vort, vel_bc, method="cg", **kw):
    """FreeSlip.solveKLE: rhs = Rw*vort + Krhs*vel ; K x = rhs.  base_problem.py:479-481."""
    rhs = mats["Rw"] @ vort + mats["Krhs"] @ vel_bc
    if method == "lu":                                          # ksp_solver.py:13-16 default
        import scipy.sparse.linalg as spla
        return spla.spsolve(mats["K"].tocsc(), rhs), rhs
    fn = pcg if method == "cg" else gmres
    x, its, hist = fn(mats["K"], rhs, **kw)
    return x, rhs
