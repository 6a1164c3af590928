#!/usr/bin/env python3
"""C5 of BASELINE.json: unstructured tetrahedral mesh (n^3 hexes cut in 6 tets, random node numbering),
written to and imported from a Gmsh file, Poisson with GMRES(30)+Jacobi.
usage: tet_case.py [n] [reorder: morton|none]   (94 -> 4,983,504 tetrahedra)"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.domain.gmsh import write_msh  # noqa: E402
from pynama_amd.elements.simplex import Simplex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 94
reorder = None if (len(sys.argv) > 2 and sys.argv[2] == "none") else "morton"


def kuhn_box(n, seed=2024):
    """n^3 unit-box hexes -> 6 n^3 positively oriented tetrahedra, nodes randomly renumbered"""
    from itertools import permutations
    lat = n + 1
    strides = np.array([1, lat, lat * lat])
    i, j, k = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    base = (i * strides[0] + j * strides[1] + k * strides[2]).ravel()
    conn = []
    for perm in permutations(range(3)):
        offs = [0]
        for d in perm:
            offs.append(offs[-1] + strides[d])
        if sum(1 for a in range(3) for b in range(a + 1, 3) if perm[a] > perm[b]) % 2:
            offs[-1], offs[-2] = offs[-2], offs[-1]
        conn.append(base[:, None] + np.array(offs)[None, :])
    conn = np.stack(conn, axis=1).reshape(-1, 4)
    ax = np.linspace(0.0, 1.0, lat)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    xyz = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    p = np.random.default_rng(seed).permutation(lat ** 3)
    return xyz[np.argsort(p)], p[conn]


t0 = time.time()
xyz, conn = kuhn_box(n)
path = os.path.join(tempfile.gettempdir(), f"pynama_c5_{os.getpid()}.msh")
write_msh(path, xyz, conn)
t1 = time.time()
dom = DMPlexDom(fileName=path, reorder=reorder)
dom.setFemIndexing(2)
os.remove(path)
t2 = time.time()
print(f"mesh {conn.shape[0]} tets, {xyz.shape[0]} nodes; write {t1 - t0:.1f} s, import+renumber {t2 - t1:.1f} s", flush=True)
ctx = dom.ctx
for t in Simplex(3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(1, bm)
n_rows, nnz = ctx.csr_symbolic()
print("rows", n_rows, "nnz", nnz, "symbolic ms", ctx.timers()["symbolic_ms"])
A = ctx.mat_create(1, 1)
for _ in range(3):
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
    ms = ctx.timers()["assemble_ms"]
B_asm = 4 * 4 * conn.shape[0] + 8 * 3 * n_rows + 4 * (n_rows + 1) + 12 * nnz
print(f"assemble ms {ms:.3f}  -> {conn.shape[0] * 4 / ms / 1e6:.2f} G element-DOFs/s, {B_asm / ms / 1e6:.0f} GB/s algorithmic")
X = dom.xyz
f = (1.0 + X[:, 0] + 2.0 * X[:, 1] ** 2 + np.exp(X[:, 0] * X[:, 1] * X[:, 2])) / n ** 3
f[bm != 0] = 0.0
vb, vx, vy = ctx.vec_create(1), ctx.vec_create(1), ctx.vec_create(1)
ctx.vec_set(vb, f)
for _ in range(3):
    ctx.spmv(A, vb, vy)
ms = ctx.timers()["spmv_ms"]
B_spmv = 12 * nnz + 4 * (n_rows + 1) + 16 * n_rows
print(f"spmv ms {ms:.4f} -> {B_spmv / ms / 1e6:.0f} GB/s algorithmic")
for og, name in ((0, "classical GS + refinement"), (1, "classical GS, no refinement"), (2, "modified GS")):
    info = ctx.solve(A, vb, vx, method=_lib.KSP_GMRES, pc=_lib.PC_JACOBI, fixed_iters=300, restart=30, gmres_orthog=og)
    print(f"gmres(30) [{name}] {info.iters} its in {info.solve_ms:.1f} ms -> {info.iters / info.solve_ms * 1e3:.0f} it/s")
    info = ctx.solve(A, vb, vx, method=_lib.KSP_GMRES, pc=_lib.PC_JACOBI, rtol=1e-10, restart=30, maxit=100000,
                     norm_type=_lib.NORM_UNPRECONDITIONED, gmres_orthog=og)
    print(f"   to 1e-10: its {info.iters} reason {info.reason} true_resid {info.true_resid:.3e} ms {info.solve_ms:.1f}")
info = ctx.solve(A, vb, vx, method=_lib.KSP_GMRES, pc=_lib.PC_JACOBI, rtol=1e-10, restart=30, maxit=100000,
                 norm_type=_lib.NORM_UNPRECONDITIONED)
print("gmres to 1e-10: its", info.iters, "reason", info.reason, "true_resid", info.true_resid, "ms", info.solve_ms)
info = ctx.solve(A, vb, vx, method=_lib.KSP_CG, pc=_lib.PC_JACOBI, rtol=1e-10, norm_type=_lib.NORM_UNPRECONDITIONED)
print("cg to 1e-10: its", info.iters, "reason", info.reason, "true_resid", info.true_resid, "ms", info.solve_ms)
ctx.close()
