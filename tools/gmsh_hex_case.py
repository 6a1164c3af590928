#!/usr/bin/env python3
"""Imported (Gmsh) hexahedral mesh: n^3 box cells with random node numbering written to a file, read back through
DMPlexDom(fileName=...) (Morton renumbering), Poisson assembly with the patch-plan kernel + CG.
usage: gmsh_hex_case.py [n] [jitter]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.domain.gmsh import write_msh  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
jit = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
box = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]}, jitter=jit)
box.setFemIndexing(2)
perm = np.random.default_rng(7).permutation(box.nOwned)
path = os.path.join(tempfile.gettempdir(), f"pynama_hex_{os.getpid()}.msh")
t0 = time.time()
write_msh(path, box.xyz[np.argsort(perm)], perm[box.conn])
dom = DMPlexDom(fileName=path)
dom.setFemIndexing(2)
os.remove(path)
print(f"{n}^3 hexes: write + import {time.time() - t0:.1f} s")
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(1, bm)
n_rows, nnz = ctx.csr_symbolic()
print("topology", ctx.mesh_topology()[0], "rows", n_rows, "nnz", nnz)
A = ctx.mat_create(1, 1)
for _ in range(3):
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
ms = ctx.timers()["assemble_ms"]
B = 4 * 8 * n ** 3 + 24 * n_rows + 4 * (n_rows + 1) + 12 * nnz
print(f"assemble ms {ms:.3f} -> {n ** 3 * 8 / ms / 1e6:.1f} G element-DOFs/s, {B / ms / 1e6:.0f} GB/s algorithmic")
vb, vx = ctx.vec_create(1), ctx.vec_create(1)
f = np.ones(n_rows) / n ** 3
f[bm != 0] = 0
ctx.vec_set(vb, f)
info = ctx.solve(A, vb, vx, fixed_iters=100, profile=1)
print(f"cg {info.solve_ms / info.iters * 1e3:.1f} us/iter, spmv {info.spmv_ms * 1e3:.1f} us")
ctx.close()
