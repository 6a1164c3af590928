#!/usr/bin/env python3
"""Assembled (SELL-64) vs matrix-free Laplacian on the bench mesh: one product and the Jacobi-PCG iteration.
usage: matfree_case.py [nel=215] [jitter=0] [iters=100]   (PYNAMA_MATFREE_TILE selects the tile shape)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
jitter = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dom = DMPlexDom(boxMesh={"nelem": [n] * 3, "lower": [0] * 3, "upper": [1] * 3}, jitter=jitter)
dom.setFemIndexing(2)
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(1, bm)
n_rows, nnz = ctx.csr_symbolic()
A = ctx.mat_create(1, 1)
ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
ctx.matfree_set(_lib.MATFREE_LAPLACE)
f = np.random.default_rng(0).standard_normal(n_rows) / n ** 3
f[bm != 0] = 0
vb, vx, vy, vz = (ctx.vec_create(1) for _ in range(4))
ctx.vec_set(vb, f)
for _ in range(3):
    ctx.spmv(A, vb, vy)
t_sell = ctx.timers()["spmv_ms"]
for _ in range(3):
    ctx.matfree_apply(vb, vz)
t_mf = ctx.timers()["spmv_ms"]
y0, y1 = ctx.vec_get(vy, 1), ctx.vec_get(vz, 1)
err = np.abs(y0 - y1).max() / np.abs(y0).max()
mf_bytes = n_rows * (8 + 8 + 24 + 1)
print(f"nel {n} jitter {jitter} tile {os.environ.get('PYNAMA_MATFREE_TILE', '0')}: rows {n_rows} nnz {nnz}; product: SELL {t_sell:.3f} ms, "
      f"matrix-free {t_mf:.3f} ms ({mf_bytes / t_mf / 1e6:.0f} GB/s of its own {mf_bytes / 1e6:.0f} MB), max rel diff {err:.2e}")
for mf in (0, 1):
    for _ in range(2):
        info = ctx.solve(A, vb, vx, fixed_iters=iters, norm_type=_lib.NORM_UNPRECONDITIONED, profile=1, matfree=mf)
    print(f"  CG matfree={mf}: {info.solve_ms / info.iters * 1e3:.1f} us/iter ({info.iters / info.solve_ms * 1e3:.0f} it/s), "
          f"product {info.spmv_ms * 1e3:.1f} us")
for mf in (0, 1):
    info = ctx.solve(A, vb, vx, rtol=1e-10, maxit=20000, norm_type=_lib.NORM_UNPRECONDITIONED, matfree=mf)
    print(f"  solve to 1e-10 matfree={mf}: {info.iters} its, {info.solve_ms:.1f} ms, true residual {info.true_resid:.2e}")
ctx.close()
