#!/usr/bin/env python3
"""Cost of a step outside the two metered phases on the bench mesh: assembly -> first solve (SELL image refresh + CG start-up)
-> second solve of the same matrix (nothing to refresh).  usage: sell_fill_case.py [nel]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
dom = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]})
dom.setFemIndexing(2)
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(1, bm)
ctx.csr_symbolic()
A = ctx.mat_create(1, 1)
vb, vx = ctx.vec_create(1), ctx.vec_create(1)
b = np.random.default_rng(0).standard_normal(dom.nOwned)
b[bm != 0] = 0
ctx.vec_set(vb, b)
for rep in range(4):
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1)
    ctx.sync()
    t0 = time.perf_counter()
    i1 = ctx.solve(A, vb, vx, fixed_iters=10, norm_type=_lib.NORM_UNPRECONDITIONED)
    t1 = time.perf_counter()
    i2 = ctx.solve(A, vb, vx, fixed_iters=10, norm_type=_lib.NORM_UNPRECONDITIONED)
    t2 = time.perf_counter()
    print(f"first solve after assembly: wall {1e3 * (t1 - t0):.3f} ms (10 iterations {i1.solve_ms:.3f} ms) | second solve: wall "
          f"{1e3 * (t2 - t1):.3f} ms ({i2.solve_ms:.3f} ms) | set-up of a fresh matrix = {1e3 * ((t1 - t0) - (t2 - t1)):.3f} ms", flush=True)
ctx.close()
