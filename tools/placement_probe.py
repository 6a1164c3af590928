#!/usr/bin/env python3
"""Does the product's speed depend on WHERE the arrays landed, or on WHEN it runs?  Several matrices / vector sets in ONE process, same
kernel, same data: spmv_ms of each (the bimodal 0.49 / 0.55 ms seen between processes on one box).  WARM=<s>: that many seconds of CG first."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

n = 215
dom = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]})
dom.setFemIndexing(2)
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(1, bm)
ctx.csr_symbolic()
b = np.random.default_rng(0).standard_normal(dom.nOwned)
b[bm != 0] = 0
mats, pads = [], []
if os.environ.get("PRE_GB"):        # PRE_GB=<GiB>: a dummy device allocation of that size is made (and kept) before the first matrix
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    dummy = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(dummy), ctypes.c_size_t(int(float(os.environ["PRE_GB"]) * 2 ** 30)))
    print("dummy allocation rc", rc, hex(dummy.value or 0), flush=True)
if os.environ.get("WARM"):          # WARM=<seconds of CG before the first measured matrix>, on a matrix that is destroyed again
    import time
    A = ctx.mat_create(1, 1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1)
    vb, vx = ctx.vec_create(1), ctx.vec_create(1)
    ctx.vec_set(vb, b)
    t0 = time.time()
    while time.time() - t0 < float(os.environ["WARM"]):
        ctx.solve(A, vb, vx, fixed_iters=30)
    ctx.mat_destroy(A)
for trial in range(6):
    A = ctx.mat_create(1, 1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1)
    vb, vx = ctx.vec_create(1), ctx.vec_create(1)
    ctx.vec_set(vb, b)
    ts = []
    for rep in range(3):
        info = ctx.solve(A, vb, vx, fixed_iters=30, profile=1)
        ts.append(info.spmv_ms)
    print(f"matrix {trial}: spmv_ms {ts[0]:.4f} {ts[1]:.4f} {ts[2]:.4f}   cg ms/iter {info.solve_ms / info.iters:.4f}", flush=True)
    mats.append(A)
    pads.append(ctx.vec_create(1))      # shift the next allocations a little
ctx.close()
