#!/usr/bin/env python3
"""Per-GPU share of the strong-scaling bench on ONE GPU: a 215 x 215 x (215/N) slab, with a real one-rank
RCCL communicator (PYNAMA_FORCE_COMM=1) so that the collective code path runs.  Gives the compute+launch
floor of a rank at N GPUs (no wire time).   usage: slab_case.py [N] [cg_iters]"""
import os
import sys

import numpy as np

os.environ.setdefault("PYNAMA_FORCE_COMM", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nz = max(1, round(215 / N))
dom = DMPlexDom(boxMesh={"nelem": [215, 215, nz], "lower": [0, 0, 0], "upper": [1, 1, nz / 215]})
dom.setFemIndexing(2)
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(1, bm)
n_rows, nnz = ctx.csr_symbolic()
if os.environ.get("PYNAMA_SLAB_PLAN"):      # the patch-plan kernels instead of the plan-free lattice ones the bench runs
    ctx.patch_plan_set(*dom.patchPlan((7, 7, 7)))
A = ctx.mat_create(1, 1)
for _ in range(3):
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
print(f"N={N}: rows {n_rows} nnz {nnz} assemble ms {ctx.timers()['assemble_ms']:.3f}")
f = np.random.default_rng(0).standard_normal(n_rows) / 215 ** 3
f[bm != 0] = 0
vb, vx = ctx.vec_create(1), ctx.vec_create(1)
ctx.vec_set(vb, f)
for variant in (1, 2):
    for _ in range(2):
        info = ctx.solve(A, vb, vx, fixed_iters=iters, cg_variant=variant, norm_type=_lib.NORM_UNPRECONDITIONED, profile=1)
    print(f"  cg variant {variant}: {info.solve_ms / info.iters * 1e3:.1f} us/iter ({info.iters / info.solve_ms * 1e3:.0f} it/s), spmv {info.spmv_ms * 1e3:.1f} us")
ctx.close()
