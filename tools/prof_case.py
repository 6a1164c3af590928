#!/usr/bin/env python3
"""Small driver for rocprofv3 counter passes: the bench workload, a few launches of one phase.
usage: prof_case.py [asm|cg|kle] [nel] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "asm"
if what.startswith("ho3"):
    # second-order cells (bench legs HO3_2D / HO3_3D): prof_case.py ho3k|ho3rw|ho3cg dim nel reps -- ONE kernel family per process, so that
    # the counters of a kernel name belong to one matrix shape
    dim, nel, reps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 3
    dw = 1 if dim == 2 else 3
    dom = DMPlexDom(boxMesh={"nelem": [nel] * dim, "lower": [0] * dim, "upper": [1] * dim})
    dom.setFemIndexing(3)
    ctx = dom.ctx
    for t in Spectral(3, dim).deviceTables():
        ctx.tables_set(*t)
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(dim, np.repeat(bm[:, None], dim, axis=1))
    ctx.csr_symbolic()
    K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create_rhs(dim, dim), ctx.mat_create(dim, dw)
    os.environ["PYNAMA_HO3_REQUIRE"] = "1"
    for _ in range(reps):
        if what == "ho3rw":
            ctx.assemble_kle(1e3, 1e2, -1, -1, Rw, -1)
        else:
            ctx.assemble_kle(1e3, 1e2, K, Krhs if what == "ho3k" else -1, -1, -1)
        print("assemble_ms", ctx.timers()["assemble_ms"])
    if what == "ho3cg":
        vb, vx = ctx.vec_create(dim), ctx.vec_create(dim)
        ctx.vec_set(vb, np.random.default_rng(0).standard_normal(dom.nOwned * dim))
        info = ctx.solve(K, vb, vx, fixed_iters=reps * 10, profile=1)
        print("cg ms/iter", info.solve_ms / info.iters, "spmv_ms", info.spmv_ms)
    ctx.close()
    sys.exit(0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 215
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
variant = int(os.environ.get("PYNAMA_VARIANT", "1"))
dom = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]},
                jitter=float(os.environ.get("PYNAMA_JITTER", "0")))
dom.setFemIndexing(2)
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(1, bm)
ctx.csr_symbolic()
if variant == 2:      # explicit plan: the patch-plan kernel instead of the plan-free lattice kernel
    tile = tuple(int(v) for v in os.environ.get("PYNAMA_TILE", "7,7,7").split(","))
    ctx.patch_plan_set(*dom.patchPlan(tile))
A = ctx.mat_create(1, 1)
for _ in range(reps if what == "asm" else 1):
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1, variant=variant)
    print("assemble_ms", ctx.timers()["assemble_ms"])
if what == "cg":
    vb, vx = ctx.vec_create(1), ctx.vec_create(1)
    b = np.random.default_rng(0).standard_normal(dom.nOwned)
    b[bm != 0] = 0
    ctx.vec_set(vb, b)
    info = ctx.solve(A, vb, vx, fixed_iters=reps * 10, profile=1, cg_variant=int(os.environ.get('PYNAMA_CG_VARIANT', '0')))
    print("cg ms/iter", info.solve_ms / info.iters, "spmv_ms", info.spmv_ms)
    if os.environ.get("PYNAMA_MATFREE", "1") != "0" and ctx.mesh_topology()[0] == "lattice":
        ctx.matfree_set(_lib.MATFREE_LAPLACE)
        info = ctx.solve(A, vb, vx, fixed_iters=reps * 10, profile=1, matfree=_lib.MATFREE_LAPLACE)
        print("matrix-free cg ms/iter", info.solve_ms / info.iters, "product_ms", info.spmv_ms)
if what == "kle":     # C3: 3 DOFs per node, assembled block SpMV and the matrix-free K product
    mask = np.repeat(bm[:, None], 3, axis=1)
    ctx.bc_set(3, mask)
    K, Krhs, Rw = ctx.mat_create(3, 3), ctx.mat_create_rhs(3, 3), ctx.mat_create(3, 3)     # compact Krhs, as Mat.createEmptyKLEMats makes it
    for _ in range(reps):
        ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
        print("assemble_kle_ms", ctx.timers()["assemble_ms"])
    ctx.matfree_set(_lib.MATFREE_KLE, 1e3, 1e2)
    vel = np.zeros((dom.nOwned, 3))
    vel[bm != 0] = [1.0, 0.0, 0.0]
    vv, vr, vx = ctx.vec_create(3), ctx.vec_create(3), ctx.vec_create(3)
    ctx.vec_set(vv, vel.ravel())
    ctx.spmv(Krhs, vv, vr)
    for mf in (0, _lib.MATFREE_KLE):
        info = ctx.solve(K, vr, vx, fixed_iters=reps * 10, profile=1, matfree=mf)
        print("kle cg matfree", mf, "ms/iter", info.solve_ms / info.iters, "product_ms", info.spmv_ms)
ctx.close()
