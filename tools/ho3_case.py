#!/usr/bin/env python3
"""Second-order (ngl = 3) KLE on a box mesh, the element order of every reference case (src/cases/*.yaml): symbolic phase, assembly of
K + Krhs + Rw by the row-run kernels (whole and per matrix, against the generic atomics kernel), block product, Jacobi-PCG rate and the
uniform-flow solve to 1e-10.   usage: ho3_case.py dim nel [reps]      (PYNAMA_HO3_RUN: rows per run; HO3_GENERIC=1 also times variant 0)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nel = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if dim == 2 else 64)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dw = 1 if dim == 2 else 3
t0 = time.time()
dom = DMPlexDom(boxMesh={"nelem": [nel] * dim, "lower": [0] * dim, "upper": [1] * dim})
dom.setFemIndexing(3)
ctx = dom.ctx
for t in Spectral(3, dim).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(dim, np.repeat(bm[:, None], dim, axis=1))
t1 = time.time()
n_rows, nnzb = ctx.csr_symbolic()
n_elem = nel ** dim
nn = 3 ** dim
print(f"{dim}-D {nel}^{dim} ngl 3: {n_rows} nodes, {n_rows * dim} DOFs, nnzb {nnzb} ({nnzb / n_elem:.1f} per element), topology {ctx.mesh_topology()}, "
      f"host mesh {t1 - t0:.2f} s, symbolic {ctx.timers()['symbolic_ms']:.2f} ms", flush=True)
K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)


def timed(k, kr, rw, variant=1, fresh=False):
    ts = []
    for _ in range(reps + 1):
        if fresh and kr >= 0:
            ctx.mat_zero(kr)            # a zeroed matrix fits any Dirichlet set: the skip of Krhs' zero blocks stays legal, nothing else is cached
        ctx.assemble_kle(1e3, 1e2, k, kr, rw, -1, variant)
        ts.append(ctx.timers()["assemble_ms"])
    return float(np.median(ts[1:]))


# bytes: SURVEY.md 8(d) per matrix: conn + xyz + rowptr + colidx + values
def b_asm(br, bc):
    return 4 * nn * n_elem + 8 * dim * n_rows + 4 * (n_rows + 1) + 4 * nnzb + 8 * nnzb * br * bc


os.environ["PYNAMA_HO3_REQUIRE"] = "1"
os.environ["PYNAMA_RHS_FULL_WRITE"] = "1"
t_all_full = timed(K, Krhs, Rw)
del os.environ["PYNAMA_RHS_FULL_WRITE"]
t_all = timed(K, Krhs, Rw)
t_kk = timed(K, Krhs, -1)
t_k = timed(K, -1, -1)
t_rw = timed(-1, -1, Rw)
del os.environ["PYNAMA_HO3_REQUIRE"]
B3 = 2 * b_asm(dim, dim) + b_asm(dim, dw)
edofs = n_elem * nn * dim
print(f"assembly K+Krhs+Rw: {t_all_full:.3f} ms with Krhs written in full = {B3 / t_all_full / 1e6:.0f} GB/s of the three-matrix model ({B3 / t_all_full / 8e9:.3f} of 8 TB/s), "
      f"{edofs / t_all_full / 1e6:.2f} G element-DOFs/s", flush=True)
print(f"          Krhs' zero blocks skipped: {t_all:.3f} ms | K+Krhs {t_kk:.3f} | K {t_k:.3f} ({b_asm(dim, dim) / t_k / 8e9:.3f}) | Rw {t_rw:.3f} ({b_asm(dim, dw) / t_rw / 8e9:.3f})",
      flush=True)
if os.environ.get("HO3_GENERIC"):
    reps_save, reps = reps, 1
    tg = timed(K, Krhs, Rw, variant=0)
    reps = reps_save
    print(f"generic atomics kernel (variant 0): {tg:.2f} ms ({tg / t_all_full:.1f} x)", flush=True)
    ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)

# ---- product and CG on K
vb, vx, vy = ctx.vec_create(dim), ctx.vec_create(dim), ctx.vec_create(dim)
rng = np.random.default_rng(0)
ctx.vec_set(vb, rng.standard_normal(n_rows * dim))
ctx.spmv(K, vb, vy)
ts = []
for _ in range(20):
    ctx.spmv(K, vb, vy)
    ts.append(ctx.timers()["spmv_ms"])
t_sp = float(np.median(ts))
nnz = nnzb * dim * dim
N = n_rows * dim
b_bcsr = 8 * nnz + 4 * nnzb + 4 * (n_rows + 1) + 16 * N
b_csr = 12 * nnz + 4 * (N + 1) + 16 * N
print(f"K product: {t_sp:.3f} ms = {b_bcsr / t_sp / 1e6:.0f} GB/s block-CSR bytes ({b_bcsr / t_sp / 8e9:.3f}); scalar-CSR model {b_csr / t_sp / 8e9:.3f}", flush=True)
info = ctx.solve(K, vb, vx, rtol=1e-30, fixed_iters=100, norm_type=_lib.NORM_UNPRECONDITIONED)
b_cg = b_bcsr + 132 * N
print(f"Jacobi-PCG: {info.solve_ms / 100:.3f} ms per iteration = {100e3 / info.solve_ms:.0f} it/s, {b_cg * 100 / info.solve_ms / 1e6:.0f} GB/s ({b_cg * 100 / info.solve_ms / 8e9:.3f})",
      flush=True)
# uniform flow (src/cases/uniform.py:35-37): rhs = Krhs v_bc, exact solution v = const
cte = np.array([1.0, 0.5, -0.25][:dim])
vel = np.zeros((n_rows, dim))
vel[bm.astype(bool)] = cte
vv, vr = ctx.vec_create(dim), ctx.vec_create(dim)
ctx.vec_set(vv, vel.ravel())
ctx.spmv(Krhs, vv, vr)
t_kr = []
for _ in range(5):
    ctx.spmv(Krhs, vv, vr)
    t_kr.append(ctx.timers()["spmv_ms"])
info = ctx.solve(K, vr, vx, rtol=1e-10, norm_type=_lib.NORM_UNPRECONDITIONED, maxit=100000)
err = np.abs(ctx.vec_get(vx, dim).reshape(-1, dim) - cte).max()
print(f"uniform flow: Krhs product {np.median(t_kr):.3f} ms, CG {info.iters} iterations in {info.solve_ms:.1f} ms, reason {info.reason}, true residual {info.true_resid:.2e}, max error {err:.2e}",
      flush=True)
ctx.close()
