#!/usr/bin/env python3
"""The KLE stiffness product y = K x on a box mesh under the product's variants, one process: block-CSR values with 8 / 16 / 32 / 64 lanes per
node row (PYNAMA_BCSR_LANES), the SELL-64 image (PYNAMA_BLOCK_SELL=1), and the Jacobi-PCG iteration on top of the default.
usage: block_spmv_case.py dim ngl nel [variants, comma separated: d (default), 8, 16, 32, 64, sell]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

dim, ngl, nel = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variants = (sys.argv[4] if len(sys.argv) > 4 else "d,8,16,32,64,sell").split(",")
dom = DMPlexDom(boxMesh={"nelem": [nel] * dim, "lower": [0] * dim, "upper": [1] * dim})
dom.setFemIndexing(ngl)
ctx = dom.ctx
for t in Spectral(ngl, dim).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(dim, np.repeat(bm[:, None], dim, axis=1))
n_rows, nnzb = ctx.csr_symbolic()
K = ctx.mat_create(dim, dim)
ctx.assemble_kle(1e3, 1e2, K, -1, -1, -1)
N, nnz = n_rows * dim, nnzb * dim * dim
b_bcsr = 8 * nnz + 4 * nnzb + 4 * (n_rows + 1) + 16 * N
vb, vx, vy = ctx.vec_create(dim), ctx.vec_create(dim), ctx.vec_create(dim)
ctx.vec_set(vb, np.random.default_rng(0).standard_normal(N))
print(f"{dim}-D ngl {ngl} {nel}^{dim}: {N} DOFs, nnz {nnz}, block-CSR bytes per product {b_bcsr / 1e9:.3f} GB", flush=True)
ref = None
for v in variants:
    if v == "d":
        env = {}
    elif v == "sell":
        env = {"PYNAMA_BLOCK_SELL": "1"}
    elif v == "bcsr":                       # block-CSR kernel with its default lanes (no LDS-staged lane-per-row kernel)
        env = {"PYNAMA_NO_CSRLB": "1"}
    elif v.startswith("lb"):                # LDS-staged lane-per-row kernel with N workgroups per CU
        env = {"PYNAMA_CSRLB_WGS_PER_CU": v[2:]} if len(v) > 2 else {}
    else:                                   # lanes[uUNROLL][wWGS_PER_CU], e.g. 32u3w6
        import re
        m = re.fullmatch(r"(\d+)(?:u(\d+))?(?:w(\d+))?", v)
        env = {"PYNAMA_BCSR_LANES": m.group(1), "PYNAMA_NO_CSRLB": "1"}
        if m.group(2):
            env["PYNAMA_BCSR_UNROLL"] = m.group(2)
        if m.group(3):
            env["PYNAMA_BCSR_WGS_PER_CU"] = m.group(3)
    os.environ.update(env)
    try:
        ts = []
        for _ in range(25):
            ctx.spmv(K, vb, vy)
            ts.append(ctx.timers()["spmv_ms"])
        y = ctx.vec_get(vy, dim)
        if ref is None:
            ref = y
        t = float(np.median(ts[5:]))
        info = ctx.solve(K, vb, vx, rtol=1e-30, fixed_iters=50, norm_type=_lib.NORM_UNPRECONDITIONED)
        print(f"  {v:8s}: product {t:.4f} ms = {b_bcsr / t / 1e6:6.0f} GB/s ({b_bcsr / t / 8e9:.3f} of 8 TB/s) | CG {info.solve_ms / 50:.4f} ms/iteration "
              f"({(b_bcsr + 132 * N) * 50 / info.solve_ms / 8e9:.3f}) | vs first variant {np.abs(y - ref).max() / np.abs(ref).max():.1e}", flush=True)
    finally:
        for k in env:
            del os.environ[k]
ctx.close()
