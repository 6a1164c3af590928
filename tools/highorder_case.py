#!/usr/bin/env python3
"""KLE assembly through the generic (any ngl / dim) kernel on high-order box meshes -- the reference's usual regime
(2-D, ngl 3..11).  usage: highorder_case.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

for dim, nel, ngl in ((2, 64, 3), (2, 64, 5), (2, 32, 8), (2, 16, 11), (3, 32, 3), (3, 16, 4), (3, 16, 5)):
    dom = DMPlexDom(boxMesh={"nelem": [nel] * dim, "lower": [0] * dim, "upper": [1] * dim})
    dom.setFemIndexing(ngl)
    ctx = dom.ctx
    for t in Spectral(ngl, dim).deviceTables():
        ctx.tables_set(*t)
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(dim, np.repeat(bm[:, None], dim, axis=1))
    n_rows, nnz = ctx.csr_symbolic()
    dw = 1 if dim == 2 else 3
    K, Kr, Rw = ctx.mat_create(dim, dim), ctx.mat_create(dim, dim), ctx.mat_create(dim, dw)
    for _ in range(2):
        ctx.assemble_kle(1e3, 1e2, K, Kr, Rw, -1)
    ms = ctx.timers()["assemble_ms"]
    nn = ngl ** dim
    print(f"{dim}-D nel {nel}^{dim} ngl {ngl}: nodes {n_rows}, K_e {dim * nn}x{dim * nn}: KLE assemble {ms:.2f} ms "
          f"({ms / nel ** dim * 1e3:.2f} us/element)")
    ctx.close()
