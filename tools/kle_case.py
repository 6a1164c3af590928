#!/usr/bin/env python3
"""128^3 KLE assembly (K + Krhs + Rw) under the diagnostic switches, same process: time per call of the whole assembly and of K / Rw alone.
usage: kle_case.py [nel] [reps] [jitter]   (PYNAMA_LATTICE_ABLATE / PYNAMA_KLE_LATTICE_TILE are read per call by the library)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
jit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dom = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]}, jitter=jit)
dom.setFemIndexing(2)
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(3, np.repeat(bm[:, None], 3, axis=1))
ctx.csr_symbolic()
K, Krhs, Rw = ctx.mat_create(3, 3), ctx.mat_create(3, 3), ctx.mat_create(3, 3)


def timed(env, k, kr, rw):
    for a, b in env.items():
        os.environ[a] = b
    try:
        ts = []
        for _ in range(reps + 1):
            ctx.assemble_kle(1e3, 1e2, k, kr, rw, -1)
            ts.append(ctx.timers()["assemble_ms"])
        return float(np.median(ts[1:]))
    finally:
        for a in env:
            del os.environ[a]


cases = [("full", {}), ("no element phase", {"PYNAMA_LATTICE_ABLATE": "1"}), ("all tiles through the CSR-slot store", {"PYNAMA_LATTICE_ABLATE": "4"}),
         ("Krhs written in full", {"PYNAMA_RHS_FULL_WRITE": "1"})]
if os.environ.get("KLE_QUICK"):
    cases = cases[:1]
for tl in os.environ.get("KLE_TILES", "").split(","):
    if tl:
        cases.append((f"tile {tl}", {"PYNAMA_KLE_LATTICE_TILE": tl}))
for name, env in cases:
    print(f"{name:40s} K+Krhs+Rw {timed(env, K, Krhs, Rw):.3f} ms | K+Krhs {timed(env, K, Krhs, -1):.3f} | K {timed(env, K, -1, -1):.3f} | Rw {timed(env, -1, -1, Rw):.3f}",
          flush=True)
ctx.close()
