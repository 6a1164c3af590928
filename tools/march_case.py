#!/usr/bin/env python3
"""General-geometry (jittered) scalar assembly on the bench mesh: z-marching kernel variants against the 7x7x7 tile
kernel, values compared in-process, interleaved timing rounds (same box, same process).
usage: march_case.py [nel] [rounds] [variants: comma list of PYNAMA_MARCH_TILE ids, 't' = tile kernel]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = (sys.argv[3] if len(sys.argv) > 3 else "t,0,1,2,3,4").split(",")
dom = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]},
                jitter=float(os.environ.get("PYNAMA_JITTER", "0.2")))
dom.setFemIndexing(2)
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(1, bm)
n_rows, nnz = ctx.csr_symbolic()
A, Ar = ctx.mat_create(1, 1), ctx.mat_create(1, 1)


def run(v, with_rhs=False):
    for k in ("PYNAMA_NO_MARCH", "PYNAMA_MARCH_TILE", "PYNAMA_NO_LEAN", "PYNAMA_LATTICE_TILE", "PYNAMA_LATTICE_ABLATE"):
        os.environ.pop(k, None)
    if "a" in v and not v.startswith("l"):     # <march tile>a<ablate code>, e.g. 0a16 = default shape, boundary columns through the CSR-slot decode
        v, ab = v.split("a")
        os.environ["PYNAMA_LATTICE_ABLATE"] = ab
    if v == "t":          # 7x7x7 tile kernel with the table-driven Gauss loop (round 1)
        os.environ["PYNAMA_NO_MARCH"] = "1"
        os.environ["PYNAMA_NO_LEAN"] = "1"
    elif v.startswith("l"):    # tile kernel with the lean closed-form element routine, l<tile id>
        os.environ["PYNAMA_NO_MARCH"] = "1"
        if len(v) > 1:
            os.environ["PYNAMA_LATTICE_TILE"] = v[1:]
    else:
        os.environ["PYNAMA_MARCH_TILE"] = v
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, Ar if with_rhs else -1)
    return ctx.timers()["assemble_ms"]


run("t", True)
ref, ref_r = ctx.mat_values(A, 1, 1).copy(), ctx.mat_values(Ar, 1, 1).copy()
scale = np.abs(ref).max()
for v in variants:
    if v == "t":
        continue
    ctx.mat_zero(A) if hasattr(ctx, "mat_zero") else None
    run(v, True)
    e = np.abs(ctx.mat_values(A, 1, 1) - ref).max() / scale
    er = np.abs(ctx.mat_values(Ar, 1, 1) - ref_r).max() / scale
    print(f"variant {v}: max rel diff vs tile kernel A {e:.2e} Arhs {er:.2e}", flush=True)
times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        times[v].append(run(v))
B_asm = 4 * 8 * n ** 3 + 24 * (n + 1) ** 3 + 4 * (n_rows + 1) + 12 * nnz
for v in variants:
    t = np.array(times[v])
    print(f"variant {v}: median {np.median(t):.3f} ms min {t.min():.3f} ms  -> {B_asm / np.median(t) / 1e6:.0f} GB/s algorithmic "
          f"({B_asm / np.median(t) / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
ctx.close()
