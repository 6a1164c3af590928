#!/usr/bin/env python3
"""Assembly time of ONE rank's slab of the strong-scaling bench (215^3 mesh split in N z-slabs), processed in
isolation through the detached communicator.   usage: slab_asm.py [N] [rank]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd import _lib  # noqa: E402
from pynama_amd.common.comm import Comm  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = int(sys.argv[2]) if len(sys.argv) > 2 else N // 2
dom = DMPlexDom(boxMesh={"nelem": [215, 215, 215], "lower": [0, 0, 0], "upper": [1, 1, 1]}, comm=Comm(r, N))
dom.setFemIndexing(2)
ctx = _lib.Context(0)
ctx.comm_init(r, N, None)
ctx.halo_set(*dom._halo_plan())
ctx.mesh_set(3, dom.conn, dom.xyz)
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
ctx.bc_set(1, dom.boundaryMaskLocal())
n_rows, nnz = ctx.csr_symbolic()
A = ctx.mat_create(1, 1)
ts = []
for _ in range(6):
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
    ts.append(ctx.timers()["assemble_ms"])
print(f"N={N} rank {r}: topology {ctx.mesh_topology()} rows {n_rows} assemble ms {min(ts):.4f} (runs {[round(t, 4) for t in ts]})"
      f" -> {215 ** 3 * 8 / N / min(ts) / 1e6:.1f} G elem-DOF/s per rank-share")
ctx.close()
