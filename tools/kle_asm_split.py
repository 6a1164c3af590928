import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynama_amd.domain.dmplex import DMPlexDom
from pynama_amd.elements.spectral import Spectral
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
jit = float(os.environ.get("PYNAMA_JITTER", "0"))
dom = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]}, jitter=jit)
dom.setFemIndexing(2)
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
bm = dom.boundaryMaskLocal()
ctx.bc_set(3, np.repeat(bm[:, None], 3, axis=1))
ctx.csr_symbolic()
tile = os.environ.get("PYNAMA_KLE_TILE")       # explicit patch plan -> patch-plan kernels; default: the library's choice
if tile:
    tile = tuple(int(v) for v in tile.split(","))
    ctx.patch_plan_set(*dom.patchPlan(tile), kind=1)
K, Krhs, Rw = ctx.mat_create(3, 3), ctx.mat_create(3, 3), ctx.mat_create(3, 3)
for name, args in (("K only", (K, -1, -1)), ("K+Krhs", (K, Krhs, -1)), ("K+Krhs+Rw", (K, Krhs, Rw))):
    for _ in range(2):
        ctx.assemble_kle(1e3, 1e2, args[0], args[1], args[2], -1, variant=1)
    print(tile, "jitter", jit, name, "ms", round(ctx.timers()["assemble_ms"], 3))
ctx.close()
