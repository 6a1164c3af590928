"""Worker of tests/test_gpu_dist.py: one rank of a world_size-N job whose ranks SHARE ONE GPU.

The product's distributed device path end to end -- slab partition, owner-computes assembly, halo exchange, single-reduction
CG with all-reduced scalars, matrix-free products on slabs, GMRES -- with the library's shared-memory TEST transport in the
place of RCCL (which refuses two ranks on one device).  Every rank checks its owned part against the serial oracle.
usage: dist_gpu_worker.py <rank> <size> <shm file> <nx,ny,nz> <case>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

rank, size, shm = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nelem = [int(v) for v in sys.argv[4].split(",")]
case = sys.argv[5]
os.environ["PYNAMA_SHM_TRANSPORT"] = shm

from oracle import fem_oracle as fo  # noqa: E402
from pynama_amd import _lib  # noqa: E402
from pynama_amd.common.comm import Comm  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402
from pynama_amd.elements.spectral import Spectral  # noqa: E402

jitter = 0.2 if "jitter" in case else 0.0
dom = DMPlexDom(boxMesh={"nelem": nelem, "lower": [0.0] * 3, "upper": [1.0] * 3}, comm=Comm(rank, size), jitter=jitter)
dom.setFemIndexing(2)
ctx = dom.ctx
for t in Spectral(2, 3).deviceTables():
    ctx.tables_set(*t)
st = ctx.comm_selftest()              # the start-up self-test of bench.py --gpus N on the test transport (RCCL counts nothing here)
assert st["transport"].startswith("shm") and "nranks_seen_by_rccl" not in st and st["allreduce_sum_ones"] == size, st
assert st["allreduce_beside_exchange_sum_ranks_plus_1"] == size * (size + 1) / 2, st
glob = fo.box_mesh(nelem, [0.0] * 3, [1.0] * 3, 2, jitter=jitter)
tb = fo.Tables(2, 3)
rng = np.random.default_rng(5)
ok = True
msg = []

if case.startswith("poisson"):
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(1, bm)
    ctx.csr_symbolic()
    A = ctx.mat_create(1, 1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
    ctx.matfree_set(_lib.MATFREE_LAPLACE)
    ref = fo.assemble_scalar(glob, tb, "laplace", dirichlet=glob.boundary)
    b_glob = rng.standard_normal(glob.n_node)
    b_glob[glob.boundary] = 0.0
    x_ref, it_ref, _ = fo.pcg(ref["A"], b_glob, rtol=1e-10, norm_type=fo.NORM_UNPRECONDITIONED)
    sl = slice(dom.rStart, dom.rEnd)
    vb, vx, vy = ctx.vec_create(1), ctx.vec_create(1), ctx.vec_create(1)
    ctx.vec_set(vb, b_glob[sl])
    # products: ghosts travel through the halo exchange
    y_ref = (ref["A"] @ b_glob)[sl]
    ctx.spmv(A, vb, vy)
    e_sp = np.abs(ctx.vec_get(vy, 1) - y_ref).max() / np.abs(y_ref).max()
    ctx.matfree_apply(vb, vy)
    e_mf = np.abs(ctx.vec_get(vy, 1) - y_ref).max() / np.abs(y_ref).max()
    ok &= e_sp < 2e-13 and e_mf < 2e-13
    msg.append(f"spmv {e_sp:.1e} matfree {e_mf:.1e}")
    kw = dict(rtol=1e-10, norm_type=_lib.NORM_UNPRECONDITIONED)
    for name, extra in (("cg-sr", {}), ("cg-std", dict(cg_variant=1)), ("cg-matfree", dict(matfree=_lib.MATFREE_LAPLACE)),
                        ("gmres", dict(method=_lib.KSP_GMRES, restart=30))):
        info = ctx.solve(A, vb, vx, **dict(kw, **extra))
        err = np.abs(ctx.vec_get(vx, 1) - x_ref[sl]).max() / np.abs(x_ref).max()
        good = info.reason == 2 and info.true_resid <= 1.05e-10 and err < 1e-7
        if name.startswith("cg"):
            good &= abs(info.iters - it_ref) <= 1
        ok &= bool(good)
        msg.append(f"{name}: {info.iters} its (serial {it_ref}) err {err:.1e}")
else:   # kle: the reference's solveKLE system with uniform-flow boundary data, assembled and matrix-free
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(3, np.repeat(bm[:, None], 3, axis=1))
    ctx.csr_symbolic()
    K, Krhs = ctx.mat_create(3, 3), ctx.mat_create(3, 3)
    ctx.assemble_kle(1e3, 1e2, K, Krhs)
    ctx.matfree_set(_lib.MATFREE_KLE, 1e3, 1e2)
    vel = np.zeros((dom.nOwned, 3))
    vel[bm[:dom.nOwned] != 0] = [1.0, 0.5, -0.25]
    vv, vb, vx = ctx.vec_create(3), ctx.vec_create(3), ctx.vec_create(3)
    ctx.vec_set(vv, vel.ravel())
    ctx.spmv(Krhs, vv, vb)
    for name, mf in (("kle", 0), ("kle-matfree", _lib.MATFREE_KLE)):
        info = ctx.solve(K, vb, vx, rtol=1e-10, norm_type=_lib.NORM_UNPRECONDITIONED, matfree=mf)
        err = np.abs(ctx.vec_get(vx, 3).reshape(-1, 3) - [1.0, 0.5, -0.25]).max()
        ok &= bool(info.reason == 2 and info.true_resid <= 1.05e-10 and err < 1e-7)
        msg.append(f"{name}: {info.iters} its err {err:.1e}")

tot = ctx.allreduce([1.0 if ok else 0.0])[0]
print(f"rank {rank}/{size}: {'; '.join(msg)} ok={ok}", flush=True)
ctx.close()
sys.exit(0 if tot == size else 1)
