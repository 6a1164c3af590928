"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N CPU job (gloo).

Runs the PRODUCT's host logic for N > 1 (slab partition, local numbering, halo plan of
pynama_amd.domain.dmplex) and drives the ORACLE's numerics through it: owner-computes assembly of
the local rows, halo exchange of ghost entries, all-reduced dot products -- the same communication
pattern libpynama_hip.so executes over RCCL (pyn_halo_exchange + ncclAllReduce)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import fem_oracle as fo  # noqa: E402
from pynama_amd.common.comm import Comm  # noqa: E402
from pynama_amd.domain.dmplex import DMPlexDom  # noqa: E402


def main():
    nelem = [int(v) for v in sys.argv[1].split(",")]
    ngl = int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    dim = len(nelem)
    msh = sys.argv[3] if len(sys.argv) > 3 else None          # imported (Gmsh) mesh instead of the box
    if msh:
        dom = DMPlexDom(fileName=msh, comm=Comm(rank, size))
    else:
        dom = DMPlexDom(boxMesh={"nelem": nelem, "lower": [0.0] * dim, "upper": [1.0] * dim}, comm=Comm(rank, size), jitter=0.2 if ngl == 2 else 0.0)
    dom.setFemIndexing(ngl)
    n_owned, n_ghost, neigh, send_ptr, send_idx, recv_ptr = dom._halo_plan()

    # ---- local owner-computes assembly with the oracle on the LOCAL mesh
    class Local:
        pass
    m = Local()
    m.dim, m.conn, m.xyz = dim, dom.conn, dom.xyz
    m.n_node, m.n_elem = dom.nLocal, dom.conn.shape[0]
    simplex = dom.conn.shape[1] == dim + 1
    tb = fo.SimplexTables(dim) if simplex else fo.Tables(ngl, dim)
    m.corners = lambda: m.xyz[m.conn[:, :tb.nc]].reshape(m.n_elem, -1)
    bmask = dom.boundaryMaskLocal()
    A_loc = fo.assemble_scalar(m, tb, "laplace", dirichlet=np.nonzero(bmask)[0])["A"][:n_owned]   # owned rows, local cols

    def halo(x):
        """fill the ghost part of x from the owners (same plan as pyn_halo_exchange)"""
        reqs, bufs = [], []
        for k, nb in enumerate(neigh):
            s = torch.from_numpy(np.ascontiguousarray(x[send_idx[send_ptr[k]:send_ptr[k + 1]]]))
            r = torch.empty(int(recv_ptr[k + 1] - recv_ptr[k]), dtype=torch.float64)
            bufs.append((k, r))
            reqs.append(dist.isend(s, int(nb)))
            reqs.append(dist.irecv(r, int(nb)))
        for q in reqs:
            q.wait()
        for k, r in bufs:
            x[n_owned + recv_ptr[k]:n_owned + recv_ptr[k + 1]] = r.numpy()

    def gdot(a, b):
        t = torch.tensor([float(a @ b)], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0])

    # ---- right-hand side: same global random vector on every rank, zero on the boundary
    if msh:                      # the whole mesh in the build's numbering, read by a one-rank domain
        one = DMPlexDom(fileName=msh, comm=Comm())
        one.setFemIndexing(ngl)
        glob = fo.BoxMesh(dim, 2, tuple(nelem), (), one.conn, one.xyz, np.nonzero(one.boundaryMaskLocal())[0], {}, nc=tb.nc)
    else:
        glob = fo.box_mesh(nelem, [0.0] * dim, [1.0] * dim, ngl, jitter=0.2 if ngl == 2 else 0.0)
    rng = np.random.default_rng(5)
    b_glob = rng.standard_normal(glob.n_node)
    b_glob[glob.boundary] = 0.0
    b = b_glob[dom.rStart:dom.rEnd].copy()

    # ---- distributed Jacobi-PCG (unpreconditioned norm, rtol 1e-10), owned entries only
    dinv = 1.0 / A_loc.diagonal()
    x = np.zeros(n_owned)
    r = b.copy()
    z = dinv * r
    p = np.zeros(dom.nLocal)
    p[:n_owned] = z
    rz = gdot(r, z)
    r0 = np.sqrt(gdot(r, r))
    its = 0
    for its in range(1, 5001):
        halo(p)
        Ap = A_loc @ p
        alpha = rz / gdot(p[:n_owned], Ap)
        x += alpha * p[:n_owned]
        r -= alpha * Ap
        z = dinv * r
        rzn = gdot(r, z)
        if np.sqrt(gdot(r, r)) <= 1e-10 * r0:
            break
        p[:n_owned] = z + (rzn / rz) * p[:n_owned]
        rz = rzn

    # ---- serial oracle on the global mesh
    ref = fo.assemble_scalar(glob, tb, "laplace", dirichlet=glob.boundary)
    x_ref, it_ref, _ = fo.pcg(ref["A"], b_glob, rtol=1e-10, norm_type=fo.NORM_UNPRECONDITIONED)
    err = np.abs(x - x_ref[dom.rStart:dom.rEnd]).max() / np.abs(x_ref).max()
    # owned rows of the distributed matrix == rows of the serial matrix (columns mapped to global ids)
    cols_glob = dom._local2global(np.arange(dom.nLocal))
    Ag = ref["A"][dom.rStart:dom.rEnd][:, cols_glob]
    derr = abs(Ag - A_loc).max() / abs(ref["A"]).max()
    ok = err < 1e-8 and abs(its - it_ref) <= 1 and derr < 1e-13
    print(f"rank {rank}: its {its} (serial {it_ref}) err {err:.2e} matrix err {derr:.2e} ok={ok}", flush=True)
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if t[0] == 1.0 else 1)


if __name__ == "__main__":
    main()
