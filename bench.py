#!/usr/bin/env python3
"""Headline benchmark of the Pynama hot path on MI355X (contract: see the driver's prompt).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric): 3-D Poisson, Q1 hexahedra, 215^3 elements = 216^3 =
10,077,696 DOFs, FP64, synthetic structured box mesh, homogeneous Dirichlet on all faces.
One *step* = one pass of the hot path over that mesh:
    (1) numeric assembly  (zero values + per-element quadrature + scatter with Dirichlet
        elimination; symbolic phase excluded)           -> element-DOFs/s
    (2) `--cg-iters` Jacobi-PCG iterations on the assembled matrix    -> CG iterations/s
Inputs (connectivity, coordinates, CSR pattern, vectors) are resident in HBM before timing.
After the timed steps the same CG runs once more with the matrix-free operator (`matrix_free` in the line).
N > 1: the SAME mesh is row-partitioned in z-slabs over the ranks ("scaling": "strong"), halo
planes and dot products go over RCCL inside libpynama_hip.so.

The product path is torch-free: ranks/LOCAL_RANK come from the launcher's environment, the RCCL
id is exchanged through a node-local file (pynama_amd/common/comm.py).
Only the `cpu_baseline` leg touches oracle/ (the C restatement, as the thing timed on the CPU).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def algorithmic_bytes(n_elem, n_node, nnz, nn=8, dim=3):
    """compulsory traffic, FP64 values, int32 indices."""
    asm = 4 * nn * n_elem + 8 * dim * n_node + 4 * (n_node + 1) + 4 * nnz + 8 * nnz
    spmv = 12 * nnz + 4 * (n_node + 1) + 16 * n_node
    cg_iter = 12 * nnz + 148 * n_node
    return asm, spmv, cg_iter


def cpu_baseline(dom, rp, ci, bmask, b, budget_s=8.0):
    """C/OpenMP restatement of the same path (oracle/c/fem_oracle.c) on the host cores, on a bounded
    sample of the SAME workload: whole-mesh assembly passes and Jacobi-PCG iterations on the
    assembled matrix, each sized to about `budget_s` seconds of CPU work."""
    from oracle import c_oracle as co
    from oracle import fem_oracle as fo

    class M:
        pass
    m = M()
    m.conn, m.xyz, m.n_elem, m.n_node = dom.conn, dom.xyz, dom.conn.shape[0], dom.xyz.shape[0]
    tb = fo.Tables(2, dom.dim)
    # the GPU box advertises every host core but grants a share (16 per GPU): do not oversubscribe
    cores = co.set_threads(min(co.usable_cores(), int(os.environ.get("PYNAMA_CPU_THREADS", "16"))))
    t0 = time.perf_counter()
    A, _ = co.assemble_laplace(m, tb, rp, ci, bmask, with_rhs=False)          # calibration pass (also the matrix)
    t_one = time.perf_counter() - t0
    reps = max(1, int(budget_s / max(t_one, 1e-3)))
    t0 = time.perf_counter()
    for _ in range(reps):
        co.assemble_laplace(m, tb, rp, ci, bmask, with_rhs=False)
    t_asm = time.perf_counter() - t0
    t0 = time.perf_counter()
    co.pcg(rp, ci, A, b, fixed_iters=5, norm_type=1)
    t5 = time.perf_counter() - t0
    iters = max(5, int(budget_s / max(t5 / 5, 1e-4)))
    t0 = time.perf_counter()
    co.pcg(rp, ci, A, b, fixed_iters=iters, norm_type=1)
    t_cg = time.perf_counter() - t0
    return {"value": reps * m.n_elem * 8 / t_asm, "unit": "element-DOFs/s", "cores": cores, "kind": "port",
            "cg_iters_per_s": iters / t_cg,
            "sample": f"C/OpenMP oracle on {cores} threads: {reps} assembly passes over all {m.n_elem} elements "
                      f"({t_asm:.1f} s) + {iters} Jacobi-PCG iterations on the full {m.n_node}-row matrix ({t_cg:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nel", type=int, default=215, help="elements per side (215 -> 10,077,696 DOFs)")
    ap.add_argument("--cg-iters", type=int, default=100, help="fixed CG iterations per step")
    ap.add_argument("--variant", type=int, default=1, help="assembly kernel: 0 generic atomics, 1 auto (plan-free lattice kernel on box meshes), "
                    "2 patch-plan kernel on 7x7x7 tiles")
    ap.add_argument("--jitter", type=float, default=0.0, help="diagnostics: perturb interior nodes by jitter*h (general geometry)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-matfree", action="store_true", help="skip the matrix-free CG leg (reported alongside, not the headline)")
    args = ap.parse_args()

    from pynama_amd import _lib
    from pynama_amd.common.comm import get_world
    from pynama_amd.domain.dmplex import DMPlexDom
    from pynama_amd.elements.spectral import Spectral

    world = get_world()
    if world.size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world.size}: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
    n = args.nel
    dom = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]}, jitter=args.jitter)
    dom.setFemIndexing(2)
    ctx = dom.ctx                                     # creates the context, RCCL comm, uploads the mesh
    for t in Spectral(2, 3).deviceTables():
        ctx.tables_set(*t)
    bmask = dom.boundaryMaskLocal()
    ctx.bc_set(1, bmask)
    n_rows, nnz = ctx.csr_symbolic()
    symbolic_ms = ctx.timers()["symbolic_ms"]
    if args.variant == 2:
        ctx.patch_plan_set(*dom.patchPlan((7, 7, 7)))       # explicit plan: the patch-plan kernel instead of the lattice one
    A = ctx.mat_create(1, 1)
    vb, vx = ctx.vec_create(1), ctx.vec_create(1)
    h = 1.0 / n
    X = dom.xyz[:dom.nOwned]
    # lumped load of a smooth non-separable source (NOT a discrete eigenvector: with
    # f = sin sin sin on a uniform grid CG would converge in one iteration)
    f = (1.0 + X[:, 0] + 2.0 * X[:, 1] ** 2 + np.exp(X[:, 0] * X[:, 1] * X[:, 2]) * np.cos(3.0 * X[:, 2])) * h ** 3
    f[bmask[:dom.nOwned] != 0] = 0.0
    ctx.vec_set(vb, f)

    n_elem_global = n ** 3
    n_node_global = (n + 1) ** 3
    nnz_global = int(ctx.allreduce([nnz])[0]) if world.size > 1 else nnz
    # units processed per step by THIS rank (owner-computes: a rank also integrates the one
    # element layer it shares with each neighbour; only globally distinct elements are counted)
    elem_dofs_per_step = n_elem_global * 8

    def step(profile):
        ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1, variant=args.variant)
        t_asm = ctx.timers()["assemble_ms"]
        info = ctx.solve(A, vb, vx, method=_lib.KSP_CG, pc=_lib.PC_JACOBI, fixed_iters=args.cg_iters,
                         norm_type=_lib.NORM_UNPRECONDITIONED, profile=1 if profile else 0)
        return t_asm, info.solve_ms, info.spmv_ms, info.reduce_ms, info.halo_ms

    for _ in range(args.warmup):
        step(False)
    ctx.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    asm_ms, cg_ms, spmv_ms, red_ms, halo_ms = [], [], [], [], []
    for _ in range(args.steps):
        a, c, s, rd, hl = step(True)
        halo_ms.append(hl)
        asm_ms.append(a)
        cg_ms.append(c)
        spmv_ms.append(s)
        red_ms.append(rd)
    ctx.sync()
    ctx.barrier()
    wall = time.perf_counter() - t0
    # max over ranks
    loc = [wall, np.mean(asm_ms), np.mean(cg_ms), np.mean(spmv_ms), np.mean(red_ms), np.mean(halo_ms)]
    red = ctx.allreduce(loc, op="max") if world.size > 1 else np.array(loc)
    wall, asm_mean, cg_mean, spmv_mean, red_mean, halo_mean = [float(v) for v in red]

    # ---- correctness on the SAME workload (outside the timed region): solve to 1e-10
    check = None
    if not args.no_check:
        info = ctx.solve(A, vb, vx, method=_lib.KSP_CG, pc=_lib.PC_JACOBI, rtol=1e-10, maxit=20000,
                         norm_type=_lib.NORM_UNPRECONDITIONED)
        check = {"cg_iters_to_rtol_1e-10": int(info.iters), "reason": int(info.reason),
                 "true_residual": float(info.true_resid), "solve_ms": float(info.solve_ms)}

    # ---- the same iteration with the matrix-free operator (outside the timed region; reported alongside, the
    # headline stays on the assembled matrix): no matrix values streamed, identical iterates
    mfree = None
    if not args.no_matfree:
        try:
            ctx.matfree_set(_lib.MATFREE_LAPLACE)
            for _ in range(2):
                mi = ctx.solve(A, vb, vx, method=_lib.KSP_CG, pc=_lib.PC_JACOBI, fixed_iters=args.cg_iters,
                               norm_type=_lib.NORM_UNPRECONDITIONED, profile=1, matfree=_lib.MATFREE_LAPLACE)
            mf_red = ctx.allreduce([mi.solve_ms, mi.spmv_ms], op="max") if world.size > 1 else [mi.solve_ms, mi.spmv_ms]
            mf_bytes = 41.0 * n_node_global / world.size        # x 8 + y 8 + xyz 24 + Dirichlet flag 1 per row
            mfree = {"kernel": "lattice_matfree_laplace_march_kernel (element products recomputed from the node coordinates, no matrix values)",
                     "cg_iters_per_s": args.cg_iters / (float(mf_red[0]) * 1e-3), "product_ms": float(mf_red[1]),
                     "bytes_per_launch": mf_bytes, "achieved_GBs": mf_bytes / (float(mf_red[1]) * 1e-3) / 1e9,
                     "bound": "FP64 VALU / latency (not HBM)"}
            if not args.no_check:
                mi = ctx.solve(A, vb, vx, method=_lib.KSP_CG, pc=_lib.PC_JACOBI, rtol=1e-10, maxit=20000,
                               norm_type=_lib.NORM_UNPRECONDITIONED, matfree=_lib.MATFREE_LAPLACE)
                mfree["check"] = {"cg_iters_to_rtol_1e-10": int(mi.iters), "reason": int(mi.reason),
                                  "true_residual_vs_assembled_matrix": float(mi.true_resid), "solve_ms": float(mi.solve_ms)}
        except _lib.PynamaHipError as e:
            mfree = {"error": str(e)}

    cpu = None
    if world.rank == 0 and world.size == 1 and not args.no_cpu_baseline:
        rp, ci = ctx.csr_get()
        cpu = cpu_baseline(dom, rp, ci, bmask, f)

    if world.rank == 0:
        # HBM traffic per launch from the PMC counters: collected in separate rocprofv3 --pmc passes
        # (never together with the timed run) and committed under profiles/; only valid for the
        # default single-GPU workload they were measured on
        traffic = {}
        try:
            if world.size == 1 and n == 215:
                for fn in ("pmc_traffic.json", "pmc_assembly.json"):    # the assembly entry comes from its own passes
                    tag = "r01e" if os.path.exists(os.path.join(ROOT, "profiles", "r01e_" + fn)) else "r01d"
                    with open(os.path.join(ROOT, "profiles", f"{tag}_{fn}")) as fh:
                        traffic.update({k: v["hbm_bytes_per_launch"] for k, v in json.load(fh)["kernels"].items()})
        except (OSError, KeyError, ValueError):
            traffic = {}
        asm_kernel = ("assemble_q1_hex_lattice_kernel" if args.variant == 1 and ctx.mesh_topology()[0] == "lattice"
                      else "assemble_q1_hex_tiled_kernel" if args.variant != 0 else "assemble_generic_kernel")
        B_asm, B_spmv, B_cg = algorithmic_bytes(n_elem_global, n_node_global, nnz_global)
        # per-rank share of the algorithmic bytes (strong scaling: each GPU streams 1/N of them)
        share = 1.0 / world.size
        asm_gbs = B_asm * share / (asm_mean * 1e-3) / 1e9
        spmv_gbs = B_spmv * share / (spmv_mean * 1e-3) / 1e9 if spmv_mean > 0 else 0.0
        cg_gbs = B_cg * share / (cg_mean / args.cg_iters * 1e-3) / 1e9
        out = {
            "metric": "assembled element-DOFs/sec + CG iterations/sec, 10M-DOF Poisson, 1/2/4/8 MI355X",
            "value": elem_dofs_per_step / (asm_mean * 1e-3),
            "unit": "element-DOFs/s (assembly phase); cg_iters_per_s alongside",
            "cg_iters_per_s": args.cg_iters / (cg_mean * 1e-3),
            "n_gpus": world.size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "breakdown_ms": {"assembly": asm_mean, "cg": cg_mean, "cg_iters": args.cg_iters,
                             "spmv_kernel": spmv_mean, "symbolic_once": symbolic_ms,
                             # N > 1 (single-reduction CG): end of the product -> all-reduced sums ready = partial sums + the one
                             # all-reduce, per iteration (the scalar step rides in the update kernel; 0 on one GPU)
                             "cg_reduction_per_iter": red_mean,
                             # overlapped halo exchange (pack + grouped send/recv on the communication stream), per iteration
                             "cg_halo_per_iter": halo_mean},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D Poisson, {n}^3 Q1 hex elements, {n_node_global} DOFs, nnz {nnz_global}, "
                                   f"FP64 assembly + Jacobi-PCG ({args.cg_iters} its/step)",
                       "partition": f"z-slabs x{world.size}", "assembly_variant": args.variant},
            "roofline": {"kernel": "sellp_spmv_kernel (SELL-64 + column-pattern dictionary SpMV inside CG)", "bound": "hbm", "achieved": spmv_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_gbs / HBM_PEAK_GBS, "traffic": traffic.get("sellp_spmv_kernel"),
                         "algorithmic_bytes_per_launch": B_spmv * share},
            "roofline_assembly": {"kernel": asm_kernel + " (zero + integration + scatter, one launch)", "bound": "hbm",
                                  "achieved": asm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": asm_gbs / HBM_PEAK_GBS,
                                  "traffic": traffic.get(asm_kernel) if args.jitter == 0.0 else None,
                                  "algorithmic_bytes_per_launch": B_asm * share},
            "roofline_cg_iteration": {"bound": "hbm", "achieved": cg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": cg_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": B_cg * share},
            "check": check,
            "matrix_free": dict(mfree, traffic=traffic.get("lattice_matfree_laplace_march_kernel")) if mfree and "error" not in mfree else mfree,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    world.cleanup()
    ctx.close()


if __name__ == "__main__":
    main()
