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
N > 1: the SAME mesh is row-partitioned in z-slabs over the ranks ("scaling": "strong"), halo
planes and dot products go over RCCL inside libpynama_hip.so; a start-up self-test of the
communicator runs first under a bounded wait (pynama_amd/common/comm.py).

Next to the headline the same run reports, outside the timed region (N = 1 only):
  * `roofline_assembly_general`: the assembly of the SAME mesh with jittered nodes -- the
    quadrature path (the reference integrates every cell, spectral.py:117-156), not the
    parallelepiped closed form the uniform box admits;
  * `configs`: the other single-GPU configurations of BASELINE.json (C1 2-D plumbing case, C2
    128^3 Poisson, C3 128^3 KLE with 3 DOFs per node, C5 5 M tetrahedra with GMRES(30)+Jacobi)
    and the reference's own element order (HO3_2D: 1024^2, HO3_3D: 64^3 second-order cells, KLE),
    each with its residual check;
  * `matrix_free`: the same CG with the operator recomputed from the mesh.

The product path is torch-free: ranks/LOCAL_RANK come from the launcher's environment, the RCCL
id is exchanged through a node-local file (pynama_amd/common/comm.py).
Only the `cpu_baseline` leg touches oracle/ (the C restatement, as the thing timed on the CPU).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def algorithmic_bytes(n_elem, n_node, nnz, nn=8, dim=3, ndof=1):
    """compulsory traffic, FP64 values, int32 indices (block CSR for ndof > 1:
    one node-level column index per ndof x ndof block)."""
    nv = nnz * ndof * ndof
    asm = 4 * nn * n_elem + 8 * dim * n_node + 4 * (n_node + 1) + 4 * nnz + 8 * nv
    spmv = 8 * nv + 4 * nnz + 4 * (n_node + 1) + 16 * n_node * ndof
    cg_iter = 8 * nv + 4 * nnz + 148 * n_node * ndof
    return asm, spmv, cg_iter


def sell_entries(rowptr, ndof=1):
    """stored entries of the SELL-64 image (slice width = longest scalar row of the slice)"""
    ln = np.repeat(np.diff(rowptr).astype(np.int64) * ndof, ndof)
    pad = (-len(ln)) % 64
    if pad:
        ln = np.concatenate([ln, np.zeros(pad, np.int64)])
    return int(ln.reshape(-1, 64).max(axis=1).sum() * 64)


def roofline(kernel, bytes_alg, ms, **extra):
    gbs = bytes_alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    out = {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": bytes_alg, "ms_per_launch": ms}
    out.update(extra)
    return out


class Traffic:
    """HBM bytes per launch from the PMC counters: collected in separate rocprofv3 --pmc passes (never together with the timed
    run), committed under profiles/ together with the hash of the kernel sources they were taken on.  Quoted only when that
    hash is the one embedded in the library that just ran."""

    def __init__(self, lib_hash, passes=("traffic", "assembly", "assembly_general")):
        # `passes`: the headline workload's counter passes (tools/profile_pass_r03.sh part a); the other configurations' passes hold
        # kernels of the same names at other sizes and are not mixed in
        self.lib_hash, self.k, self.why = lib_hash, {}, "no PMC summary under profiles/ was taken on these kernel sources"
        files = [fn for p in passes for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{p}.json")))]
        for fn in files:
            try:
                with open(fn) as fh:
                    d = json.load(fh)
            except (OSError, ValueError):
                continue
            if d.get("source_hash") != lib_hash:
                continue
            for name, v in d.get("kernels", {}).items():
                if "hbm_bytes_per_launch" in v:
                    self.k[name] = {"bytes": v["hbm_bytes_per_launch"], "file": os.path.basename(fn)}

    def get(self, kernel):
        v = self.k.get(kernel)
        return (v["bytes"], v["file"]) if v else (None, self.why)


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
                      f"({t_asm:.1f} s) + {iters} Jacobi-PCG iterations on the full {m.n_node}-row matrix ({t_cg:.1f} s); "
                      "PETSc is not installable on the box, so the reference's own KSP path cannot be timed"}


def smooth_load(X, h3, mask):
    f = (1.0 + X[:, 0] + 2.0 * X[:, 1] ** 2 + np.exp(np.prod(X, axis=1)) * np.cos(3.0 * X[:, -1])) * h3
    f[mask != 0] = 0.0
    return f


def median_assembly(ctx, fn, reps):
    t = []
    for _ in range(reps):
        fn()
        t.append(ctx.timers()["assemble_ms"])
    return float(np.median(t)), float(np.min(t))


# ---- the other single-GPU configurations of BASELINE.json (outside the timed region) ------------------------------------------
def general_geometry_leg(_lib, DMPlexDom, Spectral, n, traffic, reps=9, warm=10):
    """assembly of the bench mesh with jittered nodes (0.2 h): the quadrature path"""
    dom = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]}, jitter=0.2)
    dom.setFemIndexing(2)
    ctx = dom.ctx
    for t in Spectral(2, 3).deviceTables():
        ctx.tables_set(*t)
    ctx.bc_set(1, dom.boundaryMaskLocal())
    n_rows, nnz = ctx.csr_symbolic()
    A = ctx.mat_create(1, 1)
    for _ in range(warm):        # the GPU idled during the CPU baseline (seconds): let the clocks come back before timing 1 ms launches
        ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1)
    med, best = median_assembly(ctx, lambda: ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1), reps)
    B_asm, _, _ = algorithmic_bytes(n ** 3, (n + 1) ** 3, nnz)
    kern = "assemble_q1_hex_march_kernel"
    tb, src = traffic.get(kern)
    out = roofline(kern + " (z-marching quadrature path: sum-factorised closed form of the 2x2x2 rule per element, LDS rows, x-line stores)",
                   B_asm, med, min_ms=best, element_dofs_per_s=n ** 3 * 8 / (med * 1e-3), traffic=tb, traffic_source=src,
                   mesh=f"{n}^3 Q1 hex, nodes jittered by 0.2 h (general geometry, every element integrated with 8 Gauss points)",
                   bound_note="FP64 VALU and the store path together: about 1,050 FP64 instructions per element (1.31 elements integrated per "
                              "element of the mesh) against 387 B of compulsory traffic; VALU alone 0.57 ms, stores alone 0.76 ms (DESIGN.md 5b)")
    ctx.close()
    return out


def config_poisson(_lib, DMPlexDom, Spectral, nelem, cg_iters, name):
    dim = len(nelem)
    dom = DMPlexDom(boxMesh={"nelem": nelem, "lower": [0] * dim, "upper": [1] * dim})
    dom.setFemIndexing(2)
    ctx = dom.ctx
    for t in Spectral(2, dim).deviceTables():
        ctx.tables_set(*t)
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(1, bm)
    n_rows, nnz = ctx.csr_symbolic()
    A = ctx.mat_create(1, 1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1)
    med, best = median_assembly(ctx, lambda: ctx.assemble_scalar(_lib.FORM_LAPLACE, A, -1), 5)
    ne, nn_e = int(np.prod(nelem)), 2 ** dim
    B_asm, B_spmv, B_cg = algorithmic_bytes(ne, n_rows, nnz, nn=nn_e, dim=dim)
    vb, vx = ctx.vec_create(1), ctx.vec_create(1)
    ctx.vec_set(vb, smooth_load(dom.xyz[:dom.nOwned], 1.0 / ne, bm[:dom.nOwned]))
    for _ in range(2):
        info = ctx.solve(A, vb, vx, fixed_iters=cg_iters, norm_type=_lib.NORM_UNPRECONDITIONED, profile=1)
    chk = ctx.solve(A, vb, vx, rtol=1e-10, maxit=20000, norm_type=_lib.NORM_UNPRECONDITIONED)
    out = {"config": name, "n_elem": ne, "n_dof": n_rows, "nnz": nnz,
           "assembly_ms": med, "element_dofs_per_s": ne * nn_e / (med * 1e-3),
           "assembly_frac_of_hbm_peak": B_asm / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "cg_iters_per_s": cg_iters / (info.solve_ms * 1e-3), "spmv_ms": info.spmv_ms,
           "spmv_frac_of_hbm_peak": B_spmv / (info.spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if info.spmv_ms > 0 else None,
           "cg_iteration_frac_of_hbm_peak": B_cg / (info.solve_ms / cg_iters * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "algorithmic_bytes": {"assembly": B_asm, "spmv": B_spmv, "cg_iteration": B_cg},
           "check": {"cg_iters_to_rtol_1e-10": int(chk.iters), "reason": int(chk.reason), "true_residual": float(chk.true_resid)}}
    ctx.close()
    return out


def config_kle(_lib, DMPlexDom, Spectral, n, cg_iters, jitter=0.0):
    """C3: 3 DOFs per node on the 128^3 mesh -- the reference's KLE system stands in for 'linear elasticity' (SURVEY.md 0.3:
    the reference has no elasticity form; K is its vector-valued stiffness with 3x3 blocks).  jitter > 0: general geometry
    (every cell integrated with 8 + 1 Gauss points, as spectral.py:117-156 does).  Around the solve: the one-off products of
    solveKLE (base_problem.py:481), the operator chain of evalRHS (:212-232), the matrix-free K, the refresh of K's solver image."""
    dom = DMPlexDom(boxMesh={"nelem": [n, n, n], "lower": [0, 0, 0], "upper": [1, 1, 1]}, jitter=jitter)
    dom.setFemIndexing(2)
    ctx = dom.ctx
    sp = Spectral(2, 3)
    for t in sp.deviceTables():
        ctx.tables_set(*t)
    bm = dom.boundaryMaskLocal()
    mask3 = np.repeat(bm[:, None], 3, axis=1)
    ctx.bc_set(3, mask3)
    n_rows, nnz = ctx.csr_symbolic()
    # Krhs is a COMPACT imposed-column matrix: the node rows next to an imposed node, as the reference preallocates it
    # (mat_generator.py:42-58, 91) -- every assembly writes all of it, nothing is skipped, nothing else is stored
    K, Krhs, Rw = ctx.mat_create(3, 3), ctx.mat_create_rhs(3, 3), ctx.mat_create(3, 3)
    B_asm1, B_spmv, B_cg = algorithmic_bytes(n ** 3, n_rows, nnz, ndof=3)
    kr_blocks, kr_rows = ctx.mat_stored(Krhs)
    B_krhs = 8.0 * 9 * kr_blocks + 4.0 * n_rows          # its values + the per-row start the kernels look up
    B_asm = 2 * B_asm1 + B_krhs             # K and Rw by SURVEY.md 8d's model (4.46 GB each) + the compact Krhs: the bytes that move
    ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
    med, best = median_assembly(ctx, lambda: ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1), 3)
    # the same with a full-pattern Krhs (the layout of rounds 1-2: 4.1 GB more to write, or to skip once it is known to hold zeros)
    Kfull = ctx.mat_create(3, 3)
    os.environ["PYNAMA_RHS_FULL_WRITE"] = "1"
    try:
        ctx.assemble_kle(1e3, 1e2, K, Kfull, Rw, -1)
        med_full, _ = median_assembly(ctx, lambda: ctx.assemble_kle(1e3, 1e2, K, Kfull, Rw, -1), 3)
    finally:
        del os.environ["PYNAMA_RHS_FULL_WRITE"]
    ctx.mat_destroy(Kfull)
    vel = np.zeros((dom.nOwned, 3))
    vel[bm[:dom.nOwned] != 0] = [1.0, 0.0, 0.0]
    vv, vr, vx, vw, vy = ctx.vec_create(3), ctx.vec_create(3), ctx.vec_create(3), ctx.vec_create(3), ctx.vec_create(3)
    ctx.vec_set(vv, vel.ravel())
    ctx.vec_set(vw, np.random.default_rng(1).standard_normal(n_rows * 3))
    t_kr, t_rw = [], []
    for _ in range(5):                      # rhs = Rw w + Krhs v (base_problem.py:479-481): block-CSR values read directly, no image
        ctx.spmv(Rw, vw, vy)
        t_rw.append(ctx.timers()["spmv_ms"])
        ctx.spmv(Krhs, vv, vr)              # zero vorticity: rhs = Krhs v_bc
        t_kr.append(ctx.timers()["spmv_ms"])
    # first solve after an assembly: the refresh of K's solver image is inside this wall time and nowhere else
    ctx.assemble_kle(1e3, 1e2, K, -1, -1, -1)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.solve(K, vr, vx, fixed_iters=5)
    t1 = time.perf_counter()
    ctx.solve(K, vr, vx, fixed_iters=5)
    t2 = time.perf_counter()
    refresh_ms = 1e3 * ((t1 - t0) - (t2 - t1))
    for _ in range(2):
        info = ctx.solve(K, vr, vx, fixed_iters=cg_iters, profile=1)
    chk = ctx.solve(K, vr, vx, rtol=1e-10, maxit=20000, norm_type=_lib.NORM_UNPRECONDITIONED)
    err = float(np.abs(ctx.vec_get(vx, 3).reshape(-1, 3) - [1.0, 0.0, 0.0]).max())
    # the same solve with K recomputed from the mesh instead of read from HBM (PETSc: Amat = shell, Pmat = assembled)
    mfree = None
    try:
        ctx.matfree_set(_lib.MATFREE_KLE, 1e3, 1e2)
        tm = []
        for _ in range(5):
            ctx.matfree_apply(vv, vy, _lib.MATFREE_KLE)
            tm.append(ctx.timers()["spmv_ms"])
        for _ in range(2):
            mi = ctx.solve(K, vr, vx, fixed_iters=cg_iters, profile=1, matfree=_lib.MATFREE_KLE)
        mc = ctx.solve(K, vr, vx, rtol=1e-10, maxit=20000, norm_type=_lib.NORM_UNPRECONDITIONED, matfree=_lib.MATFREE_KLE)
        mfree = {"product_ms": float(np.median(tm[1:])), "cg_iters_per_s": cg_iters / (mi.solve_ms * 1e-3),
                 "kernel": "lattice_matfree_kle_kernel (Laplacian per component + rank-9 penalty correction per cell, no matrix values)",
                 "check": {"cg_iters_to_rtol_1e-10": int(mc.iters), "reason": int(mc.reason), "true_residual_vs_assembled_matrix": float(mc.true_resid),
                           "solve_ms": float(mc.solve_ms)},
                 "default_path": True,
                 "note": "what KspSolver multiplies with by default when Mat.K carries the shell (structured Q1 hexahedra; CG and the "
                         "symmetric preonly substitute; -pynama_mat_free 0 for the assembled product): the library refuses the shell unless "
                         "it reproduces the assembled product on b, the facade then warns and uses the assembled matrix"}
    except _lib.PynamaHipError as e:
        mfree = {"error": str(e)}
    # operator chain of evalRHS after the solve (base_problem.py:216-232): v(x)v, SrT v, axpy, DivSrT, scale, Curl
    chain = None
    try:
        ops = sp.operatorTerms()
        mats = {}
        for name in ("SrT", "DivSrT", "Curl"):
            br, bc, terms, coef = ops[name]
            mats[name] = (ctx.mat_create(br, bc), br, bc)
            ctx.assemble_operator(_lib.Q_NODAL, terms, coef, mats[name][0])
        v6a, v6b, v3 = ctx.vec_create(6), ctx.vec_create(6), ctx.vec_create(3)
        parts = {}

        def timed(label, fn, is_spmv):
            ts = []
            for _ in range(4):
                ctx.sync()
                t_a = time.perf_counter()
                fn()
                ctx.sync()
                ts.append(ctx.timers()["spmv_ms"] if is_spmv else 1e3 * (time.perf_counter() - t_a))
            parts[label] = float(np.median(ts[1:]))
        timed("vtensv", lambda: ctx.vec_vtensv(vx, v6b), False)
        timed("SrT_product", lambda: ctx.spmv(mats["SrT"][0], vx, v6a), True)
        timed("axpby", lambda: ctx.vec_axpby(v6a, 2.0 * 0.01, v6a, -1.0, v6b), False)
        timed("DivSrT_product", lambda: ctx.spmv(mats["DivSrT"][0], v6a, v3), True)
        timed("Curl_product", lambda: ctx.spmv(mats["Curl"][0], v3, vy), True)
        fr = {}
        for name, key in (("SrT", "SrT_product"), ("DivSrT", "DivSrT_product"), ("Curl", "Curl_product")):
            _, br, bc = mats[name]
            b_op = 8.0 * nnz * br * bc + 4.0 * nnz + 4.0 * (n_rows + 1) + 8.0 * n_rows * (br + bc)
            fr[name] = {"ms": parts[key], "bytes": b_op, "frac_of_hbm_peak": b_op / (parts[key] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        chain = {"operator_products": fr, "vtensv_ms_wall": parts["vtensv"], "axpby_ms_wall": parts["axpby"],
                 "rhs_products_ms": {"Rw": float(np.median(t_rw[1:])), "Krhs": float(np.median(t_kr[1:]))},
                 "sum_without_solve_ms": float(np.median(t_rw[1:]) + np.median(t_kr[1:]) + sum(parts.values())),
                 "note": "everything evalRHS does around the KLE solve; products straight from the block-CSR values (bcsr_spmv_kernel), "
                         "vector kernels as host wall time including the launch"}
    except _lib.PynamaHipError as e:
        chain = {"error": str(e)}
    out = {"config": f"C3: 3D KLE (3 DOF/node, alpha_d 1e3, alpha_w 1e2) on {n}^3 Q1 hex, uniform-flow boundary data"
                     + (f", nodes jittered by {jitter} h (general geometry)" if jitter else ""), "n_elem": n ** 3,
           "n_dof": 3 * n_rows, "nnz_blocks": nnz, "assembly_ms_K_Krhs_Rw": med,
           "element_dofs_per_s": n ** 3 * 24 / (med * 1e-3),
           "assembly_frac_of_hbm_peak": B_asm / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "assembly_note": f"K and Rw in full + the COMPACT Krhs ({kr_rows} of {n_rows} node rows, {8.0 * 9 * kr_blocks / 1e9:.2f} GB instead of "
                            f"{8.0 * 9 * nnz / 1e9:.2f} GB): {B_asm / 1e9:.2f} GB = the bytes every call moves, first call included -- nothing is "
                            f"skipped or cached.  With a full-pattern Krhs written in full (three matrices of SURVEY.md 8d's model, "
                            f"{3 * B_asm1 / 1e9:.2f} GB) the call takes {med_full:.3f} ms = {3 * B_asm1 / (med_full * 1e-3) / 1e9 / HBM_PEAK_GBS:.3f} of the roofline",
           "assembly_ms_full_pattern_krhs": med_full,
           "krhs_compact": {"node_rows_stored": kr_rows, "node_rows": n_rows, "values_GB": 8.0 * 9 * kr_blocks / 1e9,
                            "product_ms": float(np.median(t_kr[1:])),
                            "product_frac_of_hbm_peak": (8.0 * 9 * kr_blocks + 4.0 * kr_blocks + 8.0 * kr_rows + 48.0 * n_rows)
                            / (float(np.median(t_kr[1:])) * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "kernel": "assemble_q1_hex_kle_lattice_kernel (four waves per tile; "
                     + ("general geometry: closed form of the 2x2x2 rule, Gauss points split over the waves for K, node columns for Rw)"
                        if jitter else "closed-form blocks on parallelepipeds)"),
           "cg_iters_per_s": cg_iters / (info.solve_ms * 1e-3), "block_spmv_ms": info.spmv_ms,
           "spmv_frac_of_hbm_peak": B_spmv / (info.spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if info.spmv_ms > 0 else None,
           "cg_iteration_frac_of_hbm_peak": B_cg / (info.solve_ms / cg_iters * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "solver_image_refresh_ms": refresh_ms,
           "format": "CG: block SELL-64 image of K (lane per scalar row, one node-level column pattern id per row), refreshed by the first solve "
                     "after an assembly (solver_image_refresh_ms, paid once per K); Krhs, Rw and the operators are multiplied from their "
                     "block-CSR values and have no image (block-CSR byte model of SURVEY.md 8d: 5.29 GB / iteration)",
           "matrix_free": mfree, "evalRHS_chain": chain,
           "algorithmic_bytes": {"assembly_3_matrices": B_asm, "spmv": B_spmv, "cg_iteration": B_cg},
           "check": {"cg_iters_to_rtol_1e-10": int(chk.iters), "reason": int(chk.reason), "true_residual": float(chk.true_resid),
                     "max_error_vs_exact_uniform_flow": err}}
    ctx.close()
    return out


def config_ho3(_lib, DMPlexDom, Spectral, dim, nel, cg_iters):
    """Second-order elements (ngl = 3), the order EVERY case file of the reference sets (src/cases/*.yaml `ngl: 3`; eight of nine are 2-D):
    KLE system on a structured box, 9-node quadrilaterals / 27-node hexahedra, Gauss(3)^dim + Gauss(2)^dim rules
    (src/elements/spectral.py:41-43).  Assembly by the row-run kernels (pyn_assemble_ho3.hip), products from the block-CSR values."""
    dw = 1 if dim == 2 else 3
    dom = DMPlexDom(boxMesh={"nelem": [nel] * dim, "lower": [0] * dim, "upper": [1] * dim})
    dom.setFemIndexing(3)
    ctx = dom.ctx
    for t in Spectral(3, dim).deviceTables():
        ctx.tables_set(*t)
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(dim, np.repeat(bm[:, None], dim, axis=1))
    n_rows, nnzb = ctx.csr_symbolic()
    symbolic_ms = ctx.timers()["symbolic_ms"]
    topo = ctx.mesh_topology()[0]
    K, Krhs, Rw = ctx.mat_create(dim, dim), ctx.mat_create_rhs(dim, dim), ctx.mat_create(dim, dw)     # Krhs compact: rows next to imposed nodes
    ne, nn = nel ** dim, 3 ** dim
    kr_blocks, kr_rows = ctx.mat_stored(Krhs)

    def b_asm(br, bc):          # SURVEY.md 8(d) per matrix: conn + xyz + rowptr + colidx + values
        return 4 * nn * ne + 8 * dim * n_rows + 4 * (n_rows + 1) + 4 * nnzb + 8 * nnzb * br * bc

    os.environ["PYNAMA_HO3_REQUIRE"] = "1"          # the generic atomics kernel must not stand in silently
    try:
        ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1)
        med, best = median_assembly(ctx, lambda: ctx.assemble_kle(1e3, 1e2, K, Krhs, Rw, -1), 5)
        med_k, _ = median_assembly(ctx, lambda: ctx.assemble_kle(1e3, 1e2, K, -1, -1, -1), 3)
        med_rw, _ = median_assembly(ctx, lambda: ctx.assemble_kle(1e3, 1e2, -1, -1, Rw, -1), 3)
    finally:
        del os.environ["PYNAMA_HO3_REQUIRE"]
    B3 = b_asm(dim, dim) + b_asm(dim, dw) + 8.0 * dim * dim * kr_blocks + 4.0 * n_rows     # K, Rw in full + the compact Krhs
    N, nnz = n_rows * dim, nnzb * dim * dim
    B_spmv = 8 * nnz + 4 * nnzb + 4 * (n_rows + 1) + 16 * N          # block-CSR model of SURVEY.md 8(d): one column index per block
    B_cg = 8 * nnz + 4 * nnzb + 148 * N
    cte = np.array([1.0, 0.0, 0.0][:dim])
    vel = np.zeros((dom.nOwned, dim))
    vel[bm[:dom.nOwned] != 0] = cte
    vv, vr, vx, vw, vy = ctx.vec_create(dim), ctx.vec_create(dim), ctx.vec_create(dim), ctx.vec_create(dw), ctx.vec_create(dim)
    ctx.vec_set(vv, vel.ravel())
    ctx.vec_set(vw, np.random.default_rng(1).standard_normal(n_rows * dw))
    t_kr, t_rw, t_k = [], [], []
    for _ in range(6):          # rhs = Rw w + Krhs v (base_problem.py:481): one-off products, straight from the block-CSR values
        ctx.spmv(Rw, vw, vy)
        t_rw.append(ctx.timers()["spmv_ms"])
        ctx.spmv(Krhs, vv, vr)
        t_kr.append(ctx.timers()["spmv_ms"])
        ctx.spmv(K, vv, vy)
        t_k.append(ctx.timers()["spmv_ms"])
    for _ in range(2):
        info = ctx.solve(K, vr, vx, fixed_iters=cg_iters, profile=1)
    chk = ctx.solve(K, vr, vx, rtol=1e-10, maxit=100000, norm_type=_lib.NORM_UNPRECONDITIONED)
    err = float(np.abs(ctx.vec_get(vx, dim).reshape(-1, dim) - cte).max())
    avg_row = nnzb * dim / n_rows
    out = {"config": f"HO3_{dim}D: {dim}-D KLE ({dim} DOF/node, alpha_d 1e3, alpha_w 1e2) on {nel}^{dim} second-order (ngl 3, {nn}-node) cells, "
                     "uniform-flow boundary data -- the element order of every reference case (src/cases/*.yaml)",
           "topology": topo, "n_elem": ne, "n_dof": N, "nnz_blocks": nnzb, "symbolic_ms_closed_form": symbolic_ms,
           "assembly_ms_K_Krhs_Rw": med, "assembly_ms_min": best,
           "element_dofs_per_s": ne * nn * dim / (med * 1e-3),
           "assembly_frac_of_hbm_peak": B3 / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "assembly_note": f"K and Rw in full (SURVEY.md 8d's model per matrix) + the compact Krhs ({kr_rows} of {n_rows} node rows, "
                            f"{8.0 * dim * dim * kr_blocks / 1e9:.2f} GB): {B3 / 1e9:.2f} GB = the bytes every call moves",
           "assembly_ms_K_alone": med_k, "assembly_frac_K_alone": b_asm(dim, dim) / (med_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "assembly_ms_Rw_alone": med_rw, "assembly_frac_Rw_alone": b_asm(dim, dw) / (med_rw * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "kernel": "assemble_ho3_lattice_kernel (persistent workgroups walk runs of consecutive node rows of an x-line: LDS image of that "
                     "piece of the block-CSR values, (row, element, column node) triples one per lane, closed-form blocks (diagonal J^-1 forms "
                     "on axis-aligned boxes), ds_add_f64, one coalesced copy out; the next run's prologue through the scalar cache / closed "
                     "forms one run ahead; no HBM atomics, nothing integrated twice)",
           "cg_iters_per_s": cg_iters / (info.solve_ms * 1e-3), "block_spmv_ms_in_cg": info.spmv_ms,
           "spmv_frac_of_hbm_peak": B_spmv / (info.spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if info.spmv_ms > 0 else None,
           "cg_iteration_frac_of_hbm_peak": B_cg / (info.solve_ms / cg_iters * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "product_ms_one_off": {"K": float(np.median(t_k[1:])), "Krhs": float(np.median(t_kr[1:])), "Rw": float(np.median(t_rw[1:]))},
           "format": ("block-CSR values read directly (bcsr_spmv_kernel: 16 lanes per node row)"
                      if avg_row >= 128 else
                      "CG and K v: block-CSR values read directly by csrlb_spmv_kernel (lane per scalar row, the 64 rows of a slice are one "
                      "contiguous run of the values: global -> LDS by LDS-DMA, x entries through the node-level column-pattern dictionary, the two "
                      "entries of a column node by one 16-byte load); Krhs v, Rw w: bcsr_spmv_kernel; no image") +
                     "; byte model: 8 B per value + 4 B per block of column index + vectors",
           "algorithmic_bytes": {"assembly_3_matrices": B3, "spmv": B_spmv, "cg_iteration": B_cg},
           "check": {"cg_iters_to_rtol_1e-10": int(chk.iters), "reason": int(chk.reason), "true_residual": float(chk.true_resid),
                     "max_error_vs_exact_uniform_flow": err}}
    ctx.close()
    return out


def kuhn_box(n, seed=2024):
    """n^3 unit-box hexes -> 6 n^3 positively oriented tetrahedra, nodes randomly renumbered"""
    from itertools import permutations
    lat = n + 1
    strides = np.array([1, lat, lat * lat])
    i, j, k = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    base = (i * strides[0] + j * strides[1] + k * strides[2]).ravel()
    conn = []
    for perm in permutations(range(3)):
        offs = [0]
        for d in perm:
            offs.append(offs[-1] + strides[d])
        if sum(1 for a in range(3) for b in range(a + 1, 3) if perm[a] > perm[b]) % 2:
            offs[-1], offs[-2] = offs[-2], offs[-1]
        conn.append(base[:, None] + np.array(offs)[None, :])
    conn = np.stack(conn, axis=1).reshape(-1, 4)
    ax = np.linspace(0.0, 1.0, lat)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    xyz = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    p = np.random.default_rng(seed).permutation(lat ** 3)
    return xyz[np.argsort(p)], p[conn].astype(np.int32)


def config_tets(_lib, DMPlexDom, n, gmres_iters):
    """C5: unstructured tetrahedra in random node numbering, imported-mesh path (Morton renumbering, patch-plan P1 kernel),
    GMRES(30)+Jacobi.  The mesh enters through the in-memory form of the Gmsh reader's result; the file round trip itself is
    tests/test_gpu_fullsize.py::test_c5_unstructured_tets_gmsh_gmres (12 s of text parsing, not a device measurement)."""
    from pynama_amd.elements.simplex import Simplex
    xyz, conn = kuhn_box(n)
    dom = DMPlexDom(mesh={"dim": 3, "xyz": xyz, "conn": conn, "facets": [], "cell": "simplex"})
    dom.setFemIndexing(2)
    ctx = dom.ctx
    for t in Simplex(3).deviceTables():
        ctx.tables_set(*t)
    bm = dom.boundaryMaskLocal()
    ctx.bc_set(1, bm)
    n_rows, nnz = ctx.csr_symbolic()
    A = ctx.mat_create(1, 1)
    ctx.assemble_scalar(_lib.FORM_LAPLACE, A)
    med, best = median_assembly(ctx, lambda: ctx.assemble_scalar(_lib.FORM_LAPLACE, A), 5)
    ne = conn.shape[0]
    npatch, maxrows, maxlen, npe = ctx.patch_plan_info(0)
    B_asm, B_spmv, _ = algorithmic_bytes(ne, n_rows, nnz, nn=4)
    vb, vx, vy = ctx.vec_create(1), ctx.vec_create(1), ctx.vec_create(1)
    ctx.vec_set(vb, smooth_load(dom.xyz[:dom.nOwned], 1.0 / n ** 3, bm[:dom.nOwned]))
    sp = []
    for _ in range(5):
        ctx.spmv(A, vb, vy)
        sp.append(ctx.timers()["spmv_ms"])
    spmv_ms = float(np.median(sp))
    for _ in range(2):
        info = ctx.solve(A, vb, vx, method=_lib.KSP_GMRES, pc=_lib.PC_JACOBI, fixed_iters=gmres_iters, restart=30, gmres_orthog=1)
    # GMRES(m): inner step j costs 12 nnz + 20 N + 40 N (j + 1) bytes; mean over a cycle of m = 30 steps
    m = 30
    B_gmres = 12 * nnz + 20 * n_rows + 40 * n_rows * (m + 1) / 2.0
    chk = ctx.solve(A, vb, vx, method=_lib.KSP_GMRES, pc=_lib.PC_JACOBI, rtol=1e-10, restart=30, maxit=100000,
                    norm_type=_lib.NORM_UNPRECONDITIONED, gmres_orthog=1)
    basis_mb = 8.0 * n_rows * (m + 1) / 1e6
    out = {"config": f"C5: {ne} linear tetrahedra ({n}^3 hexes cut in 6), {n_rows} nodes in random numbering -> Morton order, GMRES(30)+Jacobi",
           "n_elem": ne, "n_dof": n_rows, "nnz": nnz, "assembly_ms": med, "element_dofs_per_s": ne * 4 / (med * 1e-3),
           "assembly_frac_of_hbm_peak": B_asm / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "kernel": "assemble_p1_tet_tiled_kernel (patches of 343 consecutive rows, LDS adds, rows written once)",
           # the automatic plan: every element that touches a patch's rows is integrated by that patch -- (patch, element) pairs / elements
           # is the redundancy of the integration AND of the 84 B of plan record + connectivity read per pair
           "plan": {"patches": npatch, "rows_per_patch_max": maxrows, "row_length_max": maxlen, "patch_element_pairs": npe,
                    "redundancy": npe / ne},
           "spmv_ms": spmv_ms, "spmv_algorithmic_GBs": B_spmv / (spmv_ms * 1e-3) / 1e9,
           "gmres_iters_per_s": gmres_iters / (info.solve_ms * 1e-3),
           "gmres_iteration_algorithmic_GBs": B_gmres / (info.solve_ms / gmres_iters * 1e-3) / 1e9,
           "gmres_iteration_frac_of_hbm_peak": B_gmres / (info.solve_ms / gmres_iters * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "residency_note": f"matrix ({12 * nnz / 1e6:.0f} MB) and Krylov basis ({basis_mb:.0f} MB) fit the 256 MiB Infinity Cache: "
                             "these rates are cache rates, the HBM roofline is not the bound here (launch latency of the five "
                             "kernels per inner step is)",
           "algorithmic_bytes": {"assembly": B_asm, "spmv": B_spmv, "gmres_iteration_mean": B_gmres},
           "check": {"gmres_iters_to_rtol_1e-10": int(chk.iters), "reason": int(chk.reason), "true_residual": float(chk.true_resid)}}
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nel", type=int, default=215, help="elements per side (215 -> 10,077,696 DOFs)")
    ap.add_argument("--cg-iters", type=int, default=100, help="fixed CG iterations per step")
    ap.add_argument("--variant", type=int, default=1, help="assembly kernel: 0 generic atomics, 1 auto (plan-free lattice kernel on box meshes), "
                    "2 patch-plan kernel on 7x7x7 tiles")
    ap.add_argument("--jitter", type=float, default=0.0, help="perturb interior nodes of the HEADLINE mesh by jitter*h (diagnostics; the "
                    "general-geometry leg always runs on its own jittered mesh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-matfree", action="store_true", help="skip the matrix-free CG leg (reported alongside, not the headline)")
    ap.add_argument("--no-extra", action="store_true", help="skip the general-geometry leg and the other BASELINE configurations")
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
    selftest = None
    if world.size > 1:                                # fail loudly BEFORE the timed region: rank echo, neighbours, halo planes
        selftest = world.selftest(ctx, dom)
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
    # lumped load of a smooth non-separable source (NOT a discrete eigenvector: with
    # f = sin sin sin on a uniform grid CG would converge in one iteration)
    f = smooth_load(dom.xyz[:dom.nOwned], h ** 3, bmask[:dom.nOwned])
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

    for k in range(args.warmup):
        if world.size > 1 and k == 0:    # the first distributed step: overlapped halo exchange + one all-reduce per iteration
            with world.bounded("first assembly + CG step across ranks (overlapped halo exchange, all-reduce per iteration)", 300):
                step(False)
                ctx.sync()
        else:
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

    # ---- N > 1: the single-reduction (Chronopoulos-Gear) iteration runs across ranks; its one-rank time separates the cost of
    # the algorithm from the cost of the wire in the scaling curve.  N = 1: the same form through a one-rank RCCL communicator
    # is `tools/slab_case.py 1` (a second context; not repeated here)
    cg_variant = "standard PCG, 3 launches per iteration (one rank)" if world.size == 1 else \
                 "single-reduction PCG (Chronopoulos-Gear): one 24-byte all-reduce + one halo exchange per iteration"

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

    one = world.rank == 0 and world.size == 1
    cpu = None
    rp = None
    if one:
        rp, ci = ctx.csr_get(cols=not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(dom, rp, ci, bmask, f)
            del ci
    topo = ctx.mesh_topology()[0]
    world_size, rank = world.size, world.rank
    nranks_rccl = selftest.get("nranks_seen_by_rccl") if selftest else 1      # None: shared-memory test transport, RCCL counted nothing
    transport = selftest["transport"] if selftest else "none (one rank, no communicator)"

    # the headline context is released before the other configurations allocate theirs
    extra, general = None, None
    traffic = Traffic(_lib.source_hash())
    if one and not args.no_extra:
        ctx.close()
        general = general_geometry_leg(_lib, DMPlexDom, Spectral, n, traffic)
        extra = {}
        for key, fn in (("C1", lambda: config_poisson(_lib, DMPlexDom, Spectral, [32, 32], 100,
                                                       "C1: 2D Poisson on 32x32 Q1 quads (the reference's CPU-runnable case; plumbing: launch-bound at 1,089 DOFs)")),
                        ("C2", lambda: config_poisson(_lib, DMPlexDom, Spectral, [128, 128, 128], 100,
                                                       "C2: 3D Poisson on 128^3 Q1 hex, FP64 assembly + Jacobi-PCG")),
                        ("C3", lambda: config_kle(_lib, DMPlexDom, Spectral, 128, 50)),
                        ("C3_general_geometry", lambda: config_kle(_lib, DMPlexDom, Spectral, 128, 50, jitter=0.2)),
                        ("C5", lambda: config_tets(_lib, DMPlexDom, 94, 300)),
                        ("HO3_2D", lambda: config_ho3(_lib, DMPlexDom, Spectral, 2, 1024, 50)),
                        ("HO3_3D", lambda: config_ho3(_lib, DMPlexDom, Spectral, 3, 64, 50))):
            try:
                extra[key] = fn()
            except Exception as e:                      # a failing side configuration must not take the headline line with it
                extra[key] = {"error": f"{type(e).__name__}: {e}"}
        extra["C4"] = {"config": "C4: 256^3 over 8 GPUs", "note": "multi-GPU only: measured by `bench.py --gpus 8` at its own size class "
                                                                     "(the N-GPU runs of this bench partition the 215^3 headline mesh)"}

    if rank == 0:
        asm_kernel = ("assemble_q1_hex_lattice_kernel" if args.variant == 1 and topo == "lattice" and args.jitter == 0.0
                      else "assemble_q1_hex_march_kernel" if args.variant == 1 and topo == "lattice"
                      else "assemble_q1_hex_tiled_kernel" if args.variant != 0 else "assemble_generic_kernel")
        B_asm, B_spmv, B_cg = algorithmic_bytes(n_elem_global, n_node_global, nnz_global)
        # per-rank share of the algorithmic bytes (strong scaling: each GPU streams 1/N of them)
        share = 1.0 / world_size
        cg_gbs = B_cg * share / (cg_mean / args.cg_iters * 1e-3) / 1e9
        # the format the SpMV actually reads: the CSR values (8 B per stored entry, no padding) + row offset and column-pattern id
        # (4 + 4 B per row, dictionary mode) + x and y; quoted NEXT to the CSR-model `achieved` the metric definition prescribes
        fmt = None
        spmv_kernel = "sellp_spmv_kernel" if os.environ.get("PYNAMA_SELL_IMAGE") else "csrl_spmv_kernel"
        if rp is not None:
            image = spmv_kernel == "sellp_spmv_kernel"
            fmt_bytes = (8.0 * sell_entries(rp) + 4.0 * n_rows if image else 8.0 * float(rp[-1]) + 8.0 * n_rows) + 16.0 * n_rows
            fmt = {"bytes_per_launch": fmt_bytes, "GBs": fmt_bytes / (spmv_mean * 1e-3) / 1e9,
                   "frac": fmt_bytes / (spmv_mean * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "model": ("8 B x SELL-64 stored entries (slice padding included) + 4 B pattern id per row + 16 B per row (x, y); " if image else
                             "8 B x CSR entries (the assembly's own array, no second image) + 4 B row offset + 4 B pattern id + 16 B (x, y) per row; ")
                            + "no per-entry column index is streamed"}
        tr_spmv, src_spmv = traffic.get(spmv_kernel)
        tr_asm, src_asm = traffic.get(asm_kernel)
        other = wall / args.steps * 1e3 - asm_mean - cg_mean
        cg_traffic = None
        if world_size == 1:
            parts = [traffic.get(k)[0] for k in (spmv_kernel, "cg_update_selfred_kernel", "cg_p_selfred_kernel")]
            if all(p is not None for p in parts):
                gbs = sum(parts) / (cg_mean / args.cg_iters * 1e-3) / 1e9
                cg_traffic = {"bytes_per_iteration": sum(parts), "GBs": gbs, "frac": gbs / HBM_PEAK_GBS, "source": src_spmv,
                              "kernels": [spmv_kernel, "cg_update_selfred_kernel", "cg_p_selfred_kernel"]}
        out = {
            "metric": "assembled element-DOFs/sec + CG iterations/sec, 10M-DOF Poisson, 1/2/4/8 MI355X",
            "value": elem_dofs_per_step / (asm_mean * 1e-3),
            "unit": "element-DOFs/s (assembly phase); cg_iters_per_s alongside",
            "cg_iters_per_s": args.cg_iters / (cg_mean * 1e-3),
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "breakdown_ms": {"assembly": asm_mean, "cg": cg_mean, "cg_iters": args.cg_iters,
                             "spmv_kernel": spmv_mean, "symbolic_once": symbolic_ms,
                             # what a step spends outside the two metered phases: CG start-up and launch gaps (the product reads the
                             # assembly's CSR values directly and 1 / diagonal leaves the assembly with the rows: no set-up pass)
                             "cg_setup_and_gaps": other,
                             # N > 1 (single-reduction CG): end of the product -> all-reduced sums ready = partial sums + the one
                             # all-reduce, per iteration (the scalar step rides in the update kernel; 0 on one GPU)
                             "cg_reduction_per_iter": red_mean,
                             # overlapped halo exchange (pack + grouped send/recv on the communication stream), per iteration
                             "cg_halo_per_iter": halo_mean},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D Poisson, {n}^3 Q1 hex elements, {n_node_global} DOFs, nnz {nnz_global}, "
                                   f"FP64 assembly + Jacobi-PCG ({args.cg_iters} its/step)",
                       "partition": f"z-slabs x{world_size}", "assembly_variant": args.variant, "cg_variant": cg_variant,
                       "transport": transport, "nranks_seen_by_rccl": nranks_rccl, "kernel_source_hash": _lib.source_hash()},
            "roofline": roofline(spmv_kernel + (" (SELL-64 + column-pattern dictionary SpMV inside CG)" if spmv_kernel.startswith("sellp") else
                                                " (SpMV inside CG straight from the CSR values: coalesced 64-row runs transposed through LDS, "
                                                "column-pattern dictionary)"), B_spmv * share, spmv_mean,
                                 traffic=tr_spmv, traffic_source=src_spmv, format_actually_read=fmt,
                                 note="`achieved` = SURVEY.md 8(d)'s CSR byte model (12 nnz + 4 (N+1) + 16 N) / measured launch time, as the "
                                      "metric prescribes; `format_actually_read` prices the bytes this kernel really streams"),
            "roofline_assembly": roofline(asm_kernel + " (zero + integration + scatter, one launch"
                                          + (": parallelepiped closed form, exact on the uniform box)" if asm_kernel.endswith("lattice_kernel") else ")"),
                                          B_asm * share, asm_mean, traffic=tr_asm, traffic_source=src_asm,
                                          bytes_actually_moved_model={
                                              "bytes_per_launch": (8.0 * nnz_global + 25.0 * n_node_global + 8.0 * n_node_global) * share,
                                              "frac": (8.0 * nnz_global + 33.0 * n_node_global) * share / (asm_mean * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "model": "values written (8 nnz) + coordinates and flags read (25 N) + 1/diagonal written (8 N); the "
                                                       "lattice kernels read neither rowptr nor colidx (4 nnz + 4 N of the algorithmic count)"}),
            "roofline_assembly_general": general,
            "roofline_cg_iteration": {"bound": "hbm", "achieved": cg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": cg_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": B_cg * share,
                                      # the same iteration time priced on the bytes the PMC counters saw its three kernels move
                                      # (product + the two fused vector kernels): the CSR model above charges 4 B of column index
                                      # per entry that the dictionary product never streams, so this is the fraction to steer by
                                      "on_measured_traffic": cg_traffic},
            "check": check,
            "selftest": selftest,
            "matrix_free": mfree,
            "configs": extra,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    world.cleanup()
    if not (one and not args.no_extra):
        ctx.close()


if __name__ == "__main__":
    main()
