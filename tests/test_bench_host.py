"""bench.py's host-side pieces (no GPU): the byte models of SURVEY.md 8(d), the SELL entry count, the rule that PMC traffic is only
quoted for the kernel sources it was measured on, and the tetrahedral test mesh of configuration C5."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_match_survey_8d():
    # C2: 128^3 Q1 hex, scalar -- SURVEY.md 8(d): 67.1 + 51.5 + 8.6 + 228.3 + 456.5 = 812 MB; CG 1.0025 GB / iteration
    ne, nn_, nnz = 128 ** 3, 129 ** 3, 385 ** 3
    asm, spmv, cg = bench.algorithmic_bytes(ne, nn_, nnz)
    assert asm == 4 * 8 * ne + 8 * 3 * nn_ + 4 * (nn_ + 1) + 4 * nnz + 8 * nnz
    assert abs(asm / 1e6 - 812.0) < 1.0
    assert abs(cg / 1e9 - 1.0025) < 0.001
    assert spmv == 12 * nnz + 4 * (nn_ + 1) + 16 * nn_
    # C3: 3 x 3 blocks on the same graph: 4.46 GB per assembled matrix, 5.29 GB per CG iteration (block CSR)
    asm3, _, cg3 = bench.algorithmic_bytes(ne, nn_, nnz, ndof=3)
    assert abs(asm3 / 1e9 - 4.46) < 0.01 and abs(cg3 / 1e9 - 5.29) < 0.01
    # the 10 M-DOF headline: 4.73 GB per iteration
    _, _, cg10 = bench.algorithmic_bytes(215 ** 3, 216 ** 3, 646 ** 3)
    assert abs(cg10 / 1e9 - 4.73) < 0.01


def test_sell_entries_counts_slice_padding():
    rowptr = np.concatenate([[0], np.cumsum([3] * 64 + [5] + [1] * 10)])
    # slice 0: 64 rows of width 3; slice 1: 11 rows, longest 5, padded to 64 rows
    assert bench.sell_entries(rowptr) == 64 * 3 + 64 * 5
    assert bench.sell_entries(rowptr, ndof=3) == 3 * 64 * 9 + 64 * 15                # 225 scalar rows: three slices of width 9, one of 15


def test_traffic_is_quoted_only_for_the_measured_sources(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r09_pmc_traffic.json").write_text(json.dumps(
        {"source_hash": "aaaa", "kernels": {"csrl_spmv_kernel": {"hbm_bytes_per_launch": 3.0e9}, "other": {"launches": 1}}}))
    (prof / "r09_pmc_assembly.json").write_text(json.dumps(
        {"source_hash": "bbbb", "kernels": {"assemble_q1_hex_lattice_kernel": {"hbm_bytes_per_launch": 2.5e9}}}))
    (prof / "r08_pmc_traffic.json").write_text("{not json")
    # another configuration's pass holds kernels of the same name at another size: never mixed into the headline's figures
    (prof / "r09_pmc_zz_other_config.json").write_text(json.dumps(
        {"source_hash": "aaaa", "kernels": {"csrl_spmv_kernel": {"hbm_bytes_per_launch": 1.0e9}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    t = bench.Traffic("aaaa")
    assert t.get("csrl_spmv_kernel") == (3.0e9, "r09_pmc_traffic.json")
    b, why = t.get("assemble_q1_hex_lattice_kernel")            # measured on other sources: not quoted
    assert b is None and "kernel sources" in why
    assert bench.Traffic("cccc").get("csrl_spmv_kernel")[0] is None


def test_roofline_record():
    r = bench.roofline("k", 8.0e9, 2.0, traffic=None)
    assert r["achieved"] == 4000.0 and r["frac"] == 0.5 and r["peak"] == 8000.0 and r["bound"] == "hbm" and r["unit"] == "GB/s"


def test_kuhn_box_is_a_conforming_positive_tet_mesh():
    xyz, conn = bench.kuhn_box(3)
    assert conn.shape == (6 * 27, 4) and xyz.shape == (64, 3)
    X = xyz[conn]
    vol = np.linalg.det(X[:, 1:] - X[:, :1]) / 6.0
    assert (vol > 0).all() and abs(vol.sum() - np.prod(xyz.max(0) - xyz.min(0))) < 1e-12
    # conforming: every interior triangle is shared by exactly two tetrahedra
    faces = np.sort(conn[:, [[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]]].reshape(-1, 3), axis=1)
    _, counts = np.unique(faces, axis=0, return_counts=True)
    assert set(counts) == {1, 2} and (counts == 1).sum() == 6 * 2 * 9
