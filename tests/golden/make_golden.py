#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the *reference itself*.

Run ONLY in the build container (the reference lives at /root/reference and never
travels to the GPU box):

    python tests/golden/make_golden.py

What it does: imports the reference's numpy-only element modules
(`src/elements/{utilities,element,spectral}.py`) unmodified, under two harness-side
shims (a stub `mpi4py` whose COMM_WORLD has rank 0 -- `elements/element.py:3,7,13`
reads nothing else -- and nothing more), calls them on fixed inputs, and stores
inputs + outputs as `.npz` data.  No reference source text is stored.

Fixture families:
  G1  quadrature rules            gaussPoints / lobattoPoints
  G2  element tables + orderings  Spectral(ngl, dim) H/Hrs/gps (6 variants) + HCoo1D
  G3  element matrices            getElemKLEMatrices / getElemKLEOperators
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_shims():
    mpi4py = types.ModuleType("mpi4py")

    class _Comm:
        rank = 0
        size = 1

    class _MPI:
        COMM_WORLD = _Comm()

    mpi4py.MPI = _MPI
    sys.modules["mpi4py"] = mpi4py
    sys.modules["mpi4py.MPI"] = _MPI
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)


def _gps_array(gps):
    return np.array([tuple(float(v) for v in g) for g in gps], dtype=np.float64)


VARIANTS = ["", "Red", "Op", "Coo", "CooRed", "CooOp"]


def tables(sp):
    out = {}
    for v in VARIANTS:
        out["H" + v] = np.array(getattr(sp, "H" + v), dtype=np.float64)
        out["Hrs" + v] = np.array(getattr(sp, "Hrs" + v), dtype=np.float64)
        out["gps" + v] = _gps_array(getattr(sp, "gps" + v))
    out["HCoo1D"] = np.array(sp.HCoo1D, dtype=np.float64)
    return out


def element_inputs(dim, rng):
    """name -> flat corner coordinates (DMPlex closure order)."""
    if dim == 2:
        unit = np.array([0, 0, 1, 0, 1, 1, 0, 1], dtype=np.float64)
        ref_test = np.array([1, 1, 0, 1, 0, 0, 1, 0], dtype=np.float64)  # tests/test_element.py:276
    else:
        unit = np.array([0, 0, 0, 0, 1, 0, 1, 1, 0, 1, 0, 0,
                         0, 0, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1], dtype=np.float64)
        ref_test = np.array([1, 1, 1, 0, 1, 1, 0, 0, 1, 1, 0, 1,
                             1, 1, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0], dtype=np.float64)  # test_element.py:270
    brick = unit / 128.0
    stretched = unit.reshape(-1, dim) * np.array([0.3, 1.7, 0.9][:dim]) + np.array([2.0, -1.0, 0.5][:dim])
    jitter = unit.reshape(-1, dim) + 0.15 * rng.uniform(-1, 1, size=(2 ** dim, dim))
    return {
        "unit": unit,
        "reftest": ref_test,
        "brick128": brick,
        "stretched": stretched.ravel(),
        "jitter": jitter.ravel(),
    }


def main():
    _install_shims()
    from elements.utilities import gaussPoints, lobattoPoints
    from elements.spectral import Spectral

    # ---- G1
    g1 = {}
    for n in range(2, 13):
        x, w = gaussPoints(n)
        g1[f"gauss_x_{n}"], g1[f"gauss_w_{n}"] = np.asarray(x), np.asarray(w)
        x, w = lobattoPoints(n)
        g1[f"lobatto_x_{n}"], g1[f"lobatto_w_{n}"] = np.asarray(x), np.asarray(w)
    np.savez_compressed(os.path.join(OUT, "g1_rules.npz"), **g1)

    # ---- G2
    g2 = {}
    for dim, ngls in ((2, (2, 3, 4, 5)), (3, (2, 3, 4))):
        for ngl in ngls:
            sp = Spectral(ngl, dim)
            for k, v in tables(sp).items():
                g2[f"d{dim}_n{ngl}_{k}"] = v
    # node / gauss orderings only (positions of nodal points), cheap, up to higher order
    for dim, ngls in ((2, (6, 7, 11)), (3, (5, 6))):
        for ngl in ngls:
            sp = Spectral(ngl, dim)
            g2[f"d{dim}_n{ngl}_gpsOp"] = _gps_array(sp.gpsOp)
            g2[f"d{dim}_n{ngl}_gps"] = _gps_array(sp.gps)
            g2[f"d{dim}_n{ngl}_gpsRed"] = _gps_array(sp.gpsRed)
    np.savez_compressed(os.path.join(OUT, "g2_tables.npz"), **g2)

    # ---- G3
    g3 = {}
    for dim in (2, 3):
        rng = np.random.default_rng(20260 + dim)
        inputs = element_inputs(dim, rng)
        for ngl in ((2, 3, 5) if dim == 2 else (2, 3)):
            sp = Spectral(ngl, dim)
            for name, c in inputs.items():
                key = f"d{dim}_n{ngl}_{name}"
                g3[key + "_coords"] = c.copy()
                K, Rw, Rd = sp.getElemKLEMatrices(c.copy())
                g3[key + "_K"], g3[key + "_Rw"], g3[key + "_Rd"] = K, Rw, Rd
                SrT, DivSrT, Curl, wei = sp.getElemKLEOperators(c.copy())
                g3[key + "_SrT"], g3[key + "_DivSrT"] = SrT, DivSrT
                g3[key + "_Curl"], g3[key + "_wei"] = Curl, wei
    np.savez_compressed(os.path.join(OUT, "g3_elem.npz"), **g3)
    for f in ("g1_rules.npz", "g2_tables.npz", "g3_elem.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
