"""Dense-LU direct solve (pyn_solve_direct) next to the Krylov substitute of preonly/lu: factor + solve times by size.
usage: python tools/direct_case.py  (on the GPU box)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import pynama_amd
from pynama_amd import _lib

pynama_amd.install_reference_layout()
import os
import yaml
from cases.uniform import UniformFlow

CASES = os.path.join(os.path.dirname(pynama_amd.__file__), "cases")
with open(os.path.join(CASES, 'uniform.yaml')) as f:
    y = yaml.load(f, Loader=yaml.Loader)
for nelem in ([6, 6, 6], [9, 9, 9], [12, 12, 12]):
    fem = UniformFlow(y, case='uniform', lower=[0, 0, 0], upper=[1, 1, 1], nelem=nelem, ngl=2)
    fem.setUp()
    fem.setUpSolver()
    K = fem.mat.K
    n = K.ctx.n_owned * K.br
    b, x = K.createVecLeft(), K.createVecRight()
    b.setArray(np.random.default_rng(1).standard_normal(n))
    t0 = time.perf_counter()
    i1 = K.ctx.solve_direct(K.id, b.id, x.id)
    t1 = time.perf_counter()
    i2 = K.ctx.solve_direct(K.id, b.id, x.id)
    t2 = time.perf_counter()
    ik = K.ctx.solve(K.id, b.id, x.id, method=_lib.KSP_CG, pc=_lib.PC_JACOBI, rtol=1e-14, atol=1e-300, dtol=1e8, maxit=200000,
                     norm_type=_lib.NORM_UNPRECONDITIONED)
    t3 = time.perf_counter()
    print(f"n={n}: factor+solve {1e3 * (t1 - t0):.1f} ms (resid {i1.true_resid:.1e}), cached solve {1e3 * (t2 - t1):.2f} ms "
          f"(device {i2.solve_ms:.2f}), PCG substitute {1e3 * (t3 - t2):.2f} ms / {ik.iters} its (resid {ik.true_resid:.1e})", flush=True)
