"""N > 1 path on CPU: world_size-2 (and 3) gloo jobs exercising the product's slab partition and
halo plan with the oracle's numerics (see tests/dist_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("nproc,nelem,ngl", [(2, "5,4,6", 2), (3, "6,9", 2), (2, "3,4", 3)])
def test_distributed_cg_matches_serial(nproc, nelem, ngl):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), nelem, str(ngl)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok=True") == nproc


@pytest.mark.parametrize("nproc,cell,nelem", [(2, "tet", "4,3,5"), (3, "hex", "4,4,5")])
def test_distributed_cg_on_imported_mesh(tmp_path, nproc, cell, nelem):
    """row-block partition of an imported (Gmsh) mesh: ghost index lists and a halo plan between arbitrary
    rank pairs, driven with the oracle's numerics"""
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import fem_oracle as fo
    from pynama_amd.domain.gmsh import write_msh
    ne = [int(v) for v in nelem.split(",")]
    src = fo.simplex_box_mesh(ne, [0.0] * 3, [1.0] * 3, jitter=0.2) if cell == "tet" else fo.box_mesh(ne, [0.0] * 3, [1.0] * 3, 2, jitter=0.2)
    perm = np.random.default_rng(2).permutation(src.n_node)
    path = str(tmp_path / "mesh.msh")
    write_msh(path, src.xyz[np.argsort(perm)], perm[src.conn])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), nelem, "2", path]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok=True") == nproc
