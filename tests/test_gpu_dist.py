"""world_size > 1 on ONE GPU: the ranks are processes sharing the device, the library's shared-memory test transport
replaces RCCL (duplicate devices are refused there).  Runs the product's distributed path end to end on the device."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("size,nelem,case", [(2, "9,8,11", "poisson-jitter"), (3, "16,15,17", "poisson"), (2, "7,6,9", "kle"),
                                             (3, "8,7,10", "kle-jitter"),
                                             # tall slabs: several layers of tiles per rank, so that the matrix-free products split
                                             # into the tiles without ghost planes (overlapped with the halo exchange) and the rest
                                             (2, "6,5,47", "poisson"), (3, "5,4,80", "poisson-jitter"), (2, "5,4,45", "kle")])
def test_ranks_sharing_one_gpu(size, nelem, case):
    from pynama_amd import _lib
    cap = 4 << 20
    with tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, prefix="pynama_shm_") as f:
        f.truncate(_lib.Context.shm_size(size, cap))
        f.flush()
        env = dict(os.environ, PYNAMA_SHM_CAP=str(cap))
        if int(nelem.split(",")[2]) >= 40:       # tall slabs: the halo / product overlap must be engaged (assembled and matrix-free)
            env["PYNAMA_OVERLAP_REQUIRE"] = "1"
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(r), str(size), f.name, nelem, case],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(size)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=280)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                pytest.fail("distributed GPU worker timed out")
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
