#!/usr/bin/env python3
"""Rehearsal of `bench.py --gpus N` on a ONE-GPU box: N ranks as processes sharing the device over the shared-memory test
transport (timings are meaningless -- the ranks share the GPU and every collective is host-staged; what is exercised is
bench.py's N > 1 logic: partition, reductions of the timings, the check solve, the matrix-free leg, the JSON line).
usage: bench_rehearsal.py N [bench.py arguments]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pynama_amd import _lib  # noqa: E402

n = int(sys.argv[1])
cap = 64 << 20
with tempfile.NamedTemporaryFile(dir="/dev/shm", prefix="pynama_bench_") as f:
    f.truncate(_lib.Context.shm_size(n, cap))
    f.flush()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK="0", PYNAMA_SHM_TRANSPORT=f.name, PYNAMA_SHM_CAP=str(cap))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + sys.argv[2:], env=env))
    rc = [p.wait() for p in procs]
sys.exit(max(rc))
