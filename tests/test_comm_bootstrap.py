"""Bootstrap of the one-process-per-GPU launch (pynama_amd/common/comm.py), on the CPU with two real processes: every
communicator gets its own RCCL id (a cached id was handed to the second context of a run), stale files of an
earlier launch are never accepted, and a step that never returns ends the process with the phase named on stderr."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
from pynama_amd.common.comm import get_world
w = get_world()
ids = []
for k in range(3):                       # three communicators in one run (e.g. two domains + a re-created context)
    uid = w.unique_id(lambda: os.urandom(128), timeout=30)
    ids.append(uid.hex())
    time.sleep(0.05 * (w.rank + 1))      # ranks drift apart between communicators
print("IDS", w.rank, " ".join(ids), flush=True)
if w.rank == 0:
    time.sleep(0.5)
w.cleanup()
"""


def _launch(tmp_path, nranks, extra_env=None, code=None):
    env = dict(os.environ, WORLD_SIZE=str(nranks), MASTER_PORT="29555", PYNAMA_RDZV_DIR=str(tmp_path), PYNAMA_RDZV_TAG="t1")
    env.update(extra_env or {})
    procs = []
    for r in range(nranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, "-c", code or WORKER.format(root=ROOT)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    return [p.communicate(timeout=120) + (p.returncode,) for p in procs]


def test_every_communicator_gets_a_fresh_id(tmp_path):
    res = _launch(tmp_path, 2)
    got = {}
    for out, err, rc in res:
        assert rc == 0, err
        line = [l for l in out.splitlines() if l.startswith("IDS")][0].split()
        got[int(line[1])] = line[2:]
    assert got[0] == got[1]                          # both ranks hold the same id for the same communicator ...
    assert len(set(got[0])) == 3                     # ... and no id is handed out twice
    assert not os.path.exists(os.path.join(tmp_path, f"pynama_rdzv_t1_{os.getuid()}"))   # rank 0 cleaned up


def test_stale_id_file_is_not_accepted(tmp_path):
    d = os.path.join(tmp_path, f"pynama_rdzv_t1_{os.getuid()}")
    os.makedirs(d)
    stale = os.path.join(d, "uid_0000.bin")
    with open(stale, "wb") as f:
        f.write(b"\x55" * 128)
    old = time.time() - 3600
    os.utime(stale, (old, old))
    res = _launch(tmp_path, 2)
    for out, err, rc in res:
        assert rc == 0, err
        ids = [l for l in out.splitlines() if l.startswith("IDS")][0].split()[2:]
        assert ("55" * 128) not in ids


def test_missing_rank_exits_with_the_phase(tmp_path):
    """rank 1 alone (rank 0 never starts): no id appears -> bounded exit 124, message on stderr"""
    code = WORKER.format(root=ROOT).replace("timeout=30", "timeout=1.0")
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_PORT="29556", PYNAMA_RDZV_DIR=str(tmp_path),
               PYNAMA_RDZV_TAG="t2")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert p.returncode == 124 and "TIMEOUT" in p.stderr and "unique id" in p.stderr


def test_bounded_names_the_phase(tmp_path):
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from pynama_amd.common.comm import bounded\n"
            "with bounded('halo exchange self-test', 0.5, 3):\n    time.sleep(30)\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 124 and "rank 3" in p.stderr and "halo exchange self-test" in p.stderr
