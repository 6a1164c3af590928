"""Process group of the one-process-per-GPU launch (replaces ``MPI.COMM_WORLD``,
``src/cases/base_problem.py:22``).  Ranks come from the launcher's environment (RANK,
LOCAL_RANK, WORLD_SIZE -- `python -m torch.distributed.run` / torchrun set them); the RCCL
unique id is exchanged through files in a per-launch node-local directory (one node, xGMI), after
which all collectives run inside libpynama_hip.so over RCCL.  No MPI, no torch.

Every communicator gets its OWN id (a sequence number per process: contexts are created in the same
order on every rank), so a second domain / problem in the same run never re-uses a consumed id.
Every blocking step of the bootstrap runs under `bounded(...)`: a watchdog that names the phase on
stderr and ends the process (plain exit, never a re-exec) instead of hanging the node.
"""
import os
import sys
import tempfile
import threading
import time

_T_IMPORT = time.time()


class bounded:
    """`with bounded("phase", seconds, rank): blocking_call()` -- if the block does not finish in time the process prints
    the phase and exits with status 124.  The C calls release the GIL, so the timer thread runs while RCCL waits."""

    def __init__(self, phase, seconds=None, rank=0):
        self.phase, self.rank = phase, rank
        self.seconds = float(os.environ.get("PYNAMA_COMM_TIMEOUT", "180")) if seconds is None else float(seconds)

    def _fire(self):
        sys.stderr.write(f"[pynama rank {self.rank}] TIMEOUT: '{self.phase}' did not finish within {self.seconds:.0f} s -- "
                         "a rank is missing, the RCCL id is stale, or the halo plans of two ranks disagree; exiting\n")
        sys.stderr.flush()
        os._exit(124)

    def __enter__(self):
        self._t = threading.Timer(self.seconds, self._fire)
        self._t.daemon = True
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._t.cancel()
        return False


class Comm:
    """Minimal communicator facade with the attributes the reference reads (rank, size)."""

    def __init__(self, rank=0, size=1, local_rank=0):
        self.rank, self.size, self.local_rank = rank, size, local_rank
        self._seq = 0          # communicators created so far by this process

    # mpi4py / petsc4py spellings used by the reference
    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def getRank(self):
        return self.rank

    def getSize(self):
        return self.size

    def tompi4py(self):
        return self

    def allgather(self, obj):
        if self.size != 1:
            raise NotImplementedError("python-object allgather is not part of the device path; "
                                      "global sets are computed redundantly on every rank")
        return [obj]

    def bounded(self, phase, seconds=None):
        return bounded(phase, seconds, self.rank)

    # ---- RCCL bootstrap
    def _rdzv_dir(self):
        # all ranks of one launch share the launcher as parent: its pid (+ port + run id) names the launch
        tag = os.environ.get("PYNAMA_RDZV_TAG") or "{}_{}_{}".format(
            os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())
        d = os.environ.get("PYNAMA_RDZV_DIR", tempfile.gettempdir())
        return os.path.join(d, f"pynama_rdzv_{tag}_{os.getuid()}")

    def _uid_path(self, seq):
        return os.path.join(self._rdzv_dir(), f"uid_{seq:04d}.bin")

    def unique_id(self, make_id, timeout=None):
        """A FRESH id for the next communicator: rank 0 creates it and publishes it atomically as uid_<seq>, the others poll
        for that name.  A file older than this process (left by a crashed launch that happened to share the tag) is never
        accepted: rank 0 overwrites it, the others wait for the overwrite."""
        if self.size == 1:
            return None
        timeout = float(os.environ.get("PYNAMA_COMM_TIMEOUT", "180")) if timeout is None else timeout
        seq, self._seq = self._seq, self._seq + 1
        path = self._uid_path(seq)
        if self.rank == 0:
            os.makedirs(self._rdzv_dir(), exist_ok=True)
            if seq == 0:         # stale files of an earlier launch with the same tag
                for fn in os.listdir(self._rdzv_dir()):
                    fp = os.path.join(self._rdzv_dir(), fn)
                    try:
                        if os.stat(fp).st_mtime < _T_IMPORT - 120.0:
                            os.remove(fp)
                    except OSError:
                        pass
            uid = make_id()
            tmp = path + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(uid)
            os.replace(tmp, path)
            return uid
        t0 = time.time()
        while True:
            try:
                st = os.stat(path)
                if st.st_size >= 128 and st.st_mtime >= _T_IMPORT - 120.0:
                    with open(path, "rb") as f:
                        return f.read()
            except FileNotFoundError:
                pass
            if time.time() - t0 > timeout:
                sys.stderr.write(f"[pynama rank {self.rank}] TIMEOUT: no fresh RCCL unique id at {path} after {timeout:.0f} s "
                                 "(did rank 0 start?); exiting\n")
                sys.stderr.flush()
                os._exit(124)
            time.sleep(0.02)

    def selftest(self, ctx, dom=None):
        """Start-up check of a freshly created communicator (bench.py --gpus N): every phase bounded, every mismatch an error
        that names the rank.  Returns what rank 0 puts into the JSON line."""
        with self.bounded("communicator self-test: all-reduce of 1 and of the rank, rank-stamped halo exchange on both streams"):
            info = ctx.comm_selftest()
        with self.bounded("barrier after the self-test"):
            ctx.barrier()
        info["neighbours"] = [int(r) for r in getattr(dom, "_neigh_ranks", [])] if dom is not None else []
        return info

    def cleanup(self):
        if self.size > 1 and self.rank == 0:
            d = self._rdzv_dir()
            try:
                for fn in os.listdir(d):
                    os.remove(os.path.join(d, fn))
                os.rmdir(d)
            except OSError:
                pass


_world = None


def get_world() -> Comm:
    global _world
    if _world is None:
        _world = Comm(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                      int(os.environ.get("LOCAL_RANK", "0")))
    return _world


COMM_WORLD = None  # resolved lazily through get_world()
