"""Process group of the one-process-per-GPU launch (replaces ``MPI.COMM_WORLD``,
``src/cases/base_problem.py:22``).  Ranks come from the launcher's environment (RANK,
LOCAL_RANK, WORLD_SIZE -- `python -m torch.distributed.run` / torchrun set them); the RCCL
unique id is exchanged through a file in a node-local directory (one node, xGMI), after which
all collectives run inside libpynama_hip.so over RCCL.  No MPI, no torch.
"""
import os
import tempfile
import time


class Comm:
    """Minimal communicator facade with the attributes the reference reads (rank, size)."""

    def __init__(self, rank=0, size=1, local_rank=0):
        self.rank, self.size, self.local_rank = rank, size, local_rank
        self._uid = None

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

    # ---- RCCL bootstrap
    def _uid_path(self):
        # all ranks of one launch share the launcher as parent: its pid keeps files of earlier
        # (possibly crashed) jobs on the same port from being picked up
        tag = os.environ.get("PYNAMA_RDZV_TAG") or "{}_{}_{}".format(
            os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())
        d = os.environ.get("PYNAMA_RDZV_DIR", tempfile.gettempdir())
        return os.path.join(d, f"pynama_rccl_uid_{tag}_{os.getuid()}.bin")

    def unique_id(self, make_id, timeout=300.0):
        """rank 0 creates the id and publishes it atomically; the others poll for it."""
        if self.size == 1:
            return None
        if self._uid is not None:
            return self._uid
        path = self._uid_path()
        if self.rank == 0:
            uid = make_id()
            tmp = path + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(uid)
            os.replace(tmp, path)
        else:
            t0 = time.time()
            while True:
                try:
                    if os.stat(path).st_size >= 128:
                        with open(path, "rb") as f:
                            uid = f.read()
                        break
                except FileNotFoundError:
                    pass
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"rank {self.rank}: no RCCL unique id at {path}")
                time.sleep(0.05)
        self._uid = uid
        return uid

    def cleanup(self):
        if self.size > 1 and self.rank == 0:
            try:
                os.remove(self._uid_path())
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
