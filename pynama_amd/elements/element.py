"""Base element.  Mirrors ``src/elements/element.py`` (``Element.__init__`` :7-15,
``interpFun1D`` :17-49) without the mpi4py import: the communicator only supplied ``rank``."""
import logging

import numpy as np


class Element(object):
    def __init__(self, dim, comm=None):
        self.comm = comm
        self.dim = dim
        self.dim_w = 1 if dim == 2 else 3
        self.dim_s = 3 if dim == 2 else 6
        rank = getattr(comm, "rank", 0) if comm is not None else 0
        self.logger = logging.getLogger("[{}] Class".format(rank))

    def interpFun1D(self, Nodes, evalPoi):
        """Lagrange cardinal functions on `Nodes` and their derivatives at `evalPoi`.
        Returns (hFun, dhFun), each [len(evalPoi), len(Nodes)]."""
        x = np.asarray(Nodes, dtype=np.float64)
        t = np.asarray(evalPoi, dtype=np.float64)
        m = x.size
        diff = t[:, None] - x[None, :]                       # [npt, m]
        den = np.array([np.prod(np.delete(x[a] - x, a)) for a in range(m)])
        hFun = np.empty((t.size, m))
        dhFun = np.zeros((t.size, m))
        for a in range(m):
            rest = np.delete(diff, a, axis=1)                # factors (t - x_b), b != a
            hFun[:, a] = np.prod(rest, axis=1) / den[a]
            for skip in range(m - 1):
                dhFun[:, a] += np.prod(np.delete(rest, skip, axis=1), axis=1)
            dhFun[:, a] /= den[a]
        return (hFun, dhFun)
