"""ctypes binding of libpynama_hip.so (include/pynama_hip.h).

There is NO CPU fallback: if the library is missing or no MI355X is visible, every compute
entry raises.  Loading the library and checking its symbols works without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYNAMA_LIB_PATH") or os.path.join(_HERE, "libpynama_hip.so")   # (the override: A/B of two builds, tools/)
CSRC = os.path.join(_HERE, "csrc")

# enums of include/pynama_hip.h
Q_FULL, Q_RED, Q_NODAL = 0, 1, 2
FORM_LAPLACE, FORM_MASS_NODAL, FORM_MASS_FULL, FORM_KLE = 0, 1, 2, 3
KSP_CG, KSP_GMRES = 0, 1
MATFREE_OFF, MATFREE_LAPLACE, MATFREE_KLE = 0, 1, 2
PC_NONE, PC_JACOBI = 0, 1
NORM_PRECONDITIONED, NORM_UNPRECONDITIONED, NORM_NATURAL = 0, 1, 2
T_SYMBOLIC, T_ASSEMBLE, T_SPMV, T_SOLVE = 0, 1, 2, 3


class PynamaHipError(RuntimeError):
    pass


class SolveOpts(C.Structure):
    _fields_ = [("method", C.c_int), ("pc", C.c_int), ("norm_type", C.c_int), ("maxit", C.c_int),
                ("restart", C.c_int), ("fixed_iters", C.c_int), ("profile", C.c_int), ("cg_variant", C.c_int),
                ("gmres_orthog", C.c_int), ("matfree", C.c_int), ("rtol", C.c_double), ("atol", C.c_double), ("dtol", C.c_double)]


class SolveInfo(C.Structure):
    _fields_ = [("iters", C.c_int), ("reason", C.c_int), ("rnorm", C.c_double), ("rnorm0", C.c_double),
                ("true_resid", C.c_double), ("solve_ms", C.c_double), ("spmv_ms", C.c_double),
                ("spmv_launches", C.c_int), ("reduce_ms", C.c_double), ("halo_ms", C.c_double)]


_P = C.c_void_p
_I, _L, _D = C.c_int, C.c_int64, C.c_double
_pi32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_pi64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_pf64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_pu8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

# name -> argtypes ; every function returns int except pyn_last_error
SIGNATURES = {
    "pyn_version": [],
    "pyn_device_count": [C.POINTER(_I)],
    "pyn_ctx_create": [_I, C.POINTER(_P)],
    "pyn_ctx_destroy": [_P],
    "pyn_sync": [_P],
    "pyn_comm_unique_id": [C.c_char_p, _I],
    "pyn_comm_init": [_P, _I, _I, C.c_char_p, _I],
    "pyn_comm_init_shm": [_P, _I, _I, C.c_char_p, _L],
    "pyn_comm_barrier": [_P],
    "pyn_comm_allreduce_f64": [_P, _pf64, _I, _I],
    "pyn_comm_selftest": [_P, _pf64, _I],
    "pyn_halo_set": [_P, _L, _L, _I, _pi32, _pi64, _pi32, _pi64],
    "pyn_mesh_set": [_P, _I, _I, _L, _L, _pi32, _pf64],
    "pyn_mesh_topology": [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)],
    "pyn_elem_tables_set": [_P, _I, _I, _pf64, _pf64, _pf64, _pf64],
    "pyn_bc_set": [_P, _I, _P],
    "pyn_csr_symbolic": [_P],
    "pyn_csr_info": [_P, C.POINTER(_L), C.POINTER(_L)],
    "pyn_csr_get": [_P, C.c_void_p, C.c_void_p],
    "pyn_patch_plan_set": [_P, _I, _P, _P],
    "pyn_patch_plan_set_kind": [_P, _I, _I, _P, _P],
    "pyn_patch_plan_info": [_P, _I, _P],
    "pyn_mesh_box": [_P, _I, _I, _P, _L, _P, _P, _L, _P, _P],
    "pyn_mesh_get": [_P, _P, _P],
    "pyn_mat_create": [_P, _I, _I, C.POINTER(_I)],
    "pyn_mat_create_rhs": [_P, _I, _I, C.POINTER(_I)],
    "pyn_mat_stored_blocks": [_P, _I, C.POINTER(_L), C.POINTER(_L)],
    "pyn_mat_destroy": [_P, _I],
    "pyn_mat_zero": [_P, _I],
    "pyn_mat_add_values": [_P, _I, _I, _pi32, _I, _pi32, _pf64, _I],
    "pyn_mat_get_values": [_P, _I, _pf64],
    "pyn_mat_get_diagonal": [_P, _I, _I],
    "pyn_mat_axpy": [_P, _I, _D, _I],
    "pyn_mat_row_scale": [_P, _I, _I],
    "pyn_vec_create": [_P, _I, C.POINTER(_I)],
    "pyn_vec_destroy": [_P, _I],
    "pyn_vec_set_host": [_P, _I, _pf64],
    "pyn_vec_set_local_host": [_P, _I, _pf64],
    "pyn_vec_get_host": [_P, _I, _pf64],
    "pyn_vec_fill": [_P, _I, _D],
    "pyn_vec_scatter_host": [_P, _I, _L, _pi32, _pf64, _I],
    "pyn_vec_axpby": [_P, _I, _D, _I, _D, _I],
    "pyn_vec_pointwise_mult": [_P, _I, _I, _I],
    "pyn_vec_reciprocal": [_P, _I],
    "pyn_vec_vtensv": [_P, _I, _I],
    "pyn_vec_dot": [_P, _I, _I, C.POINTER(_D)],
    "pyn_vec_norm": [_P, _I, _I, C.POINTER(_D)],
    "pyn_assemble_kle": [_P, _D, _D, _I, _I, _I, _I, _I],
    "pyn_assemble_kle_noslip": [_P, _D, _D, C.POINTER(_I)],
    "pyn_assemble_scalar": [_P, _I, _I, _I, _I],
    "pyn_elem_local": [_P, _I, _D, _D, _pf64, _P, _P, _P],
    "pyn_assemble_operator": [_P, _I, _I, _pi32, _pf64, _I],
    "pyn_elem_operator_local": [_P, _I, _I, _I, _I, _pi32, _pf64, _pf64, _pf64],
    "pyn_spmv": [_P, _I, _I, _I],
    "pyn_matfree_apply": [_P, _I, _I, _I],
    "pyn_matfree_set": [_P, _I, _D, _D],
    "pyn_solve": [_P, _I, _I, _I, C.POINTER(SolveOpts), C.POINTER(SolveInfo)],
    "pyn_direct_max_rows": [],
    "pyn_solve_direct": [_P, _I, _I, _I, C.POINTER(SolveInfo)],
    "pyn_timers_get": [_P, _pf64, _I],
}

_lib = None


def build_library(force=False):
    """Compile libpynama_hip.so for gfx950 with hipcc (in-tree)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    return LIB_PATH


def load_library():
    """dlopen the library and bind every symbol of the header.  No GPU needed."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PynamaHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pynama_amd/csrc`). pynama_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = C.c_int
    for name in ("pyn_last_error", "pyn_source_hash"):
        getattr(lib, name).argtypes = []
        getattr(lib, name).restype = C.c_char_p
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise PynamaHipError(f"[{rc}] {_lib.pyn_last_error().decode(errors='replace')}")


def source_hash() -> str:
    """identity of the kernel sources the loaded library was built from (profiles/*.json record it)"""
    return load_library().pyn_source_hash().decode()


def device_count() -> int:
    lib = load_library()
    n = _I(0)
    _check(lib.pyn_device_count(C.byref(n)))
    return n.value


class _stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: keeps C-level chatter of third-party libraries (RCCL's version
    banner) out of a program's stdout, which callers such as bench.py reserve for their own output."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        try:
            C.CDLL(None).fflush(None)       # C stdio buffers of the library must drain while fd 1 is still redirected
        except OSError:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def default_device() -> int:
    """GPU of this process: LOCAL_RANK under a one-process-per-GPU launcher, else 0."""
    n = device_count()
    if n <= 0:
        raise PynamaHipError("no MI355X visible: pynama_amd has no CPU fallback")
    return int(os.environ.get("LOCAL_RANK", "0")) % n


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Context:
    """One GPU, one process.  Thin, explicit wrapper: every method is one C-ABI call."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = _P()
        _check(self.lib.pyn_ctx_create(device, C.byref(h)))
        self.h = h
        self.rank, self.nranks = 0, 1
        self.dim = self.nn = 0
        self.n_owned = self.n_ghost = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.pyn_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- communicator
    @staticmethod
    def unique_id() -> bytes:
        lib = load_library()
        buf = C.create_string_buffer(128)
        with _stdout_to_stderr():
            _check(lib.pyn_comm_unique_id(buf, 128))
        return buf.raw

    def comm_init(self, rank, nranks, uid: bytes | None):
        with _stdout_to_stderr():       # RCCL prints a version banner on stdout when a communicator comes up
            _check(self.lib.pyn_comm_init(self.h, rank, nranks, uid, len(uid) if uid else 0))
        self.rank, self.nranks = rank, nranks

    @staticmethod
    def shm_size(nranks, cap_bytes):
        return 4096 + nranks * 512 + nranks * nranks * 16 + nranks * cap_bytes

    def comm_init_shm(self, rank, nranks, path, cap_bytes):
        """TEST transport (several ranks on one GPU): collectives staged through the shared-memory file `path`"""
        _check(self.lib.pyn_comm_init_shm(self.h, rank, nranks, path.encode(), cap_bytes))
        self.rank, self.nranks = rank, nranks

    def barrier(self):
        _check(self.lib.pyn_comm_barrier(self.h))

    def allreduce(self, values, op="sum"):
        a = _f64(np.atleast_1d(values)).copy()
        _check(self.lib.pyn_comm_allreduce_f64(self.h, a, a.size, 1 if op == "max" else 0))
        return a

    def comm_selftest(self):
        """rank count seen by RCCL, all-reduce of 1 / of the rank, rank-stamped halo exchange on both streams (raises on a mismatch)"""
        info = np.zeros(8)
        _check(self.lib.pyn_comm_selftest(self.h, info, info.size))
        out = {"transport": "rccl" if info[0] > 0 else "shm (test transport: ranks share one GPU, no RCCL)",
               "allreduce_sum_ones": info[1], "allreduce_sum_ranks": info[2],
               "halo_ghosts_checked_main_stream": int(info[3]), "halo_ghosts_checked_comm_stream": int(info[4]),
               "allreduce_beside_exchange_sum_ranks_plus_1": info[5]}
        if info[0] > 0:
            out["nranks_seen_by_rccl"] = int(info[0])      # counted by RCCL itself (ncclCommCount of both communicators)
            out["halo_communicator"] = "own (ncclCommSplit)" if info[6] == 1 else "shared with the all-reduces (split refused)"
        return out

    def halo_set(self, n_owned, n_ghost, neigh, send_ptr, send_idx, recv_ptr):
        neigh = _i32(neigh)
        _check(self.lib.pyn_halo_set(self.h, n_owned, n_ghost, len(neigh), neigh if len(neigh) else np.zeros(1, np.int32),
                                     np.ascontiguousarray(send_ptr, np.int64),
                                     _i32(send_idx) if len(send_idx) else np.zeros(1, np.int32),
                                     np.ascontiguousarray(recv_ptr, np.int64)))
        self._halo = True
        self.n_owned, self.n_ghost = int(n_owned), int(n_ghost)

    def sync(self):
        _check(self.lib.pyn_sync(self.h))

    # -- mesh / tables / bc
    def mesh_set(self, dim, conn, xyz):
        conn = _i32(conn)
        xyz = _f64(xyz)
        n_elem, nn = conn.shape
        n_node = xyz.shape[0]
        assert xyz.shape[1] == dim
        _check(self.lib.pyn_mesh_set(self.h, dim, nn, n_elem, n_node, conn, xyz))
        self.dim, self.nn, self.n_elem, self.n_node = dim, nn, n_elem, n_node
        self._bc_last = None
        if not getattr(self, "_halo", False):
            self.n_owned, self.n_ghost = n_node, 0

    def mesh_box(self, dim, ngl, nel_local, layer0, lattice, loc, planes, axes):
        """this rank's block of a structured box mesh, generated on the device (pyn_mesh_box): no host conn / xyz"""
        nel = np.ascontiguousarray(nel_local, dtype=np.int64)
        lat = np.ascontiguousarray(lattice, dtype=np.int64)
        loc = _i32(loc)
        planes = np.ascontiguousarray(planes, dtype=np.int64)
        axes = _f64(np.concatenate([np.asarray(a, dtype=np.float64) for a in axes]))
        assert loc.shape == (ngl ** dim, dim) and nel.size == dim and lat.size == dim and axes.size == int(lat.sum())
        _check(self.lib.pyn_mesh_box(self.h, dim, ngl, nel.ctypes.data_as(_P), int(layer0), lat.ctypes.data_as(_P),
                                     loc.ctypes.data_as(_P), planes.size, planes.ctypes.data_as(_P), axes.ctypes.data_as(_P)))
        per = int(np.prod(lat[:-1]))
        self.dim, self.nn, self.n_elem, self.n_node = dim, ngl ** dim, int(np.prod(nel)), per * planes.size
        self._bc_last = None
        if not getattr(self, "_halo", False):
            self.n_owned, self.n_ghost = self.n_node, 0

    def mesh_get(self, conn=True, xyz=True):
        """host copies of the local mesh as the device holds it"""
        cn = np.empty((self.n_elem, self.nn), np.int32) if conn else None
        xy = np.empty((self.n_node, self.dim), np.float64) if xyz else None
        _check(self.lib.pyn_mesh_get(self.h, cn.ctypes.data_as(_P) if conn else None, xy.ctypes.data_as(_P) if xyz else None))
        return cn, xy

    def patch_plan_info(self, kind=0):
        """(patches, longest patch in rows, longest row in entries, patch-element pairs) of the plan in use"""
        info = np.zeros(4, np.int64)
        _check(self.lib.pyn_patch_plan_info(self.h, kind, info.ctypes.data_as(_P)))
        return tuple(int(v) for v in info)

    def mesh_topology(self):
        """('lattice', nx, ny, nz) for structured Q1 hex meshes, ('general', 0, 0, 0) otherwise"""
        k, a, b, c = _I(0), _I(0), _I(0), _I(0)
        _check(self.lib.pyn_mesh_topology(self.h, C.byref(k), C.byref(a), C.byref(b), C.byref(c)))
        return (("general", "lattice", "lattice-ngl3", "lattice-q1-2d")[k.value], a.value, b.value, c.value)

    def tables_set(self, which, w, H, Hrs, HrsCoo):
        w = _f64(w)
        _check(self.lib.pyn_elem_tables_set(self.h, which, w.size, w, _f64(H), _f64(Hrs), _f64(HrsCoo)))

    def bc_set(self, ndof, mask):
        """Dirichlet mask per local DOF.  A mask identical to the one on the device is not sent again: the library stamps every
        pyn_bc_set as a new Dirichlet set (compact Krhs layouts, skipped zero blocks and cached element lists are tied to the stamp)."""
        if mask is None:
            self._bc_last = None
            _check(self.lib.pyn_bc_set(self.h, 0, None))
            return
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.size == self.n_node * ndof
        last = getattr(self, "_bc_last", None)
        if last is not None and last[0] == ndof and last[1].shape == m.shape and np.array_equal(last[1], m):
            return
        _check(self.lib.pyn_bc_set(self.h, ndof, m.ctypes.data_as(_P)))
        self._bc_last = (ndof, m.copy())

    # -- graph
    def csr_symbolic(self):
        _check(self.lib.pyn_csr_symbolic(self.h))
        self.graph_gen = getattr(self, "graph_gen", 0) + 1     # every matrix handle of the previous graph is dead now
        nr, nz = _L(0), _L(0)
        _check(self.lib.pyn_csr_info(self.h, C.byref(nr), C.byref(nz)))
        self.n_rows, self.nnzb = nr.value, nz.value
        return nr.value, nz.value

    def csr_get(self, cols=True):
        """node graph on the host; cols=False copies the row offsets only (the column array is 27 x larger)"""
        rp = np.empty(self.n_rows + 1, np.int32)
        ci = np.empty(self.nnzb, np.int32) if cols else None
        _check(self.lib.pyn_csr_get(self.h, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p) if cols else None))
        return rp, ci

    def patch_plan_set(self, patch_ptr, patch_rows, kind=0):
        """kind 0: scalar forms (tiles of <= 352 rows), kind 1: tiled KLE assembly (<= 36 rows)"""
        if patch_ptr is None:
            _check(self.lib.pyn_patch_plan_set_kind(self.h, kind, 0, None, None))
            return
        pp = _i32(patch_ptr)
        pr = _i32(patch_rows)
        _check(self.lib.pyn_patch_plan_set_kind(self.h, kind, len(pp) - 1, pp.ctypes.data_as(_P), pr.ctypes.data_as(_P)))

    # -- matrices
    def mat_create(self, br, bc) -> int:
        i = _I(-1)
        _check(self.lib.pyn_mat_create(self.h, br, bc, C.byref(i)))
        return i.value

    def mat_create_rhs(self, br, bc) -> int:
        """compact imposed-column matrix (Krhs / Krhsfs / Arhs): only the node rows next to an imposed node are stored"""
        i = _I(-1)
        _check(self.lib.pyn_mat_create_rhs(self.h, br, bc, C.byref(i)))
        return i.value

    def mat_stored(self, mid):
        """(graph blocks, node rows) the matrix stores: the whole graph, or the boundary layer of a compact imposed-column matrix"""
        b, r = _L(0), _L(0)
        _check(self.lib.pyn_mat_stored_blocks(self.h, mid, C.byref(b), C.byref(r)))
        return b.value, r.value

    def mat_values(self, mid, br, bc):
        v = np.empty(self.nnzb * br * bc, np.float64)
        _check(self.lib.pyn_mat_get_values(self.h, mid, v))
        return v

    def mat_zero(self, mid):
        _check(self.lib.pyn_mat_zero(self.h, mid))

    def mat_axpy(self, y, a, x):
        _check(self.lib.pyn_mat_axpy(self.h, y, a, x))

    def mat_row_scale(self, mid, vid):
        _check(self.lib.pyn_mat_row_scale(self.h, mid, vid))

    def mat_add_values(self, mid, rows, cols, vals, insert=False):
        """dense block into the matrix at scalar DOF indices (Mat.setValues); a scalar `vals` fills the block"""
        rows, cols = _i32(np.atleast_1d(rows)), _i32(np.atleast_1d(cols))
        v = np.asarray(vals, dtype=np.float64)
        if v.size == 1:
            v = np.full(rows.size * cols.size, float(v.reshape(-1)[0]))
        v = np.ascontiguousarray(v.reshape(-1))
        assert v.size == rows.size * cols.size, "vals must hold len(rows) x len(cols) entries"
        _check(self.lib.pyn_mat_add_values(self.h, mid, rows.size, rows, cols.size, cols, v, 1 if insert else 0))

    def mat_destroy(self, mid):
        if self.h:
            _check(self.lib.pyn_mat_destroy(self.h, mid))

    def mat_diagonal(self, mid, vid):
        _check(self.lib.pyn_mat_get_diagonal(self.h, mid, vid))

    # -- vectors
    def vec_create(self, bs) -> int:
        i = _I(-1)
        _check(self.lib.pyn_vec_create(self.h, bs, C.byref(i)))
        return i.value

    def vec_destroy(self, vid):
        if self.h:
            _check(self.lib.pyn_vec_destroy(self.h, vid))

    def vec_set(self, vid, arr):
        _check(self.lib.pyn_vec_set_host(self.h, vid, _f64(arr)))

    def vec_set_local(self, vid, arr):
        a = _f64(arr)
        assert a.size % (self.n_owned + self.n_ghost) == 0
        _check(self.lib.pyn_vec_set_local_host(self.h, vid, a))

    def vec_get(self, vid, bs):
        out = np.empty(self.n_owned * bs, np.float64)
        _check(self.lib.pyn_vec_get_host(self.h, vid, out))
        return out

    def vec_fill(self, vid, value):
        _check(self.lib.pyn_vec_fill(self.h, vid, float(value)))

    def vec_scatter(self, vid, idx, vals, add=False):
        idx = _i32(idx)
        vals = _f64(vals)
        assert idx.size == vals.size
        _check(self.lib.pyn_vec_scatter_host(self.h, vid, idx.size, idx, vals, 1 if add else 0))

    def vec_axpby(self, w, a, x, b, y):
        _check(self.lib.pyn_vec_axpby(self.h, w, float(a), x, float(b), y))

    def vec_pointwise_mult(self, w, x, y):
        _check(self.lib.pyn_vec_pointwise_mult(self.h, w, x, y))

    def vec_reciprocal(self, x):
        _check(self.lib.pyn_vec_reciprocal(self.h, x))

    def vec_vtensv(self, v, out):
        _check(self.lib.pyn_vec_vtensv(self.h, v, out))

    def vec_dot(self, x, y) -> float:
        d = _D(0)
        _check(self.lib.pyn_vec_dot(self.h, x, y, C.byref(d)))
        return d.value

    def vec_norm(self, x, norm_type=2) -> float:
        d = _D(0)
        _check(self.lib.pyn_vec_norm(self.h, x, norm_type, C.byref(d)))
        return d.value

    # -- hot loops
    def assemble_kle(self, alpha_d, alpha_w, K=-1, Krhs=-1, Rw=-1, Rd=-1, variant=1):
        _check(self.lib.pyn_assemble_kle(self.h, alpha_d, alpha_w, K, Krhs, Rw, Rd, variant))

    def assemble_kle_noslip(self, alpha_d, alpha_w, mat_ids):
        ids = (_I * 8)(*[int(m) for m in mat_ids])
        _check(self.lib.pyn_assemble_kle_noslip(self.h, alpha_d, alpha_w, ids))

    def assemble_scalar(self, form, A=-1, Arhs=-1, variant=1):
        _check(self.lib.pyn_assemble_scalar(self.h, form, A, Arhs, variant))

    def elem_local(self, form, corners, alpha_d=1e3, alpha_w=1e2):
        dim, nn = self.dim, self.nn
        dw = 1 if dim == 2 else 3
        c = _f64(corners).ravel()
        assert c.size == (nn if nn == dim + 1 else 2 ** dim) * dim
        if form == FORM_KLE:
            o0 = np.empty((dim * nn, dim * nn))
            o1 = np.empty((dim * nn, dw * nn))
            o2 = np.empty((dim * nn, nn))
            _check(self.lib.pyn_elem_local(self.h, form, alpha_d, alpha_w, c, o0.ctypes.data_as(_P),
                                           o1.ctypes.data_as(_P), o2.ctypes.data_as(_P)))
            return o0, o1, o2
        o0 = np.empty((nn, nn))
        _check(self.lib.pyn_elem_local(self.h, form, 0.0, 0.0, c, o0.ctypes.data_as(_P), None, None))
        return o0

    def assemble_operator(self, rule, terms, coef, mid):
        t = _i32(terms).reshape(-1, 3)
        _check(self.lib.pyn_assemble_operator(self.h, rule, t.shape[0], t, _f64(coef), mid))

    def elem_operator_local(self, rule, br, bc, terms, coef, corners):
        t = _i32(terms).reshape(-1, 3)
        out = np.empty((br * self.nn, bc * self.nn))
        _check(self.lib.pyn_elem_operator_local(self.h, rule, br, bc, t.shape[0], t, _f64(coef), _f64(corners).ravel(), out))
        return out

    def spmv(self, mid, x, y):
        _check(self.lib.pyn_spmv(self.h, mid, x, y))

    def matfree_apply(self, x, y, op=1):
        """y = A x without an assembled matrix (op: MATFREE_LAPLACE scalar / MATFREE_KLE 3 DOFs per node; structured Q1
        hex meshes)"""
        _check(self.lib.pyn_matfree_apply(self.h, op, x, y))

    def matfree_set(self, op=1, alpha_d=0.0, alpha_w=0.0):
        """define the matrix-free operator from the mesh, the tables and a snapshot of the CURRENT Dirichlet mask"""
        _check(self.lib.pyn_matfree_set(self.h, op, alpha_d, alpha_w))

    def matfree_kle_set(self, alpha_d, alpha_w):
        self.matfree_set(MATFREE_KLE, alpha_d, alpha_w)

    def solve(self, mid, b, x, method=KSP_CG, pc=PC_JACOBI, rtol=1e-5, atol=1e-50, dtol=1e5, maxit=10000,
              restart=30, norm_type=NORM_PRECONDITIONED, fixed_iters=0, profile=0, cg_variant=0, gmres_orthog=0,
              matfree=0) -> SolveInfo:
        o = SolveOpts(method, pc, norm_type, maxit, restart, fixed_iters, profile, cg_variant, gmres_orthog, matfree,
                      rtol, atol, dtol)
        info = SolveInfo()
        _check(self.lib.pyn_solve(self.h, mid, b, x, C.byref(o), C.byref(info)))
        return info

    def direct_max_rows(self):
        return int(self.lib.pyn_direct_max_rows())

    def solve_direct(self, mid, b, x) -> SolveInfo:
        """dense LU with partial pivoting (small systems, one rank); the factors stay cached until the matrix changes"""
        info = SolveInfo()
        _check(self.lib.pyn_solve_direct(self.h, mid, b, x, C.byref(info)))
        return info

    def timers(self):
        t = np.zeros(8)
        _check(self.lib.pyn_timers_get(self.h, t, 8))
        return {"symbolic_ms": t[0], "assemble_ms": t[1], "spmv_ms": t[2], "solve_ms": t[3]}
