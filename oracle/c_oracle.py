"""ctypes access to oracle/c/fem_oracle.c -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench
cpu_baseline).  Never imported by pynama_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libfem_oracle_c.so")
_lib = None
_f = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_V = C.c_void_p


def build():
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")])
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_num_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads.restype = None
        L.orc_csr_pattern.restype = C.c_int64
        L.orc_csr_pattern.argtypes = [C.c_int, C.c_int64, C.c_int64, _i, _i, _V]
        L.orc_elem_kle.restype = None
        L.orc_elem_kle.argtypes = [C.c_int, C.c_int, C.c_int, _f, _f, _f, _f, C.c_int, _f, _f, _f, _f, _f,
                                   C.c_double, C.c_double, _f, _f, _f]
        L.orc_assemble_laplace.restype = None
        L.orc_assemble_laplace.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, _i, _f, C.c_int, _f, _f, _f, _i, _i,
                                           _V, C.c_int64, _f, _V]
        L.orc_assemble_kle.restype = None
        L.orc_assemble_kle.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, _i, _f, C.c_int, _f, _f, _f, _f, C.c_int,
                                       _f, _f, _f, _f, C.c_double, C.c_double, _i, _i, _V, C.c_int64, _f, _V, _V]
        L.orc_spmv.restype = None
        L.orc_spmv.argtypes = [C.c_int64, C.c_int, C.c_int, _i, _i, _f, _f, _f]
        L.orc_pcg.restype = C.c_int
        L.orc_pcg.argtypes = [C.c_int64, C.c_int, _i, _i, _f, _f, _f, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                              C.POINTER(C.c_double)]
        _lib = L
    return _lib


def num_threads():
    return lib().orc_num_threads()


def usable_cores():
    """cores this process may really use: affinity mask and cgroup quota, not the host's core count"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, q // int(g.read())))
        except (OSError, ValueError, IndexError):
            pass
    return n


def set_threads(n):
    lib().orc_set_threads(int(n))
    return num_threads()


def _tabs(q):
    return (np.ascontiguousarray(q.w), np.ascontiguousarray(q.H), np.ascontiguousarray(q.Hrs))


def csr_pattern(conn, n_node):
    L = lib()
    conn = np.ascontiguousarray(conn, np.int32)
    rp = np.zeros(n_node + 1, np.int32)
    nnz = L.orc_csr_pattern(conn.shape[1], conn.shape[0], n_node, conn, rp, None)
    ci = np.zeros(nnz, np.int32)
    L.orc_csr_pattern(conn.shape[1], conn.shape[0], n_node, conn, rp, ci.ctypes.data_as(_V))
    return rp, ci


def elem_kle(tb, X, alpha_d=1e3, alpha_w=1e2):
    L = lib()
    dim, nn = tb.dim, tb.nn
    dw = tb.dim_w
    K = np.zeros((dim * nn, dim * nn))
    Rw = np.zeros((dim * nn, dw * nn))
    Rd = np.zeros((dim * nn, nn))
    wf, Hf, Hrsf = _tabs(tb.full)
    wr, Hr, Hrsr = _tabs(tb.red)
    L.orc_elem_kle(dim, nn, len(wf), wf, Hf, Hrsf, np.ascontiguousarray(tb.coo.Hrs), len(wr), wr, Hr, Hrsr,
                   np.ascontiguousarray(tb.coo_red.Hrs), np.ascontiguousarray(X, np.float64).ravel(),
                   alpha_d, alpha_w, K, Rw, Rd)
    return K, Rw, Rd


def assemble_laplace(mesh, tb, rp, ci, bc_mask=None, e0=0, e1=None, with_rhs=True):
    L = lib()
    e1 = mesh.n_elem if e1 is None else e1
    A = np.zeros(len(ci))
    Ar = np.zeros(len(ci)) if with_rhs else None
    w, _, Hrs = _tabs(tb.full)
    bc = None if bc_mask is None else np.ascontiguousarray(bc_mask, np.uint8)
    L.orc_assemble_laplace(tb.dim, tb.nn, e0, e1, np.ascontiguousarray(mesh.conn, np.int32),
                           np.ascontiguousarray(mesh.xyz), len(w), w, Hrs, np.ascontiguousarray(tb.coo.Hrs), rp, ci,
                           None if bc is None else bc.ctypes.data_as(_V), mesh.n_node, A,
                           None if Ar is None else Ar.ctypes.data_as(_V))
    return A, Ar


def assemble_kle(mesh, tb, rp, ci, bc_mask=None, alpha_d=1e3, alpha_w=1e2, e0=0, e1=None, with_rhs=True, with_rw=True):
    L = lib()
    dim, dw = tb.dim, tb.dim_w
    e1 = mesh.n_elem if e1 is None else e1
    K = np.zeros(len(ci) * dim * dim)
    Kr = np.zeros(len(ci) * dim * dim) if with_rhs else None
    Rw = np.zeros(len(ci) * dim * dw) if with_rw else None
    wf, Hf, Hrsf = _tabs(tb.full)
    wr, Hr, Hrsr = _tabs(tb.red)
    bc = None if bc_mask is None else np.ascontiguousarray(bc_mask, np.uint8)
    L.orc_assemble_kle(dim, tb.nn, e0, e1, np.ascontiguousarray(mesh.conn, np.int32), np.ascontiguousarray(mesh.xyz),
                       len(wf), wf, Hf, Hrsf, np.ascontiguousarray(tb.coo.Hrs), len(wr), wr, Hr, Hrsr,
                       np.ascontiguousarray(tb.coo_red.Hrs), alpha_d, alpha_w, rp, ci,
                       None if bc is None else bc.ctypes.data_as(_V), mesh.n_node, K,
                       None if Kr is None else Kr.ctypes.data_as(_V), None if Rw is None else Rw.ctypes.data_as(_V))
    return K, Kr, Rw


def spmv(rp, ci, val, x, br=1, bc=1):
    y = np.zeros((len(rp) - 1) * br)
    lib().orc_spmv(len(rp) - 1, br, bc, rp, ci, np.ascontiguousarray(val), np.ascontiguousarray(x), y)
    return y


def pcg(rp, ci, val, b, bs=1, rtol=1e-5, atol=1e-50, maxit=10000, norm_type=0, fixed_iters=0):
    x = np.zeros_like(b)
    rn = C.c_double(0)
    it = lib().orc_pcg(len(rp) - 1, bs, rp, ci, np.ascontiguousarray(val), np.ascontiguousarray(b), x, rtol, atol, maxit,
                       norm_type, fixed_iters, C.byref(rn))
    return x, it, rn.value
