"""Helpers shared by the GPU parity tests."""
import numpy as np
import scipy.sparse as sp


def device_available():
    try:
        from pynama_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


def block_csr_to_scipy(rowptr, colidx, val, br, bc, n_cols_nodes=None):
    """Layout of include/pynama_hip.h: val[(rowptr[i]*br + p*len_i + k)*bc + q]."""
    n = len(rowptr) - 1
    lens = np.diff(rowptr).astype(np.int64)
    rows, cols = [], []
    # scalar rows (i,p): entries (k,q)
    node_of_entry = np.repeat(np.arange(n), lens)                      # [nnzb]
    for p in range(br):
        for q in range(bc):
            pass
    # build index arrays in storage order
    out_r = np.empty(val.size, dtype=np.int64)
    out_c = np.empty(val.size, dtype=np.int64)
    pos = 0
    starts = rowptr[:-1].astype(np.int64)
    for i in range(n):
        ln = lens[i]
        cj = colidx[starts[i]:starts[i] + ln].astype(np.int64)
        blk_c = (cj[:, None] * bc + np.arange(bc)[None, :]).ravel()   # (k,q)
        for p in range(br):
            out_r[pos:pos + ln * bc] = i * br + p
            out_c[pos:pos + ln * bc] = blk_c
            pos += ln * bc
    ncn = n if n_cols_nodes is None else n_cols_nodes
    return sp.coo_matrix((val, (out_r, out_c)), shape=(n * br, ncn * bc)).tocsr()


def mat_to_scipy(ctx, mid, br, bc):
    rp, ci = ctx.csr_get()
    return block_csr_to_scipy(rp, ci, ctx.mat_values(mid, br, bc), br, bc, n_cols_nodes=ctx.n_node)


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def sp_rel_err(A, B):
    d = abs(A - B)
    s = abs(B).max()
    return (d.max() if d.nnz else 0.0) / (s if s > 0 else 1.0)
