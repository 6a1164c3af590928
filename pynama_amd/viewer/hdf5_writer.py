"""Minimal HDF5 writer / reader over the system's libhdf5 (ctypes).

The reference writes its fields with PETSc's HDF5 viewer (src/viewer/paraviewer.py:18-50).  Neither petsc4py
nor h5py exists in this image, but libhdf5 does (/opt/conda/lib): the handful of C calls needed for
``/<group>/<name>`` float64 datasets are bound here.  If the library cannot be loaded the writer raises
``RuntimeError`` naming what it looked for -- nothing is written in another format silently."""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

_lib = None
H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 0, 2, 0, 0
hid_t = C.c_int64


def _load():
    global _lib
    if _lib is not None:
        return _lib
    tried = []
    cands = [os.environ.get("PYNAMA_HDF5_LIB"), ctypes.util.find_library("hdf5")]
    cands += sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    for c in cands:
        if not c:
            continue
        try:
            lib = C.CDLL(c)
            lib.H5open.restype = C.c_int
            if lib.H5open() < 0:
                raise OSError("H5open failed")
            break
        except OSError as e:
            tried.append(f"{c}: {e}")
    else:
        raise RuntimeError("libhdf5 not found (set PYNAMA_HDF5_LIB); tried " + "; ".join(tried or ["nothing"]))
    for name, res, args in [
            ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]),
            ("H5Fclose", C.c_int, [hid_t]), ("H5Gcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
            ("H5Gclose", C.c_int, [hid_t]), ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]),
            ("H5Sclose", C.c_int, [hid_t]), ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Dget_space", hid_t, [hid_t]),
            ("H5Sget_simple_extent_npoints", C.c_int64, [hid_t]),
            ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), ("H5Dclose", C.c_int, [hid_t])]:
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    lib.f64 = hid_t.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
    _lib = lib
    return lib


def _ok(v, what):
    if v < 0:
        raise RuntimeError(f"HDF5: {what} failed")
    return v


def write_datasets(path, group, arrays):
    """create `path` with one 1-D float64 dataset /<group>/<name> per entry of `arrays` (name -> array)"""
    lib = _load()
    f = _ok(lib.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"H5Fcreate({path})")
    try:
        g = _ok(lib.H5Gcreate2(f, ("/" + group.strip("/")).encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2")
        try:
            for name, arr in arrays.items():
                a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
                dims = (C.c_uint64 * 1)(a.size)
                sp = _ok(lib.H5Screate_simple(1, dims, None), "H5Screate_simple")
                d = _ok(lib.H5Dcreate2(g, name.encode(), lib.f64, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({name})")
                _ok(lib.H5Dwrite(d, lib.f64, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), "H5Dwrite")
                lib.H5Dclose(d)
                lib.H5Sclose(sp)
        finally:
            lib.H5Gclose(g)
    finally:
        lib.H5Fclose(f)


def read_dataset(path, name):
    """1-D float64 dataset `name` (e.g. '/fields/velocity') of `path`"""
    lib = _load()
    f = _ok(lib.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), f"H5Fopen({path})")
    try:
        d = _ok(lib.H5Dopen2(f, name.encode(), H5P_DEFAULT), f"H5Dopen2({name})")
        sp = lib.H5Dget_space(d)
        n = lib.H5Sget_simple_extent_npoints(sp)
        out = np.empty(n, dtype=np.float64)
        _ok(lib.H5Dread(d, lib.f64, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)), "H5Dread")
        lib.H5Sclose(sp)
        lib.H5Dclose(d)
        return out
    finally:
        lib.H5Fclose(f)
