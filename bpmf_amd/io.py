"""Matrix file IO through the library's C ABI (include/bpmf_io.h): the same readers / writers
the `bpmf` executable uses.  Formats by extension (.mtx .mm .sdm .sbm .ddm .csv, optional .gz)."""
import ctypes as C

import numpy as np

from . import _lib


class BpmfIoError(IOError):
    pass


def _check(lib, rc):
    if rc:
        raise BpmfIoError(lib.bpmf_io_last_error().decode("utf-8", "replace"))


def read_sparse(path):
    """Returns (nrows, ncols, (colptr int64, rowidx int32, vals f64)): CSC, rows sorted, duplicates summed."""
    lib = _lib.load_library()
    nr, nc, nnz = C.c_int64(), C.c_int64(), C.c_int64()
    cp, ri, va = _lib.c_i64p(), _lib.c_i32p(), _lib.c_f64p()
    _check(lib, lib.bpmf_io_read_sparse(str(path).encode(), C.byref(nr), C.byref(nc), C.byref(nnz), C.byref(cp), C.byref(ri), C.byref(va)))
    try:
        colptr = np.ctypeslib.as_array(cp, shape=(nc.value + 1,)).copy()
        rowidx = np.ctypeslib.as_array(ri, shape=(max(nnz.value, 1),))[:nnz.value].copy()
        vals = np.ctypeslib.as_array(va, shape=(max(nnz.value, 1),))[:nnz.value].copy()
    finally:
        for p in (cp, ri, va):
            lib.bpmf_io_free(p)
    return nr.value, nc.value, (colptr, rowidx, vals)


def write_sparse(path, nrows, ncols, csc):
    lib = _lib.load_library()
    colptr = np.ascontiguousarray(csc[0], np.int64); rowidx = np.ascontiguousarray(csc[1], np.int32); vals = np.ascontiguousarray(csc[2], np.float64)
    _check(lib, lib.bpmf_io_write_sparse(str(path).encode(), int(nrows), int(ncols), colptr.ctypes.data, rowidx.ctypes.data, vals.ctypes.data))


def read_dense(path):
    """Returns an [nrows, ncols] array (the file is column-major)."""
    lib = _lib.load_library()
    nr, nc = C.c_int64(), C.c_int64()
    d = _lib.c_f64p()
    _check(lib, lib.bpmf_io_read_dense(str(path).encode(), C.byref(nr), C.byref(nc), C.byref(d)))
    try:
        n = nr.value * nc.value
        a = np.ctypeslib.as_array(d, shape=(max(n, 1),))[:n].copy()
    finally:
        lib.bpmf_io_free(d)
    return a.reshape((nc.value, nr.value)).T.copy()


def write_dense(path, a):
    lib = _lib.load_library()
    a = np.asarray(a, np.float64)
    cm = np.ascontiguousarray(a.T)                   # column-major bytes
    _check(lib, lib.bpmf_io_write_dense(str(path).encode(), a.shape[0], a.shape[1], cm.ctypes.data))
