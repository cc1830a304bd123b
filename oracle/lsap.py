"""ctypes wrapper of oracle/lsap.c (the restated scipy.optimize.linear_sum_assignment).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "liblsap_oracle.so")
_lib = None


def build():
    src = os.path.join(_DIR, "lsap.c")
    stale = lambda: not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if stale():
        import fcntl

        with open(os.path.join(_DIR, ".build.lock"), "w") as lock:  # the N ranks of a multi-GPU bench all pass through build()
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if stale():
                    subprocess.run(["make", "-C", _DIR, "-s"], check=True)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return _SO


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        lib.dfine_oracle_lsap.restype = ctypes.c_int
        lib.dfine_oracle_lsap.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        _lib = lib
    return _lib


def linear_sum_assignment(cost):
    """Same contract as scipy.optimize.linear_sum_assignment(cost) (minimisation)."""
    cost = np.ascontiguousarray(np.asarray(cost, dtype=np.float64))
    nr, nc = cost.shape
    n = min(nr, nc)
    rows = np.empty(n, dtype=np.int64)
    cols = np.empty(n, dtype=np.int64)
    rc = _load().dfine_oracle_lsap(cost.ctypes.data, nr, nc, rows.ctypes.data, cols.ctypes.data)
    if rc == -1:
        raise ValueError("cost matrix is infeasible")
    if rc != 0:
        raise MemoryError("oracle lsap failed")
    return rows, cols
