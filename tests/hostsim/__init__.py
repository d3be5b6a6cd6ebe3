"""Test-only host build of the per-lane kernel bodies (see hostsim.cpp).  Not the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from rotors_mpc_controller_amd._lib import NmpcConfig

_HERE = Path(__file__).resolve().parent
_ROOT = _HERE.parent.parent
_lib = None


def lib():
    global _lib
    if _lib is None:
        san = os.environ.get("NMPC_SANITIZE") == "1"     # ASan + UBSan build of the kernel bodies (CPU only)
        so = _HERE / ("libnmpc_hostsim_asan.so" if san else "libnmpc_hostsim.so")
        deps = [_HERE / "hostsim.cpp"] + sorted((_ROOT / "rotors_mpc_controller_amd" / "csrc").glob("*.hpp"))
        if not so.exists() or any(d.stat().st_mtime > so.stat().st_mtime for d in deps):
            flags = (["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
                     if san else ["-O2", "-fopenmp"])
            subprocess.check_call(["g++"] + flags + ["-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                                   "-o", str(so), str(_HERE / "hostsim.cpp")])
        _lib = C.CDLL(str(so))
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        _lib.hostsim_solve_batch.argtypes = [C.POINTER(NmpcConfig), C.c_int, dp, dp, dp, C.c_int, dp, dp,
                                             dp, ip, dp, dp, ip]
        _lib.hostsim_solve_batch.restype = C.c_int
        _lib.hostsim_set_threads.argtypes = [C.c_int]
        _lib.hostsim_set_threads.restype = None
        _lib.hostsim_adjoint.argtypes = [C.POINTER(NmpcConfig), C.c_int, dp, dp, dp, dp, C.c_int]
        _lib.hostsim_adjoint.restype = C.c_int
    return _lib


def _p(a, t=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def solve_batch(cfg: NmpcConfig, x0, yref, yref_e, x_init=None, u_init=None, nthreads: int = 0):
    """nthreads > 0: OpenMP threads over the instances (default: the OpenMP runtime's choice)."""
    if nthreads > 0:
        lib().hostsim_set_threads(int(nthreads))
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    yref = np.ascontiguousarray(yref, dtype=np.float64)
    yref_e = np.ascontiguousarray(yref_e, dtype=np.float64)
    B, N = x0.shape[0], cfg.N
    bcast = 1 if yref.ndim == 2 else 0
    xi = None if x_init is None else np.ascontiguousarray(x_init, dtype=np.float64)
    ui = None if u_init is None else np.ascontiguousarray(u_init, dtype=np.float64)
    u0 = np.zeros((B, 4)); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
    xo = np.zeros((B, N + 1, 13)); uo = np.zeros((B, N, 4))
    lib().hostsim_solve_batch(C.byref(cfg), B, _p(x0), _p(yref), _p(yref_e), bcast, _p(xi), _p(ui),
                              _p(u0), _p(st, C.c_int32), _p(xo), _p(uo), _p(it, C.c_int32))
    return dict(u0=u0, status=st, iters=it, x=xo, u=uo)


def adjoint(cfg: NmpcConfig, x, u, lam, continuous: bool = False):
    """(A' lam | B' lam) per instance from the host build of model_adj / erk_adjoint (nmpc_lane.hpp)."""
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    out = np.zeros((x.shape[0], 17))
    lib().hostsim_adjoint(C.byref(cfg), x.shape[0], _p(x), _p(u), _p(lam), _p(out), int(continuous))
    return out
