"""ctypes binding of the CPU oracle (oracle/nmpc_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from rotors_mpc_controller_amd (the product).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
NX, NU, NY = 13, 4, 17


class OrcConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("dt", C.c_double),
        ("W", C.c_double * NY), ("We", C.c_double * NX),
        ("lbu", C.c_double * NU), ("ubu", C.c_double * NU),
        ("lm", C.c_double), ("lm_scaled_by_dt", C.c_int), ("cost_scaled_by_dt", C.c_int),
        ("mass", C.c_double), ("gravity", C.c_double), ("J", C.c_double * 3),
        ("rotor_x", C.c_double * NU), ("rotor_y", C.c_double * NU), ("rotor_z", C.c_double * NU),
        ("sim_num_stages", C.c_int), ("sim_num_steps", C.c_int),
        ("qp_iter_max", C.c_int), ("qp_cond_N", C.c_int),
        ("qp_tol_comp", C.c_double), ("qp_tol_stat", C.c_double),
        ("qp_mu0", C.c_double), ("qp_tau", C.c_double), ("qp_thr0", C.c_double),
        ("qp_thr0_rel", C.c_double), ("qp_gamma", C.c_double),
        ("qp_polish", C.c_int), ("qp_polish_mu", C.c_double), ("qp_polish_passes", C.c_int), ("qp_polish_budget", C.c_int),
        ("qp_growth_max", C.c_double), ("qp_acc_comp", C.c_double), ("qp_acc_stat", C.c_double), ("qp_tol_step", C.c_double),
        ("qp_maxiter_status", C.c_int), ("qp_warm_start", C.c_int), ("qp_exit_mode", C.c_int), ("qp_bound_res", C.c_int),
    ]


class OrcStats(C.Structure):
    _fields_ = [
        ("qp_iter", C.c_int), ("qp_status", C.c_int),
        ("res_stat", C.c_double), ("res_eq", C.c_double), ("res_comp", C.c_double),
        ("mu", C.c_double), ("rho", C.c_double), ("hess_projected", C.c_int),
        ("polished", C.c_int), ("polish_attempts", C.c_int),
        ("growth", C.c_double), ("step_last", C.c_double), ("untrusted", C.c_int),
    ]


def build(force: bool = False) -> Path:
    # NMPC_SANITIZE=1: the AddressSanitizer + UBSan build (the process must have libasan preloaded,
    # tools/run_sanitized_tests.sh)
    san = os.environ.get("NMPC_SANITIZE") == "1"
    so = _HERE / ("libnmpc_oracle_asan.so" if san else "libnmpc_oracle.so")
    src = _HERE / "nmpc_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B" if force else "-s"] + (["asan"] if san else []),
                              stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        cp = C.POINTER(OrcConfig)
        _lib.orc_default_config.argtypes = [cp]
        _lib.orc_model_f.argtypes = [cp, dp, dp, dp]
        _lib.orc_model_jac.argtypes = [cp, dp, dp, dp, dp]
        _lib.orc_vde_forw.argtypes = [cp, dp, dp, dp, dp, dp, dp, dp]
        _lib.orc_vde_adj.argtypes = [cp, dp, dp, dp, dp]
        _lib.orc_integrate.argtypes = [cp, dp, dp, dp, dp, dp]
        _lib.orc_linearize.argtypes = [cp] + [dp] * 13 + [ip]
        _lib.orc_qp_solve.argtypes = [cp] + [dp] * 12 + [C.POINTER(OrcStats)]
        _lib.orc_qp_solve.restype = C.c_int
        _lib.orc_sqp_rti.argtypes = [cp, dp, dp, dp, dp, dp, C.POINTER(OrcStats)]
        _lib.orc_sqp_rti.restype = C.c_int
        _lib.orc_solve_batch.argtypes = [cp, C.c_int, dp, dp, dp, C.c_int, dp, dp, dp, ip, dp, dp,
                                         ip, C.c_int]
        _lib.orc_solve_batch.restype = C.c_int
        _lib.orc_solve_batch_ex.argtypes = [cp, C.c_int, dp, dp, dp, C.c_int, dp, dp, dp, ip, dp, dp, ip, ip, dp, C.c_int]
        _lib.orc_solve_batch_ex.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def default_config(**over) -> OrcConfig:
    c = OrcConfig()
    lib().orc_default_config(C.byref(c))
    for k, v in over.items():
        cur = getattr(c, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(c, k, v)
    return c


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def model_f(c, x, u):
    x, u = f64(x), f64(u)
    f = np.zeros(NX)
    lib().orc_model_f(C.byref(c), _p(x), _p(u), _p(f))
    return f


def model_jac(c, x, u):
    x, u = f64(x), f64(u)
    fx, fu = np.zeros((NX, NX)), np.zeros((NX, NU))
    lib().orc_model_jac(C.byref(c), _p(x), _p(u), _p(fx), _p(fu))
    return fx, fu


def vde_forw(c, x, Sx, Su, u):
    x, Sx, Su, u = f64(x), f64(Sx), f64(Su), f64(u)
    xd, Sxd, Sud = np.zeros(NX), np.zeros((NX, NX)), np.zeros((NX, NU))
    lib().orc_vde_forw(C.byref(c), _p(x), _p(Sx), _p(Su), _p(u), _p(xd), _p(Sxd), _p(Sud))
    return xd, Sxd, Sud


def vde_adj(c, x, lam, u):
    x, lam, u = f64(x), f64(lam), f64(u)
    adj = np.zeros(NX + NU)
    lib().orc_vde_adj(C.byref(c), _p(x), _p(lam), _p(u), _p(adj))
    return adj


def integrate(c, x, u):
    x, u = f64(x), f64(u)
    xn, A, B = np.zeros(NX), np.zeros((NX, NX)), np.zeros((NX, NU))
    lib().orc_integrate(C.byref(c), _p(x), _p(u), _p(xn), _p(A), _p(B))
    return xn, A, B


def linearize(c, xtraj, utraj, yref, yref_e):
    N = c.N
    xtraj, utraj, yref, yref_e = f64(xtraj), f64(utraj), f64(yref), f64(yref_e)
    out = dict(A=np.zeros((N, NX, NX)), B=np.zeros((N, NX, NU)), b=np.zeros((N, NX)),
               q=np.zeros((N + 1, NX)), r=np.zeros((N, NU)), lo=np.zeros((N, NU)),
               hi=np.zeros((N, NU)), Qd=np.zeros((N + 1, NX)), Rd=np.zeros((N, NU)))
    proj = C.c_int(0)
    lib().orc_linearize(C.byref(c), _p(xtraj), _p(utraj), _p(yref), _p(yref_e),
                        *[_p(out[k]) for k in ("A", "B", "b", "q", "r", "lo", "hi", "Qd", "Rd")],
                        C.byref(proj))
    out["hess_projected"] = proj.value
    return out


def qp_solve(c, qp, dx0=None):
    N = c.N
    dx0 = np.zeros(NX) if dx0 is None else f64(dx0)
    dx, du = np.zeros((N + 1, NX)), np.zeros((N, NU))
    st = OrcStats()
    s = lib().orc_qp_solve(C.byref(c), _p(dx0),
                           *[_p(f64(qp[k])) for k in ("A", "B", "b", "q", "r", "lo", "hi", "Qd", "Rd")],
                           _p(dx), _p(du), C.byref(st))
    return s, dx, du, st


def sqp_rti(c, x0, yref, yref_e, xtraj, utraj):
    """Returns (status, xtraj_new, utraj_new, stats); inputs are not modified."""
    x0, yref, yref_e = f64(x0), f64(yref), f64(yref_e)
    xt, ut = f64(xtraj).copy(), f64(utraj).copy()
    st = OrcStats()
    s = lib().orc_sqp_rti(C.byref(c), _p(x0), _p(yref), _p(yref_e), _p(xt), _p(ut), C.byref(st))
    return s, xt, ut, st


def solve_batch(c, x0, yref, yref_e, x_init=None, u_init=None, want_traj=False, nthreads=0):
    """Batched cold/warm-started RTI.  yref [N,17] (broadcast) or [B,N,17]."""
    x0 = f64(x0)
    B, N = x0.shape[0], c.N
    yref, yref_e = f64(yref), f64(yref_e)
    bcast = 1 if yref.ndim == 2 else 0
    u0 = np.zeros((B, NU))
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    xo = np.zeros((B, N + 1, NX)) if want_traj else None
    uo = np.zeros((B, N, NU)) if want_traj else None
    xi = None if x_init is None else f64(x_init)
    ui = None if u_init is None else f64(u_init)
    passes = np.zeros(B, dtype=np.int32)
    growth = np.zeros(B)
    lib().orc_solve_batch_ex(C.byref(c), B, _p(x0), _p(yref), _p(yref_e), bcast, _p(xi), _p(ui),
                             _p(u0), _ip(status), _p(xo), _p(uo), _ip(iters), _ip(passes), _p(growth), int(nthreads))
    return dict(u0=u0, status=status, iters=iters, x=xo, u=uo, passes=passes, growth=growth)


def hover_yref(c, pos=(0.0, 0.0, 1.0), yaw=0.0):
    """Hover reference as reference.py:75-91 + node:52 produce it (thrust = m g / 4)."""
    y = np.zeros(NY)
    y[0:3] = pos
    y[6] = np.cos(0.5 * yaw)
    y[9] = np.sin(0.5 * yaw)
    y[13:17] = c.mass * c.gravity / 4.0
    return np.tile(y, (c.N, 1)), y[:NX].copy()


if __name__ == "__main__":
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    c = default_config()
    yref, yref_e = hover_yref(c)
    x0 = yref_e.copy()
    x0[2] = 0.5
    out = solve_batch(c, x0[None, :], yref, yref_e)
    print(out)
