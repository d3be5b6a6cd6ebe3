"""Stand-in for the three names the reference imports from acados_template (controller.py:15):
AcadosModel, AcadosOcp, AcadosOcpSolver.  Written from scratch; see tests/levelb/README.md.

AcadosOcpSolver understands exactly the OCP that controller.py:175-264 builds (LINEAR_LS cost with
y = [x;u], input box bounds, pinned initial state, ERK 2x2, Gauss-Newton, LM) and refuses anything
else loudly.  The physical constants are recovered by probing the traced dynamics."""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np

NX, NU = 13, 4


class AcadosModel:
    def __init__(self):
        self.name = None
        self.x = self.u = self.xdot = self.z = self.p = None
        self.f_expl_expr = self.f_impl_expr = None


class AcadosOcp:
    def __init__(self):
        self.model = None
        self.dims = SimpleNamespace(N=None)
        self.solver_options = SimpleNamespace(
            tf=None, qp_solver=None, hessian_approx=None, integrator_type=None, qp_solver_cond_N=None,
            qp_solver_iter_max=None, collocation_type=None, sim_method_num_stages=4, sim_method_num_steps=1,
            regularize_method=None, levenberg_marquardt=0.0, nlp_solver_type="SQP_RTI",
            qp_solver_tol_stat=None, qp_solver_tol_eq=None, qp_solver_tol_ineq=None, qp_solver_tol_comp=None)
        self.cost = SimpleNamespace(cost_type=None, cost_type_e=None, Vx=None, Vu=None, Vx_e=None, W=None,
                                    W_e=None, yref=None, yref_e=None)
        self.constraints = SimpleNamespace(idxbu=None, lbu=None, ubu=None, idxbx_0=None, lbx_0=None, ubx_0=None,
                                           idxbx=None, lbx=None, ubx=None)
        self.code_export_directory = None
        self.json_file = None


def probe_dynamics(model):
    """Recover (mass, gravity, inertia up to a common scale, rotor geometry) from the traced f_expl."""
    def F(x, u):
        env = {f"x_{i}": float(x[i]) for i in range(NX)}
        env.update({f"u_{i}": float(u[i]) for i in range(NU)})
        return np.asarray(model.f_expl_expr.eval(env), dtype=float)
    xh = np.zeros(NX); xh[6] = 1.0
    g = -F(xh, np.zeros(NU))[5]
    inv_m = F(xh, np.eye(NU)[0])[5] + g
    abc = np.array([F(xh, np.eye(NU)[i])[10:13] for i in range(NU)])          # [ry/Jx, -rx/Jy, rz/Jz]
    def gyro(w):
        x = xh.copy(); x[10:13] = w
        return F(x, np.zeros(NU))[10:13]
    kx, ky, kz = -gyro([0, 1, 1])[0], -gyro([1, 0, 1])[1], -gyro([1, 1, 0])[2]
    Jx = 1.0
    Jy = (1.0 - kx) / (1.0 + ky)
    Jz = Jy + kx
    if abs((Jy - Jx) / Jz - kz) > 1e-9:
        raise NotImplementedError("dynamics are not the diagonal-inertia rigid body of controller.py:331-341")
    const = dict(mass=1.0 / inv_m, gravity=g, J=[Jx, Jy, Jz], rotor_y=list(abc[:, 0] * Jx),
                 rotor_x=list(-abc[:, 1] * Jy), rotor_z=list(abc[:, 2] * Jz))
    # the recovered constants must reproduce the traced function at random points
    rng = np.random.default_rng(0)
    for _ in range(5):
        x = rng.normal(size=NX); u = rng.uniform(0, 6, NU)
        if not np.allclose(F(x, u), _model_f(const, x, u), rtol=1e-10, atol=1e-10):
            raise NotImplementedError("traced dynamics do not match the rotor-level quadrotor model")
    return const


def _model_f(k, x, u):
    qw, qx, qy, qz = x[6:10]; wx, wy, wz = x[10:13]
    T = u.sum() / k["mass"]; Jx, Jy, Jz = k["J"]
    tx, ty, tz = np.dot(u, k["rotor_y"]), -np.dot(u, k["rotor_x"]), np.dot(u, k["rotor_z"])
    return np.array([x[3], x[4], x[5], 2 * (qx * qz + qw * qy) * T, 2 * (qy * qz - qw * qx) * T,
                     (1 - 2 * (qx * qx + qy * qy)) * T - k["gravity"],
                     0.5 * (-qx * wx - qy * wy - qz * wz), 0.5 * (qw * wx + qy * wz - qz * wy),
                     0.5 * (qw * wy + qz * wx - qx * wz), 0.5 * (qw * wz + qx * wy - qy * wx),
                     (tx - (Jz - Jy) * wy * wz) / Jx, (ty - (Jx - Jz) * wz * wx) / Jy, (tz - (Jy - Jx) * wx * wy) / Jz])


def _need(cond, what):
    if not cond:
        raise NotImplementedError(f"Level-B shim only covers the OCP of controller.py:175-264 ({what})")


class AcadosOcpSolver:
    def __init__(self, ocp, json_file=None, **_):
        so, cost, con = ocp.solver_options, ocp.cost, ocp.constraints
        N = int(ocp.dims.N)
        _need(cost.cost_type == "LINEAR_LS" and cost.cost_type_e == "LINEAR_LS", "LINEAR_LS cost")
        Vx, Vu = np.asarray(cost.Vx), np.asarray(cost.Vu)
        _need(np.array_equal(Vx, np.vstack([np.eye(NX), np.zeros((NU, NX))])) and
              np.array_equal(Vu, np.vstack([np.zeros((NX, NU)), np.eye(NU)])) and
              np.array_equal(np.asarray(cost.Vx_e), np.eye(NX)), "y = [x;u]")
        W, We = np.asarray(cost.W), np.asarray(cost.W_e)
        _need(np.array_equal(W, np.diag(np.diag(W))) and np.array_equal(We, np.diag(np.diag(We))), "diagonal weights")
        _need(so.integrator_type == "ERK" and so.sim_method_num_stages == 2, "ERK with 2 stages")
        _need(so.hessian_approx == "GAUSS_NEWTON" and so.nlp_solver_type == "SQP_RTI", "Gauss-Newton SQP_RTI")
        _need(list(np.asarray(con.idxbu)) == list(range(NU)) and list(np.asarray(con.idxbx_0)) == list(range(NX)),
              "input box + pinned initial state")
        _need(np.all(np.asarray(con.lbx) <= -1e5) and np.all(np.asarray(con.ubx) >= 1e5), "inactive state box")
        k = probe_dynamics(ocp.model)
        self.N, self.dt = N, float(so.tf) / N
        self._cfg = dict(N=N, dt=self.dt, W=np.diag(W), W_e=np.diag(We), lbu=np.asarray(con.lbu, float),
                         ubu=np.asarray(con.ubu, float), lm=float(so.levenberg_marquardt),
                         steps=int(so.sim_method_num_steps), iter_max=int(so.qp_solver_iter_max),
                         cond_N=int(so.qp_solver_cond_N), **k)
        # QP exit.  Default of this stand-in: the oracle's own (exact) solve - what the committed Level-B fixtures were made with.
        # LEVELB_HPIPM_EXIT=1 (the generator's dry run, tests/test_acados_golden.py): stop the interior point where HPIPM would -
        # residuals <= the four qp_solver_tol_* (1e-8 each where the OCP leaves them unset, [UPSTREAM U9]) - so that a "default
        # tolerance" golden set differs from a "tight" one the way acados' will
        tol = [getattr(so, "qp_solver_tol_" + n, None) for n in ("stat", "eq", "ineq", "comp")]
        self._hpipm_exit = None
        if os.environ.get("LEVELB_HPIPM_EXIT") == "1":
            t = [1e-8 if v is None else float(v) for v in tol]
            self._hpipm_exit = (max(t[0], t[1], t[2]), t[3])
        self._last_stats = None
        self._x = np.zeros((N + 1, NX)); self._u = np.zeros((N, NU))
        self._yref = np.tile(np.asarray(cost.yref, float), (N, 1)); self._yref_e = np.asarray(cost.yref_e, float).copy()
        self._x0 = np.asarray(con.lbx_0, float).copy()
        self.backend = os.environ.get("LEVELB_BACKEND", "oracle")
        self._make_backend()

    def _make_backend(self):
        c = self._cfg
        if self.backend == "oracle":
            from oracle import oracle as O
            self._O = O
            self._oc = O.default_config(N=c["N"], dt=c["dt"], W=c["W"], We=c["W_e"], lbu=c["lbu"], ubu=c["ubu"],
                                        lm=c["lm"], mass=c["mass"], gravity=c["gravity"], J=c["J"],
                                        rotor_x=c["rotor_x"], rotor_y=c["rotor_y"], rotor_z=c["rotor_z"],
                                        sim_num_steps=c["steps"], qp_iter_max=c["iter_max"], qp_gamma=0.0)
            if self._hpipm_exit is not None:
                self._oc.qp_exit_mode = 1
                self._oc.qp_tol_stat, self._oc.qp_tol_comp = self._hpipm_exit
        elif self.backend == "hip":
            from rotors_mpc_controller_amd import _lib
            from rotors_mpc_controller_amd.solver import NmpcOcpSolver
            cfg = _lib.default_config(N=c["N"], dt=c["dt"], W=c["W"], W_e=c["W_e"], lbu=c["lbu"], ubu=c["ubu"],
                                      levenberg_marquardt=c["lm"], mass=c["mass"], gravity=c["gravity"],
                                      inertia=c["J"], rotor_x=c["rotor_x"], rotor_y=c["rotor_y"], rotor_z=c["rotor_z"],
                                      sim_num_steps=c["steps"], qp_iter_max=c["iter_max"], max_batch=1)
            self._hip = NmpcOcpSolver(cfg)
        else:
            raise ValueError(self.backend)

    def set(self, stage, field, value):
        v = np.asarray(value, dtype=float).reshape(-1)
        if field == "x":
            self._x[stage] = v
        elif field == "u":
            self._u[stage] = v
        elif field == "yref":
            if stage < self.N:
                self._yref[stage] = v
            else:
                self._yref_e[:] = v
        elif field in ("lbx", "ubx"):
            _need(stage == 0, "state bounds at stage 0 only")
            self._x0[:] = v
        else:
            raise NotImplementedError(field)

    def get(self, stage, field):
        return (self._x if field == "x" else self._u)[stage].copy()

    def get_stats(self, field):
        if field == "qp_iter" and self._last_stats is not None:
            return np.array([self._last_stats.qp_iter])
        raise NotImplementedError(field)

    def get_residuals(self):
        """(res_stat, res_eq, res_ineq, res_comp) of the QP the last solve ended on"""
        st = self._last_stats
        if st is None:
            raise NotImplementedError("no solve yet")
        return np.array([st.res_stat, st.res_eq, 0.0, st.res_comp])

    def solve(self):
        if self.backend == "oracle":
            xt = self._x.copy(); xt[0] = self._x0
            s, xn, un, self._last_stats = self._O.sqp_rti(self._oc, self._x0, self._yref, self._yref_e, xt, self._u)
        else:
            out = self._hip.solve_batch(self._x0[None], self._yref, self._yref_e, x_init=self._x[None],
                                        u_init=self._u[None], want_traj=True)
            s, xn, un = int(out["status"][0]), out["x"][0], out["u"][0]
        if s == 0:
            self._x, self._u = xn, un
        return int(s)
