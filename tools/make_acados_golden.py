#!/usr/bin/env python3
"""acados golden harness: the hook SURVEY.md 8(c) asks for ("hook for a future real oracle").

Run on ANY machine where `acados_template` and `casadi` import (this image has neither and no
network, so it has never run here -- parity with acados stays "unpinned" until it has):

    python tools/make_acados_golden.py            # writes tests/golden/acados_rti.npz
    python -m pytest tests/test_acados_golden.py  # then compares the oracle and, with a GPU, the HIP path

What it does -- with code of this repository, not the reference's files: it constructs the optimal control
problem that /root/reference/src/rotors_mpc_controller/controller.py:175-264 hands to acados (model of
:267-355, all solver options of :179-190, cost :223-245, constraints :248-261) from this package's own
parameter derivation (`derive_params` on the shipped defaults), runs ONE cold-start SQP real-time iteration per
fixture state exactly as PositionNMPC.solve stages it (:412-447: lbx/ubx/x at stage 0, x_k = x0 and u_k = 0 on
every stage, yref on stages 0..N), and records u0, status, the full trajectories, the QP iteration count,
solve times and the acados / casadi versions.  Inputs: tests/golden/rti_cold_start.npz (x0, yref, yref_e:
the committed fixture set) plus the two known-answer states K2 / K3 of SURVEY 8(c).

TWO solution sets per state (round 5), because two different things are to be learnt from acados:
  * `u0`, `x`, `u`, `status`, `qp_iter`          the reference's options VERBATIM (controller.py:179-190 set no QP tolerance, so
                                                 HPIPM stops on its default residual tolerances, ~1e-8 [UPSTREAM U9]): what the
                                                 reference flies.  Its distance from the exact QP solution is acados' own
                                                 accuracy floor on this OCP - predicted at 3e-7 N on this fixture set and up to
                                                 8e-6 N on the bench sample by tests/test_acados_floor.py.
  * `u0_tight`, `x_tight`, `u_tight`, ...        the same OCP with HPIPM's four exit tolerances at 1e-12
                                                 (qp_solver_tol_stat / _eq / _ineq / _comp): the QP solved to rounding.  THIS set
                                                 decides the version-dependent conventions (U4, U5, U7) and is what the oracle
                                                 and the HIP path are held to at 1e-6 relative - convention and convergence
                                                 are then separate questions.
Plus, where the installed acados exposes them, the SQP residuals after the step (`residuals*`: res_stat, res_eq, res_ineq,
res_comp) of both runs.

The three acados-version-dependent conventions (U4 stage cost x dt, U5 Levenberg-Marquardt x dt, U7 x0 handling)
are whatever the installed acados does: the consumer test reports which of this repository's switch settings
(lm_scaled_by_dt, cost_scaled_by_dt) reproduces the golden commands.
"""
from __future__ import annotations

import argparse
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def build_model(p):
    """The 13-state / 4-input rigid-body model (controller.py:267-355) in this repository's own formulation:
    thrust along the body z axis rotated by the (non-normalised) quaternion, quaternion kinematics, Euler's
    equation with a diagonal inertia, rotor geometry as three torque rows."""
    import casadi as ca
    from acados_template import AcadosModel
    x = ca.SX.sym("x", 13)
    u = ca.SX.sym("u", 4)
    xdot = ca.SX.sym("xdot", 13)
    # scalar arithmetic only (vertcat / dot / sum1 / DM): the subset every CasADi version has
    qw, qx, qy, qz = x[6], x[7], x[8], x[9]
    w0, w1, w2 = x[10], x[11], x[12]
    thrust_acc = ca.sum1(u) / float(p.mass)
    # R(q) e3 * T/m - g e3  (third column of the rotation matrix of the non-normalised quaternion)
    acc = ca.vertcat(2 * (qx * qz + qw * qy) * thrust_acc,
                     2 * (qy * qz - qw * qx) * thrust_acc,
                     (1 - 2 * (qx * qx + qy * qy)) * thrust_acc - float(p.gravity))
    # qdot = 1/2 Omega(w) q
    qdot = ca.vertcat(0.5 * (-w0 * qx - w1 * qy - w2 * qz),
                      0.5 * (w0 * qw + w2 * qy - w1 * qz),
                      0.5 * (w1 * qw - w2 * qx + w0 * qz),
                      0.5 * (w2 * qw + w1 * qx - w0 * qy))
    # rotor geometry as three torque rows; Euler's equation with a diagonal inertia
    tau = [ca.dot(ca.DM(np.asarray(p.rotor_y_offsets, float)), u),
           ca.dot(ca.DM(-np.asarray(p.rotor_x_offsets, float)), u),
           ca.dot(ca.DM(np.asarray(p.rotor_z_torque, float)), u)]
    J = [float(t) for t in p.inertia]
    wdot = ca.vertcat((tau[0] - (J[2] - J[1]) * w1 * w2) / J[0],
                      (tau[1] - (J[0] - J[2]) * w2 * w0) / J[1],
                      (tau[2] - (J[1] - J[0]) * w0 * w1) / J[2])
    f = ca.vertcat(x[3], x[4], x[5], acc, qdot, wdot)
    m = AcadosModel()
    m.name = "rotors_nmpc_golden"
    m.x, m.u, m.xdot = x, u, xdot
    m.f_expl_expr = f
    m.f_impl_expr = xdot - f
    m.z = ca.SX.sym("z", 0)
    m.p = ca.SX.sym("p", 0)
    return m


def build_solver(p, workdir: Path, qp_iter_max=None, qp_tol=None):
    """qp_iter_max: override of solver_iter_max (the U10 probe: what status does this acados return when the QP hits its cap?)
    qp_tol: HPIPM's four exit tolerances (None: untouched - the reference sets none, controller.py:179-190)"""
    from acados_template import AcadosOcp, AcadosOcpSolver
    N = int(p.horizon_steps)
    ocp = AcadosOcp()
    ocp.model = build_model(p)
    ocp.dims.N = N
    so = ocp.solver_options
    so.tf = N * float(p.dt)                                  # controller.py:180
    so.qp_solver = "PARTIAL_CONDENSING_HPIPM"                # :181
    so.hessian_approx = "GAUSS_NEWTON"                       # :182
    so.integrator_type = "ERK"                               # :183
    so.qp_solver_cond_N = min(N, 5)                          # :184
    so.qp_solver_iter_max = int(p.solver_iter_max if qp_iter_max is None else qp_iter_max)           # :185
    so.collocation_type = "GAUSS_RADAU_IIA"                  # :186 (no effect on ERK)
    so.sim_method_num_stages = 2                             # :187
    so.sim_method_num_steps = 2                              # :188
    so.regularize_method = "PROJECT_REDUC_HESS"              # :189
    so.levenberg_marquardt = float(p.regularization)         # :190
    if qp_tol is not None:                                   # NOT in the reference: the tight-tolerance golden set only
        so.qp_solver_tol_stat = so.qp_solver_tol_eq = so.qp_solver_tol_ineq = so.qp_solver_tol_comp = float(qp_tol)
    # nlp_solver_type is NOT set by the reference: the acados default applies (recorded below)
    ocp.model.name = "rotors_nmpc_golden" + ("" if qp_tol is None else "_tight") + ("" if qp_iter_max is None else "_cap")
    ocp.code_export_directory = str(workdir / "c_generated_code")
    ocp.cost.cost_type = "LINEAR_LS"
    ocp.cost.cost_type_e = "LINEAR_LS"
    ocp.cost.Vx = np.vstack([np.eye(13), np.zeros((4, 13))])
    ocp.cost.Vu = np.vstack([np.zeros((13, 4)), np.eye(4)])
    ocp.cost.Vx_e = np.eye(13)
    ocp.cost.W = np.diag(np.concatenate([p.position_weight, p.velocity_weight, p.quaternion_weight, p.rate_weight,
                                         p.control_weight]))
    ocp.cost.W_e = np.diag(p.terminal_weight)
    ocp.cost.yref = np.zeros(17)
    ocp.cost.yref_e = np.zeros(13)
    ocp.constraints.idxbu = np.arange(4)
    ocp.constraints.lbu = np.asarray(p.input_lower_bounds, float)
    ocp.constraints.ubu = np.asarray(p.input_upper_bounds, float)
    ocp.constraints.idxbx_0 = np.arange(13)
    ocp.constraints.lbx_0 = np.zeros(13)
    ocp.constraints.ubx_0 = np.zeros(13)
    ocp.constraints.idxbx = np.arange(13)
    ocp.constraints.lbx = -1e6 * np.ones(13)
    ocp.constraints.ubx = 1e6 * np.ones(13)
    solver = AcadosOcpSolver(ocp, json_file=str(workdir / "rotors_nmpc_golden.json"))
    return ocp, solver


def cold_start_rti(solver, N, x0, yref, yref_e):
    """controller.py:412-447 for one instance; returns (u0, status, x[N+1,13], u[N,4], seconds of solve())."""
    solver.set(0, "lbx", x0)
    solver.set(0, "ubx", x0)
    solver.set(0, "x", x0)
    for k in range(N):
        if k > 0:
            solver.set(k, "x", x0)
        solver.set(k, "u", np.zeros(4))
    solver.set(N, "x", x0)
    for k in range(N):
        solver.set(k, "yref", yref[k])
    solver.set(N, "yref", yref_e)
    t = time.perf_counter()
    status = solver.solve()
    dt = time.perf_counter() - t
    xs = np.stack([solver.get(k, "x") for k in range(N + 1)])
    us = np.stack([solver.get(k, "u") for k in range(N)])
    return us[0].copy(), int(status), xs, us, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden" / "acados_rti.npz"))
    args = ap.parse_args()
    try:
        import acados_template
        import casadi
    except ImportError as e:
        sys.exit(f"make_acados_golden.py needs acados_template and casadi ({e}); nothing written")
    from rotors_mpc_controller_amd.controller import derive_params
    from rotors_mpc_controller_amd.params import load_params
    p = derive_params(load_params())
    N = int(p.horizon_steps)
    fx = np.load(ROOT / "tests" / "golden" / "rti_cold_start.npz")
    hov = p.mass * p.gravity / 4.0
    xh = np.zeros(13); xh[2] = 1.0; xh[6] = 1.0             # K2: hover state
    xz = xh.copy(); xz[2] = 0.5                              # K3: pure z offset
    x0 = np.concatenate([fx["x0"], xh[None], xz[None]])
    yref, yref_e = fx["yref"], fx["yref_e"]
    assert abs(yref[0, 13] - hov) < 1e-12, "fixture hover thrust differs from the shipped parameters"
    def run_set(qp_tol):
        """one cold-start RTI per state: (u0, status, x, u, seconds, qp iterations, residuals [n,4] or NaN, nlp_solver_type)"""
        with tempfile.TemporaryDirectory() as td:
            ocp, solver = build_solver(p, Path(td), qp_tol=qp_tol)
            u0, st, xs, us, ts, it, rs = [], [], [], [], [], [], []
            for x in x0:
                solver.reset() if hasattr(solver, "reset") else None
                a, s, xx, uu, dt = cold_start_rti(solver, N, x, yref, yref_e)
                u0.append(a); st.append(s); xs.append(xx); us.append(uu); ts.append(dt)
                try:
                    it.append(int(np.asarray(solver.get_stats("qp_iter")).reshape(-1)[-1]))
                except Exception:
                    it.append(-1)
                try:
                    rs.append(np.asarray(solver.get_residuals(), float).reshape(-1)[:4])
                except Exception:
                    rs.append(np.full(4, np.nan))
            return (np.array(u0), np.array(st, dtype=np.int32), np.array(xs), np.array(us), np.array(ts), np.array(it, dtype=np.int32),
                    np.array(rs), str(getattr(ocp.solver_options, "nlp_solver_type", "?")))
    u0, st, xs, us, ts, it, rs, nlp_type = run_set(None)                 # the reference's options verbatim
    TIGHT = 1e-12
    u0_t, st_t, xs_t, us_t, ts_t, it_t, rs_t, _ = run_set(TIGHT)         # the QP solved to rounding
    # U10 probe (nmpc_config.qp_maxiter_status): the same OCP with the QP capped at ONE iteration, on the first eight
    # fixture states - the status acados returns for "QP hit its iteration cap" (0 = tolerated, 2 = reported) and the command it
    # leaves are version-dependent
    st_cap, u0_cap = [], []
    with tempfile.TemporaryDirectory() as td:
        _, solver1 = build_solver(p, Path(td), qp_iter_max=1)
        for x in x0[:8]:
            solver1.reset() if hasattr(solver1, "reset") else None
            a, s, _, _, _ = cold_start_rti(solver1, N, x, yref, yref_e)
            st_cap.append(s); u0_cap.append(a)
    np.savez_compressed(
        args.out, x0=x0, yref=yref, yref_e=yref_e, u0=u0, status=st, x=xs, u=us, qp_iter=it, solve_seconds=ts, residuals=rs,
        u0_tight=u0_t, status_tight=st_t, x_tight=xs_t, u_tight=us_t, qp_iter_tight=it_t, solve_seconds_tight=ts_t, residuals_tight=rs_t,
        qp_tol_tight=np.array(TIGHT),
        acados_version=np.array(str(getattr(acados_template, "__version__", "unknown"))),
        casadi_version=np.array(str(getattr(casadi, "__version__", "unknown"))), nlp_solver_type=np.array(nlp_type),
        n_fixture=np.array(fx["x0"].shape[0]), status_itercap=np.array(st_cap, dtype=np.int32), u0_itercap=np.array(u0_cap))
    gap = float(np.abs(u0 - u0_t).max())
    print(f"wrote {args.out}: {len(x0)} instances, status histogram {np.bincount(st, minlength=5)} (tight {np.bincount(st_t, minlength=5)}), "
          f"nlp_solver_type {nlp_type}, median solve() {1e6 * float(np.median(ts)):.0f} us, QP iterations {it.mean():.1f} (tight {it_t.mean():.1f}), "
          f"max |u0(default tolerances) - u0(tolerances {TIGHT:g})| = {gap:.2e} N")


if __name__ == "__main__":
    main()
