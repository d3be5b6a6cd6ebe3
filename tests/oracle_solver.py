"""Test-only: an AcadosOcpSolver-shaped object over the CPU oracle, injected into the PositionNMPC
façade (solver_factory=...) so that the façade's own staging code can be exercised without a GPU."""
import numpy as np

from oracle import oracle as O


class OracleOcpSolver:
    def __init__(self, cfg):
        self.N = cfg.N
        self.c = O.default_config(N=cfg.N, dt=cfg.dt, W=list(cfg.W), We=list(cfg.W_e), lbu=list(cfg.lbu),
                                  ubu=list(cfg.ubu), lm=cfg.levenberg_marquardt, lm_scaled_by_dt=cfg.lm_scaled_by_dt,
                                  cost_scaled_by_dt=cfg.cost_scaled_by_dt, mass=cfg.mass, gravity=cfg.gravity,
                                  J=list(cfg.inertia), rotor_x=list(cfg.rotor_x), rotor_y=list(cfg.rotor_y),
                                  rotor_z=list(cfg.rotor_z), sim_num_steps=cfg.sim_num_steps,
                                  qp_iter_max=cfg.qp_iter_max, qp_gamma=0.0, qp_growth_max=cfg.qp_growth_max,
                                  qp_acc_comp=cfg.qp_acc_comp, qp_acc_stat=cfg.qp_acc_stat, qp_tol_step=cfg.qp_tol_step,
                                  qp_maxiter_status=cfg.qp_maxiter_status, qp_warm_start=cfg.qp_warm_start)
        self.x = np.zeros((self.N + 1, 13)); self.u = np.zeros((self.N, 4))
        self.yref = np.zeros((self.N, 17)); self.yref_e = np.zeros(13); self.x0 = np.zeros(13)

    def set(self, stage, field, value):
        v = np.asarray(value, float).reshape(-1)
        if field == "x": self.x[stage] = v
        elif field == "u": self.u[stage] = v
        elif field == "yref":
            if stage < self.N: self.yref[stage] = v
            else: self.yref_e[:] = v
        elif field in ("lbx", "ubx"): self.x0[:] = v
        else: raise KeyError(field)

    def get(self, stage, field):
        return (self.x if field == "x" else self.u)[stage].copy()

    def solve(self):
        xt = self.x.copy(); xt[0] = self.x0
        s, xn, un, _ = O.sqp_rti(self.c, self.x0, self.yref, self.yref_e, xt, self.u)
        if s == 0:
            self.x, self.u = xn, un
        return int(s)

    def close(self):
        pass
