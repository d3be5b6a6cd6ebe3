"""Seeded random draws of everything `reconfigure` can change (vehicle, tuning, horizon, integrator steps, batch, references):
the generator behind tools/dev/fuzz_parity.py / fuzz_one.py and behind the regression tests that pin single draws.
Test infrastructure only."""
from __future__ import annotations

import numpy as np

from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, sample_x0

WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
DISTS = (NEAR_HOVER, AGGRESSIVE, WILD)


def draw(seed: int, materialise_refs: bool = False, horizons=None, max_batch=None):
    """Returns (over, x0, yref, yref_e, hover_thrust, dist_index): `over` are the nmpc_config overrides of the draw (the
    random stream is consumed in the order tools/dev/fuzz_parity.py always used, so seeds name the same draws as in
    profiles/r02f_fuzz_*).  materialise_refs: per-instance [B,N,17] references even where the draw broadcasts one.
    horizons / max_batch (late round 5, tools/dev/fuzz_parity.py --long): the horizon drawn from this list instead - the block-parallel
    tail of N >= 160 on random vehicles - and the batch capped (the oracle's time); the default draws are unchanged."""
    rng = np.random.default_rng(7000 + seed)
    N = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 20, 24, 31, 40, 57]))
    if horizons is not None:
        N = int(horizons[seed % len(horizons)])
    mass = float(rng.uniform(0.3, 4.0))
    arm = float(rng.uniform(0.08, 0.5))
    km = float(rng.uniform(0.003, 0.04))
    hov = mass * 9.81 / 4.0
    B = int(rng.choice([1, 3, 4, 5, 63, 64, 65, 130, 257, 511]))
    if max_batch is not None:
        B = min(B, int(max_batch))
    over = dict(N=N, dt=float(rng.choice([0.01, 0.02, 0.05, 0.08, 0.1])), mass=mass,
                inertia=[float(v) for v in rng.uniform(0.002, 0.04, 3) * mass],
                rotor_x=[arm, 0.0, -arm, 0.0], rotor_y=[0.0, arm, 0.0, -arm], rotor_z=[-km, km, -km, km],
                lbu=[float(hov * rng.uniform(0.0, 0.5))] * 4, ubu=[float(hov * rng.uniform(1.3, 4.0))] * 4,
                W=[float(v) for v in 10.0 ** rng.uniform(-2, 2, 17)], W_e=[float(v) for v in 10.0 ** rng.uniform(-1, 2.5, 13)],
                levenberg_marquardt=float(rng.choice([0.0, 1e-4, 7e-3, 0.1, 1.0])), sim_num_steps=int(rng.choice([1, 2, 2, 3])),
                lm_scaled_by_dt=int(rng.integers(0, 2)), cost_scaled_by_dt=int(rng.integers(0, 2)),
                flags=_lib.FLAG_TEAM_MAPPING | int(rng.integers(0, 2)), max_batch=B,
                qp_polish_ckpt=int(rng.choice([0, 1, 4, 12, 100])))
    di = int(rng.integers(0, 3))
    x0 = sample_x0(B, 9000 + seed, **DISTS[di])
    per_inst = bool(rng.integers(0, 2))
    goal = rng.normal(0.0, 1.0, (B, 3)) + np.array([0.0, 0.0, 1.0])
    vel = rng.normal(0.0, 0.3, (B, 3))
    yref = np.zeros((B, N, 17)); ye = np.zeros((B, 13))
    for k in range(N + 1):
        row = np.zeros((B, 13)); row[:, 0:3] = goal + vel * (k * over["dt"]); row[:, 3:6] = vel; row[:, 6] = 1.0
        if k < N:
            yref[:, k, :13] = row; yref[:, k, 13:] = hov
        else:
            ye[:] = row
    if not per_inst:
        yref, ye = yref[0], ye[0]
        if materialise_refs:
            yref, ye = np.tile(yref, (B, 1, 1)), np.tile(ye, (B, 1))
    return over, x0, yref, ye, hov, di, rng


def oracle_config(over, **extra):
    """The oracle configured as the library would be by `over` (the fields both sides share)."""
    from oracle import oracle as O
    cfg = _lib.default_config(**over) if not isinstance(over, _lib.NmpcConfig) else over
    c = O.default_config(N=cfg.N, dt=cfg.dt, W=list(cfg.W), We=list(cfg.W_e), lbu=list(cfg.lbu), ubu=list(cfg.ubu),
                         lm=cfg.levenberg_marquardt, lm_scaled_by_dt=cfg.lm_scaled_by_dt, cost_scaled_by_dt=cfg.cost_scaled_by_dt,
                         mass=cfg.mass, gravity=cfg.gravity, J=list(cfg.inertia), rotor_x=list(cfg.rotor_x),
                         rotor_y=list(cfg.rotor_y), rotor_z=list(cfg.rotor_z), sim_num_steps=cfg.sim_num_steps,
                         qp_iter_max=cfg.qp_iter_max, qp_gamma=0.0, qp_polish=cfg.qp_polish,
                         qp_growth_max=cfg.qp_growth_max, qp_acc_comp=cfg.qp_acc_comp, qp_acc_stat=cfg.qp_acc_stat,
                         qp_tol_step=cfg.qp_tol_step, qp_maxiter_status=cfg.qp_maxiter_status, qp_warm_start=cfg.qp_warm_start)
    for k, v in extra.items():
        setattr(c, k, v)
    return c
