"""`PositionNMPC`-shaped façade over the MI355X-native solver.

Mirrors the reference's controller class (controller.py:52-463): constructor from the params
dict-of-dicts, ``reconfigure``, ``solve(state, reference) -> (u0[4], status)`` with the same
quaternion normalisation, warm-start caching (unshifted, :419-424, :455-461) and failure
behaviour (:448-450), and the properties :358-382.  It differs in what is behind it: no
AcadosOcp / CasADi / code generation -- ``reconfigure`` fills an ``nmpc_config`` and creates a
native handle -- and in ``solve_batch``, which has no counterpart in the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Mapping, Optional, Tuple

import numpy as np

from . import _lib
from .reference import stack_yref
from .solver import NmpcOcpSolver


@dataclass
class ControllerParams:
    """Same fields as the reference's dataclass (controller.py:24-49), minus codegen_directory."""
    horizon_steps: int
    dt: float
    position_weight: np.ndarray
    velocity_weight: np.ndarray
    quaternion_weight: np.ndarray
    rate_weight: np.ndarray
    control_weight: np.ndarray
    terminal_weight: np.ndarray
    regularization: float
    solver_iter_max: int
    mass: float
    inertia: np.ndarray
    gravity: float
    rotor_force_constant: float
    rotor_moment_constant: float
    rotor_x_offsets: np.ndarray
    rotor_y_offsets: np.ndarray
    rotor_z_torque: np.ndarray
    input_lower_bounds: np.ndarray
    input_upper_bounds: np.ndarray
    hover_thrust_per_motor: float
    motor_min_speed: float
    motor_max_speed: float


def derive_params(params: Mapping[str, Mapping[str, object]]) -> ControllerParams:
    """params dict -> physical and tuning constants (controller.py:69-155)."""
    sol, veh = params["solver"], params["vehicle"]
    layout = str(veh.get("rotor_configuration", "+")).strip()
    if layout != "+":
        raise ValueError(f'rotors_mpc_controller only supports a "+" rotor lay-out, got "{layout}".')
    J = np.asarray(veh.get("inertia", [0.007, 0, 0, 0, 0.007, 0, 0, 0, 0.012]), dtype=float).reshape(3, 3)
    arm = float(veh.get("arm_length", 0.17))
    kf = float(veh.get("rotor_force_constant", 8.54858e-6))
    km = float(veh.get("rotor_moment_constant", 0.016))
    w_min = float(veh.get("motor_min_speed", 0.0))
    w_max = float(veh.get("motor_max_speed", 2000.0))
    mass = float(veh["mass"])
    g = float(params.get("world", {}).get("gravity", 9.81))
    # rotor 0 front, 1 left, 2 back, 3 right; CW, CCW, CW, CCW (controller.py:98-103)
    spin = np.array([-1.0, 1.0, -1.0, 1.0])
    vec = lambda key, default: np.asarray(sol.get(key, default), dtype=float)  # noqa: E731
    return ControllerParams(
        horizon_steps=int(sol["horizon_steps"]), dt=float(sol["dt"]),
        position_weight=vec("position_weight", [10.0, 10.0, 8.0]),
        velocity_weight=vec("velocity_weight", [1.0, 1.0, 0.2]),
        quaternion_weight=vec("quaternion_weight", [3.2] * 4),
        rate_weight=vec("rate_weight", [1.4, 1.4, 0.4]),
        control_weight=vec("control_weight", [1.75] * 4),
        terminal_weight=vec("terminal_weight", [5.0, 5.0, 3.0, 2.0, 2.0, 2.0, 12.0, 12.0, 12.0, 18.5, 2.0, 2.0, 1.8]),
        regularization=float(sol.get("regularization", 7.0e-3)),
        solver_iter_max=int(sol.get("iter_max", 600)),
        mass=mass, inertia=np.diag(J).copy(), gravity=g,
        rotor_force_constant=kf, rotor_moment_constant=km,
        rotor_x_offsets=np.array([arm, 0.0, -arm, 0.0]), rotor_y_offsets=np.array([0.0, arm, 0.0, -arm]),
        rotor_z_torque=spin * km,
        input_lower_bounds=np.full(4, max(0.0, kf * w_min ** 2)), input_upper_bounds=np.full(4, kf * w_max ** 2),
        hover_thrust_per_motor=mass * g / 4.0, motor_min_speed=w_min, motor_max_speed=w_max)


def to_nmpc_config(p: ControllerParams, **over) -> _lib.NmpcConfig:
    """The OCP of controller.py:175-264 as numbers."""
    cfg = _lib.default_config()
    cfg.update(
        N=p.horizon_steps, dt=p.dt,
        W=np.concatenate([p.position_weight, p.velocity_weight, p.quaternion_weight, p.rate_weight, p.control_weight]),
        W_e=p.terminal_weight, lbu=p.input_lower_bounds, ubu=p.input_upper_bounds,
        levenberg_marquardt=p.regularization, mass=p.mass, gravity=p.gravity, inertia=p.inertia,
        rotor_x=p.rotor_x_offsets, rotor_y=p.rotor_y_offsets, rotor_z=p.rotor_z_torque,
        sim_num_stages=2, sim_num_steps=2,                       # controller.py:187-188
        qp_iter_max=p.solver_iter_max, qp_cond_N=min(p.horizon_steps, 5))   # controller.py:184-185
    return cfg.update(**over)


class PositionNMPC:
    """Nonlinear MPC for the rotor-level quadrotor model, solved on the GPU."""

    def __init__(self, params: Mapping[str, Mapping[str, object]], *, max_batch: int = 1, device: int = 0,
                 dtype: int = _lib.DTYPE_F64, solver_factory=None, one_call: bool = True, **solver_overrides) -> None:
        # one_call: a tick crosses the Python -> C boundary once (nmpc_solve_batch with B = 1: linearisation
        # point, references, solve, trajectories) instead of the reference's 64 set + 1 solve + 42 get
        # round trips (controller.py:412-460) -- same arrays, same results; one_call=False keeps
        # the reference's call sequence (also used automatically for a solver object without solve_batch)
        self._one_call = bool(one_call)
        # solver_factory(NmpcConfig) -> object with set/solve/get/close; default: the HIP library.
        # (Tests inject an oracle-backed stand-in to exercise this class without a GPU.)
        self._solver_factory = solver_factory or NmpcOcpSolver
        self._solver: Optional[NmpcOcpSolver] = None
        self._opts = dict(max_batch=max_batch, device=device, dtype=dtype, **solver_overrides)
        self._prev_solution: Optional[Dict[str, np.ndarray]] = None
        self._prev_solution_valid = False
        self.reconfigure(params)

    def reconfigure(self, params: Mapping[str, Mapping[str, object]]) -> None:
        """Rebuild the solver for new parameters (controller.py:63-172): a struct fill plus one
        allocation instead of code generation and a C compile."""
        old = self._solver
        self.config = derive_params(params)
        self.mass, self.gravity = self.config.mass, self.config.gravity
        self.nx, self.nu = 13, 4
        self.ny, self.ny_e = 17, 13
        self.nmpc_config = to_nmpc_config(self.config, **self._opts)
        self._solver = self._solver_factory(self.nmpc_config)
        N = self.config.horizon_steps
        self._prev_solution = {"u": np.zeros((N, self.nu)), "x": np.zeros((N + 1, self.nx))}
        self._prev_solution_valid = False
        if old is not None:
            old.close()

    # -- properties, controller.py:358-382 -------------------------------------------------
    @property
    def horizon(self) -> int:
        return self.config.horizon_steps

    @property
    def dt(self) -> float:
        return self.config.dt

    @property
    def hover_thrust(self) -> float:
        return self.config.hover_thrust_per_motor

    @property
    def rotor_force_constant(self) -> float:
        return self.config.rotor_force_constant

    @property
    def motor_speed_limits(self) -> Tuple[float, float]:
        return self.config.motor_min_speed, self.config.motor_max_speed

    @property
    def input_bounds(self) -> Tuple[np.ndarray, np.ndarray]:
        return self.config.input_lower_bounds, self.config.input_upper_bounds

    @property
    def solver(self) -> NmpcOcpSolver:
        return self._solver

    # -- one control tick, controller.py:385-463 -------------------------------------------
    @staticmethod
    def _state_vector(state: Mapping[str, np.ndarray]) -> np.ndarray:
        q = np.asarray(state["quaternion"], dtype=float).reshape(4)
        n = np.linalg.norm(q)
        if n == 0.0:
            raise ValueError("Quaternion norm must be non-zero.")
        return np.concatenate((np.asarray(state["position"], dtype=float).reshape(3),
                               np.asarray(state["velocity"], dtype=float).reshape(3), q / n,
                               np.asarray(state["body_rates"], dtype=float).reshape(3)))

    def solve(self, state: Mapping[str, np.ndarray], reference: Mapping[str, np.ndarray]) -> Tuple[np.ndarray, int]:
        N, s = self.config.horizon_steps, self._solver
        x0 = self._state_vector(state)
        if self._one_call and hasattr(s, "solve_batch"):
            yref, yref_e = stack_yref(reference, N)
            if self._prev_solution_valid:        # previous solution, not shifted; stage 0 is pinned to x0 (:416)
                xi = self._prev_solution["x"][None].copy()
                xi[0, 0] = x0
                out = s.solve_batch(x0[None], yref, yref_e, xi, self._prev_solution["u"][None], want_traj=True)
            else:                                # cold start: x_k = x0, u_k = 0 (:425-431)
                out = s.solve_batch(x0[None], yref, yref_e, want_traj=True)
            status = int(out["status"][0])
            if status != 0:
                self._prev_solution_valid = False
                return np.zeros(self.nu), status
            self._prev_solution["x"][:] = out["x"][0]
            self._prev_solution["u"][:] = out["u"][0]
            self._prev_solution_valid = True
            return out["u0"][0].copy(), status
        s.set(0, "lbx", x0)
        s.set(0, "ubx", x0)
        s.set(0, "x", x0)
        if self._prev_solution_valid:            # previous solution, not shifted
            s.set(0, "u", self._prev_solution["u"][0])
            for k in range(1, N):
                s.set(k, "x", self._prev_solution["x"][k])
                s.set(k, "u", self._prev_solution["u"][k])
            s.set(N, "x", self._prev_solution["x"][-1])
        else:                                    # cold start: x_k = x0, u_k = 0
            zero_u = np.zeros(self.nu)
            s.set(0, "u", zero_u)
            for k in range(1, N):
                s.set(k, "x", x0)
                s.set(k, "u", zero_u)
            s.set(N, "x", x0)
        yref, yref_e = stack_yref(reference, N)
        for k in range(N):
            s.set(k, "yref", yref[k])
        s.set(N, "yref", yref_e)
        status = s.solve()
        if status != 0:
            self._prev_solution_valid = False
            return np.zeros(self.nu), status
        u0 = s.get(0, "u").reshape(-1)
        for k in range(N):
            self._prev_solution["u"][k] = s.get(k, "u")
            self._prev_solution["x"][k] = s.get(k, "x")
        self._prev_solution["x"][-1] = s.get(N, "x")
        self._prev_solution_valid = True
        return u0, status

    # -- batched variant (new) ---------------------------------------------------------------
    def solve_batch(self, x0: np.ndarray, yref: np.ndarray, yref_e: np.ndarray, x_init=None, u_init=None,
                    want_traj: bool = False):
        """Many independent instances in one call; quaternions are normalised row-wise like :406-409."""
        x0 = np.array(x0, dtype=float)
        n = np.linalg.norm(x0[:, 6:10], axis=1)
        if (n == 0.0).any():
            raise ValueError("Quaternion norm must be non-zero.")
        x0[:, 6:10] /= n[:, None]
        return self._solver.solve_batch(x0, yref, yref_e, x_init, u_init, want_traj)
