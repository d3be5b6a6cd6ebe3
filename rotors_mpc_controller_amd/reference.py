"""Constant-setpoint reference horizons -- the input contract of the hot path.

Counterpart of the reference's ``ReferenceGenerator`` (reference.py:16-91): same method names,
argument meaning and output dictionary (keys ``positions, velocities, quaternions, body_rates,
thrusts, yaws``; N+1 rows except ``thrusts`` with N), pinned by tests/golden/reference_horizon.json.
Adds :func:`stack_yref`, the [N,17] / [13] packing that controller.py:433-445 performs stage by
stage, and a batched variant for many setpoints at once.
"""
from __future__ import annotations

import threading
from typing import Dict, Mapping, Optional, Tuple

import numpy as np


def yaw_quaternion(yaw) -> np.ndarray:
    """(w, x, y, z) of a rotation by `yaw` about world z (reference.py:11-13); vectorised."""
    half = 0.5 * np.asarray(yaw, dtype=float)
    z = np.zeros_like(half)
    return np.stack([np.cos(half), z, z, np.sin(half)], axis=-1)


class ReferenceGenerator:
    def __init__(self, config: Mapping[str, object]) -> None:
        self.frame = config.get("frame", "world")
        self._lock = threading.Lock()
        self._p = np.array(config.get("default_position", [0.0, 0.0, 1.0]), dtype=float)
        self._v = np.array(config.get("default_velocity", [0.0, 0.0, 0.0]), dtype=float)
        self._yaw = float(config.get("default_yaw", 0.0))
        self._q = yaw_quaternion(self._yaw)
        self._w = np.zeros(3)
        self._thrust = np.zeros(4)

    def set_target(self, position, velocity=None, yaw: Optional[float] = None, quaternion=None,
                   body_rates=None, thrust=None) -> None:
        with self._lock:
            self._p = np.array(position, dtype=float).reshape(3)
            if velocity is not None:
                self._v = np.array(velocity, dtype=float).reshape(3)
            if quaternion is not None:
                # an explicit quaternion wins; yaw is only recorded alongside it
                q = np.array(quaternion, dtype=float).reshape(4)
                n = np.linalg.norm(q)
                self._q = q / n if n != 0.0 else q
                if yaw is not None:
                    self._yaw = yaw
            elif yaw is not None:
                self._yaw = float(yaw)
                self._q = yaw_quaternion(self._yaw)
            if body_rates is not None:
                self._w = np.array(body_rates, dtype=float).reshape(3)
            if thrust is not None:
                t = np.array(thrust, dtype=float).reshape(-1)
                if t.shape[0] != 4:
                    raise ValueError("Thrust reference must have four components.")
                self._thrust = t

    def update_defaults(self, position, velocity, yaw: float, frame: Optional[str] = None) -> None:
        with self._lock:
            self._p = np.array(position, dtype=float).reshape(3)
            self._v = np.array(velocity, dtype=float).reshape(3)
            self._yaw = float(yaw)
            self._q = yaw_quaternion(self._yaw)
            self._w = np.zeros(3)
            if frame is not None:
                self.frame = frame

    def update_hover_thrust(self, thrust_per_motor: float) -> None:
        with self._lock:
            self._thrust = np.full(4, float(thrust_per_motor))

    def build_horizon(self, horizon: int, dt: float) -> Dict[str, np.ndarray]:
        with self._lock:
            p, v, q, w, t, yaw = self._p, self._v, self._q, self._w, self._thrust, self._yaw
            n1 = horizon + 1
            return {
                "positions": np.broadcast_to(p, (n1, 3)).copy(),
                "velocities": np.broadcast_to(v, (n1, 3)).copy(),
                "quaternions": np.broadcast_to(q, (n1, 4)).copy(),
                "body_rates": np.broadcast_to(w, (n1, 3)).copy(),
                "thrusts": np.broadcast_to(t, (horizon, 4)).copy(),
                "yaws": np.full(n1, yaw, dtype=float),
            }


def stack_yref(reference: Mapping[str, np.ndarray], horizon: int) -> Tuple[np.ndarray, np.ndarray]:
    """(yref [N,17], yref_e [13]) exactly as controller.py:433-445 assembles them."""
    yref = np.concatenate([np.asarray(reference[k], dtype=float)[:horizon]
                           for k in ("positions", "velocities", "quaternions", "body_rates", "thrusts")], axis=1)
    yref_e = np.concatenate([np.asarray(reference[k], dtype=float)[-1]
                             for k in ("positions", "velocities", "quaternions", "body_rates")])
    return np.ascontiguousarray(yref), np.ascontiguousarray(yref_e)


def batched_hover_yref(positions: np.ndarray, yaws: np.ndarray, hover_thrust: float, horizon: int):
    """Many constant setpoints at once: positions [B,3], yaws [B] -> yref [B,N,17], yref_e [B,13]."""
    positions = np.asarray(positions, dtype=float)
    B = positions.shape[0]
    y = np.zeros((B, 17))
    y[:, 0:3] = positions
    y[:, 6:10] = yaw_quaternion(np.asarray(yaws, dtype=float))
    y[:, 13:17] = hover_thrust
    return np.ascontiguousarray(np.repeat(y[:, None, :], horizon, axis=1)), np.ascontiguousarray(y[:, :13])
