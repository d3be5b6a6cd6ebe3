"""Synthetic initial-state distributions for the benchmark and the parity tests.

The recipe (order and shape of the RNG calls) is the measurement contract of SURVEY.md 8(d):
one `numpy.random.default_rng(seed)`, whole batch at once per field, in the order
p, v, axis, angle, omega.  There is no counterpart in the reference (it has no benchmark);
the hover setpoint mirrors reference config/params.yaml:36-40.
"""
from __future__ import annotations

import numpy as np

NEAR_HOVER = dict(sigma_p=0.3, sigma_v=0.5, max_angle_deg=20.0, sigma_w=0.5)
AGGRESSIVE = dict(sigma_p=1.0, sigma_v=1.0, max_angle_deg=30.0, sigma_w=1.0)


def sample_x0(batch: int, seed: int, *, sigma_p: float = 0.3, sigma_v: float = 0.5,
              max_angle_deg: float = 20.0, sigma_w: float = 0.5,
              setpoint=(0.0, 0.0, 1.0), dtype=np.float64) -> np.ndarray:
    """[batch, 13] states (p, v, q=(w,x,y,z), omega) scattered around the hover setpoint."""
    rng = np.random.default_rng(seed)
    p = np.asarray(setpoint, dtype=float) + rng.normal(0.0, sigma_p, (batch, 3))
    v = rng.normal(0.0, sigma_v, (batch, 3))
    axis = rng.normal(size=(batch, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    a = rng.uniform(0.0, np.deg2rad(max_angle_deg), batch)
    q = np.concatenate([np.cos(a / 2)[:, None], np.sin(a / 2)[:, None] * axis], axis=1)
    w = rng.normal(0.0, sigma_w, (batch, 3))
    return np.ascontiguousarray(np.concatenate([p, v, q, w], axis=1), dtype=dtype)


def hover_reference(horizon: int, hover_thrust: float, position=(0.0, 0.0, 1.0), yaw: float = 0.0,
                    dtype=np.float64):
    """(yref [N,17], yref_e [13]) for a constant hover setpoint: what reference.py:75-91
    tiles and controller.py:433-445 stacks, with thrusts = m g / 4 (node:52)."""
    y = np.zeros(17)
    y[0:3] = position
    y[6] = np.cos(0.5 * yaw)
    y[9] = np.sin(0.5 * yaw)
    y[13:17] = hover_thrust
    return (np.ascontiguousarray(np.tile(y, (horizon, 1)), dtype=dtype),
            np.ascontiguousarray(y[:13], dtype=dtype))
