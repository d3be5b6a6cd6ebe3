"""Parameter loading: YAML -> the dict-of-dicts that ``PositionNMPC(params)`` consumes.

Counterpart of the reference's ``load_params`` (params.py:153-183) for the keys on the hot path:
same search order (``$ROTORS_MPC_PARAMS``, then the packaged default), same sections, same
defaults and type coercion (params.py:70-150) -- pinned by tests/golden/params_coerced.json.
The ROS parameter overlay and the dynamic_reconfigure mapper (params.py:165-172,186-294) are
ROS plumbing and out of scope.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Any, Dict, List

import yaml

PACKAGED_DEFAULT = Path(__file__).resolve().parent / "config" / "params.yaml"
SECTIONS = ("solver", "vehicle", "controller", "world", "reference", "topics", "node")

# (key, kind, default); kinds: i int, f float, s str, vN list of N floats
_SCHEMA = {
    "solver": [("horizon_steps", "i", 20), ("dt", "f", 0.05), ("position_weight", "v3", [10.0, 10.0, 8.0]),
               ("velocity_weight", "v3", [1.0, 1.0, 0.2]), ("quaternion_weight", "v4", [3.2] * 4),
               ("rate_weight", "v3", [1.4, 1.4, 0.4]), ("control_weight", "v4", [1.75] * 4),
               ("terminal_weight", "v13", [5.0, 5.0, 3.0, 2.0, 2.0, 2.0, 12.0, 12.0, 12.0, 18.5, 2.0, 2.0, 1.8]),
               ("regularization", "f", 7.0e-3), ("iter_max", "i", 600)],
    "vehicle": [("mass", "f", 0.68), ("inertia", "v9", [0.007, 0, 0, 0, 0.007, 0, 0, 0, 0.012]),
                ("arm_length", "f", 0.17), ("rotor_force_constant", "f", 8.54858e-6),
                ("rotor_moment_constant", "f", 0.016), ("motor_min_speed", "f", 0.0),
                ("motor_max_speed", "f", 2000.0), ("drag_coefficients", "v3", [0.0, 0.0, 0.0]),
                ("rotor_configuration", "s", "+")],
    "controller": [("thrust_limits", "v2", [4.0, 20.0])],
    "world": [("gravity", "f", 9.81)],
    "reference": [("frame", "s", "world"), ("default_position", "v3", [1.0, 1.0, 1.0]),
                  ("default_velocity", "v3", [0.0] * 3), ("default_acceleration", "v3", [0.0] * 3),
                  ("default_yaw", "f", 0.0)],
    "node": [("rate", "f", 50.0), ("log_interval", "f", 3.0)],
}
_DROP = {"controller": ("attitude_gains", "max_tilt_deg", "max_tilt_angle"),
         "node": ("max_tilt_deg", "yaw_rate_gain", "yaw_rate_limit")}


def candidate_paths() -> List[Path]:
    out: List[Path] = []
    env = os.environ.get("ROTORS_MPC_PARAMS")
    if env:
        out.append(Path(env).expanduser())
    out.append(PACKAGED_DEFAULT)
    return out


def _coerce(section: str, cfg: Dict[str, Any]) -> None:
    for key, kind, default in _SCHEMA.get(section, ()):
        val = cfg.get(key, default)
        if kind == "i":
            cfg[key] = int(val)
        elif kind == "f":
            cfg[key] = float(val)
        elif kind == "s":
            cfg[key] = str(val).strip() if key == "rotor_configuration" else val
        else:
            n = int(kind[1:])
            if kind in ("v9", "v3", "v2") and len(val) != n and key in ("inertia", "drag_coefficients", "thrust_limits"):
                raise ValueError(f"{section}.{key} must contain {n} values.")
            cfg[key] = [float(x) for x in val]
    for key in _DROP.get(section, ()):
        cfg.pop(key, None)
    if section == "solver" and "codegen_directory" in cfg:
        cfg["codegen_directory"] = str(Path(cfg["codegen_directory"]).expanduser())
    if section == "topics":
        for key in ("state", "motor", "reference"):
            if key not in cfg:
                raise ValueError(f"Missing topic configuration '{key}'")
            cfg[key] = str(cfg[key])


def load_params(path: os.PathLike | None = None) -> Dict[str, Any]:
    paths = [Path(path)] if path is not None else candidate_paths()
    chosen = next((p for p in paths if p.is_file()), None)
    if chosen is None:
        raise FileNotFoundError("No configuration file found for rotors_mpc_controller.")
    data = yaml.safe_load(chosen.read_text(encoding="utf-8")) or {}
    if not isinstance(data, dict):
        raise ValueError(f"Root of {chosen} must be a mapping.")
    missing = set(SECTIONS) - data.keys()
    if missing:
        raise ValueError(f"Missing required top-level sections: {sorted(missing)}")
    for section in SECTIONS:
        _coerce(section, data[section])
    data["params_yaml"] = str(chosen)
    return data
