"""MI355X-native batched NMPC for the rotor-level quadrotor OCP of rotors_mpc_controller.

Import surface mirrors the reference package (src/rotors_mpc_controller/__init__.py:3-12) for
the parts on the hot path: PositionNMPC, ReferenceGenerator, load_params -- plus the
AcadosOcpSolver-shaped NmpcOcpSolver and its config.  Importing this package does not load the
HIP library; constructing a solver does, and fails loudly without it or without a GPU.
"""
from ._lib import DTYPE_F32, DTYPE_F32IO, DTYPE_F64, FLAG_CONDENSED_QP, FLAG_SHARE_COLD_START, FLAG_TEAM_MAPPING, NmpcConfig, default_config
from .controller import ControllerParams, PositionNMPC, derive_params, to_nmpc_config
from .params import load_params
from .reference import ReferenceGenerator, batched_hover_yref, stack_yref
from .solver import AcadosOcpSolver, NmpcError, NmpcOcpSolver

__all__ = [
    "PositionNMPC", "ReferenceGenerator", "load_params", "NmpcOcpSolver", "AcadosOcpSolver",
    "NmpcConfig", "NmpcError", "default_config", "ControllerParams", "derive_params", "to_nmpc_config",
    "stack_yref", "batched_hover_yref", "DTYPE_F64", "DTYPE_F32", "DTYPE_F32IO", "FLAG_SHARE_COLD_START",
    "FLAG_TEAM_MAPPING", "FLAG_CONDENSED_QP",
]
