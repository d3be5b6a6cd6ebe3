"""`AcadosOcpSolver`-shaped front end of the MI355X-native solver.

Mirrors exactly the surface the reference uses on the object it builds at
/root/reference/src/rotors_mpc_controller/controller.py:263 --
``set(stage, field, value)`` (:414-445), ``solve()`` (:447), ``get(stage, field)`` (:452-460)
-- and adds the batched entry points the reference does not have.  All arithmetic happens in
librotors_nmpc_hip.so on the GPU; this module only marshals arrays.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import NU, NX, NY, NmpcConfig, NmpcStats


class NmpcError(RuntimeError):
    pass


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(out.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(out.shape)}")
    return out


class NmpcOcpSolver:
    """Drop-in for ``acados_template.AcadosOcpSolver`` on the reference's hot path.

    Construct from an :class:`NmpcConfig` (numbers instead of a CasADi/AcadosOcp object: the
    reference bakes the physical constants into a symbolic expression, controller.py:311-341).
    """

    def __init__(self, config: NmpcConfig):
        self._lib = _lib.load()
        self.config = config
        self._h = self._lib.nmpc_create(C.byref(config))
        if not self._h:
            raise NmpcError(self._lib.nmpc_last_error(None).decode())
        self.N = int(config.N)
        self.status = 0

    # -- lifetime ------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.nmpc_destroy(self._h)
            self._h = None

    def __del__(self):  # controller.py:169-170 relies on `del old_solver`
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc < 0:
            raise NmpcError(self._lib.nmpc_last_error(self._h).decode())

    # -- AcadosOcpSolver surface ---------------------------------------------------------
    def set(self, stage: int, field: str, value) -> None:
        v = np.ascontiguousarray(value, dtype=np.float64).reshape(-1)
        self._check(self._lib.nmpc_set(self._h, int(stage), field.encode(), _dp(v), int(v.size)))

    def get(self, stage: int, field: str) -> np.ndarray:
        n = {"x": NX, "u": NU}.get(field)
        if n is None:
            raise NmpcError(f"get: unknown field '{field}' (known: x, u)")
        out = np.zeros(n)
        self._check(self._lib.nmpc_get(self._h, int(stage), field.encode(), _dp(out), n))
        return out

    def solve(self) -> int:
        rc = self._lib.nmpc_solve(self._h)
        self._check(rc)
        self.status = rc
        return rc

    # -- batched entry points (no counterpart in the reference) ----------------------------
    def solve_batch(self, x0, yref, yref_e, x_init=None, u_init=None, want_traj: bool = False):
        """Host arrays in, host arrays out (PCIe inclusive).  yref [N,17] (shared) or [B,N,17]."""
        x0 = _f64(x0)
        if x0.ndim != 2 or x0.shape[1] != NX:
            raise ValueError(f"x0 must be [B,{NX}]")
        B, N = x0.shape[0], self.N
        yref = _f64(yref)
        bcast = yref.ndim == 2
        yref = _f64(yref, (N, NY) if bcast else (B, N, NY))
        yref_e = _f64(yref_e, (NX,) if bcast else (B, NX))
        if (x_init is None) != (u_init is None):
            raise ValueError("x_init and u_init must both be given or both be None")
        xi = None if x_init is None else _f64(x_init, (B, N + 1, NX))
        ui = None if u_init is None else _f64(u_init, (B, N, NU))
        u0 = np.zeros((B, NU))
        status = np.zeros(B, dtype=np.int32)
        xo = np.zeros((B, N + 1, NX)) if want_traj else None
        uo = np.zeros((B, N, NU)) if want_traj else None
        rc = self._lib.nmpc_solve_batch(self._h, B, _dp(x0), _dp(yref), _dp(yref_e), int(bcast), _dp(xi),
                                        _dp(ui), _dp(u0), status.ctypes.data_as(C.POINTER(C.c_int32)),
                                        _dp(xo), _dp(uo))
        self._check(rc)
        return dict(u0=u0, status=status, x=xo, u=uo)

    def solve_batch_device(self, B: int, x0_ptr: int, yref_ptr: int, yref_e_ptr: int, yref_bcast: bool,
                           u0_ptr: int, status_ptr: int = 0, x_init_ptr: int = 0, u_init_ptr: int = 0,
                           x_out_ptr: int = 0, u_out_ptr: int = 0, stream: int = 0) -> None:
        """Device pointers (e.g. ``tensor.data_ptr()``) of element type config.dtype; only enqueues."""
        rc = self._lib.nmpc_solve_batch_device(self._h, int(B), x0_ptr, yref_ptr, yref_e_ptr, int(yref_bcast),
                                               x_init_ptr or None, u_init_ptr or None, u0_ptr,
                                               status_ptr or None, x_out_ptr or None, u_out_ptr or None,
                                               stream or None)
        self._check(rc)

    # -- the steps either side of the solve, on the device ---------------------------
    def build_hover_reference_device(self, B: int, positions_ptr: int, yaws_ptr: int, thrust_per_motor: float,
                                     yref_ptr: int, yref_e_ptr: int, stream: int = 0) -> None:
        """reference.py:75-91 + controller.py:433-445 for B constant setpoints, written in HBM."""
        self._check(self._lib.nmpc_build_hover_reference_device(self._h, int(B), positions_ptr, yaws_ptr,
                                                                float(thrust_per_motor), yref_ptr, yref_e_ptr,
                                                                stream or None))

    def odometry_to_state_device(self, B: int, pose_ptr: int, twist_ptr: int, x0_ptr: int, stream: int = 0) -> None:
        """nodes/mpc_controller_node:88-113 batched: pose [B,7] (p, q=(x,y,z,w)), body twist [B,6] -> x0 [B,13]."""
        self._check(self._lib.nmpc_odometry_to_state_device(self._h, int(B), pose_ptr, twist_ptr, x0_ptr,
                                                            stream or None))

    def commands_to_motor_speeds_device(self, B: int, u_ptr: int, rotor_force_constant: float, motor_min_speed: float,
                                        motor_max_speed: float, speeds_ptr: int, clipped_ptr: int = 0,
                                        stream: int = 0) -> None:
        """nodes/mpc_controller_node:152-164 batched: thrusts [B,4] -> motor speeds [B,4]."""
        self._check(self._lib.nmpc_commands_to_motor_speeds_device(
            self._h, int(B), u_ptr, float(rotor_force_constant), float(motor_min_speed), float(motor_max_speed),
            speeds_ptr, clipped_ptr or None, stream or None))

    def hold_command_device(self, B: int, u0_ptr: int, status_ptr: int, held_ptr: int, stream: int = 0) -> None:
        """nodes/mpc_controller_node:122-131 batched: held [B,4] <- clip(u0) where status == 0, else unchanged."""
        self._check(self._lib.nmpc_hold_command_device(self._h, int(B), u0_ptr, status_ptr, held_ptr, stream or None))

    def hold_and_step_device(self, B: int, u0_ptr: int, status_ptr: int, held_ptr: int, x_ptr: int,
                             normalize_q: bool = True, stream: int = 0) -> None:
        """hold_command_device + plant_step_device of the held command in one launch; x [B][13] is updated in place."""
        self._check(self._lib.nmpc_hold_and_step_device(self._h, int(B), u0_ptr, status_ptr, held_ptr, x_ptr,
                                                        int(normalize_q), stream or None))

    def plant_step_device(self, B: int, x_ptr: int, u_ptr: int, x_next_ptr: int, normalize_q: bool = True,
                          stream: int = 0) -> None:
        """One interval of the controller's own model/ERK as the plant of a closed-loop rollout."""
        self._check(self._lib.nmpc_plant_step_device(self._h, int(B), x_ptr, u_ptr, x_next_ptr, int(normalize_q),
                                                     stream or None))

    def adjoint_sensitivities_device(self, B: int, x_ptr: int, u_ptr: int, lam_ptr: int, out_ptr: int,
                                     continuous: bool = False, stream: int = 0) -> None:
        """(A' lam | B' lam) of one shooting interval per instance (continuous: (f_x' lam | f_u' lam), = expl_vde_adj)."""
        self._check(self._lib.nmpc_adjoint_sensitivities_device(self._h, int(B), x_ptr, u_ptr, lam_ptr, out_ptr,
                                                                int(bool(continuous)), stream or None))

    def kkt_report_device(self, B: int, x_traj_ptr: int, u_traj_ptr: int, yref_ptr: int, yref_e_ptr: int, yref_bcast: bool,
                          res_ptr: int, stream: int = 0) -> None:
        """res [B,3] = (projected input-gradient, dynamics defect, bound violation) of trajectories, by an adjoint sweep."""
        self._check(self._lib.nmpc_kkt_report_device(self._h, int(B), x_traj_ptr, u_traj_ptr, yref_ptr, yref_e_ptr,
                                                     int(bool(yref_bcast)), res_ptr, stream or None))

    def block_factor_device(self, B: int, blocks: int, x0_ptr: int, yref_ptr: int, yref_e_ptr: int, yref_bcast: bool,
                            x_init_ptr: int = 0, u_init_ptr: int = 0, factors_ptr: int = 0, boundary_ptr: int = 0,
                            check_ptr: int = 0, timed: bool = False, stream: int = 0):
        """Parallel-in-time Riccati factorisation of the LQ problem the last solve ended on (include/rotors_nmpc.h,
        nmpc_block_factor_device).  Returns (J, (ms_launch1, ms_scan, ms_launch3) or None)."""
        ms = (C.c_float * 3)() if timed else None
        rc = self._lib.nmpc_block_factor_device(self._h, int(B), int(blocks), x0_ptr, yref_ptr, yref_e_ptr, int(bool(yref_bcast)),
                                                x_init_ptr or None, u_init_ptr or None, factors_ptr or None, boundary_ptr or None,
                                                check_ptr or None, ms, stream or None)
        if rc < 0:
            self._check(rc)
        return rc, (tuple(ms) if timed else None)

    def debug_factors(self, B: int) -> np.ndarray:
        """[B][N][80] factors the last solve's own sweeps left in the workspace (complete only with NMPC_TEAM_LSTG=0)."""
        out = np.empty((int(B), self.config.N, 80))
        self._check(self._lib.nmpc_debug_factors(self._h, int(B), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def tail_states(self, B: int):
        """(blocks, states [B]) of the block-parallel tail after the last solve: 0 not in the work list, 3 finished by the tail,
        5 handed on to the sequential work-list kernel; blocks = 0 when this handle runs no tail."""
        out = np.zeros(int(B), dtype=np.int32)
        rc = self._lib.nmpc_debug_tail_states(self._h, int(B), out.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc < 0:
            self._check(rc)
        return rc, out

    def device_iterations_ptr(self) -> int:
        return int(self._lib.nmpc_device_iterations(self._h) or 0)

    def device_passes_ptr(self) -> int:
        """int32 [B] on the device: active-set passes of the last solve (> 0: accepted active-set solution)."""
        return int(self._lib.nmpc_device_passes(self._h) or 0)

    def counts(self, B: int | None = None):
        """(iterations, passes) per instance of the last solve, host int32 arrays.  passes > 0: the instance ended on an
        accepted active-set solution after that many passes; <= 0: on the interior-point iterate (magnitude = passes spent)."""
        n = int(B if B is not None else self.stats()["batch"])
        it, ps = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        self._check(self._lib.nmpc_get_counts(self._h, n, it.ctypes.data_as(ip), ps.ctypes.data_as(ip)))
        return it, ps

    def iterations(self, B: int | None = None):
        return self.counts(B)[0]

    def passes(self, B: int | None = None):
        return self.counts(B)[1]

    def set_timing(self, on: bool) -> None:
        """HIP events around the kernels of every solve (default on; stats() then reports kernel times)."""
        self._check(self._lib.nmpc_set_timing(self._h, int(bool(on))))

    def guard_check(self) -> int:
        """Damaged bytes of the canary bands around the handle's device buffers (handles created under NMPC_GUARD=<KiB>); -1 without."""
        n = int(self._lib.nmpc_debug_guard_check(self._h))
        if n > 0:
            raise RuntimeError(self._lib.nmpc_last_error(self._h).decode())
        return n

    def last_schedule(self) -> dict:
        """Launch schedule of the last solve (nmpc_debug_last_schedule): first attempt by k_team_as | failed attempts continued in place |
        block-parallel tail."""
        v = int(self._lib.nmpc_debug_last_schedule(self._h))
        return dict(split=bool(v & 1), inplace=bool(v & 2), tail=bool(v & 4))

    def stats(self) -> dict:
        st = NmpcStats()
        self._check(self._lib.nmpc_get_stats(self._h, C.byref(st)))
        return dict(batch=st.batch, iter_min=st.iter_min, iter_max=st.iter_max, iter_mean=st.iter_mean,
                    n_status=list(st.n_status), ms_prepare=st.ms_prepare, ms_solve=st.ms_solve,
                    workspace_bytes=int(st.workspace_bytes), polish_mean=st.polish_mean, polish_max=st.polish_max,
                    n_polished=st.n_polished, n_tail=st.n_tail, ms_tail=st.ms_tail)


# name a maintainer would import in place of acados_template's class
AcadosOcpSolver = NmpcOcpSolver
