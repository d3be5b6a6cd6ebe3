"""ctypes binding of include/rotors_nmpc.h (librotors_nmpc_hip.so).

The library is the product: there is no Python or CPU fallback.  `load()` raises if the
shared object is missing, and `nmpc_create` itself fails when no HIP device is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

NX, NU, NY = 13, 4, 17

DTYPE_F64, DTYPE_F32, DTYPE_F32IO = 0, 1, 2     # F32IO: float device buffers, double arithmetic and workspace
FLAG_SHARE_COLD_START = 1
FLAG_TEAM_MAPPING = 2
FLAG_CONDENSED_QP = 4

STATUS_NAMES = {0: "SUCCESS", 1: "NAN_DETECTED", 2: "MAXITER", 3: "MINSTEP", 4: "QP_FAILURE"}

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "librotors_nmpc_hip.so"
CSRC = _PKG / "csrc"


class NmpcConfig(C.Structure):
    """Mirror of `nmpc_config` (include/rotors_nmpc.h)."""
    _fields_ = [
        ("N", C.c_int32), ("dt", C.c_double),
        ("W", C.c_double * NY), ("W_e", C.c_double * NX),
        ("lbu", C.c_double * NU), ("ubu", C.c_double * NU),
        ("levenberg_marquardt", C.c_double),
        ("lm_scaled_by_dt", C.c_int32), ("cost_scaled_by_dt", C.c_int32),
        ("mass", C.c_double), ("gravity", C.c_double), ("inertia", C.c_double * 3),
        ("rotor_x", C.c_double * NU), ("rotor_y", C.c_double * NU), ("rotor_z", C.c_double * NU),
        ("sim_num_stages", C.c_int32), ("sim_num_steps", C.c_int32),
        ("qp_iter_max", C.c_int32), ("qp_cond_N", C.c_int32),
        ("qp_tol_comp", C.c_double), ("qp_tol_stat", C.c_double), ("qp_mu0", C.c_double),
        ("qp_tau", C.c_double), ("qp_thr0", C.c_double), ("qp_thr0_rel", C.c_double),
        ("dtype", C.c_int32), ("device", C.c_int32), ("max_batch", C.c_int32), ("flags", C.c_uint32),
        ("qp_polish", C.c_int32), ("qp_polish_passes", C.c_int32), ("qp_polish_budget", C.c_int32),
        ("qp_polish_mu", C.c_double), ("qp_polish_ckpt", C.c_int32), ("qp_maxiter_status", C.c_int32),
        ("qp_growth_max", C.c_double), ("qp_acc_comp", C.c_double), ("qp_acc_stat", C.c_double), ("qp_tol_step", C.c_double),
        ("qp_warm_start", C.c_int32), ("reserved_", C.c_int32),
    ]

    def update(self, **over) -> "NmpcConfig":
        for k, v in over.items():
            cur = getattr(self, k)
            if hasattr(cur, "__len__"):
                vals = list(v)
                if len(vals) != len(cur):
                    raise ValueError(f"{k} takes {len(cur)} values, got {len(vals)}")
                for i, x in enumerate(vals):
                    cur[i] = float(x)
            else:
                setattr(self, k, v)
        return self


class NmpcStats(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("iter_min", C.c_int32), ("iter_max", C.c_int32),
        ("iter_mean", C.c_double), ("n_status", C.c_int32 * 5),
        ("ms_prepare", C.c_double), ("ms_solve", C.c_double), ("workspace_bytes", C.c_uint64),
        ("polish_mean", C.c_double), ("polish_max", C.c_int32), ("n_polished", C.c_int32),
        ("n_tail", C.c_int32), ("ms_tail", C.c_double),
    ]


EXPORTS = (
    "nmpc_default_config", "nmpc_create", "nmpc_destroy", "nmpc_set", "nmpc_get", "nmpc_solve",
    "nmpc_solve_batch", "nmpc_solve_batch_device", "nmpc_device_iterations", "nmpc_device_passes", "nmpc_get_counts", "nmpc_get_stats", "nmpc_set_timing",
    "nmpc_last_error", "nmpc_version", "nmpc_abi_sizes", "nmpc_build_hover_reference_device",
    "nmpc_odometry_to_state_device", "nmpc_commands_to_motor_speeds_device", "nmpc_plant_step_device",
    "nmpc_hold_command_device", "nmpc_hold_and_step_device", "nmpc_adjoint_sensitivities_device", "nmpc_kkt_report_device", "nmpc_block_factor_device", "nmpc_debug_factors", "nmpc_debug_tail_states",
    "nmpc_debug_guard_check", "nmpc_debug_last_schedule",
)


def build(force: bool = False) -> Path:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.hpp")) + [CSRC / "Makefile"]
    srcs.append(_PKG.parent / "include" / "rotors_nmpc.h")
    stale = (not LIB_PATH.exists()) or any(p.stat().st_mtime > LIB_PATH.stat().st_mtime for p in srcs)
    if force or stale:
        subprocess.check_call(["make", "-j2", "-C", str(CSRC)] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def _bind_one_hip_runtime() -> None:
    """One HIP runtime per process, whatever the import order.

    PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64 (torch/lib, SONAME libamdhip64.so.7 -- the
    same SONAME this library's DT_NEEDED names).  Two copies in one process each run their own device
    discovery and the second one finds no device (observed on the GPU box), and a pointer allocated through one
    must never reach the other.  So, without importing torch: if a torch installation exists and has not been
    imported yet, map ITS runtime first with RTLD_GLOBAL.  The dynamic linker then resolves this library's
    DT_NEEDED to that already-loaded object, and a later `import torch` finds the same file mapped: one runtime
    in either order.  Without torch installed the library binds to /opt/rocm's copy through its RUNPATH.
    (A C program that links this library and never loads torch is in the second case by construction.)"""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                                   # torch has mapped its runtime already
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load the HIP library.  Never falls back: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    _bind_one_hip_runtime()
    path = Path(os.environ.get("ROTORS_NMPC_LIB", LIB_PATH))
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "rotors_mpc_controller_amd has no CPU fallback.")
    lib = C.CDLL(str(path))
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    cp = C.POINTER(NmpcConfig)
    lib.nmpc_default_config.argtypes = [cp]
    lib.nmpc_default_config.restype = None
    lib.nmpc_create.argtypes = [cp]
    lib.nmpc_create.restype = vp
    lib.nmpc_destroy.argtypes = [vp]
    lib.nmpc_destroy.restype = None
    lib.nmpc_set.argtypes = [vp, C.c_int, C.c_char_p, dp, C.c_int]
    lib.nmpc_set.restype = C.c_int
    lib.nmpc_get.argtypes = [vp, C.c_int, C.c_char_p, dp, C.c_int]
    lib.nmpc_get.restype = C.c_int
    lib.nmpc_solve.argtypes = [vp]
    lib.nmpc_solve.restype = C.c_int
    lib.nmpc_solve_batch.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int, dp, dp, dp, ip, dp, dp]
    lib.nmpc_solve_batch.restype = C.c_int
    lib.nmpc_solve_batch_device.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.nmpc_solve_batch_device.restype = C.c_int
    lib.nmpc_device_iterations.argtypes = [vp]
    lib.nmpc_device_iterations.restype = vp
    lib.nmpc_device_passes.argtypes = [vp]
    lib.nmpc_device_passes.restype = vp
    lib.nmpc_get_counts.argtypes = [vp, C.c_int, ip, ip]
    lib.nmpc_get_counts.restype = C.c_int
    lib.nmpc_get_stats.argtypes = [vp, C.POINTER(NmpcStats)]
    lib.nmpc_get_stats.restype = C.c_int
    lib.nmpc_set_timing.argtypes = [vp, C.c_int]
    lib.nmpc_set_timing.restype = C.c_int
    lib.nmpc_last_error.argtypes = [vp]
    lib.nmpc_last_error.restype = C.c_char_p
    lib.nmpc_build_hover_reference_device.argtypes = [vp, C.c_int, vp, vp, C.c_double, vp, vp, vp]
    lib.nmpc_build_hover_reference_device.restype = C.c_int
    lib.nmpc_odometry_to_state_device.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    lib.nmpc_odometry_to_state_device.restype = C.c_int
    lib.nmpc_commands_to_motor_speeds_device.argtypes = [vp, C.c_int, vp, C.c_double, C.c_double, C.c_double, vp, vp, vp]
    lib.nmpc_commands_to_motor_speeds_device.restype = C.c_int
    lib.nmpc_plant_step_device.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp]
    lib.nmpc_plant_step_device.restype = C.c_int
    lib.nmpc_hold_command_device.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    lib.nmpc_hold_command_device.restype = C.c_int
    lib.nmpc_hold_and_step_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    lib.nmpc_hold_and_step_device.restype = C.c_int
    lib.nmpc_adjoint_sensitivities_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    lib.nmpc_adjoint_sensitivities_device.restype = C.c_int
    lib.nmpc_kkt_report_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]
    lib.nmpc_kkt_report_device.restype = C.c_int
    lib.nmpc_block_factor_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_float), vp]
    lib.nmpc_block_factor_device.restype = C.c_int
    lib.nmpc_debug_factors.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    lib.nmpc_debug_factors.restype = C.c_int
    lib.nmpc_debug_tail_states.argtypes = [vp, C.c_int, C.POINTER(C.c_int32)]
    lib.nmpc_debug_tail_states.restype = C.c_int
    lib.nmpc_debug_guard_check.argtypes = [vp]
    lib.nmpc_debug_guard_check.restype = C.c_longlong
    lib.nmpc_debug_last_schedule.argtypes = [vp]
    lib.nmpc_debug_last_schedule.restype = C.c_int
    lib.nmpc_version.argtypes = []
    lib.nmpc_version.restype = C.c_char_p
    lib.nmpc_abi_sizes.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.nmpc_abi_sizes.restype = C.c_int
    cb, sb = C.c_int(0), C.c_int(0)
    lib.nmpc_abi_sizes(C.byref(cb), C.byref(sb))
    if cb.value != C.sizeof(NmpcConfig) or sb.value != C.sizeof(NmpcStats):
        raise RuntimeError(f"{path}: nmpc_config / nmpc_stats are {cb.value} / {sb.value} bytes in the library, "
                           f"{C.sizeof(NmpcConfig)} / {C.sizeof(NmpcStats)} in this binding - rebuild (make -C {CSRC})")
    _lib = lib
    return lib


def library_source_hash() -> str:
    """Hash of the kernel sources the loaded binary was built from (the tail of nmpc_version())."""
    v = load().nmpc_version().decode()
    return v.split("src ", 1)[1].split()[0] if "src " in v else "unknown"


def library_codegen() -> str:
    """Which builds of the kernels the loaded binary runs by default: "flag" (compiled with the two internal LLVM options of csrc/Makefile; they
    passed the build's codegen gate) or "default" (plain -O3: the gate found something, or hipcc is not the validated version)."""
    v = load().nmpc_version().decode()
    return v.split("codegen ", 1)[1].split()[0] if "codegen " in v else "flag"


def default_config(**over) -> NmpcConfig:
    cfg = NmpcConfig()
    load().nmpc_default_config(C.byref(cfg))
    return cfg.update(**over)
