#!/usr/bin/env python3
"""Throws invalid arguments at every C-ABI entry point: each call must return a negative code (never crash), leave a message
in nmpc_last_error, and the handle must still solve afterwards.  Prints one line per call; exit code 1 if any call was
accepted.  (tests/test_gpu_configs.py runs it in a child process.)"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)

lib = _lib.load()
cfg = _lib.default_config(max_batch=8)
s = NmpcOcpSolver(cfg)
h = s._h
dev = torch.device("cuda", 0)
B = 8
x0 = torch.from_numpy(sample_x0(B, 0, **NEAR_HOVER)).to(dev)
yr_h, ye_h = hover_reference(cfg.N, cfg.mass * cfg.gravity / 4)
yr = torch.from_numpy(np.broadcast_to(yr_h, (B,) + yr_h.shape).copy()).to(dev); ye = torch.from_numpy(np.broadcast_to(ye_h, (B, 13)).copy()).to(dev)
u0 = torch.zeros(B, 4, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
xt = torch.zeros(B, cfg.N + 1, 13, dtype=torch.float64, device=dev); ut = torch.zeros(B, cfg.N, 4, dtype=torch.float64, device=dev)
buf = torch.zeros(B, 32, dtype=torch.float64, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
N0 = None
accepted = 0


def expect_error(name, rc):
    global accepted
    msg = lib.nmpc_last_error(h).decode()
    ok = rc < 0
    accepted += not ok
    print(f"{'ok  ' if ok else 'ACCEPTED'} {name}: rc {rc} '{msg[:70]}'")


dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
z13 = np.zeros(13)
expect_error("set: unknown field", lib.nmpc_set(h, 0, b"nope", dp(z13), 13))
expect_error("set: stage out of range", lib.nmpc_set(h, cfg.N + 5, b"x", dp(z13), 13))
expect_error("set: negative stage", lib.nmpc_set(h, -1, b"x", dp(z13), 13))
expect_error("set: wrong size", lib.nmpc_set(h, 0, b"x", dp(z13), 5))
expect_error("set: null value", lib.nmpc_set(h, 0, b"x", None, 13))
expect_error("set: null field", lib.nmpc_set(h, 0, None, dp(z13), 13))
expect_error("get: unknown field", lib.nmpc_get(h, 0, b"yref", dp(z13), 13))
expect_error("get: null out", lib.nmpc_get(h, 0, b"x", None, 13))
expect_error("get: u at stage N", lib.nmpc_get(h, cfg.N, b"u", dp(z13), 4))
hx0 = np.zeros((B, 13)); hy = np.zeros((B, cfg.N, 17)); hye = np.zeros((B, 13)); hu0 = np.zeros((B, 4)); hst = np.zeros(B, np.int32)
ip = hst.ctypes.data_as(C.POINTER(C.c_int32))
expect_error("solve_batch: B = 0", lib.nmpc_solve_batch(h, 0, dp(hx0), dp(hy), dp(hye), 0, None, None, dp(hu0), ip, None, None))
expect_error("solve_batch: B > max_batch", lib.nmpc_solve_batch(h, 9, dp(hx0), dp(hy), dp(hye), 0, None, None, dp(hu0), ip, None, None))
expect_error("solve_batch: null x0", lib.nmpc_solve_batch(h, B, None, dp(hy), dp(hye), 0, None, None, dp(hu0), ip, None, None))
expect_error("solve_batch: null yref", lib.nmpc_solve_batch(h, B, dp(hx0), None, dp(hye), 0, None, None, dp(hu0), ip, None, None))
expect_error("solve_batch: null u0", lib.nmpc_solve_batch(h, B, dp(hx0), dp(hy), dp(hye), 0, None, None, None, ip, None, None))
expect_error("solve_batch: x_init without u_init", lib.nmpc_solve_batch(h, B, dp(hx0), dp(hy), dp(hye), 0, dp(np.zeros((B, cfg.N + 1, 13))), None, dp(hu0), ip, None, None))
expect_error("solve_batch_device: B = -1", lib.nmpc_solve_batch_device(h, -1, p(x0), p(yr), p(ye), 0, N0, N0, p(u0), p(st), N0, N0, N0))
expect_error("solve_batch_device: null x0", lib.nmpc_solve_batch_device(h, B, N0, p(yr), p(ye), 0, N0, N0, p(u0), p(st), N0, N0, N0))
expect_error("solve_batch_device: null yref_e", lib.nmpc_solve_batch_device(h, B, p(x0), p(yr), N0, 0, N0, N0, p(u0), p(st), N0, N0, N0))
expect_error("solve_batch_device: u_init without x_init", lib.nmpc_solve_batch_device(h, B, p(x0), p(yr), p(ye), 0, N0, p(ut), p(u0), p(st), N0, N0, N0))
expect_error("hover_reference: null out", lib.nmpc_build_hover_reference_device(h, B, p(buf), p(buf), 1.0, N0, N0, N0))
expect_error("odometry: null pose", lib.nmpc_odometry_to_state_device(h, B, N0, p(buf), p(buf), N0))
expect_error("motor speeds: B = 0", lib.nmpc_commands_to_motor_speeds_device(h, 0, p(u0), 8.5e-6, 50.0, 800.0, p(buf), N0, N0))
expect_error("hold: null status", lib.nmpc_hold_command_device(h, B, p(u0), N0, p(buf), N0))
expect_error("plant: null u", lib.nmpc_plant_step_device(h, B, p(x0), N0, p(buf), 1, N0))
expect_error("hold_and_step: null x", lib.nmpc_hold_and_step_device(h, B, p(u0), p(st), p(buf), N0, 1, N0))
expect_error("adjoint: null lam", lib.nmpc_adjoint_sensitivities_device(h, B, p(x0), p(u0), N0, p(buf), 0, N0))
lib.nmpc_get_stats.restype = C.c_int
expect_error("get_stats: null out", lib.nmpc_get_stats(h, None))
# null handle: a negative code, no crash
for name, call in (("solve(null)", lambda: lib.nmpc_solve(None)), ("set(null)", lambda: lib.nmpc_set(None, 0, b"x", dp(z13), 13)),
                   ("solve_batch_device(null)", lambda: lib.nmpc_solve_batch_device(None, B, p(x0), p(yr), p(ye), 0, N0, N0, p(u0), p(st), N0, N0, N0)),
                   ("plant(null)", lambda: lib.nmpc_plant_step_device(None, B, p(x0), p(u0), p(buf), 1, N0))):
    rc = call()
    accepted += not (rc < 0)
    print(f"{'ok  ' if rc < 0 else 'ACCEPTED'} {name}: rc {rc}")
lib.nmpc_destroy(None)                                  # allowed, a no-op
# bad configurations are refused by nmpc_create with a message
for over in (dict(N=0), dict(dt=0.0), dict(mass=-1.0), dict(max_batch=0), dict(sim_num_stages=4), dict(lbu=[3.0] * 4, ubu=[1.0] * 4), dict(W=[-1.0] * 17), dict(dtype=7)):
    try:
        NmpcOcpSolver(_lib.default_config(**over))
        accepted += 1
        print("ACCEPTED create", over)
    except Exception as e:
        print(f"ok   create {over}: '{str(e)[:70]}'")
# ... and the handle still works
lib.nmpc_solve_batch_device(h, B, p(x0), p(yr), p(ye), 0, N0, N0, p(u0), p(st), N0, N0, N0)
torch.cuda.synchronize()
assert (st.cpu().numpy() == 0).all() and np.isfinite(u0.cpu().numpy()).all() and (u0.cpu().numpy() > 0).all()
print("handle still solves; accepted invalid calls:", accepted)
sys.exit(1 if accepted else 0)
