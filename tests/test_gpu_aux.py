"""GPU tests of the steps either side of the solve and of a closed loop built
from them, against numpy restatements of the reference's node/reference code."""
import math

import numpy as np
import pytest

from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.reference import ReferenceGenerator, stack_yref
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, sample_x0

pytestmark = pytest.mark.gpu


def _solver(**over):
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    over.setdefault("max_batch", 256)
    over.setdefault("flags", _lib.FLAG_SHARE_COLD_START | _lib.FLAG_TEAM_MAPPING)
    return NmpcOcpSolver(_lib.default_config(**over))


def _odom_to_state_numpy(pose, twist):
    """nodes/mpc_controller_node:25-44,88-113,138-150 restated."""
    out = np.zeros((pose.shape[0], 13))
    for b, (p, tw) in enumerate(zip(pose, twist)):
        qx, qy, qz, qw = p[3:7]
        n = math.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
        roll = pitch = yaw = 0.0
        if n != 0.0:
            ax, ay, az, aw = qx / n, qy / n, qz / n, qw / n
            roll = math.atan2(2 * (aw * ax + ay * az), 1 - 2 * (ax * ax + ay * ay))
            sp = 2 * (aw * ay - az * ax)
            pitch = math.copysign(math.pi / 2, sp) if abs(sp) >= 1 else math.asin(sp)
            yaw = math.atan2(2 * (aw * az + ax * ay), 1 - 2 * (ay * ay + az * az))
        cr, sr, cp, sp_, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
        R = np.array([[cp * cy, cy * sp_ * sr - sy * cr, cy * sp_ * cr + sy * sr],
                      [cp * sy, sy * sp_ * sr + cy * cr, sy * sp_ * cr - cy * sr],
                      [-sp_, cp * sr, cp * cr]])
        out[b, 0:3] = p[0:3]
        out[b, 3:6] = R @ tw[0:3]
        out[b, 6:10] = [qw, qx, qy, qz]
        out[b, 10:13] = tw[3:6]
    return out


def test_hover_reference_builder_matches_reference_generator():
    import torch
    s = _solver()
    N, B = s.N, 37
    rng = np.random.default_rng(0)
    pos, yaw = rng.normal(0, 2, (B, 3)), rng.uniform(-3, 3, B)
    thrust = 1.6677
    d_pos, d_yaw = torch.from_numpy(pos).cuda(), torch.from_numpy(yaw).cuda()
    yref = torch.empty(B, N, 17, dtype=torch.float64, device="cuda")
    yref_e = torch.empty(B, 13, dtype=torch.float64, device="cuda")
    s.build_hover_reference_device(B, d_pos.data_ptr(), d_yaw.data_ptr(), thrust, yref.data_ptr(), yref_e.data_ptr())
    torch.cuda.synchronize()
    for b in (0, 5, B - 1):
        gen = ReferenceGenerator({})
        gen.set_target(position=pos[b], yaw=yaw[b])
        gen.update_hover_thrust(thrust)
        want, want_e = stack_yref(gen.build_horizon(N, 0.05), N)
        np.testing.assert_allclose(yref[b].cpu().numpy(), want, rtol=0, atol=1e-15)
        np.testing.assert_allclose(yref_e[b].cpu().numpy(), want_e, rtol=0, atol=1e-15)


def test_odometry_and_motor_speed_kernels_match_the_node_arithmetic():
    import torch
    s = _solver()
    B = 200
    rng = np.random.default_rng(1)
    q = rng.normal(size=(B, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    pose = np.concatenate([rng.normal(0, 2, (B, 3)), q], axis=1)
    twist = rng.normal(0, 1, (B, 6))
    d_pose, d_twist = torch.from_numpy(pose).cuda(), torch.from_numpy(twist).cuda()
    x0 = torch.empty(B, 13, dtype=torch.float64, device="cuda")
    s.odometry_to_state_device(B, d_pose.data_ptr(), d_twist.data_ptr(), x0.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(x0.cpu().numpy(), _odom_to_state_numpy(pose, twist), rtol=0, atol=1e-13)
    # thrust -> motor speed, nodes/mpc_controller_node:152-164
    kf, wmin, wmax = 8.54858e-6, 50.0, 838.0
    u = rng.uniform(-1.0, 8.0, (B, 4))
    d_u = torch.from_numpy(u).cuda()
    sp = torch.empty(B, 4, dtype=torch.float64, device="cuda")
    cl = torch.empty(B, 4, dtype=torch.float64, device="cuda")
    s.commands_to_motor_speeds_device(B, d_u.data_ptr(), kf, wmin, wmax, sp.data_ptr(), cl.data_ptr())
    torch.cuda.synchronize()
    lbu, ubu = np.array(s.config.lbu), np.array(s.config.ubu)
    clipped = np.clip(u, lbu, ubu)
    want = np.clip(np.sqrt(np.clip(clipped / max(kf, 1e-9), 0.0, wmax ** 2)), wmin, wmax)
    np.testing.assert_allclose(cl.cpu().numpy(), clipped, rtol=0, atol=0)
    np.testing.assert_allclose(sp.cpu().numpy(), want, rtol=1e-14, atol=0)


def test_closed_loop_rollout_stays_on_device_and_converges_to_hover():
    """solve -> apply u0 to the plant -> next x0, warm-started with the unshifted previous
    solution (controller.py:419-424), B instances in parallel; checked against the same loop run
    through the CPU oracle for a few instances, and for convergence to the setpoint."""
    import torch
    from oracle import oracle as O
    from rotors_mpc_controller_amd.rollout import ClosedLoopRollout
    s = _solver(max_batch=64)
    B, steps = 48, 25
    x0 = sample_x0(B, 3, **NEAR_HOVER)
    ro = ClosedLoopRollout(s, B)
    xs, us = ro.run(x0, steps, setpoint=(0.0, 0.0, 1.0), yaw=0.0)
    assert xs.shape == (steps + 1, B, 13) and us.shape == (steps, B, 4)
    # the same closed loop through the oracle (plant = the oracle's own ERK interval)
    c = O.default_config(qp_gamma=0.0, qp_polish=1)
    yref, ye = O.hover_yref(c)
    for b in (0, 17):
        x = x0[b].copy(); xt = ut = None
        for t in range(6):
            if xt is None:
                r = O.solve_batch(c, x[None], yref, ye, want_traj=True)
            else:
                r = O.solve_batch(c, x[None], yref, ye, x_init=xt, u_init=ut, want_traj=True)
            xt, ut = r["x"], r["u"]
            np.testing.assert_allclose(us[t, b], r["u0"][0], rtol=0, atol=1e-8)
            x = O.integrate(c, x, r["u0"][0])[0]
            x[6:10] /= np.linalg.norm(x[6:10])
            np.testing.assert_allclose(xs[t + 1, b], x, rtol=0, atol=1e-8)
    err0 = np.linalg.norm(xs[0, :, 0:3] - [0, 0, 1.0], axis=1)
    err1 = np.linalg.norm(xs[-1, :, 0:3] - [0, 0, 1.0], axis=1)
    assert (err1 < 0.5 * err0 + 0.02).all()


def test_closed_loop_hip_graph_replay_matches_eager_launches():
    """The tick pair (solve + plant step, one per warm-start buffer parity) replayed from a HIP graph ends
    in the same state and command as the eagerly launched loop."""
    from rotors_mpc_controller_amd.rollout import ClosedLoopRollout
    s = _solver(max_batch=64)
    s.set_timing(False)                      # no event records inside the captured region
    B, steps = 64, 21
    x0 = sample_x0(B, 9, **NEAR_HOVER)
    xs_e, us_e = ClosedLoopRollout(s, B).run(x0, steps)
    xs_g, us_g = ClosedLoopRollout(s, B).run(x0, steps, log=False, use_graph=True)
    np.testing.assert_allclose(xs_g[0], xs_e[-1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(us_g[0], us_e[-1], rtol=0, atol=1e-12)
