"""The kernel bodies (nmpc_lane.hpp / nmpc_ipm.hpp) compiled for the host and run lane by lane,
against the oracle and the golden fixture -- CPU only.  This checks the ARITHMETIC of the HIP
kernels (structured Jacobians, packed Riccati, lazy IPM update); the GPU execution itself is
covered by tests/test_gpu_parity.py.  tests/hostsim is test-only code, not a product path."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0
from tests import hostsim as H

GOLD = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("dist,seed", [(NEAR_HOVER, 0), (AGGRESSIVE, 1)])
def test_kernel_arithmetic_matches_oracle(flags, dist, seed):
    cfg = _lib.default_config(flags=flags)
    x0 = sample_x0(40, seed, **dist)
    yref, ye = hover_reference(cfg.N, cfg.mass * cfg.gravity / 4.0)
    out = H.solve_batch(cfg, x0, yref, ye)
    ref = O.solve_batch(O.default_config(qp_gamma=0.0), x0, yref, ye, want_traj=True)
    np.testing.assert_array_equal(out["status"], ref["status"])
    np.testing.assert_array_equal(out["iters"], ref["iters"])
    np.testing.assert_allclose(out["u0"], ref["u0"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out["x"], ref["x"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(out["u"], ref["u"], rtol=0, atol=1e-10)


def test_kernel_arithmetic_matches_golden_fixture():
    g = np.load(GOLD / "rti_cold_start.npz")
    out = H.solve_batch(_lib.default_config(), g["x0"], g["yref"], g["yref_e"])
    np.testing.assert_array_equal(out["status"], g["status"])
    np.testing.assert_allclose(out["u0"], g["u0"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(out["x"], g["x"], rtol=0, atol=1e-9)


def test_oracle_reproduces_its_own_golden_fixture():
    g = np.load(GOLD / "rti_cold_start.npz")
    ref = O.solve_batch(O.default_config(qp_gamma=0.0), g["x0"], g["yref"], g["yref_e"], want_traj=True)
    np.testing.assert_array_equal(ref["iters"], g["iters"])
    np.testing.assert_allclose(ref["u0"], g["u0"], rtol=0, atol=1e-12)


def test_warm_start_and_switches():
    c = O.default_config(qp_gamma=0.0, lm_scaled_by_dt=0, cost_scaled_by_dt=0)
    cfg = _lib.default_config(lm_scaled_by_dt=0, cost_scaled_by_dt=0)
    yref, ye = hover_reference(cfg.N, cfg.mass * cfg.gravity / 4.0)
    x0 = sample_x0(8, 3, **AGGRESSIVE)
    r1 = O.solve_batch(c, x0, yref, ye, want_traj=True)
    o1 = H.solve_batch(cfg, x0, yref, ye)
    np.testing.assert_allclose(o1["u0"], r1["u0"], atol=1e-11)
    r2 = O.solve_batch(c, x0, yref, ye, x_init=r1["x"], u_init=r1["u"], want_traj=True)
    o2 = H.solve_batch(cfg, x0, yref, ye, x_init=o1["x"], u_init=o1["u"])
    np.testing.assert_allclose(o2["u0"], r2["u0"], atol=1e-10)
    np.testing.assert_allclose(o2["x"], r2["x"], atol=1e-9)


@pytest.mark.parametrize("N,cond_N,flags", [(20, 5, 4), (20, 5, 5), (20, 3, 4), (7, 5, 4), (20, 1, 4)])
def test_partial_condensing_path_matches_oracle(N, cond_N, flags):
    """condensing + IPM on the condensed QP (what acados hands to HPIPM, controller.py:181,184)
    gives the oracle's condensed result and the uncondensed result (U8)."""
    cfg = _lib.default_config(N=N, qp_cond_N=cond_N, flags=flags)
    yref, ye = hover_reference(N, cfg.mass * cfg.gravity / 4.0)
    x0 = sample_x0(10, 3, **AGGRESSIVE)
    out = H.solve_batch(cfg, x0, yref, ye)
    ref_c = O.solve_batch(O.default_config(qp_gamma=0.0, N=N, qp_cond_N=cond_N), x0, yref, ye, want_traj=True)
    ref_u = O.solve_batch(O.default_config(qp_gamma=0.0, N=N), x0, yref, ye)
    np.testing.assert_array_equal(out["iters"], ref_c["iters"])
    np.testing.assert_allclose(out["u0"], ref_c["u0"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out["x"], ref_c["x"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(out["u0"], ref_u["u0"], rtol=0, atol=1e-9)
