"""Pins the oracle's model, Jacobians and integrator (K4, K6) -- CPU only.

The symbolic model below is typed from the *mathematical* definition in SURVEY 2.1 with
sympy (an independent route: the oracle's Jacobians are hand-derived C).
"""
import numpy as np
import pytest
import sympy as sp
from scipy.integrate import solve_ivp

from oracle import oracle as O

NX, NU = 13, 4


@pytest.fixture(scope="module")
def cfg():
    return O.default_config()


@pytest.fixture(scope="module")
def sym_model(cfg):
    x = sp.symbols("x0:13")
    u = sp.symbols("u0:4")
    m, g = cfg.mass, cfg.gravity
    Jx, Jy, Jz = list(cfg.J)
    rx, ry, rz = list(cfg.rotor_x), list(cfg.rotor_y), list(cfg.rotor_z)
    qw, qx, qy, qz = x[6:10]
    wx, wy, wz = x[10:13]
    T = sum(u) / m
    tau = [sum(u[i] * ry[i] for i in range(4)), sum(-u[i] * rx[i] for i in range(4)),
           sum(u[i] * rz[i] for i in range(4))]
    f = sp.Matrix([
        x[3], x[4], x[5],
        2 * (qx * qz + qw * qy) * T, 2 * (qy * qz - qw * qx) * T, (1 - 2 * (qx**2 + qy**2)) * T - g,
        sp.Rational(1, 2) * (-qx * wx - qy * wy - qz * wz),
        sp.Rational(1, 2) * (qw * wx + qy * wz - qz * wy),
        sp.Rational(1, 2) * (qw * wy + qz * wx - qx * wz),
        sp.Rational(1, 2) * (qw * wz + qx * wy - qy * wx),
        (tau[0] - (wy * Jz * wz - wz * Jy * wy)) / Jx,
        (tau[1] - (wz * Jx * wx - wx * Jz * wz)) / Jy,
        (tau[2] - (wx * Jy * wy - wy * Jx * wx)) / Jz,
    ])
    args = list(x) + list(u)
    return (sp.lambdify(args, f, "numpy"), sp.lambdify(args, f.jacobian(list(x)), "numpy"),
            sp.lambdify(args, f.jacobian(list(u)), "numpy"), f.jacobian(list(x)), f.jacobian(list(u)))


def _rand_xu(rng):
    x = rng.normal(size=NX)
    x[6:10] /= np.linalg.norm(x[6:10])
    u = rng.uniform(0.0, 6.0, NU)
    return x, u


def test_f_and_jacobians_match_sympy(cfg, sym_model):
    f_s, fx_s, fu_s, _, _ = sym_model
    rng = np.random.default_rng(0)
    for _ in range(20):
        x, u = _rand_xu(rng)
        a = list(x) + list(u)
        np.testing.assert_allclose(O.model_f(cfg, x, u), np.asarray(f_s(*a), float).ravel(), rtol=1e-13, atol=1e-13)
        fx, fu = O.model_jac(cfg, x, u)
        np.testing.assert_allclose(fx, np.asarray(fx_s(*a), float), rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(fu, np.asarray(fu_s(*a), float), rtol=1e-13, atol=1e-13)


def test_jacobian_sparsity_counts(sym_model):
    """41/169 structural non-zeros in f_x (38 with Jxx == Jyy), 20/52 in f_u."""
    _, _, _, Jx, Ju = sym_model
    nnz_x = sum(1 for e in Jx if e != 0)
    nnz_u = sum(1 for e in Ju if e != 0)
    assert nnz_x in (38, 41)      # omega_z row vanishes identically when Jxx == Jyy
    assert nnz_u == 20


def test_hover_is_equilibrium(cfg):
    x = np.zeros(NX); x[2] = 1.0; x[6] = 1.0
    u = np.full(NU, cfg.mass * cfg.gravity / 4.0)
    np.testing.assert_allclose(O.model_f(cfg, x, u), 0.0, atol=1e-14)


def test_vde_forw_and_adj_are_consistent(cfg):
    rng = np.random.default_rng(1)
    x, u = _rand_xu(rng)
    Sx, Su = rng.normal(size=(NX, NX)), rng.normal(size=(NX, NU))
    fx, fu = O.model_jac(cfg, x, u)
    xd, Sxd, Sud = O.vde_forw(cfg, x, Sx, Su, u)
    np.testing.assert_allclose(xd, O.model_f(cfg, x, u))
    np.testing.assert_allclose(Sxd, fx @ Sx, atol=1e-13)
    np.testing.assert_allclose(Sud, fx @ Su + fu, atol=1e-13)
    lam = rng.normal(size=NX)
    np.testing.assert_allclose(O.vde_adj(cfg, x, lam, u), np.concatenate([fx.T @ lam, fu.T @ lam]), atol=1e-13)


def test_erk_sensitivities_match_finite_differences(cfg):
    """U3: the propagated Sx, Su are the exact Jacobians of the discrete map."""
    rng = np.random.default_rng(2)
    for _ in range(3):
        x, u = _rand_xu(rng)
        xn, A, B = O.integrate(cfg, x, u)
        eps = 1e-6
        Afd, Bfd = np.zeros((NX, NX)), np.zeros((NX, NU))
        for j in range(NX):
            d = np.zeros(NX); d[j] = eps
            Afd[:, j] = (O.integrate(cfg, x + d, u)[0] - O.integrate(cfg, x - d, u)[0]) / (2 * eps)
        for j in range(NU):
            d = np.zeros(NU); d[j] = eps
            Bfd[:, j] = (O.integrate(cfg, x, u + d)[0] - O.integrate(cfg, x, u - d)[0]) / (2 * eps)
        np.testing.assert_allclose(A, Afd, atol=2e-9)
        np.testing.assert_allclose(B, Bfd, atol=2e-9)


def test_erk_is_two_explicit_midpoint_steps(cfg):
    """U2: num_stages=2, num_steps=2 == two midpoint steps of h = dt/2, closed form of U3."""
    rng = np.random.default_rng(3)
    x, u = _rand_xu(rng)
    h = cfg.dt / 2
    xx, Phi, Gam = x.copy(), np.eye(NX), np.zeros((NX, NU))
    for _ in range(2):
        F1, G1 = O.model_jac(cfg, xx, u)
        xm = xx + 0.5 * h * O.model_f(cfg, xx, u)
        F2, G2 = O.model_jac(cfg, xm, u)
        Px = np.eye(NX) + h * F2 @ (np.eye(NX) + 0.5 * h * F1)
        Pu = h * (F2 @ (0.5 * h * G1) + G2)
        xx = xx + h * O.model_f(cfg, xm, u)
        Phi, Gam = Px @ Phi, Px @ Gam + Pu
    xn, A, B = O.integrate(cfg, x, u)
    np.testing.assert_allclose(xn, xx, atol=1e-14)
    np.testing.assert_allclose(A, Phi, atol=1e-14)
    np.testing.assert_allclose(B, Gam, atol=1e-14)


def test_discrete_jacobian_block_structure(cfg):
    """Structure the HIP kernels rely on: A = [[I, dt I, *, *],[0, I, *, *],[0,0,*,*],[0,0,0,*]]."""
    rng = np.random.default_rng(4)
    x, u = _rand_xu(rng)
    _, A, _ = O.integrate(cfg, x, u)
    np.testing.assert_array_equal(A[:, 0:3], np.eye(NX)[:, 0:3])
    np.testing.assert_allclose(A[0:3, 3:6], cfg.dt * np.eye(3), atol=1e-16)
    np.testing.assert_array_equal(A[3:6, 3:6], np.eye(3))
    np.testing.assert_array_equal(A[6:, 3:6], 0.0)
    np.testing.assert_array_equal(A[10:, 6:10], 0.0)


def test_integrator_is_second_order(cfg):
    """K6: midpoint x2 against a tight-tolerance adaptive integrator, error ~ O(h^2)."""
    rng = np.random.default_rng(5)
    x, u = _rand_xu(rng)
    errs = []
    for steps in (2, 4, 8):
        c = O.default_config(sim_num_steps=steps)
        xn, _, _ = O.integrate(c, x, u)
        ref = solve_ivp(lambda t, y: O.model_f(c, y, u), (0, c.dt), x, rtol=1e-12, atol=1e-14).y[:, -1]
        errs.append(np.linalg.norm(xn - ref))
    assert errs[0] < 5e-2
    assert 3.0 < errs[0] / errs[1] < 5.0 and 3.0 < errs[1] / errs[2] < 5.0


def test_rk4_tableau_also_available(cfg):
    rng = np.random.default_rng(6)
    x, u = _rand_xu(rng)
    c = O.default_config(sim_num_stages=4, sim_num_steps=1)
    xn, A, _ = O.integrate(c, x, u)
    ref = solve_ivp(lambda t, y: O.model_f(c, y, u), (0, c.dt), x, rtol=1e-12, atol=1e-14).y[:, -1]
    e4 = np.linalg.norm(xn - ref)
    c8 = O.default_config(sim_num_stages=4, sim_num_steps=2)
    e8 = np.linalg.norm(O.integrate(c8, x, u)[0] - ref)
    assert e4 < 2e-3 and 10.0 < e4 / e8 < 40.0     # fourth order: halving h gains ~16x
