"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI, against the CPU
oracle on identical seeded inputs.  FP64 tolerance: |u0 - oracle| <= 1e-9 absolute (thrusts are
O(1) N, so this is well inside the 1e-6 relative target of BASELINE.json); same algorithm on
both sides, so the observed differences are rounding-level (~1e-14).
"""
import numpy as np
import pytest

from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0

pytestmark = pytest.mark.gpu

TOL_U = 1e-9
TOL_X = 1e-8
WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)


def make_solver(**over):
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    over.setdefault("max_batch", 512)
    return NmpcOcpSolver(_lib.default_config(**over))


def oracle_cfg(polish=False, **over):
    """polish=True: the oracle with the same active-set polish as the team kernel (its default);
    the lane and condensed kernels are plain interior point."""
    return O.default_config(qp_gamma=0.0, qp_polish=1 if polish else 0, **over)


def hover(cfg):
    return hover_reference(cfg.N, cfg.mass * cfg.gravity / 4.0)


@pytest.mark.parametrize("mapping", ["lane", "team"])
@pytest.mark.parametrize("dist,seed", [(NEAR_HOVER, 0), (AGGRESSIVE, 1), (WILD, 2)])
@pytest.mark.parametrize("share", [True, False])
def test_cold_start_batch_matches_oracle(dist, seed, share, mapping):
    s = make_solver(flags=(1 if share else 0) | (_lib.FLAG_TEAM_MAPPING if mapping == "team" else 0))
    x0 = sample_x0(300, seed, **dist)          # ragged: not a multiple of the wave size
    yref, ye = hover(s.config)
    out = s.solve_batch(x0, yref, ye, want_traj=True)
    ref = O.solve_batch(oracle_cfg(polish=(mapping == "team")), x0, yref, ye, want_traj=True)
    assert (out["status"] == 0).all() and (ref["status"] == 0).all()
    np.testing.assert_allclose(out["u0"], ref["u0"], rtol=0, atol=TOL_U)
    np.testing.assert_allclose(out["x"], ref["x"], rtol=0, atol=TOL_X)
    np.testing.assert_allclose(out["u"], ref["u"], rtol=0, atol=TOL_X)
    st = s.stats()
    assert st["batch"] == 300 and st["n_status"][0] == 300
    assert abs(st["iter_mean"] - ref["iters"].mean()) < 0.05
    if mapping == "team":
        assert st["n_polished"] >= 290          # the active-set path really ran


@pytest.mark.parametrize("mapping", ["lane", "team"])
def test_per_instance_yref_and_warm_start(mapping):
    """[B,N,17] references + a second RTI from the previous solution (controller.py:419-424)."""
    s = make_solver(flags=1 | (_lib.FLAG_TEAM_MAPPING if mapping == "team" else 0))
    c = oracle_cfg(polish=(mapping == "team"))
    B = 96
    x0 = sample_x0(B, 5, **AGGRESSIVE)
    rng = np.random.default_rng(5)
    yref, ye = hover(s.config)
    yref = np.tile(yref, (B, 1, 1)); ye = np.tile(ye, (B, 1))
    yref[:, :, 0:3] += rng.normal(0, 0.2, (B, 1, 3))       # per-instance setpoints
    ye[:, 0:3] = yref[:, 0, 0:3]
    o1 = s.solve_batch(x0, yref, ye, want_traj=True)
    r1 = O.solve_batch(c, x0, yref, ye, want_traj=True)
    np.testing.assert_allclose(o1["u0"], r1["u0"], rtol=0, atol=TOL_U)
    x1 = x0 + rng.normal(0, 0.01, x0.shape)                # the plant moved a little
    o2 = s.solve_batch(x1, yref, ye, x_init=o1["x"], u_init=o1["u"], want_traj=True)
    r2 = O.solve_batch(c, x1, yref, ye, x_init=r1["x"], u_init=r1["u"], want_traj=True)
    assert (o2["status"] == 0).all()
    np.testing.assert_allclose(o2["u0"], r2["u0"], rtol=0, atol=TOL_U)
    np.testing.assert_allclose(o2["x"], r2["x"], rtol=0, atol=TOL_X)


@pytest.mark.parametrize("mapping", ["lane", "team"])
def test_golden_fixture(mapping):
    """Committed inputs/outputs (tests/golden/rti_cold_start.npz, made by make_golden.py)."""
    from pathlib import Path
    g = np.load(Path(__file__).parent / "golden" / "rti_cold_start.npz")
    s = make_solver(flags=1 | (_lib.FLAG_TEAM_MAPPING if mapping == "team" else 0))
    out = s.solve_batch(g["x0"], g["yref"], g["yref_e"], want_traj=True)
    sfx = "_polish" if mapping == "team" else ""     # the team kernel's default includes the active-set polish
    np.testing.assert_array_equal(out["status"], g["status"])
    np.testing.assert_allclose(out["u0"], g["u0" + sfx], rtol=0, atol=TOL_U)
    np.testing.assert_allclose(out["x"], g["x" + sfx], rtol=0, atol=TOL_X)


def test_known_answers_on_gpu():
    hov = 0.68 * 9.81 / 4.0
    xh = np.zeros(13); xh[2] = 1.0; xh[6] = 1.0
    # K1: hover, LM = 0 -> exactly m g / 4
    s = make_solver(levenberg_marquardt=0.0)
    yref, ye = hover(s.config)
    out = s.solve_batch(xh[None], yref, ye)
    np.testing.assert_allclose(out["u0"], hov, atol=1e-10)
    # K3: pure z offset -> four equal thrusts, 2.19447 N
    s = make_solver()
    x = xh.copy(); x[2] = 0.5
    out = s.solve_batch(x[None], yref, ye)
    assert np.ptp(out["u0"]) < 1e-12 and abs(out["u0"][0, 0] - 2.19447) < 1e-5


def test_status_paths_on_gpu():
    s = make_solver()
    yref, ye = hover(s.config)
    x0 = sample_x0(70, 7, **NEAR_HOVER)
    x0[3, 4] = np.nan                          # one poisoned lane must not hurt its wave
    out = s.solve_batch(x0, yref, ye)
    ref = O.solve_batch(oracle_cfg(polish=True), x0, yref, ye)
    assert out["status"][3] == 1 and (np.delete(out["status"], 3) == 0).all()
    np.testing.assert_array_equal(out["u0"][3], 0.0)        # controller.py:448-450
    np.testing.assert_allclose(np.delete(out["u0"], 3, 0), np.delete(ref["u0"], 3, 0), atol=TOL_U)
    # iteration cap 1: tolerated like acados RTI, result = oracle's one-iteration result
    s1 = make_solver(qp_iter_max=1, qp_polish=0)
    o = s1.solve_batch(x0[:3], yref, ye)
    r = O.solve_batch(oracle_cfg(qp_iter_max=1), x0[:3], yref, ye)
    np.testing.assert_array_equal(o["status"], r["status"])
    np.testing.assert_allclose(o["u0"][:3][o["status"] == 0], r["u0"][:3][r["status"] == 0], atol=TOL_U)


def test_single_instance_set_solve_get_surface():
    """The exact call sequence of PositionNMPC.solve (controller.py:412-460)."""
    s = make_solver(max_batch=1)
    c = oracle_cfg(polish=True)
    N = s.N
    yref, ye = hover(s.config)
    x0 = sample_x0(1, 9, **AGGRESSIVE)[0]
    s.set(0, "lbx", x0); s.set(0, "ubx", x0); s.set(0, "x", x0)
    s.set(0, "u", np.zeros(4))
    for k in range(1, N):
        s.set(k, "x", x0); s.set(k, "u", np.zeros(4))
    s.set(N, "x", x0)
    for k in range(N):
        s.set(k, "yref", yref[k])
    s.set(N, "yref", ye)
    assert s.solve() == 0
    u0 = s.get(0, "u")
    ref = O.solve_batch(c, x0[None], yref, ye, want_traj=True)
    np.testing.assert_allclose(u0, ref["u0"][0], atol=TOL_U)
    np.testing.assert_allclose(s.get(N, "x"), ref["x"][0, N], atol=TOL_X)
    np.testing.assert_allclose(s.get(N - 1, "u"), ref["u"][0, N - 1], atol=TOL_X)


def test_long_horizon_and_odd_sizes():
    """Horizons 1 and 2 (no checkpoint window, hoisted stage loops longer than the horizon), a few stages,
    a long horizon; batches that leave teams of the last wave idle; cold start, then a warm-started
    second iteration (per-stage linearisation)."""
    for N, B in ((1, 3), (2, 4), (3, 5), (60, 33)):
        s = make_solver(N=N, max_batch=64)
        c = oracle_cfg(polish=True, N=N)
        yref, ye = hover(s.config)
        x0 = sample_x0(B, N, **AGGRESSIVE)
        out = s.solve_batch(x0, yref, ye, want_traj=True)
        ref = O.solve_batch(c, x0, yref, ye, want_traj=True)
        assert np.array_equal(out["status"], ref["status"])
        np.testing.assert_allclose(out["u0"], ref["u0"], atol=TOL_U)
        np.testing.assert_allclose(out["x"], ref["x"], atol=TOL_X)
        x1 = x0 + np.random.default_rng(N).normal(0, 0.01, x0.shape)
        o2 = s.solve_batch(x1, yref, ye, x_init=out["x"], u_init=out["u"])
        r2 = O.solve_batch(c, x1, yref, ye, x_init=out["x"], u_init=out["u"])
        np.testing.assert_allclose(o2["u0"], r2["u0"], atol=TOL_U)


def test_full_size_properties_batch_4096():
    """BASELINE config 2 at full size: oracle on a sample, plus size-independent properties."""
    s = make_solver(max_batch=4096)
    yref, ye = hover(s.config)
    x0 = sample_x0(4096, 0, **NEAR_HOVER)
    out = s.solve_batch(x0, yref, ye, want_traj=True)
    assert (out["status"] == 0).all()
    lbu, ubu = np.array(s.config.lbu), np.array(s.config.ubu)
    assert (out["u"] >= lbu - 1e-12).all() and (out["u"] <= ubu + 1e-12).all()
    np.testing.assert_array_equal(out["x"][:, 0], x0)                    # x0 pin (U7)
    # permutation equivariance: instances are independent
    perm = np.random.default_rng(0).permutation(4096)
    out_p = s.solve_batch(x0[perm], yref, ye)
    np.testing.assert_array_equal(out_p["u0"], out["u0"][perm])
    # broadcast and materialised references agree bit for bit
    out_m = s.solve_batch(x0, np.tile(yref, (4096, 1, 1)), np.tile(ye, (4096, 1)))
    np.testing.assert_array_equal(out_m["u0"], out["u0"])
    idx = np.arange(0, 4096, 16)
    ref = O.solve_batch(oracle_cfg(polish=True), x0[idx], yref, ye)
    np.testing.assert_allclose(out["u0"][idx], ref["u0"], atol=TOL_U)


def test_fp32_buffers_keep_the_fp64_answer():
    """Config 3's buffers: NMPC_DTYPE_F32 (= F32IO since round 5) reads and writes float arrays, the arithmetic stays FP64 -
    1e-6 N on u0 against the oracle on the same float-representable inputs (the all-FP32 kernel this replaced was tested at 5e-3)."""
    s = make_solver(dtype=_lib.DTYPE_F32)
    yref, ye = hover(s.config)
    f = lambda a: np.asarray(a).astype(np.float32).astype(np.float64)    # noqa: E731
    x0 = f(sample_x0(256, 1, **NEAR_HOVER))
    out = s.solve_batch(x0, f(yref), f(ye))
    ref = O.solve_batch(oracle_cfg(polish=True), x0, f(yref), f(ye))
    assert (out["status"] == 0).all()
    assert np.abs(out["u0"] - ref["u0"]).max() < 1e-6


def test_argument_errors_raise():
    from rotors_mpc_controller_amd.solver import NmpcError
    s = make_solver(max_batch=8)
    yref, ye = hover(s.config)
    with pytest.raises(NmpcError):
        s.solve_batch(sample_x0(9, 0), yref, ye)             # B > max_batch
    with pytest.raises(NmpcError):
        s.set(0, "nope", np.zeros(13))
    with pytest.raises(NmpcError):
        s.set(99, "x", np.zeros(13))
    with pytest.raises(NmpcError):
        s.set(0, "x", np.zeros(5))
    with pytest.raises(NmpcError):
        s.set(3, "lbx", np.zeros(13))


def test_late_interior_point_iterations_agree_with_the_oracle_on_the_round4_mismatch_draws():
    """The nine fuzz draws of rounds 3-4 (profiles/r04u_*, r04x_*: 13 instances; plus seed 431, see below) in which the oracle ended status 1 - NaN in its
    last interior-point iteration: a slack re-formed as u - lo had rounded to exactly 0 at mu = 1e-11 - where the kernels, same
    recursion with sums in another order, returned 0 "by rounding luck".  Both sides now carry the slacks as iterates and solve
    for the input step (HPIPM's form): on every instance of every one of these draws, cold and warm-started,
    the status and the number of interior-point iterations and active-set passes must be the oracle's, and no instance ends NaN.
    Seeds 431 and 3043 also hold the other mismatch of those campaigns: a warm start from a trajectory that has left the model's
    range (|x| 1e3 .. 8e5), whose first factorisation meets a pivot that is not positive - QP failure (status 4) on both sides now:
    the kernels used to report NaN (1) because the sweep that runs on after the failed pivot overflowed, the oracle because it looked
    for NaNs in the step a failed QP leaves behind (acados returns the QP failure first).  Since late round 5 the class of a failure no longer
    follows the arithmetic at all: status 1 exactly when an input of the instance is not finite (see the test of draw 11856 below)."""
    from tests.fuzz_draws import draw, oracle_config
    for seed in (1910, 3043, 3072, 3448, 4264, 4309, 4380, 4584, 5700, 431):
        over, x0, yref, ye, hov, _, _ = draw(seed)
        s = make_solver(**over)
        c = oracle_config(s.config, qp_polish=1)
        out = s.solve_batch(x0, yref, ye, want_traj=True)
        it1, ps1 = s.counts()
        ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=8)
        out2 = s.solve_batch(x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True)
        it2, ps2 = s.counts()
        ref2 = O.solve_batch(c, x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True, nthreads=8)
        scale = max(1.0, hov)
        for tag, o, r, it, ps in (("cold", out, ref, it1, ps1), ("warm", out2, ref2, it2, ps2)):
            assert (r["status"] != 1).all(), (seed, tag)         # finite inputs: no solve ends "NaN detected"
            np.testing.assert_array_equal(o["status"], r["status"], err_msg=f"seed {seed} {tag}")
            np.testing.assert_array_equal(it[: len(x0)], r["iters"], err_msg=f"seed {seed} {tag}: interior-point iterations")
            if seed != 431:      # (draw 431's open loop amplifies by 2^31 over the horizon: one of its instances spends its passes differently -
                                 # an acceptance test decided by rounding - and ends with the same status after the same iterations)
                np.testing.assert_array_equal(np.abs(ps[: len(x0)]), np.abs(r["passes"]), err_msg=f"seed {seed} {tag}: active-set passes")
            ok = r["status"] == 0
            assert np.abs(o["u0"][ok] - r["u0"][ok]).max() <= 1e-6 * scale, (seed, tag)
        s.close()


def test_the_class_of_a_failure_follows_the_inputs_not_the_arithmetic():
    """Fuzz draw 11856 (found late in round 5, profiles/r05_fuzz_parity_draws_11800_13599.txt): instance 97 ends its cold solve with status 0
    and a trajectory at |x| 5e10 (a wild tuning on an unstable discretisation), and the warm start about THAT trajectory linearises to
    numbers of magnitude 1e135: the first pivot of the first factorisation is 8.9e269.  The oracle's Cholesky went on to an exact zero
    (status 4), the kernels' L D L' to inf - inf (status 1); a range test on the pivots moved the disagreement to draw 431 (a pivot that
    is noise of either sign in front of one of 3e116).  On such data the arithmetic event that ends a solve is decided by rounding.  The
    class of a failed solve is therefore taken from what the caller handed in: status 1 exactly when an input of the instance (x0, yref,
    yref_e, the linearisation trajectory) is not finite, status 4 otherwise - on both sides, in every kernel.  Here: every status of the
    draw equal, cold and warm; instance 97 warm a QP failure; and the same instance with a NaN put into its warm start: status 1."""
    from tests.fuzz_draws import draw, oracle_config
    over, x0, yref, ye, hov, _, _ = draw(11856)
    s = make_solver(**over)
    c = oracle_config(s.config, qp_polish=1)
    out = s.solve_batch(x0, yref, ye, want_traj=True)
    ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=8)
    np.testing.assert_array_equal(out["status"], ref["status"])
    assert ref["status"][97] == 0 and np.abs(ref["x"][97]).max() > 1e10
    out2 = s.solve_batch(x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True)
    ref2 = O.solve_batch(c, x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True, nthreads=8)
    np.testing.assert_array_equal(out2["status"], ref2["status"])
    assert out2["status"][97] == 4 and (ref2["status"] != 1).all()
    assert np.array_equal(out2["u0"][97], np.zeros(4))
    ok = ref2["status"] == 0
    assert np.abs(out2["u0"][ok] - ref2["u0"][ok]).max() <= 1e-6 * max(1.0, hov)
    # a NaN in the warm start of two instances - one that failed anyway, one that solved: both are "NaN detected" now, nothing else moves
    xi, ui = ref["x"].copy(), ref["u"].copy()
    good = int(np.nonzero(ref2["status"] == 0)[0][0])
    xi[97, 3, 4] = np.nan
    ui[good, 1, 2] = np.nan
    out3 = s.solve_batch(x0, yref, ye, x_init=xi, u_init=ui)
    ref3 = O.solve_batch(c, x0, yref, ye, x_init=xi, u_init=ui, nthreads=8)
    np.testing.assert_array_equal(out3["status"], ref3["status"])
    assert out3["status"][97] == 1 and out3["status"][good] == 1
    others = np.ones(len(x0), bool); others[[97, good]] = False
    np.testing.assert_array_equal(out3["status"][others], out2["status"][others])
    s.close()


@pytest.mark.parametrize("seed", range(8))
def test_randomised_vehicle_tuning_and_references(seed):
    """What PositionNMPC.reconfigure can change (controller.py:63-172: mass, inertia, arm length, rotor constants ->
    thrust bounds and hover thrust; the weights; the horizon, dt and the integrator's step count), drawn at random, with
    per-instance moving references and tighter thrust boxes so that bounds are active: the GPU path (active-set kernel
    for 1-2 integrator steps, the general kernel for 3; both linearisation variants) against the oracle built from the
    same numbers."""
    from tests.oracle_solver import OracleOcpSolver
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([3, 9, 20, 31]))
    mass = float(rng.uniform(0.4, 3.0))
    arm = float(rng.uniform(0.1, 0.4))
    km = float(rng.uniform(0.005, 0.03))
    hov = mass * 9.81 / 4.0
    over = dict(N=N, dt=float(rng.choice([0.02, 0.05, 0.08])), mass=mass,
                inertia=[float(v) for v in rng.uniform(0.003, 0.03, 3) * mass],
                rotor_x=[arm, 0.0, -arm, 0.0], rotor_y=[0.0, arm, 0.0, -arm], rotor_z=[-km, km, -km, km],
                lbu=[float(hov * rng.uniform(0.0, 0.3))] * 4, ubu=[float(hov * rng.uniform(1.6, 3.5))] * 4,
                W=[float(v) for v in 10.0 ** rng.uniform(-2, 1.5, 17)], W_e=[float(v) for v in 10.0 ** rng.uniform(-1, 2, 13)],
                levenberg_marquardt=float(rng.choice([0.0, 7e-3, 0.1])), sim_num_steps=int(rng.choice([1, 2, 3])),
                lm_scaled_by_dt=int(rng.integers(0, 2)), cost_scaled_by_dt=int(rng.integers(0, 2)),
                flags=_lib.FLAG_TEAM_MAPPING | int(rng.integers(0, 2)), max_batch=128)
    s = make_solver(**over)
    c = OracleOcpSolver(s.config).c
    c.qp_polish = 1
    B = 96
    x0 = sample_x0(B, 2000 + seed, **AGGRESSIVE)
    # per-instance references: a setpoint that moves along the horizon, hover thrust of THIS vehicle
    yref = np.zeros((B, N, 17)); ye = np.zeros((B, 13))
    goal = rng.normal(0.0, 1.0, (B, 3)) + np.array([0.0, 0.0, 1.0])
    vel = rng.normal(0.0, 0.3, (B, 3))
    for k in range(N + 1):
        row = np.zeros((B, 13)); row[:, 0:3] = goal + vel * (k * over["dt"]); row[:, 3:6] = vel; row[:, 6] = 1.0
        if k < N:
            yref[:, k, :13] = row; yref[:, k, 13:] = hov
        else:
            ye[:] = row
    out = s.solve_batch(x0, yref, ye, want_traj=True)
    ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=8)
    np.testing.assert_array_equal(out["status"], ref["status"])
    ok = ref["status"] == 0
    assert ok.sum() >= B - 4
    scale = max(1.0, hov)
    np.testing.assert_allclose(out["u0"][ok], ref["u0"][ok], rtol=0, atol=TOL_U * scale)
    np.testing.assert_allclose(out["x"][ok], ref["x"][ok], rtol=0, atol=TOL_X * scale)
    np.testing.assert_allclose(out["u"][ok], ref["u"][ok], rtol=0, atol=TOL_X * scale)
    at_bound = (np.abs(out["u"][ok] - over["lbu"][0]) < 1e-9) | (np.abs(out["u"][ok] - over["ubu"][0]) < 1e-9)
    assert at_bound.any() or seed not in (0,)          # the draw exercises active bounds (checked on the first seed)
    # second tick, warm-started from the first (per-stage linearisation whatever the flag says)
    out2 = s.solve_batch(x0, yref, ye, x_init=out["x"], u_init=out["u"], want_traj=True)
    ref2 = O.solve_batch(c, x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True, nthreads=8)
    np.testing.assert_array_equal(out2["status"], ref2["status"])
    ok2 = ok & (ref2["status"] == 0)
    np.testing.assert_allclose(out2["u0"][ok2], ref2["u0"][ok2], rtol=0, atol=10 * TOL_U * scale)


@pytest.mark.parametrize("polish,N", [(1, 31), (0, 31), (1, 120), (0, 120)])
def test_unstable_discretised_open_loop_keeps_the_riccati_recursion_symmetric(polish, N):
    """dt = 0.1 with ONE integrator step, a light vehicle with a small inertia and body rates of several rad/s: the
    discretised open loop has a spectral radius of ~2.  The tile form of the backward sweep used to compute tile (i,j) and
    tile (j,i) of the cost-to-go independently; their rounding difference - an antisymmetric perturbation the recursion does
    not contract - grew by rho^2 per stage and ended as NaN / QP failures on 19 of 511 instances after 31 stages (and as
    wrong commands with status 0 on others), where the row form, the lane kernel and the oracle - one triangle of P - were
    fine (found by tools/dev/fuzz_parity.py, draw 21).  P is now kept exactly symmetric.
    N = 120: long saturated stretches of this plant are an open loop in the pinned recursion of an active-set pass (P grows
    by rho^2 per stage) and a pass can end in a NaN pivot; that is an attempt that failed - the interior point iteration
    takes over, as in the oracle - not status 1 (30 of 511 instances before the fix)."""
    over = dict(N=N, dt=0.1, mass=0.4738978976479069, inertia=[0.0017, 0.006, 0.012],
                rotor_x=[0.4541, 0.0, -0.4541, 0.0], rotor_y=[0.0, 0.4541, 0.0, -0.4541], rotor_z=[-0.0141, 0.0141, -0.0141, 0.0141],
                lbu=[0.0452] * 4, ubu=[2.0395] * 4,
                W=[0.1868, 4.7625, 0.1212, 70.9006, 3.7842, 57.0244, 0.0357, 32.6734, 12.5504, 0.1446, 29.7859, 5.8384, 0.409,
                   11.6702, 0.8986, 3.7607, 0.0134],
                W_e=[0.127, 13.7797, 13.6895, 2.7246, 3.6848, 4.8743, 0.2726, 140.7136, 0.5401, 23.0029, 17.7842, 0.149, 23.2557],
                levenberg_marquardt=0.0, sim_num_steps=1, lm_scaled_by_dt=1, cost_scaled_by_dt=1,
                flags=_lib.FLAG_TEAM_MAPPING, max_batch=512, qp_polish=polish)
    from tests.fuzz_draws import oracle_config
    s = make_solver(**over)
    c = oracle_config(s.config)
    B = 511
    x0 = sample_x0(B, 9021, **WILD)
    hov = over["mass"] * 9.81 / 4.0
    yref = np.zeros((N, 17)); yref[:, 2] = 1.0; yref[:, 6] = 1.0; yref[:, 13:] = hov
    ye = yref[0, :13].copy()
    out = s.solve_batch(x0, yref, ye)
    it, ps = s.counts()
    ref = O.solve_batch(c, x0, yref, ye, nthreads=8)
    ok = ref["status"] == 0
    # Where the accuracy certificate (qp_growth_max) stops trusting the factorisations - saturated stretches of this plant are an
    # open loop in which P grows by rho^2 = 4 per stage - the QP is reported as failed instead of solved to an unknown accuracy:
    # 1 instance at N = 31, 42 at N = 120.  Same verdict on both sides, instance by instance.
    assert ok.sum() >= (505 if N == 31 else 455)
    mis = np.nonzero(out["status"] != ref["status"])[0]
    assert (out["status"][ok] != 0).sum() == 0 and len(mis) == 0, (mis, out["status"][mis], ref["status"][mis])
    both = ok & (out["status"] == 0)
    acc = both & (ps > 0) & (ref["passes"] > 0)           # an accepted active-set solution on both sides: the exact QP solution
    np.testing.assert_allclose(out["u0"][acc], ref["u0"][acc], rtol=0, atol=1e-8)
    # ... the others end on the interior-point iterate on at least one side: converged to mu <= 1e-11 on trusted factorisations
    np.testing.assert_allclose(out["u0"][both], ref["u0"][both], rtol=0, atol=1e-6 * max(1.0, hov))
    if not polish:
        assert (it == ref["iters"]).mean() > 0.99        # the plain interior point takes the same iterations as the oracle's


def test_growth_certificate_gives_the_oracles_verdict_on_an_ill_conditioned_draw():
    """Draw 161 of tools/dev/fuzz_parity.py (rho(A) = 1.5, N = 40, three integrator steps): two of 63 instances saturate over
    most of the horizon; their pinned recursions lose 1e12 of accuracy and an accepted active-set answer was off by 2e-2 with
    status 0 (tests/test_oracle_qp.py has the 60-digit evidence).  With the certificate both sides refuse them."""
    from tests.fuzz_draws import draw, oracle_config
    over, x0, yref, ye, hov, _, _ = draw(161)
    for gmax, nbad in ((1e6, 2), (0.0, 0)):
        s = make_solver(**dict(over, qp_growth_max=gmax))
        out = s.solve_batch(x0, yref, ye)
        ref = O.solve_batch(oracle_config(s.config), x0, yref, ye, nthreads=8)
        np.testing.assert_array_equal(out["status"], ref["status"])
        assert (out["status"] == 4).sum() == nbad
        if gmax > 0:
            okk = out["status"] == 0
            np.testing.assert_allclose(out["u0"][okk], ref["u0"][okk], rtol=0, atol=1e-9 * max(1.0, hov))
        s.close()


def test_U10_switch_on_the_gpu():
    """nmpc_config.qp_maxiter_status (consumed at controller.py:448-450, nodes/mpc_controller_node:124): the QP stopped
    by qp_iter_max = 1 is tolerated (status 0, the iterate's command) or reported (status 2, command discarded, cold restart)."""
    yref, ye = hover(_lib.default_config())
    x0 = sample_x0(130, 6, **AGGRESSIVE)
    for polish in (0, 1):
        for sw in (0, 2):
            s = make_solver(qp_iter_max=1, qp_polish=polish, qp_polish_passes=1, qp_polish_budget=1, qp_maxiter_status=sw)
            c = oracle_cfg(polish=bool(polish), qp_iter_max=1, qp_polish_passes=1, qp_polish_budget=1, qp_maxiter_status=sw)
            out = s.solve_batch(x0, yref, ye, want_traj=True)
            ref = O.solve_batch(c, x0, yref, ye, want_traj=True)
            np.testing.assert_array_equal(out["status"], ref["status"])
            capped = s.iterations() == 1
            assert capped.sum() > (100 if not polish else 10)
            assert (out["status"][capped] == sw).all()
            np.testing.assert_allclose(out["u0"], ref["u0"], rtol=0, atol=TOL_U)
            np.testing.assert_allclose(out["x"], ref["x"], rtol=0, atol=TOL_X)
            if sw:
                assert (out["u0"][capped] == 0).all() and np.array_equal(out["x"][capped], np.tile(x0[capped][:, None, :], (1, 21, 1)))
            s.close()


@pytest.mark.parametrize("share", [1, 0])
def test_results_do_not_depend_on_wave_mates_with_trajectories_and_second_launch(share):
    """A batch in which some instances end on an accepted active-set pass and others on the interior-point iterate of the
    second launch (wild set), solved as drawn and permuted, trajectories wanted: every output bit-identical.  The general
    kernel used to choose its output path per wave - an accepted instance's state trajectory depended in its last bits on
    how its wave-mates had ended (found by tools/dev/fuzz_perm.py)."""
    s = make_solver(flags=_lib.FLAG_TEAM_MAPPING | share, max_batch=512)
    B = 511
    x0 = np.concatenate([sample_x0(300, 71, **WILD), sample_x0(B - 300, 72, **AGGRESSIVE)])
    yref, ye = hover(s.config)
    a = s.solve_batch(x0, yref, ye, want_traj=True)
    st = s.stats()
    assert st["iter_max"] > 0 and st["n_polished"] > B // 2          # both ways of ending are present
    perm = np.random.default_rng(73).permutation(B)
    b = s.solve_batch(x0[perm], yref, ye, want_traj=True)
    for key in ("u0", "status", "x", "u"):
        np.testing.assert_array_equal(a[key][perm], b[key])
    a2 = s.solve_batch(x0, yref, ye, x_init=a["x"], u_init=a["u"], want_traj=True)
    b2 = s.solve_batch(x0[perm], yref, ye, x_init=a["x"][perm], u_init=a["u"][perm], want_traj=True)
    for key in ("u0", "status", "x", "u"):
        np.testing.assert_array_equal(a2[key][perm], b2[key])


@pytest.mark.parametrize("N,cond_N", [(20, 5), (20, 3), (7, 5)])
def test_partial_condensing_kernel_matches_oracle_and_the_fast_path(N, cond_N):
    """SURVEY 8a7 on the GPU: k_cond_ipm (condense -> IPM on dense blocks -> expand) vs the oracle's
    condensed solve, and vs the team kernel on the uncondensed QP (solution invariance, U8)."""
    sc = make_solver(N=N, qp_cond_N=cond_N, flags=_lib.FLAG_CONDENSED_QP | 1, max_batch=128)
    st = make_solver(N=N, flags=_lib.FLAG_TEAM_MAPPING | 1, max_batch=128, qp_polish=0)
    yref, ye = hover(sc.config)
    x0 = sample_x0(100, 4, **AGGRESSIVE)
    oc = sc.solve_batch(x0, yref, ye, want_traj=True)
    ot = st.solve_batch(x0, yref, ye, want_traj=True)
    ref = O.solve_batch(oracle_cfg(N=N, qp_cond_N=cond_N), x0, yref, ye, want_traj=True)
    assert (oc["status"] == 0).all()
    np.testing.assert_allclose(oc["u0"], ref["u0"], rtol=0, atol=TOL_U)
    np.testing.assert_allclose(oc["x"], ref["x"], rtol=0, atol=TOL_X)
    np.testing.assert_allclose(oc["u0"], ot["u0"], rtol=0, atol=TOL_U)
    np.testing.assert_allclose(oc["u"], ot["u"], rtol=0, atol=TOL_X)


@pytest.mark.parametrize("share", [True, False])
def test_riccati_checkpoint_restart_is_transparent(share):
    """qp_polish_ckpt only decides where an active-set pass resumes its backward sweep: the stages it
    skips would reproduce the factors already stored, so every window size returns the same solution
    (and the same pass counts) as full sweeps, and all of them match the oracle's full sweeps."""
    x0 = sample_x0(512, 11, **AGGRESSIVE)
    outs = {}
    for ck in (0, 1, 3, 12, 19):
        s = make_solver(flags=(1 if share else 0) | _lib.FLAG_TEAM_MAPPING, qp_polish_ckpt=ck)
        yref, ye = hover(s.config)
        outs[ck] = s.solve_batch(x0, yref, ye, want_traj=True)
        st = s.stats()
        outs[ck]["polish"] = (st["polish_mean"], st["polish_max"], st["n_polished"])
    assert outs[0]["polish"][1] >= 3                         # the batch really needs several passes
    for ck in (1, 3, 12, 19):
        assert outs[ck]["polish"] == outs[0]["polish"]
        assert np.array_equal(outs[ck]["status"], outs[0]["status"])
        np.testing.assert_allclose(outs[ck]["u"], outs[0]["u"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(outs[ck]["x"], outs[0]["x"], rtol=0, atol=1e-12)
    ref = O.solve_batch(oracle_cfg(polish=True), x0, yref, ye)
    np.testing.assert_allclose(outs[12]["u0"], ref["u0"], rtol=0, atol=TOL_U)


def test_timing_switch_and_status_array():
    """nmpc_set_timing(0): no events, times read 0, results unchanged; the kernel writes the caller's
    status array itself."""
    s = make_solver()
    x0 = sample_x0(64, 3, **NEAR_HOVER)
    yref, ye = hover(s.config)
    a = s.solve_batch(x0, yref, ye)
    t_on = s.stats()["ms_solve"]
    s.set_timing(False)
    b = s.solve_batch(x0, yref, ye)
    st = s.stats()
    assert t_on > 0 and st["ms_solve"] == 0 and st["batch"] == 64 and st["n_status"][0] == 64
    assert np.array_equal(a["u0"], b["u0"]) and np.array_equal(a["status"], b["status"])


def test_plain_ipm_iteration_counts_match_oracle():
    """qp_polish = 0 on the team mapping (the general kernel, tile form): not only the solution but the number of
    Mehrotra iterations per instance is the oracle's (same start point, same step rule, same stopping test).  Guards the
    iteration statistics that bench.py prices the interior-point path with: a code-generation flag once left the results
    right and the counter at 1."""
    import torch
    s = make_solver(flags=1 | _lib.FLAG_TEAM_MAPPING, qp_polish=0)
    yref, ye = hover(s.config)
    x0 = np.concatenate([sample_x0(200, 31, **NEAR_HOVER), sample_x0(200, 32, **AGGRESSIVE)])
    out = s.solve_batch(x0, yref, ye)
    ref = O.solve_batch(oracle_cfg(), x0, yref, ye)
    np.testing.assert_allclose(out["u0"], ref["u0"], rtol=0, atol=TOL_U)
    it = torch.empty(400, dtype=torch.int32, device="cuda")
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    host = np.zeros(400, np.int32)
    assert hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(s.device_iterations_ptr()), 1600, 2) == 0
    del it
    st = s.stats()
    assert st["iter_max"] == ref["iters"].max() and abs(st["iter_mean"] - ref["iters"].mean()) < 0.02
    assert (host == ref["iters"]).mean() > 0.97          # (a rounding-level difference may move a stopping test by one step)


@pytest.mark.parametrize("share", [1, 0])
def test_flag_build_of_the_active_set_kernel_is_bit_equal_to_the_default_codegen_build(share, monkeypatch):
    """k_team_as is the one kernel built with -mllvm -amdgpu-mfma-vgpr-form (nmpc_as.hip): an internal compiler option, validated for this
    kernel only (nmpc_qp.hip and DESIGN.md section 4.2 have the story).  The same kernel is also built with the
    default code generation; NMPC_AS_NOFLAG=1 selects that build.  Same arithmetic, different register placement: every output
    bit must agree - cold, warm-started, trajectories, pass statistics."""
    yref, ye = hover(_lib.default_config())
    x0 = np.concatenate([sample_x0(200, 81, **NEAR_HOVER), sample_x0(200, 82, **AGGRESSIVE), sample_x0(111, 83, **WILD)])
    res = []
    for noflag in ("0", "1"):
        monkeypatch.setenv("NMPC_AS_NOFLAG", noflag)
        s = make_solver(flags=_lib.FLAG_TEAM_MAPPING | share)
        a = s.solve_batch(x0, yref, ye, want_traj=True)
        pa = s.passes()
        b = s.solve_batch(x0, yref, ye, x_init=a["x"], u_init=a["u"], want_traj=True)
        res.append((a, pa, b, s.passes()))
        s.close()
    (a0, p0, b0, q0), (a1, p1, b1, q1) = res
    for key in ("u0", "status", "x", "u"):
        np.testing.assert_array_equal(a0[key], a1[key])
        np.testing.assert_array_equal(b0[key], b1[key])
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(q0, q1)


@pytest.mark.parametrize("polish", [0, 1])
@pytest.mark.parametrize("steps", [3, 4])
def test_three_and_four_integrator_steps_per_stage_linearisation_with_trajectories(steps, polish, monkeypatch):
    """Regression test of the one GPU memory fault this repository has recorded (round 3, gpurun_out/qp_check2.log: k_team_qp<per-stage,
    trajectories>, sim_num_steps = 4, qp_polish = 0, B = 256, aggressive seed 8 - then built with -mllvm -amdgpu-mfma-vgpr-form; DESIGN.md
    section 4.2 has what was examined).  More than two integrator steps take EvLayout<4> of the evaluation-point buffer (half the
    stages per chunk, twice the room per stage; include/rotors_nmpc.h accepts sim_num_steps <= 4, controller.py:188 ships 2).
    Per-stage linearisation + trajectories, plain interior point (k_team_qp) and the default path (k_team_as + k_team_qp_list), cold
    and warm-started, against the oracle at 1e-9; statuses and interior-point iteration counts equal.  The handle's buffers sit
    between 64 KiB canary bands (NMPC_GUARD): no kernel of either solve stores outside a buffer."""
    B = 256
    monkeypatch.setenv("NMPC_GUARD", "64")
    s = make_solver(sim_num_steps=steps, qp_polish=polish, flags=_lib.FLAG_TEAM_MAPPING, max_batch=B)
    c = oracle_cfg(polish=bool(polish), sim_num_steps=steps)
    yref, ye = hover(s.config)
    x0 = sample_x0(B, 8, **AGGRESSIVE)
    o1 = s.solve_batch(x0, yref, ye, want_traj=True)
    it1 = s.iterations()
    r1 = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=8)
    o2 = s.solve_batch(x0, yref, ye, x_init=o1["x"], u_init=o1["u"], want_traj=True)
    it2 = s.iterations()
    r2 = O.solve_batch(c, x0, yref, ye, x_init=r1["x"], u_init=r1["u"], want_traj=True, nthreads=8)
    for o, r, it in ((o1, r1, it1), (o2, r2, it2)):
        np.testing.assert_array_equal(o["status"], r["status"])
        assert (o["status"] == 0).all()
        np.testing.assert_allclose(o["u0"], r["u0"], rtol=0, atol=TOL_U)
        np.testing.assert_allclose(o["x"], r["x"], rtol=0, atol=TOL_X)
        np.testing.assert_allclose(o["u"], r["u"], rtol=0, atol=TOL_X)
        np.testing.assert_array_equal(it, r["iters"])
    assert s.guard_check() == 0
    s.close()


@pytest.mark.parametrize("share", [1, 0])
def test_flag_build_of_the_interior_point_kernel_is_bit_equal_to_the_default_codegen_build(share, monkeypatch):
    """The kernels that iterate the interior point method - k_team_qp (whole QP, whole batch: qp_polish = 0, or an attempt schedule that
    starts with iterations) and k_team_qp_list (the work list of the default path) - are built twice since round 4: nmpc_qpf.hip with -mllvm -amdgpu-mfma-vgpr-form (12 % fewer instructions: no accumulation-register moves),
    nmpc_qp.hip with the default code generation (NMPC_QP_NOFLAG=1).  Same arithmetic, different register placement: every output bit and
    every iteration count must agree - plain interior point and the single-kernel attempt schedule, cold and warm-started, trajectories."""
    yref, ye = hover(_lib.default_config())
    x0 = np.concatenate([sample_x0(200, 91, **NEAR_HOVER), sample_x0(200, 92, **AGGRESSIVE), sample_x0(111, 93, **WILD)])
    # plain interior point | single-kernel attempt schedule (both k_team_qp) | default split path with tight attempts (k_team_qp_list drains
    # a well-filled work list: the wild third of the sample)
    for over in (dict(qp_polish=0), dict(qp_polish=1, qp_polish_mu=1e-2), dict(qp_polish=1, qp_polish_passes=3, qp_polish_budget=6)):
        res = []
        # (failed first attempts are continued inside k_team_as by default: the work-list launch is what this case is about)
        monkeypatch.setenv("NMPC_TEAM_INPLACE", "0" if "qp_polish_passes" in over else "1")
        for noflag in ("0", "1"):
            monkeypatch.setenv("NMPC_QP_NOFLAG", noflag)
            s = make_solver(flags=_lib.FLAG_TEAM_MAPPING | share, **over)
            a = s.solve_batch(x0, yref, ye, want_traj=True)
            ia = s.iterations()
            b = s.solve_batch(x0, yref, ye, x_init=a["x"], u_init=a["u"], want_traj=True)
            res.append((a, ia, b, s.iterations()))
            s.close()
        (a0, i0, b0, j0), (a1, i1, b1, j1) = res
        for key in ("u0", "status", "x", "u"):
            np.testing.assert_array_equal(a0[key], a1[key])
            np.testing.assert_array_equal(b0[key], b1[key])
        np.testing.assert_array_equal(i0, i1)
        np.testing.assert_array_equal(j0, j1)
        assert i0.max() >= 3                      # (interior-point iterations really ran)


@pytest.mark.parametrize("share,traj,dtype", [(1, False, "f64"), (1, True, "f64"), (0, True, "f64"), (0, False, "f64"), (1, True, "f32io"), (0, True, "f32io")])
def test_continuing_failed_attempts_inside_k_team_as_gives_the_bits_of_the_work_list_launch(share, traj, dtype, monkeypatch):
    """Default since round 4: a team whose first active-set attempt fails is continued at once by the wave that made the attempt (team_as MODE 2
    inside k_team_as) instead of being appended to a work list for a second launch (NMPC_TEAM_INPLACE=0; the per-stage build without
    trajectories and long horizons keep the list).  Which wave continues an instance, and beside which wave-mates, must not show in a
    single bit: outputs, trajectories, statuses, iteration and pass counts of the two schedules - tight pass budgets, so that the wild and
    aggressive parts of the sample fail their first attempts by the hundred - cold and warm-started."""
    yref, ye = hover(_lib.default_config())
    x0 = np.concatenate([sample_x0(300, 71, **NEAR_HOVER), sample_x0(300, 72, **AGGRESSIVE), sample_x0(211, 73, **WILD)])
    for over in (dict(qp_polish_passes=1, qp_polish_budget=2), dict(qp_polish_passes=3, qp_polish_budget=6), dict()):
        res = []
        for inplace in ("1", "0"):
            monkeypatch.setenv("NMPC_TEAM_INPLACE", inplace)
            s = make_solver(max_batch=len(x0), flags=_lib.FLAG_TEAM_MAPPING | share, dtype=_lib.DTYPE_F32IO if dtype == "f32io" else _lib.DTYPE_F64, **over)
            a = s.solve_batch(x0, yref, ye, want_traj=traj)
            assert s.last_schedule() == dict(split=True, inplace=inplace == "1", tail=False)      # (nmpc_debug_last_schedule: what ran)
            ia, pa = s.iterations(), s.passes()
            xi, ui = (a["x"], a["u"]) if traj else (None, None)
            if xi is None:
                t = s.solve_batch(x0, yref, ye, want_traj=True)
                xi, ui = t["x"], t["u"]
            b = s.solve_batch(x0, yref, ye, x_init=xi, u_init=ui, want_traj=traj)
            res.append((a, ia, pa, b, s.iterations(), s.passes()))
            s.close()
        (a0, i0, p0, b0, j0, q0), (a1, i1, p1, b1, j1, q1) = res
        for key in ("u0", "status") + (("x", "u") if traj else ()):
            np.testing.assert_array_equal(a0[key], a1[key])
            np.testing.assert_array_equal(b0[key], b1[key])
        for u, v in ((i0, i1), (p0, p1), (j0, j1), (q0, q1)):
            np.testing.assert_array_equal(u, v)
        if over:
            assert (i0 > 0).sum() >= 100              # (first attempts really failed, by the hundred)
