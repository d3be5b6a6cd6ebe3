"""Consumer of tests/golden/acados_rti.npz -- the file tools/make_acados_golden.py writes on a machine that has
acados.  It does not exist in this repository yet (acados, HPIPM, BLASFEO and CasADi are absent from the image and
from /root/reference: SURVEY 8c), so the tests on it SKIP with that reason and every accuracy figure of this
repository stays "vs the build's CPU oracle; acados parity unpinned".  Once the file is committed they pin, in
this order: (1) the oracle against acados on the fixture set, per switch setting of the three
version-dependent conventions; (2) the HIP path against acados (GPU test).

What is compared with what (round 5).  The file holds TWO acados solutions per state:
  * the TIGHT set (HPIPM's exit tolerances at 1e-12: the QP solved to rounding) DECIDES - the convention that reproduces it to
    1e-6 relative must exist and be the shipped default, and the HIP path must sit within 1e-6 relative of it;
  * the DEFAULT set (the reference's options verbatim: HPIPM stops on ~1e-8 residuals) is REPORTED, not asserted at 1e-6: its
    distance from the exact solution is acados' own accuracy floor, predicted by tests/test_acados_floor.py at 3e-7 N on this
    fixture set and up to 8e-6 N on the bench sample.  It is held to 1e-4 (sanity: a wrong convention moves u0 by 3e-4 .. 6e-3).
A file written before round 5 has only the default set: it then decides at 1e-6, as before.

The same checks run TODAY, end to end, on a file the generator writes against the Level-B stand-ins (tests/levelb: an
acados_template whose solver forwards to the CPU oracle, here with HPIPM's exit rule emulated - LEVELB_HPIPM_EXIT=1): that
validates generator + consumer, not acados parity.
"""
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).parent / "golden" / "acados_rti.npz"
REASON = ("tests/golden/acados_rti.npz not present: run tools/make_acados_golden.py where acados_template and casadi "
          "import (not possible in this image); parity with acados is unpinned until then")

needs_golden = pytest.mark.skipif(not GOLDEN.exists(), reason=REASON)


def _load():
    g = np.load(GOLDEN)
    ok = g["status"] == 0
    return g, ok


def _rel_err(a, b):
    return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())


def test_harness_script_is_importable_and_refuses_without_acados():
    """The generator must exist and fail cleanly (no partial file) where acados is missing."""
    import importlib.util
    import subprocess
    import sys
    script = Path(__file__).resolve().parent.parent / "tools" / "make_acados_golden.py"
    assert script.exists()
    spec = importlib.util.spec_from_file_location("make_acados_golden", script)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                      # importing must not need acados
    assert callable(mod.build_solver) and callable(mod.cold_start_rti)
    have = importlib.util.find_spec("acados_template") is not None and importlib.util.find_spec("casadi") is not None
    if not have:
        out = Path(__file__).parent / "golden" / "_should_not_exist.npz"
        r = subprocess.run([sys.executable, str(script), "--out", str(out)], capture_output=True, text=True)
        assert r.returncode != 0 and "needs acados_template" in (r.stderr + r.stdout)
        assert not out.exists()


def _sets(g):
    """(decisive u0 / x / status, reported u0 or None, label)"""
    if "u0_tight" in g.files:
        return g["u0_tight"], g["x_tight"], g["status_tight"], g["u0"], f"tolerances {float(g['qp_tol_tight']):g}"
    return g["u0"], g["x"], g["status"], None, "default tolerances (file written before round 5)"


def check_oracle_against(g, verbose=print):
    """oracle vs the golden file: names the convention on the decisive set, reports the default-tolerance distance, U10"""
    from oracle import oracle as O
    u0_dec, _, st_dec, u0_rep, label = _sets(g)
    ok = st_dec == 0
    table = {}
    for lm_dt in (1, 0):
        for cost_dt in (1, 0):
            # (the accuracy certificate is behaviour HPIPM does not have: off when statuses are compared with acados)
            c = O.default_config(qp_gamma=0.0, qp_polish=1, lm_scaled_by_dt=lm_dt, cost_scaled_by_dt=cost_dt, qp_growth_max=0.0)
            r = O.solve_batch(c, g["x0"], g["yref"], g["yref_e"])
            table[(lm_dt, cost_dt)] = _rel_err(r["u0"][ok], u0_dec[ok])
    best = min(table, key=table.get)
    verbose(f"acados {g['acados_version']} ({g['nlp_solver_type']}), decisive set = {label}: rel. error per (lm_scaled_by_dt, cost_scaled_by_dt): {table}")
    assert table[best] <= 1e-6, f"no switch setting reproduces acados: {table}"
    assert best == (1, 1), f"the shipped defaults (1, 1) are not this acados version's convention; {best} is: {table}"
    # the shipped DEFAULT configuration (certificate on) on the same states: it must not fire on any golden instance - if it ever does,
    # the default differs from acados in status there, and this is where that shows
    cd = O.default_config(qp_gamma=0.0, qp_polish=1)
    rd = O.solve_batch(cd, g["x0"], g["yref"], g["yref_e"])
    fired = np.nonzero((rd["status"] != 0) & ok)[0]
    assert fired.size == 0, f"default configuration (qp_growth_max = {cd.qp_growth_max:g}) refuses golden instances {fired.tolist()} that acados solves"
    assert _rel_err(rd["u0"][ok], u0_dec[ok]) <= 1e-6
    if u0_rep is not None:
        okr = ok & (g["status"] == 0)
        gap_exact = float(np.abs(rd["u0"][okr] - u0_rep[okr]).max())
        gap_sets = float(np.abs(u0_dec[okr] - u0_rep[okr]).max())
        verbose(f"default-tolerance set (the reference's options verbatim): max |u0 - exact| = {gap_exact:.2e} N, |u0 - tight set| = {gap_sets:.2e} N, "
                f"QP iterations {float(np.mean(g['qp_iter'])):.1f} (tight {float(np.mean(g['qp_iter_tight'])):.1f}) - acados' accuracy floor on this "
                f"OCP; predicted 3e-7 N on the fixture set (tests/test_acados_floor.py)")
        assert gap_exact <= 1e-4, "the default-tolerance solutions are further from the exact ones than an exit tolerance explains: a convention?"
    # U10: the status this acados returns when the QP hits its iteration cap names the qp_maxiter_status setting (0 tolerated, 2 reported)
    if "status_itercap" in g.files:
        seen = sorted(set(int(v) for v in g["status_itercap"]))
        assert set(seen) <= {0, 2}, f"unexpected status for a capped QP: {seen}"
        convention = 2 if 2 in seen else 0
        verbose(f"acados {g['acados_version']}: a QP stopped by qp_solver_iter_max returns status {seen} -> qp_maxiter_status = {convention}")
        assert convention == O.default_config().qp_maxiter_status, \
            f"the shipped default qp_maxiter_status = {O.default_config().qp_maxiter_status} is not this acados version's behaviour ({convention})"
    return table


def check_hip_against(g, verbose=print):
    from rotors_mpc_controller_amd import _lib
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    u0_dec, x_dec, st_dec, u0_rep, label = _sets(g)
    ok = st_dec == 0
    # the shipped default configuration, certificate included (it must stay silent on the golden states: asserted through the statuses)
    s = NmpcOcpSolver(_lib.default_config(max_batch=int(g["x0"].shape[0])))
    out = s.solve_batch(g["x0"], g["yref"], g["yref_e"], want_traj=True)
    np.testing.assert_array_equal(out["status"][ok], 0)
    assert _rel_err(out["u0"][ok], u0_dec[ok]) <= 1e-6
    assert _rel_err(out["x"][ok], x_dec[ok]) <= 1e-5
    if u0_rep is not None:
        okr = ok & (g["status"] == 0)
        gap = float(np.abs(out["u0"][okr] - u0_rep[okr]).max())
        verbose(f"HIP path vs the default-tolerance set: max |u0 - acados| = {gap:.2e} N (acados' own floor; decisive set = {label})")
        assert gap <= 1e-4
    s.close()


@needs_golden
def test_oracle_matches_acados_and_names_the_convention():
    check_oracle_against(np.load(GOLDEN))


@needs_golden
@pytest.mark.gpu
def test_hip_path_matches_acados():
    check_hip_against(np.load(GOLDEN))


def _dry_run_file(tmp_path):
    """the generator against the Level-B stand-ins, HPIPM's exit rule emulated"""
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    out = tmp_path / "acados_shim.npz"
    env = dict(os.environ, PYTHONPATH=str(root / "tests" / "levelb") + os.pathsep + os.environ.get("PYTHONPATH", ""), LEVELB_HPIPM_EXIT="1")
    r = subprocess.run([sys.executable, str(root / "tools" / "make_acados_golden.py"), "--out", str(out)],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    return np.load(out), r.stdout


def test_consumer_end_to_end_on_a_dry_run_file(tmp_path):
    """Generator (two solution sets) -> consumer, today: the 'tight' set of the stand-in is the oracle converged to 1e-12 residuals,
    its 'default' set the oracle stopped on HPIPM's default exit (1e-8) - the two differ the way acados' will, and the consumer must
    pick the convention on the former and report the latter."""
    g, log = _dry_run_file(tmp_path)
    assert "u0_tight" in g.files and g["u0_tight"].shape == g["u0"].shape and float(g["qp_tol_tight"]) == 1e-12
    gap = float(np.abs(g["u0"] - g["u0_tight"]).max())
    assert 1e-9 < gap < 1e-5, gap                       # the default exit is visibly looser: 3e-7 N on this set (tests/test_acados_floor.py)
    assert (g["qp_iter_tight"] >= g["qp_iter"]).all() and (g["qp_iter_tight"] > g["qp_iter"]).any()
    assert np.isfinite(g["residuals"]).all() and float(g["residuals"][:, 3].max()) <= 1e-8 and float(g["residuals_tight"][:, 3].max()) <= 1e-12
    lines = []
    table = check_oracle_against(g, verbose=lines.append)
    assert table[(1, 1)] <= 1e-6 and min(table[k] for k in table if k != (1, 1)) > 1e-5     # the other three conventions are rejected
    assert any("accuracy floor" in ln for ln in lines)


@pytest.mark.gpu
def test_hip_consumer_end_to_end_on_a_dry_run_file(tmp_path):
    g, _ = _dry_run_file(tmp_path)
    check_hip_against(g)


def test_harness_dry_run_on_the_levelb_shims(tmp_path):
    """The generator end to end on the dev-only stand-ins of tests/levelb (a CasADi-subset tracer and an
    acados_template whose solver probes the traced dynamics and forwards to the CPU oracle): its model, OCP
    construction and set/solve/get staging must reproduce the committed oracle fixture.  This validates the
    SCRIPT (it would feed acados the same problem), not acados parity -- the file it writes here is discarded."""
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    out = tmp_path / "acados_shim.npz"
    env = dict(os.environ, PYTHONPATH=str(root / "tests" / "levelb") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, str(root / "tools" / "make_acados_golden.py"), "--out", str(out)],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    g = np.load(out)
    f = np.load(root / "tests" / "golden" / "rti_cold_start.npz")
    n = int(g["n_fixture"])
    assert n == f["x0"].shape[0] and g["x0"].shape[0] == n + 2 and str(g["nlp_solver_type"]) == "SQP_RTI"
    np.testing.assert_array_equal(g["status"], 0)
    np.testing.assert_allclose(g["u0"][:n], f["u0"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(g["x"][:n], f["x"], rtol=0, atol=1e-11)
    hov = 0.68 * 9.81 / 4.0
    assert np.ptp(g["u0"][n]) < 1e-12 and abs(abs(g["u0"][n, 0] - hov) - 3.4e-4) < 2e-5      # K2
    assert np.ptp(g["u0"][n + 1]) < 1e-12 and abs(g["u0"][n + 1, 0] - 2.19447) < 1e-5        # K3
