"""Consumer of tests/golden/acados_rti.npz -- the file tools/make_acados_golden.py writes on a machine that has
acados.  It does not exist in this repository yet (acados, HPIPM, BLASFEO and CasADi are absent from the image and
from /root/reference: SURVEY 8c), so these tests SKIP with that reason and every accuracy figure of this
repository stays "vs the build's CPU oracle; acados parity unpinned".  Once the file is committed they pin, in
this order: (1) the oracle against acados on the fixture set, per switch setting of the three
version-dependent conventions; (2) the HIP path against acados (GPU test).

Tolerance: BASELINE.json asks for 1e-6 relative on u0.  acados' own u0 is defined to HPIPM's exit tolerances
(about 1e-8 on the residuals: SURVEY U9), so the assertion is |u0 - acados| <= 1e-6 * max(1, |u0|).
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


@needs_golden
def test_oracle_matches_acados_and_names_the_convention():
    from oracle import oracle as O
    g, ok = _load()
    table = {}
    for lm_dt in (1, 0):
        for cost_dt in (1, 0):
            # (the accuracy certificate is behaviour HPIPM does not have: off when statuses are compared with acados)
            c = O.default_config(qp_gamma=0.0, qp_polish=1, lm_scaled_by_dt=lm_dt, cost_scaled_by_dt=cost_dt, qp_growth_max=0.0)
            r = O.solve_batch(c, g["x0"], g["yref"], g["yref_e"])
            table[(lm_dt, cost_dt)] = _rel_err(r["u0"][ok], g["u0"][ok])
    best = min(table, key=table.get)
    print(f"acados {g['acados_version']} ({g['nlp_solver_type']}): rel. error per (lm_scaled_by_dt, cost_scaled_by_dt): {table}")
    assert table[best] <= 1e-6, f"no switch setting reproduces acados: {table}"
    assert best == (1, 1), f"the shipped defaults (1, 1) are not this acados version's convention; {best} is: {table}"
    # U10: the status this acados returns when the QP hits its iteration cap names the qp_maxiter_status setting (0 tolerated, 2 reported)
    if "status_itercap" in g.files:
        seen = sorted(set(int(v) for v in g["status_itercap"]))
        assert set(seen) <= {0, 2}, f"unexpected status for a capped QP: {seen}"
        convention = 2 if 2 in seen else 0
        print(f"acados {g['acados_version']}: a QP stopped by qp_solver_iter_max returns status {seen} -> qp_maxiter_status = {convention}")
        assert convention == O.default_config().qp_maxiter_status, \
            f"the shipped default qp_maxiter_status = {O.default_config().qp_maxiter_status} is not this acados version's behaviour ({convention})"


@needs_golden
@pytest.mark.gpu
def test_hip_path_matches_acados():
    from rotors_mpc_controller_amd import _lib
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    g, ok = _load()
    s = NmpcOcpSolver(_lib.default_config(max_batch=int(g["x0"].shape[0]), qp_growth_max=0.0))     # certificate off: see above
    out = s.solve_batch(g["x0"], g["yref"], g["yref_e"], want_traj=True)
    np.testing.assert_array_equal(out["status"][ok], 0)
    assert _rel_err(out["u0"][ok], g["u0"][ok]) <= 1e-6
    assert _rel_err(out["x"][ok], g["x"][ok]) <= 1e-5


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
