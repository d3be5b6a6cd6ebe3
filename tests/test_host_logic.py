"""Host-side logic and the C-ABI surface -- CPU only (no compute calls into the HIP library)."""
import ctypes as C
import json
import re
from pathlib import Path

import numpy as np
import pytest

from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.controller import derive_params, to_nmpc_config
from rotors_mpc_controller_amd.params import load_params
from rotors_mpc_controller_amd.reference import ReferenceGenerator, batched_hover_yref, stack_yref, yaw_quaternion

ROOT = Path(__file__).resolve().parent.parent
GOLD = Path(__file__).resolve().parent / "golden"


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = (ROOT / "include" / "rotors_nmpc.h").read_text()
    declared = set(re.findall(r"\b(nmpc_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.nmpc_version()


def test_config_struct_matches_the_c_layout_and_defaults():
    cfg = _lib.default_config()
    assert (cfg.N, cfg.dt, cfg.sim_num_stages, cfg.sim_num_steps, cfg.qp_iter_max) == (20, 0.05, 2, 2, 600)
    assert cfg.dtype == _lib.DTYPE_F64 and cfg.max_batch == 4096 and cfg.flags & _lib.FLAG_SHARE_COLD_START
    np.testing.assert_allclose(list(cfg.lbu), 8.54858e-6 * 50.0 ** 2)
    np.testing.assert_allclose(list(cfg.ubu), 8.54858e-6 * 838.0 ** 2)      # 6.00318901352 N, SURVEY 2.1
    np.testing.assert_allclose(list(cfg.rotor_z), [-0.016, 0.016, -0.016, 0.016])
    # the last field survived the round trip => the ctypes mirror has the C struct's size/offsets
    assert cfg.qp_thr0_rel == 0.25


def test_create_fails_loudly_without_a_gpu_or_with_bad_config():
    import torch
    lib = _lib.load()
    bad = _lib.default_config(sim_num_stages=4)
    assert not lib.nmpc_create(C.byref(bad))
    assert b"sim_method_num_stages" in lib.nmpc_last_error(None)
    if not torch.cuda.is_available():
        assert not lib.nmpc_create(C.byref(_lib.default_config()))
        assert b"no CPU path" in lib.nmpc_last_error(None)


def test_params_loader_matches_reference_golden(monkeypatch):
    monkeypatch.delenv("ROTORS_MPC_PARAMS", raising=False)
    got = load_params()
    got.pop("params_yaml")
    want = json.loads((GOLD / "params_coerced.json").read_text())
    assert got == want


def test_params_loader_defaults_and_errors(tmp_path, monkeypatch):
    p = tmp_path / "p.yaml"
    p.write_text("solver: {}\nvehicle: {}\ncontroller: {}\nworld: {}\nreference: {}\n"
                 "topics: {state: a, motor: b, reference: c}\nnode: {}\n")
    monkeypatch.setenv("ROTORS_MPC_PARAMS", str(p))
    d = load_params()
    assert d["solver"]["horizon_steps"] == 20 and d["vehicle"]["motor_max_speed"] == 2000.0
    assert d["reference"]["default_position"] == [1.0, 1.0, 1.0] and d["node"]["rate"] == 50.0
    p.write_text("solver: {}\n")
    with pytest.raises(ValueError):
        load_params()


def test_reference_generator_matches_reference_golden():
    cases = json.loads((GOLD / "reference_horizon.json").read_text())
    gen = ReferenceGenerator(json.loads((GOLD / "params_coerced.json").read_text())["reference"])
    gen.update_hover_thrust(0.68 * 9.81 / 4.0)
    for case in cases:
        if case["setup"]:
            gen.set_target(**case["setup"])
        hz = gen.build_horizon(case["horizon"], case["dt"])
        assert set(hz) == set(case["out"])
        for k, v in case["out"].items():
            np.testing.assert_array_equal(hz[k], np.asarray(v), err_msg=f"{case['label']}:{k}")
    with pytest.raises(ValueError):
        gen.set_target([0, 0, 1], thrust=[1, 2, 3])


def test_yref_stacking_is_the_controller_layout():
    gen = ReferenceGenerator({"default_position": [1, 2, 3], "default_yaw": 0.4})
    gen.update_hover_thrust(1.5)
    hz = gen.build_horizon(7, 0.05)
    yref, ye = stack_yref(hz, 7)
    assert yref.shape == (7, 17) and ye.shape == (13,)
    np.testing.assert_array_equal(yref[3], np.hstack([hz[k][3] for k in
                                  ("positions", "velocities", "quaternions", "body_rates", "thrusts")]))
    np.testing.assert_array_equal(ye, yref[0, :13])
    yb, yeb = batched_hover_yref(np.array([[1, 2, 3.0]]), np.array([0.4]), 1.5, 7)
    np.testing.assert_allclose(yb[0], yref)
    np.testing.assert_allclose(yeb[0], ye)
    np.testing.assert_allclose(yaw_quaternion(0.4), [np.cos(0.2), 0, 0, np.sin(0.2)])


def test_derive_params_reproduces_survey_constants():
    p = derive_params(load_params(ROOT / "rotors_mpc_controller_amd" / "config" / "params.yaml"))
    np.testing.assert_allclose(p.input_lower_bounds, 0.02137145)
    np.testing.assert_allclose(p.input_upper_bounds, 6.00318901352)
    assert abs(p.hover_thrust_per_motor - 1.6677) < 1e-12
    np.testing.assert_array_equal(p.rotor_x_offsets, [0.17, 0, -0.17, 0])
    np.testing.assert_array_equal(p.rotor_y_offsets, [0, 0.17, 0, -0.17])
    cfg = to_nmpc_config(p, max_batch=8)
    assert cfg.qp_cond_N == 5 and cfg.max_batch == 8 and list(cfg.W)[13:] == [1.75] * 4
    bad = load_params(ROOT / "rotors_mpc_controller_amd" / "config" / "params.yaml")
    bad["vehicle"]["rotor_configuration"] = "x"
    with pytest.raises(ValueError):
        derive_params(bad)


def test_library_binds_one_hip_runtime_without_importing_torch():
    """`_lib.load()` must not depend on `import torch` having happened first: it maps
    torch's bundled HIP runtime itself when a torch installation exists, so that a later `import torch` finds
    the same object -- exactly one libamdhip64 in the process in either order."""
    import subprocess
    import sys
    code = (
        "import sys\n"
        "from rotors_mpc_controller_amd import _lib\n"
        "lib = _lib.load()\n"
        "assert 'torch' not in sys.modules, 'load() imported torch'\n"
        "maps = lambda: sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
        "a = maps(); assert len(a) == 1, a\n"
        "import torch\n"
        "b = maps(); assert b == a, (a, b)\n"
        "print('ok', a[0])\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       cwd=str(__import__('pathlib').Path(__file__).resolve().parent.parent))
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr + r.stdout


def test_shell_helpers_parse():
    """tools/*.sh are run on the GPU box only: at least their syntax is checked here (a trailing comment once swallowed a loop header)."""
    import subprocess
    from pathlib import Path
    for sh in sorted((Path(__file__).resolve().parent.parent / "tools").glob("*.sh")):
        assert subprocess.run(["bash", "-n", str(sh)], capture_output=True).returncode == 0, sh.name


def test_binding_mirrors_the_structs_of_the_loaded_library():
    """nmpc_abi_sizes: the ctypes mirrors of nmpc_config / nmpc_stats have the sizes the binary was compiled with (load() refuses a
    mismatch), the version string carries the hash of the kernel sources the binary was built from, and - after build() - that is
    the hash of the sources in the tree."""
    import ctypes as C
    import sys
    from pathlib import Path
    from rotors_mpc_controller_amd import _lib
    lib = _lib.load()
    cb, sb = C.c_int(0), C.c_int(0)
    assert lib.nmpc_abi_sizes(C.byref(cb), C.byref(sb)) == 3
    assert cb.value == C.sizeof(_lib.NmpcConfig) and sb.value == C.sizeof(_lib.NmpcStats)
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    from source_hash import source_hash
    assert _lib.library_source_hash() == source_hash()
    cfg = _lib.default_config()
    assert cfg.qp_polish_passes == 0 and cfg.qp_polish_budget == 0 and cfg.qp_growth_max == 1e6 and cfg.qp_maxiter_status == 0      # (0: the policy for the horizon)
