"""The shipped code-generation of the default path's first launch, executed on the CPU.

k_team_as (csrc/nmpc_as.hip) is the one kernel built with the internal LLVM option -amdgpu-mfma-vgpr-form.  tools/emu/gfx950_emu.py is a
functional emulator of a gfx950 wave that reads the compiler's own assembly of that translation unit (make build/nmpc_as.s: same flags as
the shipped object): these tests run workgroups of the headline configuration and of the warm-started per-stage configuration through it,
registers and LDS poisoned, every global access checked against buffers of nmpc_create's sizes and every LDS access against the launch's
allocation, and compare the commands with the oracle.  No GPU involved: what is under test is the INSTRUCTION STREAM of the flag build
(DESIGN.md section 4.2 has why; the GPU-side guard is test_flag_build_of_the_active_set_kernel_is_bit_equal_to_the_default_codegen_build).
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools" / "emu"))
CSRC = ROOT / "rotors_mpc_controller_amd" / "csrc"

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="needs hipcc for the assembly")


def _asm(target):
    subprocess.check_call(["make", "-s", "-C", str(CSRC), f"build/{target}"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return str(CSRC / "build" / target)


def _check(sfile, kernel, wgs, **kw):
    import run_team_kernel as R
    from oracle import oracle as O
    worst = 0.0
    for wg in wgs:
        r = R.emulate(sfile, kernel, wg=wg, verbose=False, **kw)
        assert r["error"] is None, r["error"]
        assert not r["violations"], [(v.kind, v.line, v.text, v.lane, hex(v.addr), v.note) for v in r["violations"][:4]]
        c = O.default_config(N=20, qp_gamma=0.0, qp_polish=kw.get("polish", 0), sim_num_steps=kw.get("steps", 2))
        sl = slice(r["inst"], r["inst"] + r["tpw"])
        ref = O.solve_batch(c, r["x0"][sl], r["yref"], r["ye"], x_init=None if r["x_init"] is None else r["x_init"][sl],
                            u_init=None if r["u_init"] is None else r["u_init"][sl])
        keep = np.array([i not in r["listed"] for i in range(r["inst"], r["inst"] + r["tpw"])])      # (the second launch finishes those)
        assert (r["status"][keep] == ref["status"][keep]).all()
        if keep.any():
            worst = max(worst, float(np.abs(r["u0"][keep] - ref["u0"][keep]).max()))
    assert worst < 1e-9, worst
    return worst


def test_shipped_flag_build_of_k_team_as_headline_configuration():
    """B >= 2048 (four instances per wave), shared cold-start linearisation, no trajectories: k_team_as<true, false, 1, double>."""
    s = _asm("nmpc_as.s")
    _check(s, "k_team_asILb1ELb0ELi1EdEE", range(0, 4), steps=2, polish=1, share=1, B=2048, dist="near_hover", seed=0, kind="as")
    _check(s, "k_team_asILb1ELb0ELi1EdEE", range(0, 3), steps=2, polish=1, share=1, B=2048, dist="aggressive", seed=1, kind="as")


def test_shipped_flag_build_of_k_team_as_warm_started_per_stage_with_trajectories():
    """What every tick after the first runs (controller.py:419-424): k_team_as<false, true, 1, double>, warm start from the oracle's first solve."""
    s = _asm("nmpc_as.s")
    _check(s, "k_team_asILb0ELb1ELi1EdEE", range(0, 2), steps=2, polish=1, share=1, B=2048, dist="aggressive", seed=1, kind="as", warm=True)


def test_shipped_flag_build_of_the_interior_point_kernel():
    """k_team_qp<shared, no trajectories, double> of nmpc_qpf.hip (what bench.py --no-polish and qp_polish = 0 run), plain interior point: two
    workgroups of four instances through ~80 k instructions each."""
    s = _asm("nmpc_qpf.s")
    _check(s, "k_team_qpILb1ELb0EdEE", range(0, 2), steps=2, polish=0, share=1, B=2048, dist="near_hover", seed=0, kind="qp")


@pytest.mark.skipif(os.environ.get("NMPC_EMU_FULL") != "1", reason="NMPC_EMU_FULL=1: compiles nmpc_qp.hip with the flag (minutes); profiles/r04_emulation_*.txt holds the full runs")
def test_the_configuration_that_faulted_in_round_3_on_the_flag_build_of_k_team_qp():
    """k_team_qp<per-stage, trajectories>, sim_num_steps = 4, qp_polish = 0, B = 256, aggressive seed 8 (gpurun_out/qp_check2.log), built WITH
    -amdgpu-mfma-vgpr-form: eight workgroups of the launch that faulted, cold and warm."""
    s = _asm("nmpc_qp_flag.s")
    _check(s, "k_team_qpILb0ELb1EdEE", range(0, 8), steps=4, polish=0, share=0, B=256, dist="aggressive", seed=8, kind="qp")
    _check(s, "k_team_qpILb0ELb1EdEE", range(0, 4), steps=4, polish=0, share=0, B=256, dist="aggressive", seed=8, kind="qp", warm=True)
