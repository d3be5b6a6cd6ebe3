"""The shipped code generation of the default path, executed on the CPU.

The solver kernels that run by default (csrc/nmpc_as.hip: k_team_as; csrc/nmpc_qpf.hip: k_team_qp, k_team_qp_list, k_team_tail) are built with
the internal LLVM option -amdgpu-mfma-vgpr-form.  tools/emu/gfx950_emu.py is a functional emulator of a gfx950 wave that reads the compiler's
own assembly of those translation units (make build/nmpc_as.s, build/nmpc_qpf.s: same flags as the shipped objects): these tests run
workgroups of the headline configuration, of the warm-started per-stage configuration, of the plain interior-point kernel and of the two
launches of the default path in sequence through it, registers, LDS and scratch poisoned, every global access checked against buffers of
nmpc_create's sizes, every LDS access against the launch's allocation, every scratch access against the kernel's private segment, and
compare the results with the oracle.  No GPU involved: what is under test is the INSTRUCTION STREAM of the flag builds (DESIGN.md section
4.2 has why; the GPU-side guards are the test_flag_build_of_..._is_bit_equal_to_the_default_codegen_build tests).
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


_ASM_TARGETS = ("nmpc_as.s", "nmpc_qpf.s", "nmpc_blockf.s")
_asm_built = []


def _asm(target):
    if not _asm_built:           # the three translation units side by side (minutes of hipcc when the sources changed, nothing otherwise)
        subprocess.check_call(["make", "-s", "-j3", "-C", str(CSRC)] + [f"build/{t}" for t in _ASM_TARGETS], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _asm_built.append(True)
    if target not in _ASM_TARGETS:
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


def _check_hand_over(as_kernel, list_kernel, wgs, warm, share=1):
    """Failed first attempts continued (pass budget of ONE pass per attempt, so that every instance that pins an input fails its first attempt):
    list_kernel = None - inside k_team_as, on the wave that made the attempt (what runs by default); else the work-list flow - k_team_as with
    the continuation switched off, then workgroup 0 of that k_team_qp_list build on the memory it left.  Addresses of the instruction
    streams, scratch reloads against their spills, statuses / iteration counts / commands / trajectories of every instance of the workgroup
    against the oracle."""
    import run_team_kernel as R
    from oracle import oracle as O
    over = dict(qp_polish_passes=1, qp_polish_budget=2)
    continued = 0
    for wg in wgs:
        r = R.emulate(_asm("nmpc_as.s"), as_kernel, wg=wg, verbose=False, steps=2, polish=1, share=share, B=2048, dist="aggressive", seed=1, kind="as",
                      warm=warm, cfg_over=over, inplace=list_kernel is None,
                      then=None if list_kernel is None else (_asm("nmpc_qpf.s"), list_kernel))
        runs = [r] if list_kernel is None else [r, r["second"]]
        for q in runs:
            assert q["error"] is None, q["error"]
            assert not q["violations"], [(x.kind, x.line, x.text, x.lane, hex(x.addr), x.note) for x in q["violations"][:4]]
        assert (list_kernel is None) == (not r["listed"])          # in place: nothing is appended to the work list
        c = O.default_config(N=20, qp_gamma=0.0, qp_polish=1, sim_num_steps=2, **over)
        sl = slice(r["inst"], r["inst"] + r["tpw"])
        ref = O.solve_batch(c, r["x0"][sl], r["yref"], r["ye"], x_init=None if not warm else r["x_init"][sl],
                            u_init=None if not warm else r["u_init"][sl], want_traj=True)
        B = len(r["x0"])
        m = r["mem"]
        assert (m.view("status", np.int32)[sl] == ref["status"]).all() and (m.view("d_iters", np.int32)[sl] == ref["iters"]).all()
        assert np.abs(m.view("u0", np.float64).reshape(B, 4)[sl] - ref["u0"]).max() < 1e-9
        if warm:
            assert np.abs(m.view("x_out", np.float64).reshape(B, 21, 13)[sl] - ref["x"]).max() < 1e-9
            assert np.abs(m.view("u_out", np.float64).reshape(B, 20, 4)[sl] - ref["u"]).max() < 1e-9
        continued += int((ref["iters"] > 0).sum())
    assert continued >= 2, continued    # (the point of the case: some first attempts failed)


def test_shipped_flag_build_continues_failed_first_attempts_inside_k_team_as():
    """k_team_as<shared> and k_team_as<per-stage, trajectories, warm>: the first attempt, then team_as MODE 2 on the same wave - interior-point
    iterations and a second attempt - in one instruction stream (the default schedule of short horizons: no work-list launch)."""
    _check_hand_over("k_team_asILb1ELb0ELi1EdEE", None, range(0, 2), warm=False)
    _check_hand_over("k_team_asILb0ELb1ELi1EdEE", None, range(0, 2), warm=True)
    _check_hand_over("k_team_asILb0ELb0ELi1EdEE", None, range(1, 3), warm=False, share=0)     # (the kernel of DESIGN.md section 4.2a's fault)


def test_shipped_flag_builds_of_both_launches_with_instances_handed_over():
    """The work-list flow (long horizons, NMPC_TEAM_INPLACE=0): k_team_as<shared> ->
    k_team_qp_list<shared> (nmpc_qpf.hip)."""
    _check_hand_over("k_team_asILb1ELb0ELi1EdEE", "k_team_qp_listILb1ELb0EdEE", range(0, 2), warm=False)


def test_shipped_flag_builds_of_both_launches_per_stage_with_trajectories():
    """... and the warm-started per-stage variants with trajectories, whose work-list kernel is the one shipped kernel with spills (44 B of
    scratch per lane): the emulator keeps the private segment per lane and flags a reload of a byte no spill wrote."""
    _check_hand_over("k_team_asILb0ELb1ELi1EdEE", "k_team_qp_listILb0ELb1EdEE", range(0, 1), warm=True)


def test_no_vector_instruction_sits_in_front_of_an_exec_restore_in_the_flag_builds():
    """tools/emu/exec_join_check.py on the assembly of the three translation units that are built with -amdgpu-sched-strategy=iterative-ilp:
    with that strategy LLVM has placed a register-parking copy at the top of a join block, in front of the s_or_b64 that restores EXEC (the
    copy then saves some lanes only; found when the emulated pass counts left the oracle's).  The builds that ship must have none."""
    import exec_join_check as X
    for target in ("nmpc_as.s", "nmpc_qpf.s", "nmpc_blockf.s"):
        found = X.check(_asm(target), verbose=False)
        assert not found, [(f[0][:50], f[2], f[3]) for f in found[:4]]


def test_codegen_gate_of_the_build(tmp_path):
    """tools/emu/codegen_gate.py is what `make` runs on the assembly of the flag builds before it links (round 5: the scans above moved from
    pytest into the build).  On the shipped builds with the validated hipcc it writes mode 0 (flag builds are the default); told to expect
    another hipcc it writes mode 1 and the reason - the library is then linked with -DNMPC_DEFAULT_NOFLAG=1 and nmpc_version() ends in
    'codegen default'.  The loaded library's version string names its mode."""
    import sys
    gate = str(ROOT / "tools" / "emu" / "codegen_gate.py")
    files = [_asm(t) for t in ("nmpc_as.s", "nmpc_qpf.s", "nmpc_blockf.s")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    ok, bad = tmp_path / "ok.mode", tmp_path / "bad.mode"
    subprocess.check_call([sys.executable, gate, "--hipcc", hipcc, "--expect", "7.2", "--out", str(ok)] + files, stdout=subprocess.DEVNULL)
    subprocess.check_call([sys.executable, gate, "--hipcc", hipcc, "--expect", "9.9", "--out", str(bad)] + files, stdout=subprocess.DEVNULL)
    assert ok.read_text().splitlines()[0] == "0", ok.read_text()
    lines = bad.read_text().splitlines()
    assert lines[0] == "1" and "validated on 9.9" in lines[1]
    from rotors_mpc_controller_amd import _lib
    assert _lib.library_codegen() in ("flag", "default") and ("codegen " + _lib.library_codegen()) in _lib.load().nmpc_version().decode()


@pytest.mark.skipif(os.environ.get("NMPC_EMU_FULL") != "1", reason="NMPC_EMU_FULL=1: compiles nmpc_qp.hip with the flag (minutes); profiles/r04_emulation_*.txt holds the full runs")
def test_the_configuration_that_faulted_in_round_3_on_the_flag_build_of_k_team_qp():
    """k_team_qp<per-stage, trajectories>, sim_num_steps = 4, qp_polish = 0, B = 256, aggressive seed 8 (gpurun_out/qp_check2.log), built WITH
    -amdgpu-mfma-vgpr-form: eight workgroups of the launch that faulted, cold and warm."""
    s = _asm("nmpc_qp_flag.s")
    _check(s, "k_team_qpILb0ELb1EdEE", range(0, 8), steps=4, polish=0, share=0, B=256, dist="aggressive", seed=8, kind="qp")
    _check(s, "k_team_qpILb0ELb1EdEE", range(0, 4), steps=4, polish=0, share=0, B=256, dist="aggressive", seed=8, kind="qp", warm=True)


def test_mfma_results_are_read_no_earlier_than_the_hazard_table_allows_in_the_flag_builds():
    """tools/emu/isa_checks.py: for every v_mfma_f64_4x4x4 that writes vector registers (all of them in the flag builds), the distance in
    wait states to the first reader / overwriter along every path, against LLVM's table for the DGEMM 4x4x4 result (VALU 6, memory 9,
    MFMA A/B 6).  The scheduler strategy moves instructions into those slots; the hazard recognizer pads what is left - nothing may be
    closer than the table."""
    import isa_checks as H
    for target, kernels in (("nmpc_as.s", ("k_team_asILb1ELb0ELi1EdEE", "k_team_asILb0ELb1ELi1EdEE")), ("nmpc_qpf.s", ("k_team_qpILb1ELb0EdEE",)),
                            ("nmpc_blockf.s", (None,))):
        for k in kernels:
            found = H.check(_asm(target), k, verbose=False)
            assert not found, (target, k, found[:3])


def _run_snippet(tmp_path, body, scratch_bytes=0):
    """Emulate a hand-written instruction sequence: s[0:1] = address of a 64-byte read-only block holding the double 2.5 at offset 8,
    s[2:3] = address of a 512-byte output buffer."""
    import gfx950_emu as E
    f = tmp_path / "snippet.s"
    f.write_text("_Zsnippet:\n" + body + "\n\ts_endpgm\n.Lfunc_end0:\n")
    mem = E.Memory()
    src = np.zeros(8, dtype=np.float64); src[1] = 2.5
    a_in = mem.add("in", np.frombuffer(src.tobytes(), dtype=np.uint8).copy(), False)
    a_out = mem.add("out", np.zeros(512, dtype=np.uint8), True)
    insts, labels = E.parse_kernel(str(f), "snippet")
    w = E.Wave(insts, labels, mem, 0, a_in, 0, {})
    w.S[2], w.S[3] = a_out & 0xFFFFFFFF, a_out >> 32
    w.scratch_bytes = scratch_bytes
    w.run()
    return w, mem.view("out", np.float64)


def test_emulator_treats_vcc_as_an_ordinary_register_pair(tmp_path):
    """The iterative-ilp builds load constants into vcc (s_load_dwordx2 vcc, ...) and write vcc_hi on its own; the emulator once put such a load
    into s[0:1] - found because the emulated work-list kernel then disagreed with the oracle."""
    w, out = _run_snippet(tmp_path, """
\ts_load_dwordx2 vcc, s[0:1], 0x8
\ts_waitcnt lgkmcnt(0)
\tv_mul_f64 v[2:3], vcc, 2.0
\tv_lshlrev_b32_e32 v4, 3, v0
\tglobal_store_dwordx2 v4, v[2:3], s[2:3]
\ts_mov_b32 vcc_hi, 0
\ts_mov_b32 vcc_lo, 5
\tv_mov_b32_e32 v5, vcc_lo""")
    assert not w.viol and (out == 5.0).all()
    assert int(w.S[0]) | (int(w.S[1]) << 32) == w.mem.find(int(w.S[0]) | (int(w.S[1]) << 32), 8).base     # the kernarg pointer is untouched
    assert (w.R[5] == 5).all() and int(w.S[107]) == 0


def test_emulator_runs_the_sdwa_form_of_an_integer_add(tmp_path):
    """v_add_u32_sdwa with a byte select (how the compiler adds a bool to an integer: npol hand-over of nmpc_team_as.hpp)."""
    w, out = _run_snippet(tmp_path, """
\tv_mov_b32_e32 v1, 0x10203
\tv_mov_b32_e32 v2, 7
\tv_add_u32_sdwa v3, v2, v1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0
\tv_add_u32_sdwa v4, v2, v1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1
\tv_add_u32_sdwa v5, v2, v1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1""")
    assert not w.viol and (w.R[3] == 10).all() and (w.R[4] == 9).all() and (w.R[5] == 8).all()


def test_emulator_checks_scratch_against_the_private_segment_and_its_spills(tmp_path):
    body = """
\tv_mov_b32_e32 v1, 7
\tscratch_store_dword off, v1, off offset:4
\tscratch_load_dword v2, off, off offset:4
\tscratch_load_dword v3, off, off offset:8
\tscratch_store_dword off, v1, off offset:16"""
    w, _ = _run_snippet(tmp_path, body, scratch_bytes=16)
    assert (w.R[2] == 7).all()
    kinds = sorted({(v.kind, v.note.split()[0]) for v in w.viol})
    assert kinds == [("scratch-read", "never"), ("scratch-write", "private")], kinds
