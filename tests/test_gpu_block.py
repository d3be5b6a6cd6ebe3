"""Parallel-in-time Riccati factorisation (csrc/nmpc_block.hpp, C ABI nmpc_block_factor_device; SURVEY 8a7: controller.py:184
asks for partial condensing into min(N, 5) blocks, cfg/rotors_mpc.cfg:9 lets the horizon reach 600).

The blocks of the horizon are swept by their own teams at the same time.  What is checked, through the C ABI on the GPU:
  * blocks = 1 (the sequential sweep in the new code) reproduces the factors the solver's own sweeps left in the workspace;
  * blocks = J reproduces the factors of blocks = 1 on every stage of >= 32 instances at N = 600 - relative 1e-9;
  * the value function at every block boundary as the sequential boundary scan computes it equals the one the block's own final
    sweep arrives at (the linear-fractional composition P_s = J + Psi' T Psi against the stage-by-stage recursion).
The constant of the value function (entry (15,15) of the padded homogeneous form) is excluded: no gain depends on it."""
import numpy as np
import pytest

from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0

pytestmark = pytest.mark.gpu


def _tile_matrix(t):
    """[..., 256] (16 tiles x 16 lanes: tile (it,jt) at (4 it + jt) 16 + 4 a + c) -> [..., 16, 16]"""
    m = t.reshape(t.shape[:-1] + (4, 4, 4, 4))            # it, jt, a, c
    return np.moveaxis(m, -3, -2).reshape(t.shape[:-1] + (16, 16))


def _solve_and_factor(N, B, blocks, dist, monkeypatch, seed=5):
    import torch
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    monkeypatch.setenv("NMPC_TEAM_LSTG", "0")             # the solver's own factors all reach HBM (debug_factors reads them there)
    s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B, flags=_lib.FLAG_TEAM_MAPPING))   # per-stage linearisation
    yref, ye = hover_reference(N, s.config.mass * s.config.gravity / 4.0)
    x0 = sample_x0(B, seed, **dist)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_x0, d_yr, d_ye = dev(x0), dev(yref), dev(ye)
    d_u0 = torch.zeros(B, 4, dtype=torch.float64, device="cuda")
    d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
    s.solve_batch_device(B, d_x0.data_ptr(), d_yr.data_ptr(), d_ye.data_ptr(), True, d_u0.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    status = d_st.cpu().numpy()
    passes = s.passes(B)
    own = s.debug_factors(B)

    def factor(J, check=False):
        fac = torch.zeros(B, N, 80, dtype=torch.float64, device="cuda")
        bnd = torch.zeros(B, J + 1, 256, dtype=torch.float64, device="cuda")
        chk = torch.zeros(B, J + 1, 256, dtype=torch.float64, device="cuda") if check else None
        Jeff, ms = s.block_factor_device(B, J, d_x0.data_ptr(), d_yr.data_ptr(), d_ye.data_ptr(), True, factors_ptr=fac.data_ptr(),
                                         boundary_ptr=bnd.data_ptr(), check_ptr=chk.data_ptr() if check else 0, timed=True)
        torch.cuda.synchronize()
        return Jeff, ms, fac.cpu().numpy(), bnd.cpu().numpy()[:, :Jeff], (chk.cpu().numpy()[:, :Jeff] if check else None)
    return s, status, passes, own, factor


@pytest.mark.parametrize("N,B,blocks,dist", [(60, 48, 5, AGGRESSIVE), (600, 64, 25, NEAR_HOVER), (600, 64, 5, NEAR_HOVER)])
def test_block_parallel_factorisation_reproduces_the_sequential_factors(N, B, blocks, dist, monkeypatch):
    s, status, passes, own, factor = _solve_and_factor(N, B, blocks, dist, monkeypatch)
    acc = (status == 0) & (passes > 0)                    # ended on an accepted active-set pass: the pin set in the workspace is the final one
    assert acc.sum() >= min(B, 32), (status, passes)
    assert (passes[acc] > 1).any()                        # some of them with pinned inputs
    _, _, seq, bseq, _ = factor(1)
    scale = np.abs(seq[acc]).max(axis=(1, 2), keepdims=True)
    # the sequential sweep of the new code against the solver's own (same products, separately compiled)
    assert np.abs(seq[acc] - own[acc]).max() <= 1e-11 * scale.max()
    J, ms, blk, bnd, chk = factor(blocks, check=True)
    assert J == blocks and np.isfinite(blk[acc]).all() and np.isfinite(bnd[acc]).all()
    rel = np.abs(blk[acc] - seq[acc]) / scale
    assert rel.max() <= 1e-9, rel.max()
    # boundary values: scan against each block's own recomputation (blocks 0 .. J-2), off the constant
    Pb, Pc = _tile_matrix(bnd[acc][:, :J - 1]), _tile_matrix(chk[acc][:, :J - 1])
    Pb[..., 15, 15] = 0; Pc[..., 15, 15] = 0
    assert np.abs(Pb - Pc).max() <= 1e-9 * np.abs(Pc).max()
    assert np.abs(Pb - np.swapaxes(Pb, -1, -2)).max() == 0.0        # kept exactly symmetric
    s.close()


def test_block_factor_rejects_what_it_cannot_do():
    import torch
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    s = NmpcOcpSolver(_lib.default_config(max_batch=8))              # shared cold-start linearisation: no per-stage tiles in the workspace
    yref, ye = hover_reference(20, s.config.mass * s.config.gravity / 4.0)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x0 = d(sample_x0(8, 1, **NEAR_HOVER)); yr = d(yref); ye_d = d(ye)
    with pytest.raises(RuntimeError, match="no solve"):
        s.block_factor_device(8, 4, x0.data_ptr(), yr.data_ptr(), ye_d.data_ptr(), True)
    u0 = torch.zeros(8, 4, dtype=torch.float64, device="cuda")
    s.solve_batch_device(8, x0.data_ptr(), yr.data_ptr(), ye_d.data_ptr(), True, u0.data_ptr())
    with pytest.raises(RuntimeError, match="per-stage linearisation"):
        s.block_factor_device(8, 4, x0.data_ptr(), yr.data_ptr(), ye_d.data_ptr(), True)
    with pytest.raises(RuntimeError, match="blocks"):
        s.block_factor_device(8, 0, x0.data_ptr(), yr.data_ptr(), ye_d.data_ptr(), True)
    s.close()


WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)


def _solve_both(monkeypatch, over, x0, yref, ye, J, warm_from=None):
    """the same solve with the work list continued sequentially (k_team_qp_list) and by the block-parallel tail"""
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    res = []
    for tail in ("0", "1"):
        monkeypatch.setenv("NMPC_BLOCK_TAIL", tail)
        if J:
            monkeypatch.setenv("NMPC_BLOCK_J", str(J))
        s = NmpcOcpSolver(_lib.default_config(**over))
        kw = dict(x_init=warm_from["x"], u_init=warm_from["u"]) if warm_from else {}
        out = s.solve_batch(x0, yref, ye, want_traj=True, **kw)
        it, ps = s.counts()
        st = s.stats()
        st["tail_blocks"], st["tail_states"] = s.tail_states(len(x0))
        res.append((out, it.copy(), ps.copy(), st))
        s.close()
    return res


TWO_ATTEMPTS = dict(qp_polish_passes=8, qp_polish_budget=16)     # the schedule with an interior-point iteration between two attempts (the default below N = 160)


@pytest.mark.parametrize("N,B,dist,J,over", [
    (600, 1024, NEAR_HOVER, 0, TWO_ATTEMPTS),                                # config 5 under the two-attempt schedule: ~100 instances take the iteration
    (600, 1024, NEAR_HOVER, 0, {}),                                          # config 5 as shipped: ONE attempt of N / 16 = 32 passes (from N = 160 up, round 5)
    (600, 256, NEAR_HOVER, 12, TWO_ATTEMPTS),
    (57, 512, WILD, 6, {}),                                                  # short horizon, tail forced on: most of the batch passes through it
    (20, 512, WILD, 4, dict(qp_polish_passes=3, qp_polish_budget=6)),        # tight attempts: many instances leave the tail for the fallback list
])
def test_block_parallel_tail_gives_the_results_of_the_sequential_work_list(N, B, dist, J, over, monkeypatch):
    """The long-horizon tail (one warm interior-point iteration, one more attempt, every factorisation cut into blocks - DESIGN 4.6)
    against the sequential continuation of the same work list, cold (one shared linearisation) and warm-started (per stage):
    statuses, interior-point iterations and pass counts equal on every instance; commands and trajectories to 1e-9 / 1e-8."""
    over = dict(over, N=N, max_batch=B)
    yref, ye = hover_reference(N, 0.68 * 9.81 / 4.0)
    x0 = sample_x0(B, 5, **dist)
    (a, ita, psa, sta), (b, itb, psb, stb) = _solve_both(monkeypatch, over, x0, yref, ye, J)
    one_attempt = N >= 160 and not over.get("qp_polish_passes")   # the shipped long-horizon policy: nobody needs the interior point on this set
    assert sta["n_tail"] == stb["n_tail"] and (sta["n_tail"] > 0 or one_attempt)
    assert sta["tail_blocks"] == 0 and stb["tail_blocks"] == (J or round(0.7 * N ** 0.5))
    fin, fb = int((stb["tail_states"] == 3).sum()), int((stb["tail_states"] == 5).sum())
    assert fin + fb >= stb["n_tail"] and fin > 0                 # every work-list instance went through the tail, some to the end
    if over.get("qp_polish_passes", 8) < 8:
        assert fb > 0                                            # ... and with tight attempts some to the fallback list
    if one_attempt:
        # ONE attempt of N / 16 passes (32 at N = 600) and no iteration in between: the one instance of the 1024 that ran out of 16 passes - and
        # then cost the batch 25 ms of sequential interior point from the fallback list - is accepted after 23 (late round 5)
        assert np.abs(psb).max() > 16 and int((itb > 0).sum()) == 0
    np.testing.assert_array_equal(a["status"], b["status"])
    np.testing.assert_array_equal(ita, itb)
    np.testing.assert_array_equal(psa, psb)
    ok = a["status"] == 0
    np.testing.assert_allclose(b["u0"][ok], a["u0"][ok], rtol=0, atol=1e-9)
    # (trajectories: the tail's forward sweep starts every block from a state the boundary scan reconstructs - relative 1e-9 of states
    # that reach hundreds of metres over a 30 s horizon)
    np.testing.assert_allclose(b["x"][ok], a["x"][ok], rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(b["u"][ok], a["u"][ok], rtol=0, atol=1e-8)
    # second solve from the first one's trajectories: the per-stage linearisation through the same two paths
    if N <= 100:
        (a2, ita2, psa2, sta2), (b2, itb2, psb2, stb2) = _solve_both(monkeypatch, over, x0, yref, ye, J, warm_from=a)
        np.testing.assert_array_equal(a2["status"], b2["status"])
        np.testing.assert_array_equal(ita2, itb2)
        np.testing.assert_array_equal(psa2, psb2)
        ok2 = a2["status"] == 0
        np.testing.assert_allclose(b2["u0"][ok2], a2["u0"][ok2], rtol=0, atol=1e-9)


@pytest.mark.parametrize("seed", [30001, 30003, 30004, 30005])
def test_long_horizons_on_random_vehicles_against_the_oracle(seed):
    """Four draws of the long-horizon fuzz (tools/dev/fuzz_parity.py --long, profiles/r05_fuzz_parity_long_horizon_draws_*: horizons
    200 / 320 / 400 / 600 on random vehicles, tunings and references; wild and aggressive initial states, shared and per-stage
    linearisation, up to 32 passes, interior-point endings and failed instances among them): the block-parallel tail against the oracle,
    cold and warm-started - statuses equal on every instance, commands of the solved ones to 1e-8 of the hover thrust."""
    from oracle import oracle as O
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    from tests.fuzz_draws import draw, oracle_config
    over, x0, yref, ye, hov, _, _ = draw(seed, horizons=[160, 200, 256, 320, 400, 600], max_batch=65)
    s = NmpcOcpSolver(_lib.default_config(**over))
    c = oracle_config(s.config, qp_polish=1)
    out = s.solve_batch(x0, yref, ye, want_traj=True)
    assert s.tail_states(len(x0))[0] > 0                     # the handle has the block-parallel tail (N >= 160)
    ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=8)
    out2 = s.solve_batch(x0, yref, ye, x_init=ref["x"], u_init=ref["u"])
    ref2 = O.solve_batch(c, x0, yref, ye, x_init=ref["x"], u_init=ref["u"], nthreads=8)
    for o, r in ((out, ref), (out2, ref2)):
        np.testing.assert_array_equal(o["status"], r["status"])
        ok = r["status"] == 0
        assert ok.any() and np.abs(o["u0"][ok] - r["u0"][ok]).max() <= 1e-8 * max(1.0, hov)
    s.close()


def test_block_parallel_tail_on_a_per_stage_linearisation_at_the_long_horizon(monkeypatch):
    """N = 600 without the shared cold-start linearisation (what a warm-started tick runs): tail against sequential work list."""
    N, B = 600, 256
    over = dict(N=N, max_batch=B, flags=_lib.FLAG_TEAM_MAPPING, **TWO_ATTEMPTS)      # (the schedule with the interior-point step in it)
    yref, ye = hover_reference(N, 0.68 * 9.81 / 4.0)
    x0 = sample_x0(B, 5, **NEAR_HOVER)
    (a, ita, psa, sta), (b, itb, psb, stb) = _solve_both(monkeypatch, over, x0, yref, ye, 0)
    assert sta["n_tail"] >= 8
    np.testing.assert_array_equal(a["status"], b["status"])
    np.testing.assert_array_equal(ita, itb)
    np.testing.assert_array_equal(psa, psb)
    np.testing.assert_allclose(b["u0"], a["u0"], rtol=0, atol=1e-9)


def test_tail_is_the_default_from_horizon_160_and_handles_fp32_buffers(monkeypatch):
    """Default switches (no environment): the tail runs from N = 160 up and not below; on FP32 device buffers (NMPC_DTYPE_F32IO: FP64
    arithmetic) it gives what the sequential work list gives."""
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    monkeypatch.delenv("NMPC_BLOCK_TAIL", raising=False)
    monkeypatch.delenv("NMPC_BLOCK_J", raising=False)
    for N, on in ((120, False), (200, True)):
        s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=64))
        yref, ye = hover_reference(N, 0.68 * 9.81 / 4.0)
        out = s.solve_batch(sample_x0(64, 3, **AGGRESSIVE), yref, ye)
        blocks, _ = s.tail_states(64)
        assert (blocks > 0) == on and (out["status"] == 0).all()
        s.close()
    N, B = 200, 256
    yref, ye = hover_reference(N, 0.68 * 9.81 / 4.0)
    x0 = sample_x0(B, 11, **AGGRESSIVE).astype(np.float32).astype(np.float64)
    res = []
    for tail in ("0", "1"):
        monkeypatch.setenv("NMPC_BLOCK_TAIL", tail)
        s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B, dtype=_lib.DTYPE_F32IO))
        out = s.solve_batch(x0, yref, ye, want_traj=True)
        res.append((out, s.counts(), s.tail_states(B)))
        s.close()
    (a, (ita, psa), _), (b, (itb, psb), (blocks, states)) = res
    assert blocks > 0 and (states == 3).sum() > 0
    np.testing.assert_array_equal(a["status"], b["status"])
    np.testing.assert_array_equal(psa, psb)
    np.testing.assert_array_equal(ita, itb)
    np.testing.assert_allclose(b["u0"], a["u0"], rtol=0, atol=2e-6)          # FP32 outputs: one ulp of a 6 N command is 5e-7
    np.testing.assert_allclose(b["x"], a["x"], rtol=0, atol=2e-5)


def test_flag_build_of_the_block_kernels_is_bit_equal_to_the_default_codegen_build(monkeypatch):
    """The block kernels (k_block_sweep, k_block_scan and their tail forms) and the tail's own kernel k_team_tail exist twice since round 4: nmpc_blockf.hip built with
    -mllvm -amdgpu-mfma-vgpr-form (what runs), nmpc_block.hip with the default code generation (NMPC_BLOCK_NOFLAG=1).  Same source, same
    arithmetic: a long-horizon solve through the tail and the factorisation building block must agree bit for bit."""
    import torch
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    N, B = 600, 256
    yref, ye = hover_reference(N, 0.68 * 9.81 / 4.0)
    x0 = sample_x0(B, 5, **NEAR_HOVER)
    res = []
    for noflag in ("0", "1"):
        monkeypatch.setenv("NMPC_BLOCK_NOFLAG", noflag)
        monkeypatch.setenv("NMPC_QP_NOFLAG", noflag)          # k_team_tail / k_team_qp_list of the same solve: both builds too
        s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B))
        out = s.solve_batch(x0, yref, ye, want_traj=True)
        it, ps = s.counts()
        blocks, states = s.tail_states(B)
        # the building block on the per-stage tiles of a warm-started solve
        out2 = s.solve_batch(x0, yref, ye, x_init=out["x"], u_init=out["u"], want_traj=True)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        d = [dev(x0), dev(yref), dev(ye), dev(out["x"]), dev(out["u"])]
        fac = torch.zeros(B, N, 80, dtype=torch.float64, device="cuda")
        Jeff, _ = s.block_factor_device(B, 17, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), True, x_init_ptr=d[3].data_ptr(),
                                        u_init_ptr=d[4].data_ptr(), factors_ptr=fac.data_ptr())
        torch.cuda.synchronize()
        res.append((out, it.copy(), ps.copy(), states.copy(), out2, fac.cpu().numpy(), blocks, Jeff))
        s.close()
    a, b = res
    assert a[6] > 0 and (a[3] == 3).sum() > 0 and a[7] == b[7]          # the tail ran and finished instances
    for key in ("u0", "status", "x", "u"):
        np.testing.assert_array_equal(a[0][key], b[0][key])
        np.testing.assert_array_equal(a[4][key], b[4][key])
    np.testing.assert_array_equal(a[1], b[1]); np.testing.assert_array_equal(a[2], b[2]); np.testing.assert_array_equal(a[3], b[3])
    np.testing.assert_array_equal(a[5], b[5])
