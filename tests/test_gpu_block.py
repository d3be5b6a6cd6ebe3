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
