import numpy as np, sys
sys.path.insert(0,'.')
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import sample_x0, NEAR_HOVER, AGGRESSIVE, hover_reference
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
yref, ye = hover_reference(20, 0.68*9.81/4)
for share in (1, 0):
  s = NmpcOcpSolver(_lib.default_config(max_batch=1024, flags=share|2))
  for name, dist in (('near',NEAR_HOVER),('aggr',AGGRESSIVE),('wild',WILD)):
    x0 = sample_x0(600, 3, **dist)
    out = s.solve_batch(x0, yref, ye, want_traj=True); st = s.stats()
    ref = O.solve_batch(O.default_config(qp_polish=1), x0, yref, ye, want_traj=True)
    ref0 = O.solve_batch(O.default_config(), x0, yref, ye, want_traj=True)
    print('share',share,name,'status',np.bincount(out['status']),'|u0-oracle(polish)| %.2e  |u0-oracle(ipm)| %.2e  x %.2e'%(np.abs(out['u0']-ref['u0']).max(), np.abs(out['u0']-ref0['u0']).max(), np.abs(out['x']-ref['x']).max()),
          'iters gpu %.3f oracle %.3f'%(st['iter_mean'], ref['iters'].mean()), 'passes mean %.2f max %d polished %d'%(st['polish_mean'], st['polish_max'], st['n_polished']))
