import sys, ctypes
sys.path.insert(0,'.')
mode = sys.argv[1]
if mode != 'notorch':
    import torch
    if mode == 'avail': print('avail', torch.cuda.is_available())
    if mode == 'init': torch.zeros(1, device='cuda'); print('init done')
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
cfg = _lib.default_config(max_batch=64)
try:
    s = NmpcOcpSolver(cfg); print(mode, 'create OK')
except Exception as e:
    print(mode, 'FAILED', e)
print([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime' in l][:6:2])
