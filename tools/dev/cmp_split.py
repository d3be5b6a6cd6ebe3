"""dev: split path (active-set kernel + work list) vs the single general kernel on the same inputs."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, '.')
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
N = int(sys.argv[1]); B = int(sys.argv[2]); dist = sys.argv[3] if len(sys.argv) > 3 else "near"
if len(sys.argv) > 4 and sys.argv[4] == "child":
    from rotors_mpc_controller_amd import _lib
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0
    s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B))
    yref, ye = hover_reference(N, 0.68 * 9.81 / 4)
    x0 = sample_x0(B, 5, **(NEAR_HOVER if dist == "near" else AGGRESSIVE))
    o = s.solve_batch(x0, yref, ye, want_traj=True)
    import ctypes as C
    import torch
    npol = torch.zeros(1)  # noqa
    st = s.stats()
    np.savez(sys.argv[5], u=o["u"], x=o["x"], u0=o["u0"], status=o["status"], stats=json.dumps(st))
    sys.exit(0)
outs = {}
for name, env in (("split", {}), ("single", {"NMPC_TEAM_SPLIT": "0"})):
    f = f"/tmp/cmp_{name}.npz"
    subprocess.check_call([sys.executable, __file__, str(N), str(B), dist, "child", f], env=dict(os.environ, **env))
    outs[name] = np.load(f)
a, b = outs["split"], outs["single"]
print("stats split ", a["stats"]); print("stats single", b["stats"])
du = np.abs(a["u"] - b["u"]).max(axis=(1, 2)); dx = np.abs(a["x"] - b["x"]).max(axis=(1, 2))
bad = np.where((du > 1e-9) | (a["status"] != b["status"]))[0]
print("mismatching instances:", len(bad), bad[:40])
for i in bad[:10]:
    k = np.abs(a["u"][i] - b["u"][i]).max(axis=1).argmax()
    print(i, "status", a["status"][i], b["status"][i], "du", du[i], "dx", dx[i], "worst stage", k, a["u"][i, k], b["u"][i, k])
