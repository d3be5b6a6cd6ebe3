#!/bin/bash
# the default path as two launches (k_team_as + k_team_qp_list) against one (k_team_qp, NMPC_TEAM_SPLIT=0), same box
mkdir -p gpurun_out
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$ENVTAG $*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:64s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  device {d['device_ms_per_step']:.4f}  st {d['status_histogram']}")
PY
}
{
for i in 1 2; do
ENVTAG="two launches "; row --steps 1000 --warmup 200
ENVTAG="one launch   "; NMPC_TEAM_SPLIT=0 row --steps 1000 --warmup 200
done
for a in "--no-share" "--dist aggressive" "--traj-out" "--steps 20 --warmup 5"; do
ENVTAG="two launches "; row $a
ENVTAG="one launch   "; NMPC_TEAM_SPLIT=0 row $a
done
} 2>&1 | tee gpurun_out/r04n_split.txt
