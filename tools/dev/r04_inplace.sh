#!/bin/bash
# failed first attempts continued inside k_team_as (default) against the work-list launch (NMPC_TEAM_INPLACE=0), same box, same binary
mkdir -p gpurun_out
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$ENVTAG $*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:64s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  device {d['device_ms_per_step']:.4f}  st {d['status_histogram']} ipm {d['ipm_iterations']['mean']:.3f}")
PY
}
{
for i in 1 2; do
ENVTAG="work list    "; NMPC_TEAM_INPLACE=0 row --steps 1000 --warmup 200
ENVTAG="in place     "; row --steps 1000 --warmup 200
done
for a in "--traj-out" "--dist aggressive" "--no-share" "--no-share --traj-out" "--steps 20 --warmup 5" "--batch 65536 --steps 100 --warmup 20" "--dist aggressive --polish-passes 2 --polish-budget 4"; do
ENVTAG="work list    "; NMPC_TEAM_INPLACE=0 row $a
ENVTAG="in place     "; row $a
done
} 2>&1 | tee gpurun_out/r04o_inplace.txt
