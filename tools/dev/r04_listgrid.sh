#!/bin/bash
# workgroups of the work-list launch (the schedules that keep the list: per-stage without trajectories, NMPC_TEAM_INPLACE=0)
mkdir -p gpurun_out
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$ENVTAG $*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:72s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  st {d['status_histogram']} ipm {d['ipm_iterations']['mean']:.3f}")
PY
}
{
for g in 64 256 1024; do
ENVTAG="grid $g, empty list       "; NMPC_LIST_GRID=$g row --no-share
ENVTAG="grid $g, 37 % handed over "; NMPC_LIST_GRID=$g row --no-share --dist aggressive --polish-passes 2 --polish-budget 4
ENVTAG="grid $g, shared, list kept"; NMPC_TEAM_INPLACE=0 NMPC_LIST_GRID=$g row --dist aggressive --polish-passes 2 --polish-budget 4
done
ENVTAG="in place (default), shared"; row --dist aggressive --polish-passes 2 --polish-budget 4
} 2>&1 | tee gpurun_out/r04s_list_grid.txt
