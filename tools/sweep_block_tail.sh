#!/bin/bash
# Block-parallel tail of long-horizon solves: off / on, number of blocks, horizon.
#   gpurun -- bash tools/sweep_block_tail.sh
row() {
  env $1 python bench.py --no-cpu-baseline --no-secondary --batch 1024 --steps 5 --warmup 1 --horizon $2 > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
l = d["roofline"]["launches"]
print(f"N = {sys.argv[2]:>4s}  {sys.argv[1]:34s} {d['ms_per_step']:8.3f} ms/step  first launch {l.get('k_team_as_ms_isolated', 0):7.3f}  rest {l.get('k_team_qp_list_ms_isolated', 0):7.3f}  "
      f"in work list {l.get('instances_in_second_launch')}  passes {d['active_set_passes']['mean']:.2f}/{d['active_set_passes']['max']}  ipm {d['ipm_iterations']['mean']:.3f}/{d['ipm_iterations']['max']}  st {d['status_histogram']}")
PY
}
for N in 120 250 600; do
  row NMPC_BLOCK_TAIL=0 $N
  row NMPC_BLOCK_TAIL=1 $N
done
for J in 8 12 16 26 32; do row "NMPC_BLOCK_TAIL=1 NMPC_BLOCK_J=$J" 600; done
