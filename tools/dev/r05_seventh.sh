#!/bin/bash
# round 5, seventh dev call: config 5 on the SURVEY's own sample (seed 5) and on seed 0 / 1, aggressive - with pass budgets of 16 (shipped), 24, 32
mkdir -p gpurun_out
O=gpurun_out/r05p_config5_budget.txt
python -c "from rotors_mpc_controller_amd import _lib; print('#', _lib.load().nmpc_version().decode())" > $O 2>/dev/null
row() {
  python bench.py --no-cpu-baseline --no-secondary --batch 1024 --horizon 600 --steps 5 --warmup 1 "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$*] FAILED" >> $O; tail -3 gpurun_out/bench_q.err >> $O; return; }
  python - "$*" >> $O <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:60s}] {d['ms_per_step']:8.3f} ms  ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  passes {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  status {d['status_histogram']}")
PY
}
for seed in 0 5 1; do
  row --seed $seed
  row --seed $seed --polish-passes 24 --polish-budget 24
  row --seed $seed --polish-passes 32 --polish-budget 32
done
row --dist aggressive
row --dist aggressive --polish-passes 32 --polish-budget 32
cat $O
