#!/bin/bash
# All rows of DESIGN.md section 5 in one GPU call.
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[{sys.argv[1]:44s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f}  "
      f"ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  "
      f"flop-frac exec {r['alu']['frac']:.4f} nominal {r['alu']['nominal']['frac']:.4f} hbm {r['hbm']['frac']:.5f}  st {d['status_histogram']}")
PY
}
row
row --no-share
row --dist aggressive
row --no-polish
row --mapping lane --steps 5 --warmup 1
row --batch 1024
row --batch 16384
row --batch 65536 --dtype f32io
row --batch 65536 --dtype f32io --seed 1
row --batch 65536 --dtype f32io --no-polish --steps 20 --warmup 4
row --batch 65536
row --batch 65536 --no-share
row --batch 1024 --horizon 600 --steps 5 --warmup 1
row --batch 1024 --horizon 600 --steps 5 --warmup 1 --seed 5
row --batch 1024 --horizon 600 --steps 5 --warmup 1 --polish-passes 8 --polish-budget 16
row --batch 1024 --horizon 250 --steps 10 --warmup 2
row --condensed --steps 3 --warmup 1
row --yref broadcast
row --traj-out
