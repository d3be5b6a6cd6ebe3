#!/bin/bash
# Occupancy / teams-per-wave sweep of the team kernel (env overrides read by nmpc_create).
for b in 4096 16384 65536; do for dt in f64 f32; do for occ in 1 2; do
  NMPC_TEAM_OCC=$occ python bench.py --no-cpu-baseline --batch $b --dtype $dt > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err
  python - "$b $dt occ=$occ" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[{sys.argv[1]:24s}] {d['value']/1e6:8.3f} M/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f}")
PY
done; done; done
