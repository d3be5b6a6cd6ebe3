#!/bin/bash
# Sweep of nmpc_config.qp_polish_ckpt (Riccati checkpoint window of the active-set passes).
set -e
for b in 4096 65536; do for d in "near_hover" "aggressive"; do for ck in 0 2 4 8 12 19; do
  python bench.py --no-cpu-baseline --batch $b --dist $d --polish-ckpt $ck > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err
  python - "$b $d ckpt=$ck" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[{sys.argv[1]:32s}] {d['value']/1e6:8.3f} M/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}")
PY
done; done; done
