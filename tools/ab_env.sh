#!/bin/bash
# A/B of an environment knob of the library: bash tools/ab_env.sh VAR "v1 v2" [bench args...]
VAR=$1; VALS=$2; shift 2
for b in 4096 65536; do for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline --batch $b "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err
  python - "$b $VAR=$v $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[{sys.argv[1]:36s}] {d['value']/1e6:8.3f} M/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f}  prep {r['prepare_ms']:.4f}")
PY
done; done
