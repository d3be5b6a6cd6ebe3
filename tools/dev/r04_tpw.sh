#!/bin/bash
# two waves per SIMD at B = 4096 by putting two instances in a wave instead of four (experiment)
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
ENVTAG="default            "; row --steps 500 --warmup 100
ENVTAG="TPW=2 OCC=2        "; NMPC_TEAM_TPW=2 NMPC_TEAM_OCC=2 row --steps 500 --warmup 100
ENVTAG="TPW=4 OCC=2        "; NMPC_TEAM_OCC=2 row --steps 500 --warmup 100
ENVTAG="TPW=2 OCC=1        "; NMPC_TEAM_TPW=2 row --steps 500 --warmup 100
ENVTAG="default  no-share  "; row --no-share
ENVTAG="TPW=2 no-share     "; NMPC_TEAM_TPW=2 row --no-share
} 2>&1 | tee gpurun_out/r04l_tpw.txt
