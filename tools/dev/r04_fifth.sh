#!/bin/bash
mkdir -p gpurun_out
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[{sys.argv[1]:60s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  device {d['device_ms_per_step']:.4f}  isolated {r.get('kernel_ms_isolated')}  launches {r.get('launches')}")
PY
}
{
for i in 1 2 3; do ENVTAG=""; row --steps 20 --warmup 5; done
for i in 1 2 3; do ENVTAG="PREROLL=300"; NMPC_BENCH_PREROLL=300 row --steps 20 --warmup 5; done
for i in 1 2; do ENVTAG="PREROLL=2000"; NMPC_BENCH_PREROLL=2000 row --steps 20 --warmup 5; done
ENVTAG=""; row --steps 2000 --warmup 100
} 2>&1 | tee gpurun_out/r04e_short_runs.txt
python tools/rollout_rate.py --graph > gpurun_out/r04e_rollout_graph.txt 2>&1; tail -2 gpurun_out/r04e_rollout_graph.txt
python tools/rollout_rate.py --graph --dist near_hover >> gpurun_out/r04e_rollout_graph.txt 2>&1; tail -1 gpurun_out/r04e_rollout_graph.txt
