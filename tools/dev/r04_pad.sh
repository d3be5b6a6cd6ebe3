#!/bin/bash
mkdir -p gpurun_out
{
for pad in 24 0 2 4 6 8 10 12 14 16 18 20 22 26 28 30 24; do
  NMPC_LDS_PAD=$pad python bench.py --no-cpu-baseline --no-secondary --steps 600 --warmup 100 > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "pad $pad FAILED"; tail -2 gpurun_out/bench_q.err; continue; }
  python - $pad <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"pad {sys.argv[1]:>2s}: {d['value']/1e6:8.3f} M/s  step {d['ms_per_step']:.4f} ms")
PY
done
} 2>&1 | tee gpurun_out/r04h_lds_pad_sweep.txt
