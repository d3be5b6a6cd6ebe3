#!/bin/bash
# Same-box A/B: the library as it was at the start of round 4 (commit 108b564: round-3 kernels + the round's first infrastructure) against this tree.
mkdir -p gpurun_out
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:56s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  device {d['device_ms_per_step']:.4f}  {d['binary_source_hash']}")
PY
}
OLD=$PWD/tools/dev/ab_round4_start.so
{
for i in 1 2 3; do
  ENVTAG="round-4 start"; ROTORS_NMPC_LIB=$OLD row --steps 1000 --warmup 200
  ENVTAG="this tree    "; row --steps 1000 --warmup 200
done
for a in "--no-share" "--no-polish" "--dist aggressive" "--batch 65536 --steps 100 --warmup 20" "--batch 1024 --horizon 600 --steps 5 --warmup 1"; do
  ENVTAG="round-4 start"; ROTORS_NMPC_LIB=$OLD row $a
  ENVTAG="this tree    "; row $a
done
} 2>&1 | tee gpurun_out/r04g_same_box_ab.txt
