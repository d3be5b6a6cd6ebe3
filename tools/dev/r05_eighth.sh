#!/bin/bash
# round 5, eighth dev call: the fused tail launches (scan inside launch 1, decision inside the forward launch) - the long-horizon tests three
# times over (a lost update between block waves would show as a difference to the sequential work list), then timings fused / not fused
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05q_fused_tail.txt
python -c "from rotors_mpc_controller_amd import _lib; print('#', _lib.load().nmpc_version().decode())" > $O 2>/dev/null
for i in 1 2 3; do
  timeout -k 10 400 python -m pytest tests/test_gpu_block.py tests/test_gpu_configs.py -m gpu -q -x >> $O 2>&1 || { tail -30 $O; exit 1; }
done
row() {
  python bench.py --no-cpu-baseline --no-secondary --batch 1024 --steps 5 --warmup 1 "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$*] FAILED" >> $O; tail -3 gpurun_out/bench_q.err >> $O; return; }
  python - "$NMPC_TAIL_FUSE $*" >> $O <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[fuse {sys.argv[1]:60s}] {d['ms_per_step']:8.3f} ms  ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  passes {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  status {d['status_histogram']}")
PY
}
for f in 1 0; do
  export NMPC_TAIL_FUSE=$f
  row --horizon 600 --seed 0
  row --horizon 600 --seed 5
  row --horizon 600 --dist aggressive
  row --horizon 250 --steps 10 --warmup 2
  row --horizon 160 --steps 10 --warmup 2
done
grep -v "^\.\|^$" $O | tail -20
