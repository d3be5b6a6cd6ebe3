#!/bin/bash
# One-call GPU check used during kernel work: parity tests, default bench, sweep profile.
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
for args in "" "--no-share" "--dist aggressive" "--batch 65536" "--batch 65536 --dtype f32" "--batch 1024 --horizon 600"; do
  python bench.py --no-cpu-baseline $args > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err
  python - "$args" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[{sys.argv[1]:32s}] {d['value']/1e6:8.3f} M/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f} (isolated {r['kernel_ms_isolated']:.4f})  prep {r['prepare_ms']:.4f}  "
      f"ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  st {d['status_histogram']}")
PY
done
make -s -C rotors_mpc_controller_amd/csrc prof
python tools/profile_sweeps.py > gpurun_out/prof_sweeps.txt 2>gpurun_out/prof_sweeps.err || { echo "profile_sweeps.py FAILED"; tail -5 gpurun_out/prof_sweeps.err; exit 1; }; tail -9 gpurun_out/prof_sweeps.txt; if grep -q "core dump" gpurun_out/prof_sweeps.txt gpurun_out/prof_sweeps.err; then echo "GPU FAULT in profile_sweeps"; exit 1; fi
