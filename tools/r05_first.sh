#!/bin/bash
# round 5, first GPU call: the GPU suite, then the bench rows of tools/bench_table.sh's short list
set -e
mkdir -p gpurun_out
python -c "from rotors_mpc_controller_amd import _lib; print(_lib.load().nmpc_version().decode() if hasattr(_lib.load().nmpc_version(), 'decode') else _lib.load().nmpc_version())" 2>/dev/null || true
python -m pytest tests -m gpu -q > gpurun_out/r05a_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r05a_gpu_tests.log; }
tail -1 gpurun_out/r05a_gpu_tests.log
for args in "" "--no-share" "--dist aggressive" "--no-polish" "--batch 65536" "--batch 65536 --dtype f32" "--batch 1024 --horizon 600"; do
  python bench.py --no-cpu-baseline $args > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { tail -5 gpurun_out/bench_q.err; exit 1; }
  python - "$args" <<'PY' | tee -a gpurun_out/r05a_bench_rows.txt
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[{sys.argv[1]:32s}] {d['value']/1e6:8.3f} M/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f} (isolated {r['kernel_ms_isolated']:.4f})  "
      f"ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  st {d['status_histogram']}")
PY
done
# same-box A/B against the library of round 4's final commit (994b126, built by `make` in a worktree of that commit: tools/dev/ab_round4_final.so)
OLD=$PWD/tools/dev/ab_round4_final.so
if [ -f $OLD ]; then
  for args in "--steps 1000 --warmup 200" "--no-polish" "--no-share" "--dist aggressive" "--batch 1024 --horizon 600 --steps 5 --warmup 1"; do
    for lib in old new old new; do
      if [ $lib = old ]; then export ROTORS_NMPC_LIB=$OLD; else unset ROTORS_NMPC_LIB; fi
      python bench.py --no-cpu-baseline --no-secondary $args > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { tail -3 gpurun_out/bench_q.err; continue; }
      python - "$lib $args" <<'PY' | tee -a gpurun_out/r05a_same_box_ab.txt
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:60s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  device {d['device_ms_per_step']:.4f}  ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  {d.get('binary_source_hash')}")
PY
    done
  done
  unset ROTORS_NMPC_LIB
fi
