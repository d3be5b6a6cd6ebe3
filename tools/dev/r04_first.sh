#!/bin/bash
# Round 4, first GPU call: full GPU suite (with the new 3-4 integrator-step regression test), the kernel check of the final sources
# under canary bands, and the A/B of the three builds of k_team_as on the headline workload.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04a_gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04a_gpu_tests.log
NMPC_GUARD=64 timeout -k 10 500 python tools/dev/qp_kernel_check.py > gpurun_out/r04a_qp_kernel_check.txt 2>&1; echo "qp_kernel_check rc $?"; tail -4 gpurun_out/r04a_qp_kernel_check.txt
for b in flag default v256; do
  for rep in 1 2; do
    NMPC_AS_BUILD=$b python bench.py --no-cpu-baseline --no-secondary --steps 400 --warmup 40 > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$b] FAILED"; tail -3 gpurun_out/bench_q.err; continue; }
    python - "$b" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[build {sys.argv[1]:8s}] {d['value']/1e6:8.3f} M/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}")
PY
  done
done 2>&1 | tee gpurun_out/r04a_as_builds.txt
