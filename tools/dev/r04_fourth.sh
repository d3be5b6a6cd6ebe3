#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04d_gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04d_gpu_tests.log
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[{sys.argv[1]:60s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f}  "
      f"ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  "
      f"flop-frac exec {r['alu']['frac']:.4f}  st {d['status_histogram']}")
PY
}
{
ENVTAG=""; row --steps 400 --warmup 40
ENVTAG="NMPC_TEAM_SPLIT=0"; NMPC_TEAM_SPLIT=0 row --steps 400 --warmup 40
ENVTAG="NMPC_TEAM_SPLIT=0"; NMPC_TEAM_SPLIT=0 row --dist aggressive
ENVTAG=""; row --dist aggressive
row --no-share
ENVTAG="NMPC_LDS_OVERLAP=0"; NMPC_LDS_OVERLAP=0 row --no-share
ENVTAG=""; row --batch 65536 --no-share
ENVTAG="NMPC_LDS_OVERLAP=0"; NMPC_LDS_OVERLAP=0 row --batch 65536 --no-share
ENVTAG=""; row --no-polish --no-share
row --batch 1024 --horizon 600 --steps 5 --warmup 1
ENVTAG="NMPC_QP_NOFLAG=1"; NMPC_QP_NOFLAG=1 row --batch 1024 --horizon 600 --steps 5 --warmup 1
} 2>&1 | tee gpurun_out/r04d_bench_rows.txt
python tools/rollout_rate.py > gpurun_out/r04d_rollout_rate.txt 2>&1; tail -6 gpurun_out/r04d_rollout_rate.txt
