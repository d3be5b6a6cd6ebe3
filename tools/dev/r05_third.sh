#!/bin/bash
# round 5, third GPU call: suite on the build that runs the scan's forward walk beside the final sweeps; config 5 with and without
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r05d_gpu_tests.log 2>&1 || tail -30 gpurun_out/r05d_gpu_tests.log
tail -1 gpurun_out/r05d_gpu_tests.log
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$ENVTAG $*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:64s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  device {d['device_ms_per_step']:.4f}  ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  {d.get('binary_source_hash')}", flush=True)
PY
}
{
for i in 1 2; do
  ENVTAG="NMPC_TAIL_FWD_OVERLAP=0"; NMPC_TAIL_FWD_OVERLAP=0 row --batch 1024 --horizon 600 --steps 5 --warmup 1
  ENVTAG="default                "; row --batch 1024 --horizon 600 --steps 5 --warmup 1
done
ENVTAG="NMPC_TAIL_FWD_OVERLAP=0"; NMPC_TAIL_FWD_OVERLAP=0 row --batch 1024 --horizon 250 --steps 10 --warmup 2
ENVTAG="default                "; row --batch 1024 --horizon 250 --steps 10 --warmup 2
ENVTAG="default N=160          "; row --batch 1024 --horizon 160 --steps 10 --warmup 2
ENVTAG="NMPC_BLOCK_TAIL=0 N=160"; NMPC_BLOCK_TAIL=0 row --batch 1024 --horizon 160 --steps 10 --warmup 2
ENVTAG="default N=120          "; row --batch 1024 --horizon 120 --steps 10 --warmup 2
ENVTAG="NMPC_BLOCK_TAIL=1 N=120"; NMPC_BLOCK_TAIL=1 row --batch 1024 --horizon 120 --steps 10 --warmup 2
} 2>&1 | tee gpurun_out/r05d_fwd_overlap.txt
