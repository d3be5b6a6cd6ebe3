#!/bin/bash
# round 5, fourth GPU call: the tail with blocks per step chosen from the list's count and the first pass made by the tail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_block.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/r05e_gpu_tests.log 2>&1 || tail -40 gpurun_out/r05e_gpu_tests.log
tail -1 gpurun_out/r05e_gpu_tests.log
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$ENVTAG $*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:70s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  device {d['device_ms_per_step']:.4f}  ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  {d.get('binary_source_hash')}", flush=True)
PY
}
{
A="--batch 1024 --horizon 600 --steps 5 --warmup 1"
ENVTAG="default (cap 0, waves 1024)        "; row $A
ENVTAG="NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0  "; NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0 row $A
ENVTAG="NMPC_TAIL_CAP=1                    "; NMPC_TAIL_CAP=1 row $A
ENVTAG="NMPC_TAIL_WAVES=0                  "; NMPC_TAIL_WAVES=0 row $A
for W in 768 1280 1536 2048; do ENVTAG="NMPC_TAIL_WAVES=$W                "; NMPC_TAIL_WAVES=$W row $A; done
ENVTAG="default                            "; row $A
A="--batch 1024 --horizon 250 --steps 10 --warmup 2"
ENVTAG="default                            "; row $A
ENVTAG="NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0  "; NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0 row $A
A="--batch 1024 --horizon 160 --steps 10 --warmup 2"
ENVTAG="default                            "; row $A
ENVTAG="NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0  "; NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0 row $A
ENVTAG="NMPC_BLOCK_TAIL=0                  "; NMPC_BLOCK_TAIL=0 row $A
A="--batch 256 --horizon 600 --steps 5 --warmup 1"
ENVTAG="default                            "; row $A
ENVTAG="NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0  "; NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0 row $A
A="--batch 4096 --horizon 600 --steps 3 --warmup 1"
ENVTAG="default                            "; row $A
ENVTAG="NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0  "; NMPC_TAIL_CAP=1 NMPC_TAIL_WAVES=0 row $A
} 2>&1 | tee gpurun_out/r05e_adaptive_tail.txt
