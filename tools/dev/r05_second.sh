#!/bin/bash
# round 5, second GPU call: suite, same-box A/B of the plain interior point (pair loads), config 5 under the one-attempt policy: block count and
# hand-over sweeps, and the kernel timeline of one solve
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r05c_gpu_tests.log 2>&1 || tail -30 gpurun_out/r05c_gpu_tests.log
tail -1 gpurun_out/r05c_gpu_tests.log
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$ENVTAG $*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:64s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  device {d['device_ms_per_step']:.4f}  ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  {d.get('binary_source_hash')}", flush=True)
PY
}
OLD=$PWD/tools/dev/ab_round4_final.so
{
for i in 1 2; do
  ENVTAG="round-4 final"; ROTORS_NMPC_LIB=$OLD row --no-polish
  ENVTAG="this tree    "; row --no-polish
done
ENVTAG="round-4 final"; ROTORS_NMPC_LIB=$OLD row --steps 1000 --warmup 200
ENVTAG="this tree    "; row --steps 1000 --warmup 200
ENVTAG="round-4 final"; ROTORS_NMPC_LIB=$OLD row --no-polish --batch 65536 --steps 20 --warmup 4
ENVTAG="this tree    "; row --no-polish --batch 65536 --steps 20 --warmup 4
} 2>&1 | tee gpurun_out/r05c_same_box_ab.txt
{
for J in 10 12 14 17 20 24 30; do ENVTAG="NMPC_BLOCK_J=$J"; NMPC_BLOCK_J=$J row --batch 1024 --horizon 600 --steps 5 --warmup 1; done
for C in 2 3; do ENVTAG="NMPC_TAIL_CAP=$C"; NMPC_TAIL_CAP=$C row --batch 1024 --horizon 600 --steps 5 --warmup 1; done
ENVTAG="two attempts 8/16"; row --batch 1024 --horizon 600 --steps 5 --warmup 1 --polish-passes 8 --polish-budget 16
ENVTAG="N=250"; row --batch 1024 --horizon 250 --steps 10 --warmup 2
ENVTAG="N=250 two attempts 8/16"; row --batch 1024 --horizon 250 --steps 10 --warmup 2 --polish-passes 8 --polish-budget 16
ENVTAG="N=160"; row --batch 1024 --horizon 160 --steps 10 --warmup 2
ENVTAG="N=160 two attempts 8/16"; row --batch 1024 --horizon 160 --steps 10 --warmup 2 --polish-passes 8 --polish-budget 16
} 2>&1 | tee gpurun_out/r05c_config5_sweeps.txt
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/rocprof_r05c_n600
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/rocprof_r05c_n600 -- python3 $REPO/bench.py --no-cpu-baseline --no-secondary --batch 1024 --horizon 600 --steps 2 --warmup 1 > $REPO/gpurun_out/r05c_n600_trace.log 2>&1
cd $REPO
F=$(find gpurun_out/rocprof_r05c_n600 -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $F 0 400 | grep -v "elementwise\|fillBuffer\|copyBuffer" | tail -95 > gpurun_out/r05c_config5_timeline.txt 2>&1
tail -5 gpurun_out/r05c_config5_timeline.txt
