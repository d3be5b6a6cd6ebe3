#!/bin/bash
# Same-box A/B of the machine-scheduler strategy of the flag builds: the shipped library against tools/dev/librotors_nmpc_hip_ilp.so
# (make BUILD=build_ilp OUT=../../tools/dev/librotors_nmpc_hip_ilp.so ASFLAGS_HIP="-mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=iterative-ilp").
mkdir -p gpurun_out
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$ENVTAG $*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
print(f"[{sys.argv[1]:64s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  device {d['device_ms_per_step']:.4f}  st {d['status_histogram']}  du0 {d.get('max_abs_u0_vs_oracle')}")
PY
}
ALT=${ALT:-$PWD/tools/dev/librotors_nmpc_hip_ilp.so}
{
for i in 1 2 3; do
  ENVTAG="default sched"; row --steps 1000 --warmup 200
  ENVTAG="alternative  "; ROTORS_NMPC_LIB=$ALT row --steps 1000 --warmup 200
done
for a in "--no-share" "--no-polish" "--dist aggressive" "--traj-out" "--batch 65536 --steps 100 --warmup 20" "--batch 65536 --no-share --steps 100 --warmup 20" "--batch 1024 --horizon 600 --steps 5 --warmup 1"; do
  ENVTAG="default sched"; row $a
  ENVTAG="alternative  "; ROTORS_NMPC_LIB=$ALT row $a
done
} 2>&1 | tee gpurun_out/${OUTNAME:-r04j_sched_ab}.txt
