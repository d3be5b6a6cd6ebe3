#!/bin/bash
# Round 4, second GPU call: GPU suite on the shared-stage sources, the bench rows that moved, the default line with the new cpu_baseline.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04b_gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04b_gpu_tests.log
row() {
  python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err || { echo "[$*] FAILED"; tail -3 gpurun_out/bench_q.err; return; }
  python - "$*" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_q.json"))
r = d["roofline"]
print(f"[{sys.argv[1]:44s}] {d['value']/1e6:9.4f} M/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f}  "
      f"ipm {d['ipm_iterations']['mean']:.2f}/{d['ipm_iterations']['max']}  pol {d['active_set_passes']['mean']:.3f}/{d['active_set_passes']['max']}  "
      f"flop-frac exec {r['alu']['frac']:.4f}  st {d['status_histogram']}")
PY
}
{
row --steps 400 --warmup 40
row --steps 400 --warmup 40
row --no-share
row --dist aggressive
row --no-polish
row --batch 65536
row --batch 65536 --no-share
row --batch 1024 --horizon 600 --steps 5 --warmup 1
row --batch 1024 --horizon 600 --steps 5 --warmup 1
row --traj-out
} 2>&1 | tee gpurun_out/r04b_bench_rows.txt
python bench.py --steps 200 --warmup 20 > gpurun_out/r04b_bench_default.json 2> gpurun_out/r04b_bench_default.err; echo "default bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04b_bench_default.json"))
c = d["cpu_baseline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print("cpu oracle", c["value"], "single", c["single_thread_value"], "eff", c["scaling_efficiency"], c["cores"], c.get("physical_cores"), c["sample"][:60])
s = c["structured"]; print("cpu structured", s.get("value"), s.get("single_thread_value"), s.get("scaling_efficiency"), s.get("error"))
print("secondary", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("secondary", {}).items()})
PY
