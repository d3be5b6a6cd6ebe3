#!/bin/bash
# Round 4, third GPU call: suite with the flag builds of k_team_qp / the block kernels, the rows they move, config-5 sweeps, cpu_baseline with measured thread count.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04c_gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04c_gpu_tests.log
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
row --no-polish
ENVTAG="NMPC_QP_NOFLAG=1"; NMPC_QP_NOFLAG=1 row --no-polish
ENVTAG=""; row --batch 1024 --horizon 600 --steps 5 --warmup 1
row --batch 1024 --horizon 600 --steps 5 --warmup 1
ENVTAG="NMPC_BLOCK_NOFLAG=1"; NMPC_BLOCK_NOFLAG=1 row --batch 1024 --horizon 600 --steps 5 --warmup 1
for J in 12 14 20 24; do ENVTAG="NMPC_BLOCK_J=$J"; NMPC_BLOCK_J=$J row --batch 1024 --horizon 600 --steps 5 --warmup 1; done
ENVTAG=""; row --batch 1024 --horizon 250 --steps 5 --warmup 1
row --batch 65536 --no-polish --steps 20 --warmup 3
} 2>&1 | tee gpurun_out/r04c_bench_rows.txt
python bench.py --steps 200 --warmup 20 > gpurun_out/r04c_bench_default.json 2> gpurun_out/r04c_bench_default.err; echo "default bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c_bench_default.json"))
c = d["cpu_baseline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print("cpu oracle", c["value"], "single", c["single_thread_value"], "eff", c["scaling_efficiency"], "cores", c["cores"], c["note"][:260])
s = c["structured"]; print("cpu structured", s.get("value"), s.get("single_thread_value"), s.get("scaling_efficiency"), s.get("cores"), s.get("thread_count_table"), s.get("error"))
print("secondary", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("secondary", {}).items()})
PY
