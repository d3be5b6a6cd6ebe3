#!/bin/bash
# round 5, final GPU call: the GPU suite, smoke(), the evidence captures and the bench table on the final kernel sources
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r05z_gpu_tests.log 2>&1 || tail -30 gpurun_out/r05z_gpu_tests.log
tail -1 gpurun_out/r05z_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r05z_smoke.txt
bash tools/evidence.sh r05 || exit 1
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_style_20_steps.json 2> gpurun_out/r05_bench_driver_style.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_driver_style_20_steps.json"))
print("driver-style 20 steps:", d["value"] / 1e6, "M solves/s", d["ms_per_step"], "ms; roofline frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "cpu", d["cpu_baseline"]["value"])
PY
