#!/bin/bash
# The measured evidence of a round in one GPU call: rocprofv3 captures (kernel trace + PMC passes) of the default workload, of the
# plain interior point and of config 5, the bench table of DESIGN.md section 5, the full bench.py line, the block-factorisation table.
#   gpurun --timeout 1100 -- bash tools/evidence.sh r03          (outputs under gpurun_out/; copy what is to be judged into profiles/)
TAG=${1:-r05}
mkdir -p gpurun_out
set -e
bash tools/rocprof_capture.sh $TAG > gpurun_out/${TAG}_capture.log 2>&1
python tools/summarize_pmc.py gpurun_out/rocprof_$TAG gpurun_out/${TAG}_pmc_summary.json
BENCH_ARGS="--no-polish" bash tools/rocprof_capture.sh ${TAG}ipm > gpurun_out/${TAG}ipm_capture.log 2>&1
python tools/summarize_pmc.py gpurun_out/rocprof_${TAG}ipm gpurun_out/${TAG}ipm_pmc_summary.json
BENCH_ARGS="--batch 1024 --horizon 600 --steps 4 --warmup 1" bash tools/rocprof_capture.sh ${TAG}n600 > gpurun_out/${TAG}n600_capture.log 2>&1
python tools/summarize_pmc.py gpurun_out/rocprof_${TAG}n600 gpurun_out/${TAG}n600_pmc_summary.json
BENCH_ARGS="--no-share" bash tools/rocprof_capture.sh ${TAG}ps > gpurun_out/${TAG}ps_capture.log 2>&1
python tools/summarize_pmc.py gpurun_out/rocprof_${TAG}ps gpurun_out/${TAG}ps_pmc_summary.json
echo "captures done"
bash tools/bench_table.sh > gpurun_out/${TAG}_bench_table.txt 2>&1
echo "bench table done"
python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err
python tools/block_factor_rate.py 600 100 > gpurun_out/${TAG}_block_factor_rate.txt 2>&1
python tools/block_factor_rate.py 600 1024 1 5 15 25 40 >> gpurun_out/${TAG}_block_factor_rate.txt 2>&1
tail -3 gpurun_out/${TAG}_bench_table.txt
